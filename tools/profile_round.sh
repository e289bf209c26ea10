#!/bin/bash
# rocprofv3 evidence for one round, run ON THE GPU BOX from the repo root:
#   tools/profile_round.sh gpurun_out/prof_r03 [extra bench.py args for every pass]
# Passes (each its own process; counters never share a run with --kernel-trace/--stats):
#   kt        --kernel-trace --stats over the driver-shaped command (bench.py --steps 20 --warmup 5), no legs: the
#             timed region's dispatches, their gaps and periods (tools/trace_summary.py)
#   kt_full   the same over the DEFAULT command with its legs (what the driver runs), --stats table only
#   kt16      16 777 216 boards per launch (a 105 us kernel: the profiler's per-dispatch cost no longer shows)
#   kt_gym    --mode gym;  kt_fused  --mode random-fused at 262 144 boards (BASELINE config 3 / config 4's shard)
#   kt_rows   tools/bench_rows.py: the kernels beside the step (observe, export, import, node_info, expand, rollout, ...)
#   pmc_f / pmc_w   FETCH_SIZE / WRITE_SIZE at 1 048 576 boards; pmc_f16 / pmc_w16 at 16 777 216
#   pmc_sq, pmc_sq_fused, pmc_sq_rows   SQ instruction / cycle counters: replay, random-fused, tools/bench_rows.py
# The program after `--` is python3 itself (no env/bash hop: the profiler's preload has touched the GPU).
set -u
out=$(readlink -f "$1"); shift
R=$(readlink -f .)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
SQ="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU"
run() { name=$1; shift; echo "== $name: $*"; "$@" > "$out/$name.log" 2>&1; echo "rc=$?" >> "$out/$name.log"; tail -2 "$out/$name.log"; }
N="--no-cpu-baseline --no-legs"
run kt       rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt"       -- python3 "$B" --steps 20 --warmup 5 $N "$@"
run kt_full  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_full"  -- python3 "$B" --no-cpu-baseline "$@"
run kt16     rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt16"     -- python3 "$B" --boards 16777216 --steps 10 --warmup 2 --regions 4 $N "$@"
run kt_gym   rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_gym"   -- python3 "$B" --steps 20 --warmup 5 $N --mode gym "$@"
QTTT_ROWS_MIN_S=0 run kt_rows  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_rows"  -- python3 "$R/tools/bench_rows.py"
run kt_fused rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_fused" -- python3 "$B" --boards 262144 --steps 128 --warmup 10 --regions 20 $N --mode random-fused "$@"
run pmc_f    rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_f"   -- python3 "$B" --steps 20 --warmup 5 --regions 3 $N "$@"
run pmc_w    rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_w"   -- python3 "$B" --steps 20 --warmup 5 --regions 3 $N "$@"
run pmc_f16  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_f16" -- python3 "$B" --boards 16777216 --steps 10 --warmup 2 --regions 2 $N "$@"
run pmc_w16  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_w16" -- python3 "$B" --boards 16777216 --steps 10 --warmup 2 --regions 2 $N "$@"
run pmc_sq        rocprofv3 --pmc $SQ --output-format csv -d "$out/pmc_sq"       -- python3 "$B" --steps 20 --warmup 5 --regions 3 $N "$@"
run pmc_sq_fused  rocprofv3 --pmc $SQ --output-format csv -d "$out/pmc_sq_fused" -- python3 "$B" --steps 128 --warmup 10 --regions 3 $N --mode random-fused "$@"
QTTT_ROWS_MIN_S=0 run pmc_sq_rows   rocprofv3 --pmc $SQ --output-format csv -d "$out/pmc_sq_rows"  -- python3 "$R/tools/bench_rows.py"
# keep the merge-back small (gpurun copies at most 64 MiB back): everything is summarised HERE, on the box, and the raw
# per-dispatch CSVs are dropped except the kernel traces trace_summary.py reads (kt, kt16, kt_gym) and the step kernel's
# own counter rows.
find "$out" -name "*.db" -delete 2>/dev/null
find "$out" -name "*agent_info.csv" -delete 2>/dev/null
python3 "$R/tools/trace_rows_summary.py" "$out/kt_rows" > "$out/kernel_trace_rows.csv" 2> "$out/kernel_trace_rows.err"
for d in kt_full kt_rows kt_fused; do find "$out/$d" -name "*kernel_trace.csv" -delete 2>/dev/null; done
tag=$(basename "$out" | sed 's/^prof_//')
python3 "$R/tools/pmc_summary.py" 1048576 "$out/pmc_f" "$out/pmc_w" "$tag step_kernel<1024,2,false,true,false,false> via bench.py --steps 20 --warmup 5 --regions 3 --no-legs" 16 > "$out/pmc_traffic_1M.json"
python3 "$R/tools/pmc_summary.py" 16777216 "$out/pmc_f16" "$out/pmc_w16" "$tag step_kernel<256,2,false,true,false,false> via bench.py --boards 16777216 --steps 10 --warmup 2 --regions 2 --no-legs" 16 > "$out/pmc_traffic_16M.json"
cp "$R/profiles/pmc_traffic.json" "$out/pmc_traffic.json"
for n in pmc_f pmc_w pmc_f16 pmc_w16; do python3 - "$out/$n" "$out/${n}_step_kernel_counter_collection.csv" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows[-48:])
PY
done
python3 "$R/tools/pmc_sq_summary.py" "$out/pmc_sq" 1048576 > "$out/pmc_sq_summary.csv"
python3 "$R/tools/pmc_sq_summary.py" "$out/pmc_sq_fused" 1048576 > "$out/pmc_sq_fused_summary.csv"
python3 "$R/tools/pmc_sq_summary.py" "$out/pmc_sq_rows" 1048576 > "$out/pmc_sq_rows_summary.csv"
find "$out" -name "*counter_collection.csv" -not -name "*step_kernel_counter_collection.csv" -delete 2>/dev/null
du -sh "$out"/* | sort -h | tail -5
du -sh "$out"
