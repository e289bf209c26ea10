#!/bin/bash
# rocprofv3 evidence for one round, run ON THE GPU BOX from the repo root:
#   tools/profile_round.sh gpurun_out/prof_r02 [extra bench.py args for every pass]
# Passes (each its own process; counters never share a run with --kernel-trace/--stats):
#   kt     --kernel-trace --stats over the driver-shaped command (bench.py --steps 20 --warmup 5)
#   kt16   the same at 16 777 216 boards per launch (a 112 us kernel: the profiler's per-dispatch cost, 0.5-1.2 us,
#          no longer shows, so the trace can be held against the un-profiled bench line)
#   kt_gym the same for --mode gym
#   pmc_f / pmc_w   FETCH_SIZE / WRITE_SIZE at 1 048 576 boards
#   pmc_f16 / pmc_w16  the same at 16 777 216 boards (state 256 MiB + outputs: beyond the Infinity Cache)
#   pmc_sq  SQ instruction / cycle counters
# The program after `--` is python3 itself (no env/bash hop: the profiler's preload has touched the GPU).
set -u
out=$(readlink -f "$1"); shift
R=$(readlink -f .)
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
run() { name=$1; shift; echo "== $name: $*"; "$@" > "$out/$name.log" 2>&1; echo "rc=$?" >> "$out/$name.log"; tail -2 "$out/$name.log"; }
run kt      rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt"      -- python3 "$B" --steps 20 --warmup 5 --no-cpu-baseline "$@"
run kt16    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt16"    -- python3 "$B" --boards 16777216 --steps 10 --warmup 2 --regions 4 --no-cpu-baseline "$@"
run kt_gym  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_gym"  -- python3 "$B" --steps 20 --warmup 5 --no-cpu-baseline --mode gym "$@"
run pmc_f   rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_f"   -- python3 "$B" --steps 20 --warmup 5 --regions 3 --no-cpu-baseline "$@"
run pmc_w   rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_w"   -- python3 "$B" --steps 20 --warmup 5 --regions 3 --no-cpu-baseline "$@"
run pmc_f16 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_f16" -- python3 "$B" --boards 16777216 --steps 10 --warmup 2 --regions 2 --no-cpu-baseline "$@"
run pmc_w16 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_w16" -- python3 "$B" --boards 16777216 --steps 10 --warmup 2 --regions 2 --no-cpu-baseline "$@"
run pmc_sq  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$out/pmc_sq" -- python3 "$B" --steps 20 --warmup 5 --regions 3 --no-cpu-baseline "$@"
# keep the merge-back small: the kernel traces are the only big files
find "$out" -name "*.db" -delete 2>/dev/null
du -sh "$out"
