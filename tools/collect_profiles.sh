#!/bin/bash
# Folds the outputs of tools/profile_round.sh (PROF) and tools/round_numbers.sh (NUM) into profiles/rNN (OUT):
#   tools/collect_profiles.sh gpurun_out/prof_r03_final gpurun_out/r03_final profiles/r03
set -eu
P=$1; N=$2; O=$3
mkdir -p "$O"
j() { python3 - "$1" "$2" <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith("{"):
        d = json.loads(l)
        print(d[sys.argv[2]] if sys.argv[2] != "us" else d["ms_per_step"] * 1e3)
PY
}
unprof=$(j "$N/bench_k20.json" us)
python3 tools/trace_summary.py "$P/kt" 20 5 "$(j "$P/kt.log" regions)" step_kernel "$unprof" > "$O/kernel_trace_timed_region.txt"
python3 tools/trace_summary.py "$P/kt16" 10 2 4 step_kernel > "$O/kernel_trace_timed_region_16M_boards.txt"
python3 tools/trace_summary.py "$P/kt_gym" 20 5 "$(j "$P/kt_gym.log" regions)" step_kernel > "$O/kernel_trace_timed_region_gym.txt"
cp "$P"/kt/runc/*_kernel_stats.csv "$O/kernel_stats.csv"
cp "$P"/kt_full/runc/*_kernel_stats.csv "$O/kernel_stats_default_command_with_legs.csv"
cp "$P"/kt16/runc/*_kernel_stats.csv "$O/kernel_stats_16M_boards.csv"
cp "$P"/kt_gym/runc/*_kernel_stats.csv "$O/kernel_stats_gym.csv"
cp "$P"/kt_fused/runc/*_kernel_stats.csv "$O/kernel_stats_random_fused_262144.csv"
[ -d "$P/kt_rows" ] && python3 tools/trace_rows_summary.py "$P/kt_rows" > "$O/kernel_trace_rows.csv"
for n in kt kt_full kt16 kt_gym kt_fused; do grep "^{" "$P/$n.log" > "$O/bench_${n}_under_rocprof.json"; done
python3 tools/pmc_summary.py 1048576 "$P/pmc_f" "$P/pmc_w" "$(basename "$O") step_kernel<1024,2,false,true,false,false> via bench.py --steps 20 --warmup 5 --regions 3 --no-legs" 16 > /dev/null
python3 tools/pmc_summary.py 16777216 "$P/pmc_f16" "$P/pmc_w16" "$(basename "$O") step_kernel<256,2,false,true,false,false> via bench.py --boards 16777216 --steps 10 --warmup 2 --regions 2 --no-legs" 16 > /dev/null
for n in pmc_f pmc_w pmc_f16 pmc_w16; do python3 - "$P/$n" "$O/${n}_step_kernel_counter_collection.csv" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "step_kernel" in r["Kernel_Name"]]
w = csv.DictWriter(open(sys.argv[2], "w"), fieldnames=rows[0].keys()); w.writeheader(); w.writerows(rows[-48:])
PY
done
python3 tools/pmc_sq_summary.py "$P/pmc_sq" 1048576 > "$O/pmc_sq_summary.csv"
python3 tools/pmc_sq_summary.py "$P/pmc_sq_fused" 1048576 > "$O/pmc_sq_fused_summary.csv"
python3 tools/pmc_sq_summary.py "$P/pmc_sq_rows" 1048576 > "$O/pmc_sq_rows_summary.csv"
for f in "$N"/bench_*.json rows.jsonl stepbench_262144.txt rowbench_1M.txt rowbench_64K.txt nproc.txt; do
  [ -f "$f" ] && cp "$f" "$O/" || cp "$N/$f" "$O/"
done
cp "$N/facade.json" "$O/facade_latency.json"; cp "$N/sweep.jsonl" "$O/sweep_boards.jsonl"; cp "$N/sweep_gym.jsonl" "$O/sweep_boards_gym.jsonl"
cp "$N/stepbench.txt" "$O/stepbench_1048576.txt"
ls "$O" | wc -l
