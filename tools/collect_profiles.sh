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
[ -f "$P/kernel_trace_rows.csv" ] && cp "$P/kernel_trace_rows.csv" "$O/kernel_trace_rows.csv"
for n in kt kt_full kt16 kt_gym kt_fused; do grep "^{" "$P/$n.log" > "$O/bench_${n}_under_rocprof.json"; done
cp "$P/pmc_traffic.json" profiles/pmc_traffic.json
cp "$P"/pmc_*_step_kernel_counter_collection.csv "$P"/pmc_sq_summary.csv "$P"/pmc_sq_fused_summary.csv "$P"/pmc_sq_rows_summary.csv "$O/"
for f in "$N"/bench_*.json rows.jsonl stepbench_262144.txt rowbench_1M.txt rowbench_64K.txt nproc.txt; do
  [ -f "$f" ] && cp "$f" "$O/" || cp "$N/$f" "$O/"
done
cp "$N/facade.json" "$O/facade_latency.json"; cp "$N/sweep.jsonl" "$O/sweep_boards.jsonl"; cp "$N/sweep_gym.jsonl" "$O/sweep_boards_gym.jsonl"
cp "$N/stepbench.txt" "$O/stepbench_1048576.txt"
ls "$O" | wc -l
