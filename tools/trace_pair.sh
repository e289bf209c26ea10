#!/bin/bash
# The un-profiled driver-shaped line and rocprofv3's kernel trace of the SAME command on the SAME box in ONE gpurun
# call (a trace average must not be held against a step timed on another box):
#   tools/trace_pair.sh gpurun_out/r04_pair
# Order: un-profiled, profiled, un-profiled again (the two un-profiled lines bracket the profiled pass, so a drift of
# the box shows).  Then tools/trace_summary.py over the trace.  The program after `--` is python3 itself.
set -u
out=$(readlink -f "$1"); R=$(readlink -f .)
mkdir -p "$out"
N="--steps 20 --warmup 5 --no-cpu-baseline --no-legs"
python3 "$R/bench.py" $N > "$out/bench_k20_before.json" 2> "$out/bench_k20_before.err"; echo "before rc=$?"
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 "$R/bench.py" $N > "$out/kt.log" 2>&1; echo "kt rc=$?" )
python3 "$R/bench.py" $N > "$out/bench_k20_after.json" 2> "$out/bench_k20_after.err"; echo "after rc=$?"
find "$out" -name "*.db" -delete 2>/dev/null
us() { python3 -c "import json,sys; print(json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])['ms_per_step']*1e3)" "$1"; }
reg() { python3 -c "import json,sys; print(json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])['regions'])" "$1"; }
b=$(us "$out/bench_k20_before.json"); a=$(us "$out/bench_k20_after.json")
python3 "$R/tools/trace_summary.py" "$out/kt" 20 5 "$(reg "$out/kt.log")" step_kernel "$b" > "$out/kernel_trace_timed_region.txt"
echo "un-profiled on this box: before $b us, after $a us per launch" >> "$out/kernel_trace_timed_region.txt"
grep "^{" "$out/kt.log" > "$out/bench_kt_under_rocprof.json"
cp "$out"/kt/*/*_kernel_stats.csv "$out/kernel_stats.csv" 2>/dev/null
hostname > "$out/box.txt"; rocminfo | grep -m2 -E "Marketing Name|gfx950" >> "$out/box.txt" 2>/dev/null
tail -12 "$out/kernel_trace_timed_region.txt"
