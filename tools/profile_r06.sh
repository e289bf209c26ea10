#!/bin/bash
# rocprofv3 evidence of round 6, run ON THE GPU BOX from the repo root:  tools/profile_r06.sh gpurun_out/prof_r06
#   pair          tools/trace_pair.sh: the driver-shaped command un-profiled / under --kernel-trace --stats / un-profiled
#   kt_gymdef     --kernel-trace --stats over `bench.py --mode gym-default`: the DEFAULT VecEnv.step() — the dispatch
#                 list of the timed regions must hold the observation-writing step kernel and nothing else
#   pmc_f / pmc_w FETCH_SIZE / WRITE_SIZE (separate passes) of the same command: 69 algorithmic bytes per board-step
# Counters never share a run with --kernel-trace/--stats; the program after `--` is python3 itself.
set -u
out=$(readlink -f "$1"); R=$(readlink -f .)
mkdir -p "$out"
"$R/tools/trace_pair.sh" "$out/pair"
cd /tmp && export TMPDIR=/tmp
N="--steps 20 --warmup 5 --no-cpu-baseline --no-legs --mode gym-default"
run() { name=$1; shift; echo "== $name"; "$@" > "$out/$name.log" 2>&1; echo "rc=$?" >> "$out/$name.log"; tail -1 "$out/$name.log"; }
run kt_gymdef rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt_gymdef" -- python3 "$R/bench.py" $N
run pmc_f     rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_f" -- python3 "$R/bench.py" $N --regions 3
run pmc_w     rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_w" -- python3 "$R/bench.py" $N --regions 3
find "$out" -name "*.db" -delete 2>/dev/null
find "$out" -name "*agent_info.csv" -delete 2>/dev/null
cp "$out"/kt_gymdef/*/*_kernel_stats.csv "$out/kernel_stats_gym_default.csv" 2>/dev/null
grep "^{" "$out/kt_gymdef.log" > "$out/bench_kt_gym_default_under_rocprof.json"
# every dispatch between the first and the last observation-writing step kernel, by kernel name: what a default step() enqueues
python3 - "$out/kt_gymdef" > "$out/kernel_trace_gym_default_dispatches.txt" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
import re
OBS = re.compile(r"step_kernel<\d+, \d+, (?:true|false), (?:true|false), (?:true|false), true, (?:true|false)>")   # template argument 6 = OBS
is_obs = lambda r: OBS.search(r["Kernel_Name"]) is not None
short = lambda k: re.sub(r"\(.*$", "", k.replace("void ", "").replace("(anonymous namespace)::", ""))
obs = [i for i, r in enumerate(rows) if is_obs(r)]
print("kernel trace of: bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-legs --mode gym-default  (%d dispatches in all)" % len(rows))
if obs:
    span = rows[obs[0]:obs[-1] + 1]
    c = collections.Counter(r["Kernel_Name"] for r in span)
    print("dispatches from the first to the last observation-writing step kernel (the timed regions and what lies between them):")
    for k, v in c.most_common():
        d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in span if r["Kernel_Name"] == k]
        print("  %6d x %-70s avg %8.1f ns" % (v, short(k)[:70], sum(d) / len(d)))
    # inside one timed region: K = 20 consecutive default steps -> consecutive dispatches of the step kernel only
    runs, cur = [], 0
    for r in span:
        if is_obs(r): cur += 1
        else:
            if cur: runs.append(cur)
            cur = 0
    if cur: runs.append(cur)
    print("runs of consecutive observation-writing step kernels (K = 20 timed steps per region, nothing in between): %s ... (%d runs, %d of length >= 20)"
          % (runs[:12], len(runs), sum(1 for x in runs if x >= 20)))
PY
PMC_KERNEL_FILTER="true, false>" python3 "$R/tools/pmc_summary.py" 1048576 "$out/pmc_f" "$out/pmc_w" "r06 step_kernel<1024,2,false,true,false,true> through the DEFAULT VecEnv.step(): bench.py --mode gym-default --steps 20 --warmup 5 --regions 3 --no-legs" 16 30 gym-default > "$out/pmc_traffic_gym_default.json"
cp "$R/profiles/pmc_traffic.json" "$out/pmc_traffic.json"
find "$out" -name "*kernel_trace.csv" -path "*kt_gymdef*" -delete 2>/dev/null
find "$out" -name "*counter_collection.csv" -delete 2>/dev/null
du -sh "$out"
cat "$out/kernel_trace_gym_default_dispatches.txt"
