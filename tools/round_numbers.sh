#!/bin/bash
# Every number DESIGN.md §6 quotes, in one go on the GPU box:  tools/round_numbers.sh gpurun_out/r02_final
# The stepbench A/B lines want two extra builds under gpurun_tmp/ (git-ignored, they travel with gpurun):
#   gpurun_tmp/lib20/libqttt_hip.so     round 1's 20-byte-state kernel:  git show 637ea87:qtttgym_amd/csrc/qttt_kernels.hip > /tmp/r1.hip
#                                       && hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -o gpurun_tmp/lib20/libqttt_hip.so /tmp/r1.hip
#                                       (with round 1's include/qttt.h: git show 637ea87:include/qttt.h)
#   gpurun_tmp/libstamp/libqttt_hip.so  this tree with -DQTTT_DEBUG_STAMPS (per-wave timeline)
set -u
out=$1; mkdir -p "$out"
b() { name=$1; shift; python3 bench.py "$@" > "$out/$name.json" 2> "$out/$name.err"; echo "$name rc=$? $(cut -c1-120 "$out/$name.json")"; }
b bench_default
b bench_k20 --steps 20 --warmup 5
b bench_gym --mode gym --steps 200 --warmup 10 --no-cpu-baseline
b bench_gym_k20 --mode gym --steps 20 --warmup 5 --no-cpu-baseline
b bench_random --mode random --steps 200 --warmup 10 --no-cpu-baseline
b bench_policy --mode policy --steps 200 --warmup 10 --no-cpu-baseline
b bench_262144 --boards 262144 --steps 200 --warmup 10 --no-cpu-baseline
b bench_4096 --boards 4096 --steps 200 --warmup 10 --no-cpu-baseline
QTTT_DIST_BACKEND=gloo python3 bench.py --gpus 2 --boards 262144 --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_gpus2_gloo_one_card.json" 2> "$out/bench_gpus2.err"; echo "gpus2 rc=$?"
tools/sweep_boards.sh "$out/sweep.jsonl" > /dev/null 2> "$out/sweep.err"
tools/sweep_boards.sh "$out/sweep_gym.jsonl" --mode gym > /dev/null 2>> "$out/sweep.err"
python3 tools/bench_rows.py > "$out/rows.jsonl" 2> "$out/rows.err"
python3 tools/facade_latency.py > "$out/facade.json" 2> "$out/facade.err"
timeout -k 10 300 tools/stepbench 1048576 200 12 qtttgym_amd/libqttt_hip.so:0:0 gpurun_tmp/lib20/libqttt_hip.so:2:0 gpurun_tmp/libstamp/libqttt_hip.so:0:0 > "$out/stepbench.txt" 2>&1
timeout -k 10 120 tools/stepbench 262144 400 8 qtttgym_amd/libqttt_hip.so:0:0 gpurun_tmp/lib20/libqttt_hip.so:2:0 > "$out/stepbench_262144.txt" 2>&1
rocminfo > "$out/rocminfo.txt" 2>&1; nproc > "$out/nproc.txt"; grep -m1 "model name" /proc/cpuinfo >> "$out/nproc.txt"
ls "$out"
