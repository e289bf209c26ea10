#!/bin/bash
# Every number DESIGN.md §6 quotes, in one go on the GPU box:  tools/round_numbers.sh gpurun_out/r03_final
# (un-profiled; the rocprofv3 passes are tools/profile_round.sh).  The stepbench A/B line wants one extra build under
# gpurun_tmp/ (git-ignored, travels with gpurun):
#   gpurun_tmp/libstamp/libqttt_hip.so  this tree with -DQTTT_DEBUG_STAMPS (per-wave timeline):
#       hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -mllvm -amdgpu-kernarg-preload-count=4 -DQTTT_DEBUG_STAMPS -Iinclude -o gpurun_tmp/libstamp/libqttt_hip.so qtttgym_amd/csrc/qttt_kernels.hip
set -u
out=$1; mkdir -p "$out"
b() { name=$1; shift; python3 bench.py "$@" > "$out/$name.json" 2> "$out/$name.err"; echo "$name rc=$? $(cut -c1-120 "$out/$name.json")"; }
b bench_default
b bench_k20 --steps 20 --warmup 5
N="--no-cpu-baseline --no-legs"
b bench_gym --mode gym --steps 200 --warmup 10 $N
b bench_random --mode random --steps 200 --warmup 10 $N
b bench_policy --mode policy --steps 200 --warmup 10 $N
b bench_random_fused --mode random-fused --steps 256 --warmup 10 $N
b bench_random_fused_262144 --mode random-fused --boards 262144 --steps 256 --warmup 10 $N
b bench_262144 --boards 262144 --steps 200 --warmup 10 $N
b bench_4096 --boards 4096 --steps 200 --warmup 10 $N
b bench_total_2M_one_gpu --total-boards 2097152 --steps 100 --warmup 10 $N
b bench_total_2M_one_gpu_fused --total-boards 2097152 --steps 128 --warmup 10 --mode random-fused $N
QTTT_DIST_BACKEND=gloo python3 bench.py --gpus 2 --total-boards 524288 --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_gpus2_gloo_one_card_strong.json" 2> "$out/bench_gpus2.err"; echo "gpus2 rc=$?"
tools/sweep_boards.sh "$out/sweep.jsonl" --no-legs > /dev/null 2> "$out/sweep.err"
tools/sweep_boards.sh "$out/sweep_gym.jsonl" --mode gym --no-legs > /dev/null 2>> "$out/sweep.err"
python3 tools/bench_rows.py > "$out/rows.jsonl" 2> "$out/rows.err"
python3 tools/facade_latency.py > "$out/facade.json" 2> "$out/facade.err"
timeout -k 10 300 tools/stepbench 1048576 200 12 qtttgym_amd/libqttt_hip.so:0:0 gpurun_tmp/libstamp/libqttt_hip.so:0:0 > "$out/stepbench.txt" 2>&1
timeout -k 10 120 tools/stepbench 262144 400 8 qtttgym_amd/libqttt_hip.so:0:0 > "$out/stepbench_262144.txt" 2>&1
timeout -k 10 120 tools/rowbench 1048576 5 50 9 > "$out/rowbench_1M.txt" 2>&1
timeout -k 10 120 tools/rowbench 65536 5 50 9 > "$out/rowbench_64K.txt" 2>&1
rocminfo > "$out/rocminfo.txt" 2>&1; nproc > "$out/nproc.txt"; grep -m1 "model name" /proc/cpuinfo >> "$out/nproc.txt"
ls "$out"
