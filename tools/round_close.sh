#!/bin/bash
# The round's closing sequence on ONE GPU box, one gpurun call:  tools/round_close.sh r05
#   smoke(), the whole `-m gpu` suite, then the three evidence scripts (rocprofv3 passes, un-profiled numbers, same-box
#   trace pair).  tools/collect_profiles.sh folds the outputs into profiles/rNN afterwards (in the build container).
set -u
tag=${1:-rXX}
mkdir -p gpurun_out/${tag}_close
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_close/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/${tag}_close/smoke.log
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_close/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/${tag}_close/pytest_gpu.log
rm -rf gpurun_out/prof_${tag} gpurun_out/${tag}_final gpurun_out/${tag}_pair
tools/profile_round.sh gpurun_out/prof_${tag} > gpurun_out/prof_${tag}.log 2>&1; tail -2 gpurun_out/prof_${tag}.log
tools/round_numbers.sh gpurun_out/${tag}_final > gpurun_out/${tag}_final.log 2>&1; tail -2 gpurun_out/${tag}_final.log
tools/trace_pair.sh gpurun_out/${tag}_pair > gpurun_out/${tag}_pair.log 2>&1; tail -3 gpurun_out/${tag}_pair.log
