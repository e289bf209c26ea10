#!/bin/bash
# bench.py --boards sweep: 1 M / 4 M / 16 M / 32 M boards per launch (state 16 / 64 / 256 / 512 MB at
# 16 B/board, plus the action and output streams: the last two cannot live in the 256 MB Infinity Cache).
# One JSON line per size.
#   tools/sweep_boards.sh OUT.jsonl [extra bench.py args]
set -e
out=$1; shift
: > "$out"
for b in 1048576 4194304 16777216 33554432; do
  k=200; [ $b -ge 16777216 ] && k=60
  python3 bench.py --boards $b --steps $k --warmup 10 --no-cpu-baseline --no-legs "$@" | tee -a "$out"
done
