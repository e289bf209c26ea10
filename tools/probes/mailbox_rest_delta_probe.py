#!/usr/bin/env python3
"""Why do 1 M-board launches 2..9 after a Board.make_move read ~0.1 - 0.2 us longer than alone (mailbox on), when the wave has
been asked to leave before launch 1?  Region variants, alternating region by region on one box, HIP events around launches 2..9:
  alone            retire_mailbox(wait) ; steps
  board            Board.make_move ; steps                      (the step entry retires the wave, no wait)
  board_retire     Board.make_move ; retire_mailbox(wait) ; steps   (the wave has SAID it left before launch 1)
  board_sleep      Board.make_move ; busy-wait 60 us ; steps        (the wave left by its idle exit long before)
  small_kernel     a 1-lane kernel on another stream + synchronise of that stream ; steps   (no mailbox involved at all)
If the last three read like `board`, the cost is not the wave's CU slot but what follows a small kernel's completion on
another stream (the runtime's completion handling beside the launching thread)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qtttgym_amd import Board, QEvalClassic, VecEnv, retire_mailbox
n, T, K = 1 << 20, 64, 9
env = VecEnv(n, seed=1, auto_reset=True)
acts = torch.empty((T, n, 2), dtype=torch.uint8, device="cuda")
for t in range(T):
    env.sample_actions(out=acts[t]); env.step_raw(acts[t])
torch.cuda.synchronize()
e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
MOVES = [(0, 1), (2, 3), (4, 5), (6, 7)]
st = {"b": Board(QEvalClassic()), "k": 0, "r": 0}
side = torch.cuda.Stream()
one = torch.zeros(1, device="cuda")
def board():
    if st["k"] == len(MOVES): st["b"], st["k"] = Board(QEvalClassic()), 0
    st["b"].make_move(MOVES[st["k"]]); st["k"] += 1
def pre(kind):
    if kind == "alone": retire_mailbox()
    elif kind == "board": board()
    elif kind == "board_retire": board(); retire_mailbox()
    elif kind == "board_sleep":
        board(); t0 = time.perf_counter_ns()
        while time.perf_counter_ns() - t0 < 60000: pass
    elif kind == "small_kernel":
        retire_mailbox()
        with torch.cuda.stream(side): one.add_(1.0)
        side.synchronize()
def region(kind):
    pre(kind)
    r = st["r"]; st["r"] += 1
    env.step_raw(acts[(r * K) % T])
    e1.record()
    for t in range(1, K): env.step_raw(acts[(r * K + t) % T])
    e2.record()
    torch.cuda.synchronize()
    return e1.elapsed_time(e2) * 1e3 / (K - 1)
KINDS = ["alone", "board", "board_retire", "board_sleep", "small_kernel"]
for _ in range(30):
    for k in KINDS: region(k)
out = {k: [] for k in KINDS}
for _ in range(400):
    for k in KINDS: out[k].append(region(k))
med = {k: sorted(v)[len(v) // 2] for k, v in out.items()}
print(json.dumps({"rest_us_median": med, "delta_vs_alone": {k: med[k] - med["alone"] for k in KINDS},
                  "mailbox_us": os.environ.get("QTTT_BOARD_MAILBOX_US", "20 (default)")}))
