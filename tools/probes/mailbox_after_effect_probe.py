#!/usr/bin/env python3
"""How long does the ~0.2 us that 1 M-board launches read above "alone" after a Board call (mailbox on) last, and does a
full synchronise after the wave has left remove it?  Launches enqueued from C (qttt_step_many), HIP events around launches
2..K, region kinds alternating on one box:
  alone        retire_mailbox(wait) ; steps
  board        Board.make_move ; steps                                   (the step entry retires the wave, no wait)
  board_sync   Board.make_move ; retire_mailbox(wait) ; torch.cuda.synchronize() ; steps
  side_nosync  a one-lane kernel on another stream, NOT synchronised ; steps
for K = 9 and K = 65."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qtttgym_amd import Board, QEvalClassic, VecEnv, retire_mailbox
n, T = 1 << 20, 96
env = VecEnv(n, seed=1, auto_reset=True)
acts = torch.empty((T, n, 2), dtype=torch.uint8, device="cuda")
for t in range(T):
    env.sample_actions(out=acts[t]); env.step_raw(acts[t])
torch.cuda.synchronize()
e1, e2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
MOVES = [(0, 1), (2, 3), (4, 5), (6, 7)]
st = {"b": Board(QEvalClassic()), "k": 0}
side = torch.cuda.Stream()
one = torch.zeros(1, device="cuda")
def board():
    if st["k"] == len(MOVES): st["b"], st["k"] = Board(QEvalClassic()), 0
    st["b"].make_move(MOVES[st["k"]]); st["k"] += 1
def pre(kind):
    if kind == "alone": retire_mailbox()
    elif kind == "board": board()
    elif kind == "board_sync": board(); retire_mailbox(); torch.cuda.synchronize()
    elif kind == "side_nosync":
        retire_mailbox()
        with torch.cuda.stream(side): one.add_(1.0)
def region(kind, K):
    pre(kind)
    env.step_many(acts[0:1])
    e1.record()
    env.step_many(acts[1:K])
    e2.record()
    torch.cuda.synchronize()
    return e1.elapsed_time(e2) * 1e3 / (K - 1)
KINDS = ["alone", "board", "board_sync", "side_nosync"]
res = {}
for K in (9, 65):
    for _ in range(10):
        for k in KINDS: region(k, K)
    out = {k: [] for k in KINDS}
    for _ in range(200 if K == 9 else 80):
        for k in KINDS: out[k].append(region(k, K))
    med = {k: sorted(v)[len(v) // 2] for k, v in out.items()}
    res["K=%d" % K] = {"rest_us_median": med, "delta_vs_alone": {k: med[k] - med["alone"] for k in KINDS}}
print(json.dumps(res))
