#!/usr/bin/env python3
"""Experiment (not the judged bench): the K = 20 timed step launches of bench.py's replay region enqueued eagerly from C
(qttt_step_many) against the same launches captured once in a hipGraph and replayed, per batch size.  Round 4, one
MI355X (profiles/r04/graph_vs_eager_step.txt): 1 M boards 7.15 - 7.21 us eager / 7.50 - 7.84 graph (a replay's own cost
does not amortise over 20 launches of 7 us), 262 144 boards 3.85 - 3.92 / 3.90 - 3.91, 4 096 boards 3.58 - 3.78 / 2.65.
So the headline stays on the eager C loop; graphs pay for small batches only (VecEnv.capture)."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import torch
from qtttgym_amd import VecEnv
from qtttgym_amd import recommended_env  # noqa: E402
recommended_env(apply=True)   # HIP_FORCE_DEV_KERNARG=1 etc., before the first HIP call
dev = torch.device("cuda", 0)
for B in (1 << 20, 262144, 4096):
    K, W = 20, 5
    env = VecEnv(B, device=dev, seed=1, auto_reset=True)
    T = K + W
    actions = torch.empty((T, B, 2), dtype=torch.uint8, device=dev)
    for t in range(T):
        env.sample_actions(out=actions[t]); env.step_raw(actions[t])
    torch.cuda.synchronize()
    final = env.state.clone()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    def preroll():
        env.reset_raw(); env.step_many(actions[:W])
    # eager
    def region_eager():
        preroll(); e0.record(); env.step_many(actions[W:]); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / K
    # graph of the K timed launches
    preroll(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); side = torch.cuda.Stream(device=dev); side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            env.step_many(actions[W:])
    torch.cuda.current_stream(dev).wait_stream(side); torch.cuda.synchronize()
    def region_graph():
        preroll(); e0.record(); g.replay(); e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / K
    for name, fn in (("eager", region_eager), ("graph", region_graph), ("eager", region_eager), ("graph", region_graph)):
        us = sorted(fn() for _ in range(300))
        ok = torch.equal(env.state, final)
        print(B, name, "median %.3f us  min %.3f  replay_ok %s" % (us[len(us)//2], us[0], ok), flush=True)
