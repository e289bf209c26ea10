#!/usr/bin/env python3
"""Diagnostic: replays the golden traces REPS times in one process through VecEnv.step (fresh
H2D action/bit tensors every step, like the test) and counts any mismatch, with details.
    QTTT_LIB_PATH=<other build> python tools/flake_probe.py [REPS]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from qtttgym_amd import VecEnv, _native  # noqa: E402


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    with np.load(os.path.join(ROOT, "tests", "golden", "step_traces.npz")) as d:
        g = {k: d[k] for k in d.files}
    acts, bits = g["actions"], g["bits"]
    E, T = bits.shape
    bad = 0
    for r in range(reps):
        env = VecEnv(E)
        for t in range(T):
            obs, reward, term, _, _ = env.step(torch.from_numpy(acts[:, t].copy()), torch.from_numpy(bits[:, t].copy()))
            ex = {k: v.cpu().numpy() for k, v in env.export_boards().items()}
            checks = [("board", ex["board"], g["board"][:, t]), ("n_moves", ex["n_moves"], g["n_moves"][:, t]),
                      ("moves", ex["moves"], g["moves"][:, t]),
                      ("reward", reward.cpu().numpy().view(np.uint32), g["reward"][:, t].astype(np.float32).view(np.uint32)),
                      ("terminated", term.cpu().numpy().astype(np.uint8), g["terminated"][:, t]),
                      ("obs.classical", obs["classical"].cpu().numpy(), g["board"][:, t]),
                      ("obs.q_p1", obs["q_states_p1"].cpu().numpy(), g["q_p1"][:, t]),
                      ("obs.q_p2", obs["q_states_p2"].cpu().numpy(), g["q_p2"][:, t]),
                      ("obs.turn", obs["turn"].cpu().numpy(), g["turn"][:, t])]
            for name, got, want in checks:
                if not np.array_equal(got, want):
                    idx = np.argwhere(got != want)
                    bad += 1
                    print("rep %d step %d %s: %d mismatches, first %s got %s want %s" % (
                        r, t, name, len(idx), idx[0].tolist(), got[tuple(idx[0])], want[tuple(idx[0])]), flush=True)
                    if bad > 20:
                        print("giving up"); return 1
    print("%s: %d reps x %d steps, %d mismatching checks" % (os.path.basename(os.path.dirname(_native.LIB_PATH)) or _native.LIB_PATH, reps, T, bad))
    return 0 if bad == 0 else 1


if __name__ == "__main__":
    sys.exit(main())
