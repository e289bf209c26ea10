#!/usr/bin/env python3
"""Exhaustive (not sampled) parity of the kernels beside the step, against the oracle, over EVERY position reachable from
the empty board in <= D plies (every legal action, both branches of every collapse; D = 4: 1 906 489 positions):
observe, check_win, export, node_info (winner, terminal, legal mask, CPython key), encode, and the fused playout
(result, plies) from each of them.  The positions are enumerated on the device with qttt_expand and rebuilt for the
oracle from their exported attributes (tests/test_rows_tiles_and_keys_gpu.py checks that enumeration against the oracle's own to
depth 3 and every step transition from it).      python3 tests/exhaustive_parity.py [D=4]
(Test infrastructure, kept under tests/: it checks against oracle/ like the suite does; not collected by pytest.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import oracle  # noqa: E402
from qtttgym_amd import VecEnv, _native  # noqa: E402


def npy(t):
    return t.cpu().numpy()


def cat_states(parts, seed=0):
    cols = [e.state.view(torch.int64).view(2, -1)[:, idx] for e, idx in parts]
    planes = torch.cat(cols, dim=1)
    m = planes.shape[1]
    st = torch.zeros(int(_native.lib().qttt_state_bytes(m)), dtype=torch.uint8, device=planes.device)
    st.view(torch.int64).view(2, -1)[:, :m] = planes
    return VecEnv.from_state(st, m, seed=seed)


def next_level(frontier):
    n = frontier.num_envs
    rep = frontier.take(torch.arange(n, device="cuda").repeat_interleave(36))
    act = torch.arange(36, dtype=torch.uint8, device="cuda").repeat(n)
    out = rep.expand(act, python_key=False)
    nch = out["n_children"]
    return cat_states([(out["child0"], (nch >= 1).nonzero().flatten()), (out["child1"], (nch == 2).nonzero().flatten())])


def main():
    D = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    t0 = time.time()
    levels = [VecEnv(1)]
    for _ in range(D):
        levels.append(next_level(levels[-1]))
    seed, s0 = 99, 40
    pos = cat_states([(e, torch.arange(e.num_envs, device="cuda")) for e in levels], seed=seed)
    n = pos.num_envs
    print("positions to depth %d: %d (%s per depth)" % (D, n, [e.num_envs for e in levels]), flush=True)
    ex = {k: npy(v) for k, v in pos.export_boards().items()}
    ob = oracle.boards_from_arrays(ex["board"], ex["moves"], ex["n_moves"], ex["qmask"].view(np.uint16), ex["n_q"])
    # ---- observe / check_win
    classical, q1, l1, q2, l2, turn = ob.observe()
    o = pos.observ()
    for name, ref in (("classical", classical), ("q_states_p1", q1), ("q_states_p1_len", l1), ("q_states_p2", q2),
                      ("q_states_p2_len", l2), ("turn", turn)):
        assert np.array_equal(npy(o[name]), ref), name
    print("observe ok  %.0f s" % (time.time() - t0), flush=True)
    p1, p2 = pos.check_win()
    w1, w2 = ob.check_win()
    assert np.array_equal(npy(p1), w1) and np.array_equal(npy(p2), w2)
    print("check_win ok  %.0f s" % (time.time() - t0), flush=True)
    # ---- node_info / encode
    winner, terminal, legal, key = oracle.node_info(ob)
    info = pos.node_info()
    assert np.array_equal(npy(info["winner"]), winner) and np.array_equal(npy(info["terminal"]).astype(np.uint8), terminal)
    assert np.array_equal(npy(info["legal"]).view(np.uint64), legal) and np.array_equal(npy(info["key"]), key)
    assert len(np.unique(npy(info["state_key"]))) == n == len(np.unique(key))
    print("node_info ok (and %d distinct native keys)  %.0f s" % (n, time.time() - t0), flush=True)
    CH = 1 << 18
    for a in range(0, n, CH):                                # encode: 720 B per board, in chunks
        sub = pos.take(torch.arange(a, min(a + CH, n), device="cuda"))
        vec, mask = sub.encode()
        sl = oracle.OracleBoards(sub.num_envs)
        sl.b[:] = ob.b[a:a + CH]
        assert np.array_equal(npy(vec), oracle.to_vector(sl).astype(np.float32)), a
        lm = legal[a:a + CH]
        assert np.array_equal(npy(mask), (lm[:, None] >> np.arange(36, dtype=np.uint64)[None, :] & np.uint64(1)).astype(bool)), a
    print("encode ok  %.0f s" % (time.time() - t0), flush=True)
    # ---- the fused playout from every position
    result, plies = pos.rollout(step_idx0=s0)
    res_o, plies_o, _ = oracle.rollout(ob, seed, s0, 0)
    assert np.array_equal(npy(result), res_o) and np.array_equal(npy(plies), plies_o)
    print("rollout ok (mean %.2f plies)  %.0f s" % (float(plies_o.mean()), time.time() - t0), flush=True)
    print("exhaustive parity ok: %d positions to depth %d" % (n, D))


if __name__ == "__main__":
    main()
