#!/usr/bin/env python3
"""Randomised soak of the step path against the oracle, beyond what the test suite pins: random batch
sizes (odd, around the launch-shape thresholds), board offsets (incl. across 2^32), launch shapes,
explicit / hashed collapse bits, auto-reset on / off, adversarial actions (out of range, same square,
classical squares), the fused step + observation kernel, export and check_win.  Prints one line per
case; exits non-zero on the first mismatch.     python3 tests/soak_parity.py [cases] [seed] [rows]
(`rows`: the SURVEY §8(f) rows instead — node_info, expand, rollout, encode on boards frozen at random depths.)
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


def one_case(rng, k):
    L = _native.lib()
    sizes = [1, 2, 3, 63, 64, 65, 127, 129, 511, 1000, 4097, 65535, 262143, 458751, 458753, 524289, 917503, 917505,
             1048577, 1572865, 2097153]
    n = int(rng.choice(sizes)) if rng.random() < 0.7 else int(rng.integers(1, 300000))
    off = int(rng.choice([0, 7, 2**32 - n // 2 - 1, 2**33 + 12345, int(rng.integers(0, 2**40))]))
    shape = [(0, 0), (1, 256), (2, 256), (1, 512), (2, 512), (4, 512), (1, 1024), (2, 1024)][int(rng.integers(0, 8))]
    auto_reset = bool(rng.integers(0, 2))
    use_bits = bool(rng.integers(0, 2))
    adversarial = rng.random() < 0.4
    observe = bool(rng.integers(0, 2))
    # with the observation: the zero-copy form, the DEFAULT step() (fresh tensors, one kernel) checked at once, or the
    # default step() with every step's tensors KEPT and checked after the last step
    obs_mode = ["raw", "default", "default_keep"][int(rng.integers(0, 3))] if observe else None
    kept = []
    seed = int(rng.integers(0, 2**62))
    steps = int(rng.integers(3, 14)) if n < 600000 else int(rng.integers(3, 7))
    per_call = bool(rng.integers(0, 2))               # the launch shape in the calls' own flags, or the process default
    assert L.qttt_set_tuning(*((0, 0) if per_call else shape)) == 0
    env = VecEnv(n, seed=seed, auto_reset=auto_reset, board_offset=off,
                 launch_shape=shape if per_call and shape != (0, 0) else None)
    ob = oracle.OracleBoards(n)
    fused = int(rng.integers(0, 6)) if (not adversarial and not use_bits) else 0   # a run of fused random plies first
    if fused:
        acts = torch.empty((fused, n, 2), dtype=torch.uint8, device=env.device)
        rr = torch.empty((fused, n), dtype=torch.float32, device=env.device)
        tt = torch.empty((fused, n), dtype=torch.bool, device=env.device)
        env.step_random_many(fused, actions_out=acts, reward=rr, terminated=tt)
        for t in range(fused):
            a_or = ob.sample_actions(seed, t, off, auto_reset)
            r_or, t_or = ob.step(a_or, None, seed, t, off, auto_reset)
            assert np.array_equal(npy(acts[t]), a_or), ("fused policy", t)
            assert np.array_equal(npy(rr[t]).view(np.uint32), r_or.view(np.uint32)), ("fused reward", t)
            assert np.array_equal(npy(tt[t]).astype(np.uint8), t_or), ("fused terminated", t)
    for t in range(fused, fused + steps):
        a_or = ob.sample_actions(seed, t, off, auto_reset)
        a = env.sample_actions()
        assert np.array_equal(npy(a), a_or), ("policy", t)
        if adversarial:                                   # overwrite a random third of the actions with junk
            junk = rng.integers(0, 12, size=(n, 2)).astype(np.uint8)
            junk[rng.random(n) < 0.1] = 255
            sel = rng.random(n) < 0.33
            a_or = np.where(sel[:, None], junk, a_or).astype(np.uint8)
            a = torch.from_numpy(a_or).to(env.device)
        bits_np = rng.integers(0, 2, size=n).astype(np.uint8) if use_bits else None
        bits = torch.from_numpy(bits_np).to(env.device) if use_bits else None
        if observe and obs_mode == "raw":
            o, reward, term = env.step_observe_raw(a, bits)
        elif observe:
            o, reward, term, trunc, info = env.step(a, bits)
            assert info == {} and trunc.shape == (n,)
        elif not adversarial and not use_bits and t % 3 == 2:      # policy + step in one kernel: the same step
            played = torch.empty_like(a)
            reward, term = env.step_random(actions_out=played)
            assert torch.equal(played, a), ("step_random actions", t)
        else:
            reward, term = env.step_raw(a, bits)
        r_or, t_or = ob.step(a_or, bits_np, seed, t, off, auto_reset)
        assert np.array_equal(npy(reward).view(np.uint32), r_or.view(np.uint32)), ("reward", t)
        assert np.array_equal(npy(term).astype(np.uint8), t_or), ("terminated", t)
        if observe:
            classical, q1, l1, q2, l2, turn = ob.observe()
            refs = (("classical", classical), ("q_states_p1", q1), ("q_states_p1_len", l1),
                    ("q_states_p2", q2), ("q_states_p2_len", l2), ("turn", turn))
            for name, ref in refs:
                assert np.array_equal(npy(o[name]), ref), (name, t)
            if obs_mode == "default_keep":
                kept.append((t, o, reward, term, refs, r_or, t_or))
    for t, o, reward, term, refs, r_or, t_or in kept:        # nothing the caller kept was written again
        for name, ref in refs:
            assert np.array_equal(npy(o[name]), ref), ("kept", name, t)
        assert np.array_equal(npy(reward).view(np.uint32), r_or.view(np.uint32)), ("kept reward", t)
        assert np.array_equal(npy(term).astype(np.uint8), t_or), ("kept terminated", t)
    ex = {kk: npy(v) for kk, v in env.export_boards().items()}
    assert np.array_equal(ex["board"], ob.board) and np.array_equal(ex["moves"], ob.moves)
    assert np.array_equal(ex["n_moves"], ob.n_moves) and np.array_equal(ex["qmask"].view(np.uint16), ob.qmask)
    assert np.array_equal(ex["n_q"], ob.n_q) and np.array_equal(npy(env.turn()), ob.n_moves)
    p1, p2 = env.check_win()
    w1, w2 = ob.check_win()
    assert np.array_equal(npy(p1), w1) and np.array_equal(npy(p2), w2)
    return dict(case=k, n=n, offset=off, shape=shape, per_call=per_call, auto_reset=auto_reset, bits=use_bits,
                adversarial=adversarial, observe=obs_mode, fused_plies=fused, steps=steps)


def rows_case(rng, k):
    """SURVEY §8(f) rows on boards frozen at random depths: node_info, expand, rollout, encode."""
    n = int(rng.choice([1, 2, 63, 65, 255, 257, 511, 513, 1000, 4099, 20001]))
    off = int(rng.choice([0, 5, 2**32 - n // 2 - 1, int(rng.integers(0, 2**40))]))
    seed = int(rng.integers(0, 2**62))
    env = VecEnv(n, seed=seed, board_offset=off)
    ob = oracle.OracleBoards(n)
    depth = rng.integers(0, 10, n)
    for t in range(9):
        a = npy(env.sample_actions())
        a[depth <= t] = 255
        env.step_raw(torch.from_numpy(a).to(env.device))
        ob.step(a, None, seed, t, off)
    winner, terminal, legal, key = oracle.node_info(ob)
    info = env.node_info()
    assert np.array_equal(npy(info["winner"]), winner) and np.array_equal(npy(info["terminal"]).astype(np.uint8), terminal)
    assert np.array_equal(npy(info["legal"]).view(np.uint64), legal) and np.array_equal(npy(info["key"]), key)
    vec, mask = env.encode()
    assert np.array_equal(npy(vec), oracle.to_vector(ob).astype(np.float32))
    assert np.array_equal(npy(mask), (legal[:, None] >> np.arange(36, dtype=np.uint64)[None, :] & np.uint64(1)).astype(bool))
    act = rng.integers(0, 40, n).astype(np.uint8)           # 36..39: not an action -> no children
    out = env.expand(torch.from_numpy(act))
    nch, kids, w2, t2, l2, k2 = oracle.expand(ob, act)
    assert np.array_equal(npy(out["n_children"]), nch)
    for c, child in enumerate((out["child0"], out["child1"])):
        sel = nch > c
        ex = {kk: npy(v) for kk, v in child.export_boards().items()}
        assert np.array_equal(ex["board"][sel], kids[c].board[sel]) and np.array_equal(ex["moves"][sel], kids[c].moves[sel])
        assert np.array_equal(ex["qmask"].view(np.uint16)[sel], kids[c].qmask[sel])
        assert np.array_equal(npy(out["winner"])[sel, c], w2[sel, c]) and np.array_equal(npy(out["key"])[sel, c], k2[sel, c])
        assert np.array_equal(npy(out["legal"]).view(np.uint64)[sel, c], l2[sel, c])
        assert np.array_equal(npy(out["terminal"])[sel, c].astype(np.uint8), t2[sel, c])
    s0 = int(rng.integers(0, 1000))
    result, plies, final = env.rollout(step_idx0=s0, return_final=True)
    res_o, plies_o, fin_o = oracle.rollout(ob, seed, s0, off)
    assert np.array_equal(npy(result), res_o) and np.array_equal(npy(plies), plies_o)
    exf = {kk: npy(v) for kk, v in final.export_boards().items()}
    assert np.array_equal(exf["board"], fin_o.board) and np.array_equal(exf["moves"], fin_o.moves)
    # ---- round 4: the packed state is canonical (import of the export is the state, bit for bit) ...
    exb = env.export_boards()
    back = VecEnv(n, seed=seed, board_offset=off)
    back.import_boards(exb["moves"], exb["n_moves"], exb["board"], exb["qmask"], exb["n_q"])
    pl = lambda e: e.state.view(torch.int64).view(2, -1)[:, :n]
    assert torch.equal(pl(back), pl(env)), "import(export(s)) != s"
    # ... so the native key is equal exactly where the oracle's CPython hash is, for the parents and all children
    nat = [npy(info["state_key"])] + [npy(out["state_key"])[nch > c, c] for c in range(2)]
    pyk = [key] + [k2[nch > c, c] for c in range(2)]
    nat, pyk = np.concatenate(nat), np.concatenate(pyk)
    assert len(np.unique(np.stack([nat, pyk], 1), axis=0)) == len(np.unique(nat)) == len(np.unique(pyk)), "key partition"
    # ---- qttt_expand_rollout (either mapping, by size) against the oracle's expand + playout loop
    S = int(rng.choice([1, 2, 5, 10])) if n <= 4099 else 1
    xr = env.expand_rollout(torch.from_numpy(act), n_sims=S, step_idx0=s0, with_result=True, python_key=True)
    assert np.array_equal(npy(xr["n_children"]), nch) and torch.equal(xr["state_key"], out["state_key"])
    assert torch.equal(xr["key"], out["key"]) and torch.equal(xr["winner"], out["winner"]) and torch.equal(xr["legal"], out["legal"])
    res, vs = npy(xr["result"]), npy(xr["value_sum"])
    for c in range(2):
        sel = nch > c
        tot = np.zeros(n, dtype=np.int64)
        for sim in range(S):
            r_o, _, _ = oracle.rollout(kids[c], seed, s0 + (c * S + sim) * 16, off)
            assert np.array_equal(res[sel, c, sim], r_o[sel]), ("expand_rollout result", c, sim)
            tot += r_o
        # leaf.turn (mcts.py:174,243): True after an even number of real moves (an autofill move is not one)
        mv, nm = kids[c].moves, kids[c].n_moves.astype(np.int64)
        last = mv[np.arange(n), np.maximum(nm - 1, 0)]
        real = nm - ((nm > 0) & (last[:, 0] == last[:, 1]))
        assert np.array_equal(vs[sel, c], np.where(real % 2 == 0, tot, -tot)[sel]), ("value_sum", c)
        assert (res[~sel, c] == 0).all() and (vs[~sel, c] == 0).all()
    return dict(case=k, rows=True, n=n, offset=off, sims=S)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    rows = len(sys.argv) > 3 and sys.argv[3] == "rows"
    t0 = time.time()
    for k in range(cases):
        info = rows_case(rng, k) if rows else one_case(rng, k)
        print("ok", info, "%.0f s" % (time.time() - t0), flush=True)
    _native.lib().qttt_set_tuning(0, 0)
    print("soak ok: %d cases" % cases)


if __name__ == "__main__":
    main()
