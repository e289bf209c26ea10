"""SURVEY.md §8(f) rows on the GPU: qttt_expand / qttt_node_info / qttt_rollout / qttt_encode
against the golden traces from the reference's mcts.py and against the oracle at BASELINE
config 5's batch (65 536 boards)."""
import os

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _np(t):
    return t.cpu().numpy()


@pytest.fixture(scope="module")
def gx():
    with np.load(os.path.join(ROOT, "tests", "golden", "expand_traces.npz")) as d:
        return {k: d[k] for k in d.files}


def mask_to_bits(m):
    return (m.astype(np.uint64) << np.arange(36, dtype=np.uint64)).sum(axis=-1).astype(np.uint64)


def env_from(board, moves, n_moves, qmask, n_q):
    from qtttgym_amd import VecEnv
    env = VecEnv(len(n_moves))
    env.import_boards(torch.from_numpy(moves.copy()), torch.from_numpy(n_moves.copy()),
                      torch.from_numpy(board.copy()), torch.from_numpy(qmask.view(np.int16).copy()),
                      torch.from_numpy(n_q.copy()))
    return env


def assert_export_equals(env, board, moves, n_moves, qmask, n_q, sel=None):
    ex = {k: _np(v) for k, v in env.export_boards().items()}
    sel = slice(None) if sel is None else sel
    assert np.array_equal(ex["board"][sel], board[sel])
    assert np.array_equal(ex["moves"][sel], moves[sel])
    assert np.array_equal(ex["n_moves"][sel], n_moves[sel])
    assert np.array_equal(ex["qmask"].view(np.uint16)[sel], qmask[sel])
    assert np.array_equal(ex["n_q"][sel], n_q[sel])


def test_node_info_and_encode_match_reference(gx):
    env = env_from(gx["p_board"], gx["p_moves"], gx["p_n_moves"], gx["p_qmask"], gx["p_n_q"])
    assert_export_equals(env, gx["p_board"], gx["p_moves"], gx["p_n_moves"], gx["p_qmask"], gx["p_n_q"])
    info = env.node_info()
    assert np.array_equal(_np(info["winner"]), gx["p_winner"])
    assert np.array_equal(_np(info["terminal"]), gx["p_terminal"])
    assert np.array_equal(_np(info["legal"]).view(np.uint64), mask_to_bits(gx["p_mask"]))
    assert np.array_equal(_np(info["key"]), gx["p_hash"])          # Python's hash(), bit for bit
    # out= reuses the caller's buffers (no allocation per call)
    for v in info.values():
        v.zero_()
    again = env.node_info(out=info)
    assert again is info and np.array_equal(_np(info["key"]), gx["p_hash"])
    assert np.array_equal(_np(info["winner"]), gx["p_winner"])
    pair = env.check_win()
    ref = (_np(pair[0]).copy(), _np(pair[1]).copy())
    pair[0].zero_(); pair[1].zero_()
    back = env.check_win(out=pair)
    assert back[0] is pair[0] and np.array_equal(_np(pair[0]), ref[0]) and np.array_equal(_np(pair[1]), ref[1])
    with pytest.raises(ValueError):
        env.check_win(out=(pair[0][:-1], pair[1]))
    vec, mask = env.encode()
    # reference builds float64 (mcts.py:67-85); values are 0, 1, 1/3: exact after rounding to f32
    assert np.array_equal(_np(vec), gx["p_vector"].astype(np.float32))
    assert np.array_equal(_np(mask), gx["p_mask"])


def test_expand_matches_reference_step(gx):
    idx = gx["parent"]
    env = env_from(gx["p_board"][idx], gx["p_moves"][idx], gx["p_n_moves"][idx], gx["p_qmask"][idx],
                   gx["p_n_q"][idx])
    out = env.expand(torch.from_numpy(gx["action"].copy()))
    nch = _np(out["n_children"])
    assert np.array_equal(nch, gx["n_children"])
    for c, child in enumerate((out["child0"], out["child1"])):
        sel = nch > c
        assert_export_equals(child, gx["c_board"][:, c], gx["c_moves"][:, c], gx["c_n_moves"][:, c],
                             gx["c_qmask"][:, c], gx["c_n_q"][:, c], sel)
        assert np.array_equal(_np(out["winner"])[sel, c], gx["c_winner"][sel, c])
        assert np.array_equal(_np(out["terminal"])[sel, c], gx["c_terminal"][sel, c])
        assert np.array_equal(_np(out["legal"]).view(np.uint64)[sel, c], mask_to_bits(gx["c_mask"][sel, c]))
        assert np.array_equal(_np(out["key"])[sel, c], gx["c_hash"][sel, c])
    # illegal actions leave copies of the parent
    bad = nch == 0
    assert_export_equals(out["child0"], gx["p_board"][idx], gx["p_moves"][idx], gx["p_n_moves"][idx],
                         gx["p_qmask"][idx], gx["p_n_q"][idx], bad)


def test_expand_65536_boards_vs_oracle():
    """BASELINE config 5: 65 536-board batched expand."""
    from qtttgym_amd import VecEnv
    n, seed = 65536, 17
    env = VecEnv(n, seed=seed)
    ob = oracle.OracleBoards(n)
    rng = np.random.default_rng(1)
    depth = rng.integers(0, 9, n)
    for t in range(8):
        a = _np(env.sample_actions())
        a[depth <= t] = 255                                   # freeze boards at mixed depths (noop)
        env.step_raw(torch.from_numpy(a).cuda())
        ob.step(a, None, seed, t)
    act = rng.integers(0, 36, n).astype(np.uint8)
    out = env.expand(torch.from_numpy(act))
    nch, kids, winner, terminal, legal, key = oracle.expand(ob, act)
    assert np.array_equal(_np(out["n_children"]), nch)
    assert (nch == 2).sum() > 1000 and (nch == 0).sum() > 1000
    for c, child in enumerate((out["child0"], out["child1"])):
        sel = nch > c
        assert_export_equals(child, kids[c].board, kids[c].moves, kids[c].n_moves, kids[c].qmask, kids[c].n_q, sel)
        assert np.array_equal(_np(out["winner"])[sel, c], winner[sel, c])
        assert np.array_equal(_np(out["terminal"])[sel, c].astype(np.uint8), terminal[sel, c])
        assert np.array_equal(_np(out["legal"]).view(np.uint64)[sel, c], legal[sel, c])
        assert np.array_equal(_np(out["key"])[sel, c], key[sel, c])
    # children can be stepped further like any other batch
    r, tm = out["child0"].step_raw(out["child0"].sample_actions())
    assert r.shape == (n,)


def test_node_info_and_encode_at_every_depth_vs_oracle():
    """node_info (winner, terminal, legal mask, CPython hash key) and encode on boards frozen at every
    depth 0..9 — finished games included: nine real moves, eight moves + the autofill, early wins."""
    from qtttgym_amd import VecEnv
    n, seed = 20000, 29
    env = VecEnv(n, seed=seed)
    ob = oracle.OracleBoards(n)
    rng = np.random.default_rng(4)
    depth = rng.integers(0, 10, n)
    for t in range(9):
        a = _np(env.sample_actions())
        a[depth <= t] = 255                                   # frozen boards get a noop
        env.step_raw(torch.from_numpy(a).cuda())
        ob.step(a, None, seed, t)
    nm = ob.n_moves
    assert (nm == 9).sum() > 1000 and (nm == 0).sum() > 500
    winner, terminal, legal, key = oracle.node_info(ob)
    info = env.node_info()
    assert np.array_equal(_np(info["winner"]), winner)
    assert np.array_equal(_np(info["terminal"]).astype(np.uint8), terminal)
    assert np.array_equal(_np(info["legal"]).view(np.uint64), legal)
    assert np.array_equal(_np(info["key"]), key)
    vec, _ = env.encode()
    assert np.array_equal(_np(vec), oracle.to_vector(ob).astype(np.float32))


def test_rollout_equals_stepping_to_the_end():
    from qtttgym_amd import VecEnv
    n, seed = 8192, 23
    env = VecEnv(n, seed=seed)
    ob = oracle.OracleBoards(n)
    for t in range(3):                                        # start from a mid-game position
        a = env.sample_actions()
        env.step_raw(a)
        ob.step(_np(a), None, seed, t)
    result, plies, final = env.rollout(return_final=True)
    res_o, plies_o, fin_o = oracle.rollout(ob, seed, 3)
    assert np.array_equal(_np(result), res_o)
    assert np.array_equal(_np(plies), plies_o)
    assert_export_equals(final, fin_o.board, fin_o.moves, fin_o.n_moves, fin_o.qmask, fin_o.n_q)
    # the source boards are untouched, and stepping them launch by launch gives the same end
    assert_export_equals(env, ob.board, ob.moves, ob.n_moves, ob.qmask, ob.n_q)
    for t in range(9):
        env.step_raw(env.sample_actions())
    ex_step, ex_roll = env.export_boards(), final.export_boards()
    done = _np(final.node_info()["terminal"])
    assert done.all()
    for k in ex_step:
        # boards keep accepting legal moves after a win in raw mode, so compare only
        # where the rollout ended because the board was full
        full = _np(ex_roll["n_moves"]) == 9
        assert torch.equal(ex_step[k][torch.from_numpy(full)], ex_roll[k][torch.from_numpy(full)]), k
    assert set(np.unique(_np(result))) <= {-1, 0, 1}
    # outcome split under the uniform policy from the empty board: P1-only 52.8 %,
    # both 22.2 % (tie-broken by the earlier line), none 12.8 %, P2-only 12.2 %
    env0 = VecEnv(1 << 17, seed=5)
    r0, p0 = env0.rollout()
    frac_none = float((r0 == 0).float().mean())
    assert abs(frac_none - 0.128) < 0.01, frac_none
    assert abs(float(p0.float().mean()) - 8.30) < 0.05


def test_flat_monte_carlo_on_expand_and_rollout_beats_random():
    """The rows compose: a flat Monte-Carlo mover built from expand + rollout wins clearly more
    often as P1 than the 52.8 % + (its share of the 22.2 % double-line games) a random P1 gets."""
    import subprocess
    import sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "flat_mc_selfplay.py"),
                          "--games", "512", "--sims", "8"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    pct = float(out.stdout.split("(")[2].split("%")[0])
    assert pct > 86.0, out.stdout      # measured 91.4 % (512 games, 8 playouts per child)


def test_outcome_distribution_matches_reference_statistics():
    """T3 (measured on 20 000 reference episodes under the uniform-legal policy):
    P1-only line 52.8 %, both 22.2 %, none 12.8 %, P2-only 12.2 %; autofill in 32.4 % of episodes;
    episode length 5: 0.9 %, 6: 2.9 %, 7: 11.3 %, 8: 35.1 %, 9: 49.7 %."""
    from qtttgym_amd import VecEnv
    n = 1 << 18
    env = VecEnv(n, seed=99)
    result, plies, final = env.rollout(return_final=True)
    p1, p2 = final.check_win()
    p1, p2 = p1.cpu().numpy(), p2.cpu().numpy()
    frac = lambda m: float(m.mean())
    assert abs(frac((p1 > 0) & (p2 < 0)) - 0.528) < 0.012
    assert abs(frac((p1 > 0) & (p2 > 0)) - 0.222) < 0.012
    assert abs(frac((p1 < 0) & (p2 < 0)) - 0.128) < 0.012
    assert abs(frac((p1 < 0) & (p2 > 0)) - 0.122) < 0.012
    ex = final.export_boards()
    mv = ex["moves"].cpu().numpy()
    nm = ex["n_moves"].cpu().numpy().astype(int)
    last = mv[np.arange(n), np.maximum(nm - 1, 0)]
    autofill = (nm > 0) & (last[:, 0] == last[:, 1])
    assert abs(frac(autofill) - 0.324) < 0.012
    pl = plies.cpu().numpy()
    for length, want in ((5, 0.009), (6, 0.029), (7, 0.113), (8, 0.351), (9, 0.497)):
        assert abs(frac(pl == length) - want) < 0.012, (length, frac(pl == length))
    # winner tie-break of update_winner (mcts.py:54-56): both lines -> the earlier one
    r = result.cpu().numpy()
    both = (p1 > 0) & (p2 > 0)
    assert np.array_equal(r[both], np.where(p1[both] < p2[both], 1, -1))


@pytest.mark.parametrize("n,sims", [(1, 1), (3, 7), (1000, 10), (65536, 10), (4099, 33), (500, 28), (500, 29)])   # 28 = the key table's slots
def test_rollout_many_columns_equal_single_rollouts(n, sims):
    """qttt_rollout_many = MCTS._rollout's num_simulations loop (mcts.py:170-176) in one launch: column s is
    qttt_rollout(step_idx0 + 16 s) — which the tests above hold against the oracle's playout loop."""
    from qtttgym_amd import VecEnv
    env = VecEnv(n, seed=19, board_offset=5 * n)
    rng = np.random.default_rng(n)
    depth = torch.from_numpy(rng.integers(0, 8, n).astype(np.uint8)).cuda()
    for t in range(7):                                                  # parents at mixed depths (some finished)
        a = env.sample_actions()
        a[depth <= t] = 0
        env.step_raw(a)
    s0 = 123
    res, pl = env.rollout_many(sims, step_idx0=s0, with_plies=True)
    assert res.shape == (n, sims) and pl.shape == (n, sims)
    for s in range(sims):
        r1, p1 = env.rollout(step_idx0=s0 + 16 * s)
        assert torch.equal(res[:, s], r1) and torch.equal(pl[:, s], p1), s
    only = env.rollout_many(sims, step_idx0=s0)
    assert torch.equal(only, res)
    res.zero_()
    assert env.rollout_many(sims, step_idx0=s0, out=res) is res and torch.equal(res, only)
    if n >= 1000:                                                       # simulations of one leaf differ, the mean is a value estimate
        live = pl[:, 0] > 0
        assert bool((res[live].to(torch.float32).std(dim=1) > 0).any())
    with pytest.raises(ValueError):
        env.rollout_many(0)
    L, s = env._lib, torch.cuda.current_stream().cuda_stream
    assert L.qttt_rollout_many(env.state.data_ptr(), 1, 0, 0, 4, None, None, n, s) == -1
    assert L.qttt_rollout_many(env.state.data_ptr(), 1, 0, 0, -1, res.data_ptr(), None, n, s) == -2
    assert L.qttt_rollout_many(env.state.data_ptr(), 1, 0, 0, 0, res.data_ptr(), None, n, s) == 0
