"""SURVEY.md §8(f) rows, CPU side: the oracle's restatements of MCTS._step / update_winner /
GameState.actions / to_vector / __hash__ (mcts.py) against the golden traces recorded from the
reference's own mcts.py (tests/golden/make_golden_expand.py)."""
import os

import numpy as np
import pytest

import oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gx():
    with np.load(os.path.join(ROOT, "tests", "golden", "expand_traces.npz")) as d:
        return {k: d[k] for k in d.files}


def mask_to_bits(m):
    return (m.astype(np.uint64) << np.arange(36, dtype=np.uint64)).sum(axis=-1).astype(np.uint64)


def parents_of(gx):
    return oracle.boards_from_arrays(gx["p_board"], gx["p_moves"], gx["p_n_moves"], gx["p_qmask"], gx["p_n_q"])


def test_node_info_matches_reference_gamestate(gx):
    ob = parents_of(gx)
    winner, terminal, legal, key = oracle.node_info(ob)
    assert np.array_equal(winner, gx["p_winner"])
    assert np.array_equal(terminal.astype(bool), gx["p_terminal"])
    assert np.array_equal(legal, mask_to_bits(gx["p_mask"]))
    assert np.array_equal(key, gx["p_hash"])


def test_to_vector_matches_reference(gx):
    ob = parents_of(gx)
    assert np.array_equal(oracle.to_vector(ob), gx["p_vector"])     # float64, exact


def test_expand_matches_reference_step(gx):
    par = parents_of(gx)
    idx = gx["parent"]
    ob = oracle.OracleBoards(len(idx))
    ob.b[:] = par.b[idx]
    nch, kids, winner, terminal, legal, key = oracle.expand(ob, gx["action"])
    assert np.array_equal(nch, gx["n_children"])
    for c in range(2):
        sel = nch > c
        assert np.array_equal(kids[c].board[sel], gx["c_board"][sel, c])
        assert np.array_equal(kids[c].moves[sel], gx["c_moves"][sel, c])
        assert np.array_equal(kids[c].n_moves[sel], gx["c_n_moves"][sel, c])
        assert np.array_equal(kids[c].qmask[sel], gx["c_qmask"][sel, c])
        assert np.array_equal(kids[c].n_q[sel], gx["c_n_q"][sel, c])
        assert np.array_equal(winner[sel, c], gx["c_winner"][sel, c])
        assert np.array_equal(terminal[sel, c].astype(bool), gx["c_terminal"][sel, c])
        assert np.array_equal(legal[sel, c], mask_to_bits(gx["c_mask"][sel, c]))
        assert np.array_equal(key[sel, c], gx["c_hash"][sel, c])


def test_pyhash_matches_the_running_interpreter():
    """CPython's tuple hash is the third-party arithmetic behind GameState.__hash__."""
    rng = np.random.default_rng(3)
    ob = oracle.OracleBoards(2000)
    for t in range(9):
        ob.step(ob.sample_actions(11, t), rng.integers(0, 2, 2000).astype(np.uint8))
        _, _, _, key = oracle.node_info(ob)
        for i in range(0, 2000, 37):
            b = tuple(int(x) for x in ob.board[i])
            mv = tuple((int(ob.b["moves"][i][j][0]), int(ob.b["moves"][i][j][1]), j)
                       for j in range(int(ob.b["n_moves"][i])))
            assert int(key[i]) == hash(b + mv)


# ---------------------------------------------------------------------------------------------
# The playout loop (MCTS._rollout's num_simulations loop, _simulate, _reward: mcts.py:166-176,185-209)
# recorded from the reference's own mcts.py with its three random sources keyed by the build's counter
# hash (tests/golden/make_golden_playout.py).
@pytest.fixture(scope="module")
def gp():
    with np.load(os.path.join(ROOT, "tests", "golden", "playout_traces.npz")) as d:
        return {k: d[k] for k in d.files}


def playout_groups(gp):
    """(seed, board_offset, step_idx0, slice of the parents) per recorded group."""
    start = 0
    for g in range(len(gp["g_seed"])):
        n = int(gp["g_count"][g])
        yield int(gp["g_seed"][g]), int(gp["g_offset"][g]), int(gp["g_step_idx0"][g]), slice(start, start + n)
        start += n


def test_oracle_rollout_reproduces_the_reference_simulate_loop(gp):
    S, stride = int(gp["n_sims"]), int(gp["sim_stride"])
    assert stride == 16                                              # QTTT_SIM_STRIDE, include/qttt.h
    for seed, offset, step0, sl in playout_groups(gp):
        ob = oracle.boards_from_arrays(gp["p_board"][sl], gp["p_moves"][sl], gp["p_n_moves"][sl],
                                       gp["p_qmask"][sl], gp["p_n_q"][sl])
        w, t, _, _ = oracle.node_info(ob)
        assert np.array_equal(w, gp["p_winner"][sl]) and np.array_equal(t.astype(bool), gp["p_terminal"][sl])
        total = np.zeros(ob.n, dtype=np.int64)
        for s in range(S):
            result, plies, final = oracle.rollout(ob, seed, step0 + s * stride, offset)
            assert np.array_equal(result, gp["s_result"][sl, s])              # MCTS._reward of the end node
            assert np.array_equal(plies, gp["s_plies"][sl, s])                # termination + the 9-ply cap
            assert np.array_equal(final.board, gp["s_f_board"][sl, s])
            assert np.array_equal(final.moves, gp["s_f_moves"][sl, s])
            assert np.array_equal(final.n_moves, gp["s_f_n_moves"][sl, s])
            total += result
        # what _rollout hands to _backpropogate: r_tot / num_simulations, r_tot += r if leaf.turn else -r
        sign = np.where(gp["p_turn"][sl], 1, -1)
        assert np.array_equal(sign * total / float(S), gp["p_value"][sl])
    # terminal parents play nothing; the longest playout is the nine plies from the empty board
    assert (gp["s_plies"][gp["p_terminal"]] == 0).all() and gp["s_plies"].max() == 9
