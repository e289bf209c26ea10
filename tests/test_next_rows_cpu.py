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
