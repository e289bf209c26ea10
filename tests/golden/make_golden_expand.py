#!/usr/bin/env python3
"""Generates tests/golden/expand_traces.npz from the UNMODIFIED reference's MCTS code
(`/root/reference/mcts.py`: GameState, MCTS._step, to_vector, action_mask, __hash__), loaded
through ref_shim.py.  Build container only; the .npz is data (inputs + expected outputs).

For S states reached by random play and every one of the 36 actions:
  parent state, action -> n_children (0 = make_move raised), and per child (ordered by the
  collapse bit: child 0 = closing move landed on lo) the Board attributes, winner
  (1 True / 0 False / -1 None), terminal, legal-action mask, Python hash(child).
For every parent state also GameState.to_vector() (18x10) and action_mask().
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import load_reference, REFERENCE_ROOT  # noqa: E402


class Toggle:
    """random.choice stand-in that alternates 0,1,0,1: MCTS._step (mcts.py:252-261) re-runs
    make_move until the other branch shows up, so the bit must change between calls."""

    def __init__(self):
        self.bit = 0

    def choice(self, seq):
        out = seq[self.bit]
        self.bit ^= 1
        return out


def board_arrays(gs):
    mv = [[255, 255]] * 9
    for i, m in enumerate(gs.moves):
        mv[i] = [m[0], m[1]]
    qm = [0] * 4
    for i, s in enumerate(gs.qstructs):
        qm[i] = sum(1 << x for x in s)
    return list(gs.board), mv, len(gs.moves), qm, len(gs.qstructs)


def main():
    qtttgym, _ = load_reference()
    sys.path.insert(0, REFERENCE_ROOT)
    import mcts as ref_mcts                      # the reference's mcts.py, unmodified
    tog = Toggle()
    qtttgym.qeval.random = tog
    strat = ref_mcts.MCTS(rollouts=1, num_simulations=1)
    GS = ref_mcts.MCTS.GameState
    rng = random.Random(4242)

    parents = []
    for _ in range(260):
        gs = GS([-1] * 9, [], True, None, False)
        depth = rng.randrange(0, 9)
        for _ in range(depth):
            legal = [a for a in range(36) if gs.board[ref_mcts.ind2move(a)[0]] == -1
                     and gs.board[ref_mcts.ind2move(a)[1]] == -1]
            if not legal:
                break
            tog.bit = rng.getrandbits(1)
            gs.make_move(ref_mcts.ind2move(rng.choice(legal)))
        gs.update_actions()
        gs.winner, gs.terminal = None, False
        gs.update_winner()
        parents.append(gs)

    P = {k: [] for k in ("board", "moves", "n_moves", "qmask", "n_q", "vector", "mask", "hash",
                         "winner", "terminal")}
    E = {k: [] for k in ("parent", "action", "n_children", "c_board", "c_moves", "c_n_moves", "c_qmask",
                         "c_n_q", "c_winner", "c_terminal", "c_mask", "c_hash")}
    wcode = {True: 1, False: 0, None: -1}
    for pi, gs in enumerate(parents):
        b, mv, nm, qm, nq = board_arrays(gs)
        P["board"].append(b); P["moves"].append(mv); P["n_moves"].append(nm)
        P["qmask"].append(qm); P["n_q"].append(nq)
        P["vector"].append(gs.to_vector()); P["mask"].append(gs.action_mask())
        P["hash"].append(hash(gs)); P["winner"].append(wcode[gs.winner]); P["terminal"].append(bool(gs.terminal))
        for a in range(36):
            lo, hi = ref_mcts.ind2move(a)
            tog.bit = rng.getrandbits(1)
            try:
                kids = strat._step(gs, a)
            except Exception:
                kids = []
            # order children by the collapse bit: child 0 = closing move (the newest regular
            # move, index len(parent.moves)) landed on lo
            if len(kids) == 2:
                r = len(gs.moves)
                kids.sort(key=lambda k: 0 if k.board[lo] == r else 1)
                assert kids[0].board[lo] == r and kids[1].board[hi] == r
            row = {k: [] for k in ("c_board", "c_moves", "c_n_moves", "c_qmask", "c_n_q", "c_winner",
                                   "c_terminal", "c_mask", "c_hash")}
            for c in range(2):
                if c < len(kids):
                    k = kids[c]
                    cb, cmv, cnm, cqm, cnq = board_arrays(k)
                    assert k.turn == (not gs.turn)
                    row["c_board"].append(cb); row["c_moves"].append(cmv); row["c_n_moves"].append(cnm)
                    row["c_qmask"].append(cqm); row["c_n_q"].append(cnq)
                    row["c_winner"].append(wcode[k.winner]); row["c_terminal"].append(bool(k.terminal))
                    m = np.zeros(36, dtype=bool)
                    if len(kids) == 2:          # mcts.py:248,260 update_actions only after a collapse;
                        m[k.actions] = True     # otherwise actions were enumerated in the ctor: same rule
                    else:
                        m[k.actions] = True
                    row["c_mask"].append(m); row["c_hash"].append(hash(k))
                else:
                    row["c_board"].append([0] * 9); row["c_moves"].append([[255, 255]] * 9)
                    row["c_n_moves"].append(0); row["c_qmask"].append([0] * 4); row["c_n_q"].append(0)
                    row["c_winner"].append(-1); row["c_terminal"].append(False)
                    row["c_mask"].append(np.zeros(36, dtype=bool)); row["c_hash"].append(0)
            E["parent"].append(pi); E["action"].append(a); E["n_children"].append(len(kids))
            for k, v in row.items():
                E[k].append(v)

    out = {
        "p_board": np.array(P["board"], dtype=np.int8), "p_moves": np.array(P["moves"], dtype=np.uint8),
        "p_n_moves": np.array(P["n_moves"], dtype=np.uint8), "p_qmask": np.array(P["qmask"], dtype=np.uint16),
        "p_n_q": np.array(P["n_q"], dtype=np.uint8), "p_vector": np.array(P["vector"], dtype=np.float64),
        "p_mask": np.array(P["mask"], dtype=bool), "p_hash": np.array(P["hash"], dtype=np.int64),
        "p_winner": np.array(P["winner"], dtype=np.int8), "p_terminal": np.array(P["terminal"], dtype=bool),
        "parent": np.array(E["parent"], dtype=np.int32), "action": np.array(E["action"], dtype=np.uint8),
        "n_children": np.array(E["n_children"], dtype=np.uint8),
        "c_board": np.array(E["c_board"], dtype=np.int8), "c_moves": np.array(E["c_moves"], dtype=np.uint8),
        "c_n_moves": np.array(E["c_n_moves"], dtype=np.uint8), "c_qmask": np.array(E["c_qmask"], dtype=np.uint16),
        "c_n_q": np.array(E["c_n_q"], dtype=np.uint8), "c_winner": np.array(E["c_winner"], dtype=np.int8),
        "c_terminal": np.array(E["c_terminal"], dtype=bool), "c_mask": np.array(E["c_mask"], dtype=bool),
        "c_hash": np.array(E["c_hash"], dtype=np.int64),
    }
    path = os.path.join(HERE, "expand_traces.npz")
    np.savez_compressed(path, **out)
    nc = out["n_children"]
    print("wrote %s: %d parents, %d expansions (illegal %d, one child %d, two children %d), %d B"
          % (path, len(parents), len(nc), int((nc == 0).sum()), int((nc == 1).sum()), int((nc == 2).sum()),
             os.path.getsize(path)))


if __name__ == "__main__":
    main()
