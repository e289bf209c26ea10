#!/usr/bin/env python3
"""Generates tests/golden/playout_traces.npz from the UNMODIFIED reference's playout code
(`/root/reference/mcts.py`: MCTS._rollout :166-176 incl. the `num_simulations` loop and the
`r if leaf.turn else -r` sign, MCTS._simulate :185-198, MCTS._reward :200-209, MCTS._step :233-267,
sample_action / get_action_probs :287-295), loaded through ref_shim.py.  Build container only; the
.npz is data (inputs + expected outputs).

The reference draws from three random sources; each is replaced by the build's counter hash
(`oracle.hash64` = include/qttt.h qttt_hash) of (seed, board id, step index), one step
index per ply:
  * `np.random.choice(node.actions, p=...)` in sample_action (mcts.py:295) -> the k-th legal action,
    k = (h2 * len(actions)) >> 32 (actions are enumerated in ind2move order, mcts.py:20-27);
  * `random.choice((lo, hi))` in qeval.py:35 -> seq[bit], bit = top bit of h1, for the FIRST make_move of a
    _step; _step's resample loop (mcts.py:252-261) re-runs make_move until the other branch shows up, so
    the second draw of the same ply returns seq[1 - bit];
  * `np.random.choice(nodes)` in _simulate (mcts.py:195) -> nodes[0], the node of the first make_move,
    i.e. the same collapse bit.
Nothing of the reference is edited: mcts.py's module global `np` is swapped for a stand-in whose
`random.choice` is the source above (every other attribute is numpy's), `_simulate` and `_backpropogate`
are wrapped on the instance to set the simulation's first step index and to record the values they
return / receive.

Per parent: Board attributes, turn, terminal; per simulation s (step indices step_idx0 + 16 s + ply, the
QTTT_SIM_STRIDE of include/qttt.h): _simulate's return value, plies played, the end state's board / moves;
and the value MCTS._rollout hands to _backpropogate (r_tot / num_simulations with the leaf.turn sign).
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from ref_shim import load_reference, REFERENCE_ROOT  # noqa: E402
import oracle  # noqa: E402  (test infrastructure: the counter hash only)

N_SIMS = 4
SIM_STRIDE = 16
# (seed, board_offset, step_idx0, parents): the second group's ids cross 2^32 (the hash folds the high word)
GROUPS = ((7, 0, 0, 1200), (0x1234567, (1 << 32) - 600, 100, 1200))


class Source:
    """The three random draws of one playout, keyed like the build's kernels."""

    def __init__(self):
        self.seed = self.board_id = self.step = 0
        self.bit = 0
        self.draws_this_ply = 0
        self.plies = 0
        self.last = None

    # ---- numpy.random stand-in (mcts.py:195,295)
    def choice(self, a, p=None):
        if p is not None:                                   # sample_action: one per ply, first draw of the ply
            h = oracle.hash64(self.seed, self.board_id, self.step)
            h1, h2 = h & 0xFFFFFFFF, h >> 32
            assert len(a) == len(p) and len(a) > 0
            self.bit = h1 >> 31
            self.draws_this_ply = 0
            self.plies += 1
            return a[(h2 * len(a)) >> 32]
        self.step += 1                                      # the branch pick ends the ply
        self.last = a[0]
        return a[0]

    # ---- `random` stand-in inside qeval.py (qeval.py:35)
    class _Qeval:
        def __init__(self, outer):
            self.o = outer

        def choice(self, seq):
            assert len(seq) == 2
            o = self.o
            out = seq[o.bit ^ (o.draws_this_ply & 1)]
            o.draws_this_ply += 1
            return out


class NpStandIn:
    def __init__(self, real, rnd):
        self._real = real
        self.random = rnd

    def __getattr__(self, k):
        return getattr(self._real, k)


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
    src = Source()
    qtttgym.qeval.random = Source._Qeval(src)
    ref_mcts.np = NpStandIn(np, src)
    GS = ref_mcts.MCTS.GameState
    rng = random.Random(777)
    wcode = {True: 1, False: 0, None: -1}

    P = {k: [] for k in ("group", "board", "moves", "n_moves", "qmask", "n_q", "turn", "terminal", "winner", "value")}
    S = {k: [] for k in ("result", "plies", "f_board", "f_moves", "f_n_moves", "f_winner")}
    for gi, (seed, offset, step_idx0, count) in enumerate(GROUPS):
        for pi in range(count):
            # ---- a parent reached by random play (collapse bits chosen by this script's rng)
            gs = GS([-1] * 9, [], True, None, False)
            depth = (pi % 10) if pi < 40 else rng.randrange(0, 9)
            real = 0
            for _ in range(depth):
                legal = [a for a in range(36) if gs.board[ref_mcts.ind2move(a)[0]] == -1
                         and gs.board[ref_mcts.ind2move(a)[1]] == -1]
                if not legal:
                    break
                src.bit, src.draws_this_ply = rng.getrandbits(1), 0
                gs.make_move(ref_mcts.ind2move(rng.choice(legal)))
                real += 1
            gs.turn = real % 2 == 0                          # mcts.py:243: flipped once per _step
            gs.update_actions()
            gs.winner, gs.terminal = None, False
            gs.update_winner()
            b, mv, nm, qm, nq = board_arrays(gs)
            P["group"].append(gi); P["board"].append(b); P["moves"].append(mv); P["n_moves"].append(nm)
            P["qmask"].append(qm); P["n_q"].append(nq); P["turn"].append(bool(gs.turn))
            P["terminal"].append(bool(gs.terminal)); P["winner"].append(wcode[gs.winner])

            # ---- MCTS._rollout on it: _select returns the root (its P is None), then num_simulations x _simulate
            strat = ref_mcts.MCTS(rollouts=1, num_simulations=N_SIMS)
            strat.root = gs
            strat.nodes = {hash(gs): gs}
            sims, handed = [], []
            inner = strat._simulate

            def simulate(node, _inner=inner, _sims=sims):
                s = len(_sims)
                src.seed, src.board_id, src.step = seed, offset + pi, step_idx0 + s * SIM_STRIDE
                src.plies, src.last = 0, node
                r = _inner(node)
                end = src.last
                _sims.append((r, src.plies, end))
                return r
            strat._simulate = simulate
            strat._backpropogate = lambda path, r, _h=handed: _h.append((len(path), r))
            strat._rollout()
            assert len(sims) == N_SIMS and len(handed) == 1 and handed[0][0] == 1
            P["value"].append(float(handed[0][1]))
            row = {k: [] for k in S}
            for r, plies, end in sims:
                assert end.terminal
                eb, emv, enm, _, _ = board_arrays(end)
                row["result"].append(r); row["plies"].append(plies)
                row["f_board"].append(eb); row["f_moves"].append(emv); row["f_n_moves"].append(enm)
                row["f_winner"].append(wcode[end.winner])
            for k in S:
                S[k].append(row[k])
            # the value handed on is the signed mean of the per-simulation results (mcts.py:170-176)
            sign = 1 if gs.turn else -1
            assert abs(P["value"][-1] - sign * sum(r for r, _, _ in sims) / N_SIMS) < 1e-12

    out = {
        "n_sims": np.int32(N_SIMS), "sim_stride": np.int32(SIM_STRIDE),
        "g_seed": np.array([g[0] for g in GROUPS], dtype=np.uint64),
        "g_offset": np.array([g[1] for g in GROUPS], dtype=np.int64),
        "g_step_idx0": np.array([g[2] for g in GROUPS], dtype=np.uint32),
        "g_count": np.array([g[3] for g in GROUPS], dtype=np.int64),
        "p_group": np.array(P["group"], dtype=np.uint8),
        "p_board": np.array(P["board"], dtype=np.int8), "p_moves": np.array(P["moves"], dtype=np.uint8),
        "p_n_moves": np.array(P["n_moves"], dtype=np.uint8), "p_qmask": np.array(P["qmask"], dtype=np.uint16),
        "p_n_q": np.array(P["n_q"], dtype=np.uint8), "p_turn": np.array(P["turn"], dtype=bool),
        "p_terminal": np.array(P["terminal"], dtype=bool), "p_winner": np.array(P["winner"], dtype=np.int8),
        "p_value": np.array(P["value"], dtype=np.float64),
        "s_result": np.array(S["result"], dtype=np.int8), "s_plies": np.array(S["plies"], dtype=np.uint8),
        "s_f_board": np.array(S["f_board"], dtype=np.int8), "s_f_moves": np.array(S["f_moves"], dtype=np.uint8),
        "s_f_n_moves": np.array(S["f_n_moves"], dtype=np.uint8), "s_f_winner": np.array(S["f_winner"], dtype=np.int8),
    }
    path = os.path.join(HERE, "playout_traces.npz")
    np.savez_compressed(path, **out)
    pl = out["s_plies"]
    print("wrote %s: %d parents x %d simulations; plies 0..%d (mean %.2f), terminal parents %d, results +1/0/-1 = %d/%d/%d, %d B"
          % (path, len(P["board"]), N_SIMS, int(pl.max()), float(pl.mean()), int(out["p_terminal"].sum()),
             int((out["s_result"] == 1).sum()), int((out["s_result"] == 0).sum()), int((out["s_result"] == -1).sum()),
             os.path.getsize(path)))


if __name__ == "__main__":
    main()
