#!/usr/bin/env python3
"""Generates tests/golden/step_traces.npz by running the UNMODIFIED reference
(`/root/reference/qtttgym`, loaded through ref_shim.py) on recorded action and
collapse-bit sequences.  Runs only in the build container; the .npz (data: inputs
and expected outputs, no reference source) is what is committed and what travels.

    python tests/golden/make_golden.py            # rewrites step_traces.npz

Per episode e and step t (T steps per episode, no reset in between):
  inputs   actions[e,t,2] u8, bits[e,t] u8 (offered; consumed only if a collapse happens)
  outputs  everything `Env.step` (env.py:34-53) returns or leaves behind:
           board, moves, n_moves, qstructs (list order), obs fields, reward (f64, sign
           of -0.0 preserved), terminated, check_win pair, consumed flag.
"""
import os
import random
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from ref_shim import load_reference  # noqa: E402

T = 12
NOOP = (255, 255)

# SURVEY.md Appendix A known-answer traces: (a, b, bit)
K_TRACES = {
    "K1": [(0, 1, 0), (1, 0, 0)],
    "K2": [(0, 1, 0), (1, 0, 1)],
    "K3": [(4, 4, 0), (0, 1, 0), (1, 2, 0), (2, 3, 0), (9, 1, 0), (1, 3, 1), (0, 4, 0)],
    "K4": [(1, 6, 0), (4, 6, 0), (2, 8, 0), (4, 5, 0), (1, 7, 0), (1, 8, 0), (2, 4, 1)],
    "K5": [(4, 5, 0), (3, 4, 0), (2, 3, 0), (3, 6, 0), (1, 6, 0), (5, 7, 0), (2, 4, 1),
           (0, 8, 0), (0, 8, 0)],
    "K6": [(4, 8, 0), (5, 6, 0), (4, 5, 0), (0, 5, 0), (2, 7, 0), (1, 2, 0), (6, 7, 0),
           (5, 7, 0)],
    "K7": [(4, 6, 0), (1, 2, 0), (1, 8, 0), (0, 2, 0), (4, 6, 1), (0, 2, 1)],
}


def legal_pairs(board):
    """mcts.py:20-27 rule: both squares classical-empty, lexicographic pairs."""
    empty = [i for i in range(9) if board[i] == -1]
    return [(a, b) for i, a in enumerate(empty) for b in empty[i + 1:]]


def gen_actions(kind, rng, qtttgym, src):
    """Plays one episode on a scratch reference Env to choose state-dependent actions."""
    env = qtttgym.Env()
    env.reset()
    acts, bits = [], []
    for _ in range(T):
        board = env._gameboard.board
        legal = legal_pairs(board)
        bit = rng.getrandbits(1)
        if kind == "uniform":
            if legal:
                a, b = rng.choice(legal)
                if rng.getrandbits(1):
                    a, b = b, a
            else:
                a, b = rng.randrange(0, 9), rng.randrange(0, 9)
        else:  # adversarial: noops of every flavour mixed with legal moves
            r = rng.random()
            if r < 0.45 and legal:
                a, b = rng.choice(legal)
                if rng.getrandbits(1):
                    a, b = b, a
            elif r < 0.60:
                a = b = rng.randrange(0, 9)                      # same square
            elif r < 0.75:
                a, b = rng.randrange(0, 9), rng.randrange(0, 9)  # maybe classical
            elif r < 0.85:
                a, b = rng.randrange(9, 256), rng.randrange(0, 9)   # IndexError first
            elif r < 0.95:
                a, b = rng.randrange(0, 9), rng.randrange(9, 256)   # IndexError second
            else:
                a = b = rng.randrange(9, 256)                    # same-square, out of range
        src.bit = bit
        env.step((a, b))
        acts.append((a, b))
        bits.append(bit)
    return acts, bits


def run_trace(qtttgym, src, acts, bits):
    env = qtttgym.Env()
    obs0, info = env.reset()
    assert info == {} and obs0["classical"] == [-1] * 9
    out = {k: [] for k in ("board", "moves", "n_moves", "qmask", "n_q", "q_p1", "q_p1_len",
                           "q_p2", "q_p2_len", "turn", "reward", "terminated", "p1_round",
                           "p2_round", "consumed")}
    for (a, b), bit in zip(acts, bits):
        src.bit = int(bit)
        calls0 = src.calls
        obs, r, terminated, truncated, info = env.step((int(a), int(b)))
        assert truncated is False and info == {}
        gb = env._gameboard
        assert obs["classical"] is gb.board          # env.py:71,82 aliasing
        out["board"].append(list(gb.board))
        mv = [[255, 255]] * 9
        for i, m in enumerate(gb.moves):
            assert m[2] == i
            mv[i] = [m[0], m[1]]
        out["moves"].append(mv)
        out["n_moves"].append(len(gb.moves))
        qm = [0, 0, 0, 0]
        for i, s in enumerate(gb.qstructs):
            qm[i] = sum(1 << x for x in s)
        out["qmask"].append(qm)
        out["n_q"].append(len(gb.qstructs))
        q1 = [[255, 255]] * 5
        for i, m in enumerate(obs["q_states_p1"]):
            q1[i] = list(m)
        q2 = [[255, 255]] * 4
        for i, m in enumerate(obs["q_states_p2"]):
            q2[i] = list(m)
        out["q_p1"].append(q1)
        out["q_p1_len"].append(len(obs["q_states_p1"]))
        out["q_p2"].append(q2)
        out["q_p2_len"].append(len(obs["q_states_p2"]))
        out["turn"].append(obs["turn"])
        assert isinstance(r, float)
        out["reward"].append(r)
        out["terminated"].append(bool(terminated))
        p1, p2 = gb.check_win()
        out["p1_round"].append(p1)
        out["p2_round"].append(p2)
        out["consumed"].append(src.calls - calls0)
    return out


def main():
    qtttgym, src = load_reference()
    rng = random.Random(20261004)
    episodes = []
    names = []
    for name, tr in K_TRACES.items():
        acts = [(a, b) for a, b, _ in tr] + [NOOP] * (T - len(tr))
        bits = [c for _, _, c in tr] + [0] * (T - len(tr))
        episodes.append((acts, bits))
        names.append(name)
    for _ in range(1024):
        episodes.append(gen_actions("uniform", rng, qtttgym, src))
        names.append("uniform")
    for _ in range(505):
        episodes.append(gen_actions("adversarial", rng, qtttgym, src))
        names.append("adversarial")

    cols = {}
    for acts, bits in episodes:
        tr = run_trace(qtttgym, src, acts, bits)
        for k, v in tr.items():
            cols.setdefault(k, []).append(v)
    E = len(episodes)
    data = {
        "actions": np.array([a for a, _ in episodes], dtype=np.uint8).reshape(E, T, 2),
        "bits": np.array([b for _, b in episodes], dtype=np.uint8).reshape(E, T),
        "board": np.array(cols["board"], dtype=np.int8),
        "moves": np.array(cols["moves"], dtype=np.uint8),
        "n_moves": np.array(cols["n_moves"], dtype=np.uint8),
        "qmask": np.array(cols["qmask"], dtype=np.uint16),
        "n_q": np.array(cols["n_q"], dtype=np.uint8),
        "q_p1": np.array(cols["q_p1"], dtype=np.uint8),
        "q_p1_len": np.array(cols["q_p1_len"], dtype=np.uint8),
        "q_p2": np.array(cols["q_p2"], dtype=np.uint8),
        "q_p2_len": np.array(cols["q_p2_len"], dtype=np.uint8),
        "turn": np.array(cols["turn"], dtype=np.uint8),
        "reward": np.array(cols["reward"], dtype=np.float64),
        "terminated": np.array(cols["terminated"], dtype=np.uint8),
        "p1_round": np.array(cols["p1_round"], dtype=np.int8),
        "p2_round": np.array(cols["p2_round"], dtype=np.int8),
        "consumed": np.array(cols["consumed"], dtype=np.uint8),
        "kind": np.array(names),
    }
    out = os.path.join(HERE, "step_traces.npz")
    np.savez_compressed(out, **data)
    n_steps = E * T
    print("wrote %s: %d episodes x %d steps; collapses=%d wins=%d terminated=%d size=%d B"
          % (out, E, T, int(data["consumed"].sum()), int((data["reward"] == -1.0).sum()),
             int(data["terminated"].sum()), os.path.getsize(out)))
    assert set(np.unique(data["consumed"])) <= {0, 1}
    assert n_steps == data["reward"].size


if __name__ == "__main__":
    main()
