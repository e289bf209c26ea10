#!/usr/bin/env python3
"""Wall-clock cost of the single-board façades at N = 1, and beside it — same process, same host core,
same random-play loop — the INTERPRETER running the reference's algorithm: oracle/py_env.py's PyEnv.step_full, the pure-
Python restatement with the reference's own data structures that returns what Env.step returns (a tool may import the
oracle; the product never does).  A GPU round trip per step cannot beat the interpreter on one board and need not: the
façades exist so that reference-shaped callers run unchanged.  (SURVEY.md §8a's "~12 us per Env.step" was the survey
container's Xeon @ 2.1 GHz; it is not comparable with numbers taken on the GPU box's host.)  One JSON line."""
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from qtttgym_amd import Env, Board, QEvalClassic  # noqa: E402
from qtttgym_amd import recommended_env  # noqa: E402
recommended_env(apply=True)   # HIP_FORCE_DEV_KERNARG=1 etc., before the first HIP call


def episodes(n_steps):
    """(seconds per step of the whole loop, seconds per Env.step CALL alone): the loop also pays the random legal move
    and an Env.reset per episode, which are the harness's, not the environment's"""
    rng = random.Random(3)
    env = Env()
    env.reset()
    done_steps, in_step = 0, 0
    clock = time.perf_counter_ns
    t0 = time.perf_counter()
    while done_steps < n_steps:
        empty = [i for i, v in enumerate(env._gameboard.board) if v == -1]
        if len(empty) < 2:
            env.reset()
            continue
        a = tuple(rng.sample(empty, 2))
        c0 = clock()
        _, _, term, _, _ = env.step(a)
        in_step += clock() - c0
        done_steps += 1
        if term:
            env.reset()
    return (time.perf_counter() - t0) / done_steps, in_step * 1e-9 / done_steps


def interpreter_episodes(n_steps):
    """the same loop through the pure-Python restatement of the reference (oracle/py_env.py): (s per step of the loop,
    s per step call alone)"""
    from oracle.py_env import PyEnv
    rng = random.Random(3)
    env = PyEnv()
    done_steps, in_step = 0, 0
    clock = time.perf_counter_ns
    t0 = time.perf_counter()
    while done_steps < n_steps:
        empty = [i for i, v in enumerate(env.b.board) if v == -1]
        if len(empty) < 2:
            env.reset()
            continue
        a = rng.sample(empty, 2)
        bit = rng.getrandbits(1)
        c0 = clock()
        _, _, term, _, _ = env.step_full(a[0], a[1], bit)
        in_step += clock() - c0
        done_steps += 1
        if term:
            env.reset()
    return (time.perf_counter() - t0) / done_steps, in_step * 1e-9 / done_steps


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def main():
    episodes(200)                                            # warm: library load, pinned buffers, kernels
    t_env_loop, t_env = episodes(3000)
    interpreter_episodes(500)
    t_py_loop, t_py = interpreter_episodes(20000)
    b = Board(QEvalClassic())
    b.make_move((0, 1))
    n = 3000
    t0 = time.perf_counter()
    for _ in range(n):
        b.check_win()
    t_cw = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for i in range(n):
        b2 = Board(QEvalClassic())
        b2.make_move((i % 8, 8))
    t_mm = (time.perf_counter() - t0) / n
    # the same mid-game: a board with five moves on it, one more move (copied first, like a search does)
    mid = Board(QEvalClassic())
    for mv in ((0, 1), (1, 2), (3, 4), (4, 5), (6, 7)):
        mid.make_move(mv)
    t_mid = 0
    for i in range(n):
        k = Board(QEvalClassic())
        k.board, k.moves, k.qstructs = mid.board.copy(), mid.moves.copy(), [set(q) for q in mid.qstructs]
        c0 = time.perf_counter_ns()
        k.make_move((2, 8) if i & 1 else (0, 2))             # an entangling move / a cycle (collapse of three squares)
        t_mid += time.perf_counter_ns() - c0
    t_mid = t_mid * 1e-9 / n
    # an _expand_child-style loop (mcts.py:210-221): 36 copies of a parent, one move each — one make_move at a
    # time, and as ONE Board.make_moves call
    pairs = [(i, j) for i in range(9) for j in range(i + 1, 9)]
    parent = Board(QEvalClassic())
    parent.make_move((0, 1))
    parent.make_move((1, 2))

    def copies():
        out = []
        for _ in pairs:
            k = Board(QEvalClassic())
            k.board, k.moves, k.qstructs = parent.board.copy(), parent.moves.copy(), [set(q) for q in parent.qstructs]
            out.append(k)
        return out
    reps = 200
    t0 = time.perf_counter()
    for _ in range(reps):
        for k, mv in zip(copies(), pairs):
            k.make_move(mv)
    t_loop = (time.perf_counter() - t0) / (reps * len(pairs))
    t0 = time.perf_counter()
    for _ in range(reps):
        Board.make_moves(copies(), pairs)
    t_batch = (time.perf_counter() - t0) / (reps * len(pairs))
    from qtttgym_amd import board as board_mod
    print(json.dumps({"row": "facade_latency_N1", "Env.step_us": t_env * 1e6, "Env.step_loop_us": t_env_loop * 1e6,
                      "Board.make_move_us": t_mm * 1e6, "Board.make_move_midgame_us": t_mid * 1e6,
                      "fastboard": board_mod._stage().fast is not None,
                      "board_mailbox_us": os.environ.get("QTTT_BOARD_MAILBOX_US", "20 (default)"),
                      "interpreter_Env.step_us": t_py * 1e6, "interpreter_Env.step_loop_us": t_py_loop * 1e6,
                      "interpreter_steps_per_s": 1.0 / t_py, "facade_over_interpreter": t_env / t_py,
                      "host_cpu": cpu_model(),
                      "expand_36_children_loop_of_make_move_us_per_child": t_loop * 1e6,
                      "expand_36_children_one_make_moves_call_us_per_child": t_batch * 1e6,
                      "Board.check_win_us": t_cw * 1e6,
                      "note": "interpreter_Env.step_us = oracle/py_env.py PyEnv.step_full (the reference's algorithm and data structures, observation included) timed the same way in the same process: the like-for-like host number; the facade is a device round trip per call and is SLOWER than it. Env.step_us = the env.step(action) call alone (perf_counter_ns around it); Env.step_loop_us = the whole random-play loop per step (also the random legal move and an Env.reset per episode: the figure of rounds 3 - 4). One step = one qttt_board_op_host call: a request to the resident mailbox wave (or, QTTT_BOARD_MAILBOX_US=0, a launch + polling the out record's stamp); check_win comes back in the same record"}))


if __name__ == "__main__":
    main()
