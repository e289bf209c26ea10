#!/usr/bin/env python3
"""Root-level UCB search on top of qttt_expand_rollout: G simultaneous games, and for the mover of each game a bandit over
its 36 actions — every iteration picks ONE action per game by the PUCT rule the reference uses at a node
(mcts.py:281-285: Q + c_puct * P * sqrt(Ntot) / (1 + N), uniform priors), runs one MCTS rollout below it
(`VecEnv.expand_rollout`: expansion + `sims` playouts from each child, ONE launch for all G games) and backs the value
up (mcts.py:178-184).  The loop a search needs — select with torch ops, one launch, update with torch ops — with nothing
leaving the GPU; the same budget as flat_mc_selfplay.py spends uniformly is spent where it matters.

    python examples/ucb_selfplay.py [--games 1024] [--iters 72] [--sims 4]

Player 1 (X) searches, player 2 (O) plays the uniform-legal random policy; prints P1's score.
"""
import argparse
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qtttgym_amd import VecEnv  # noqa: E402
from qtttgym_amd import recommended_env  # noqa: E402
recommended_env(apply=True)   # HIP_FORCE_DEV_KERNARG=1 etc., before the first HIP call
from qtttgym_amd.actions import action36_to_pairs  # noqa: E402


def search_actions(env, iters, sims, sweep, c_puct=1.0):
    G, dev = env.num_envs, env.device
    legal = env.node_info(python_key=False)["legal"]                                  # bit a = action a legal (mcts.py:20-27)
    mask = (legal[:, None] >> torch.arange(36, device=dev)[None, :]) & 1 == 1          # [G, 36]
    n_legal = mask.sum(1).clamp(min=1).to(torch.float32)
    N = torch.zeros((G, 36), device=dev)
    W = torch.zeros((G, 36), device=dev)
    out = None
    work = VecEnv.from_state(env.state, G, seed=env.seed + 7919 * (sweep + 1), board_offset=env.board_offset)
    for it in range(iters):
        Q = W / N.clamp(min=1)
        U = c_puct * (1.0 / n_legal)[:, None] * torch.sqrt(N.sum(1, keepdim=True) + 1.0) / (1.0 + N)   # mcts.py:283
        a = torch.where(mask, Q + U, torch.full_like(Q, -math.inf)).argmax(1)
        out = work.expand_rollout(a.to(torch.uint8), n_sims=sims, step_idx0=32 * sims * it, out=out)
        nch = out["n_children"].to(torch.float32).clamp(min=1)
        # value_sum is signed for the player to move at the leaf (the mover's opponent, mcts.py:174); both collapse
        # branches are equally likely (mcts.py:195)
        v = -out["value_sum"].to(torch.float32).sum(1) / sims / nch
        N.scatter_add_(1, a[:, None], torch.ones((G, 1), device=dev))
        W.scatter_add_(1, a[:, None], v[:, None])
    return torch.where(mask, N, torch.full_like(N, -1.0)).argmax(1)                    # the most visited action


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=72)
    ap.add_argument("--sims", type=int, default=4)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    G = args.games
    env = VecEnv(G, seed=args.seed)
    finished = torch.zeros(G, dtype=torch.bool, device=env.device)
    for ply in range(9):
        if ply % 2 == 0:
            actions = action36_to_pairs(search_actions(env, args.iters, args.sims, ply))
        else:
            actions = env.sample_actions()
        actions = torch.where(finished[:, None], torch.full_like(actions, 255), actions)   # freeze finished games
        _, term = env.step_raw(actions.contiguous())
        finished |= term
    w = env.node_info(python_key=False)["winner"]
    p1, p2, none = int((w == 1).sum()), int((w == 0).sum()), int((w == -1).sum())
    print("games %d  iterations %d x %d playouts per child :  P1 (root UCB) wins %d (%.1f %%), P2 (random) wins %d, no winner %d"
          % (G, args.iters, args.sims, p1, 100.0 * p1 / G, p2, none))
    return p1 / G


if __name__ == "__main__":
    main()
