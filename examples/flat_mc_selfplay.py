#!/usr/bin/env python3
"""Batched search on top of the §8(f) rows: G simultaneous games, the mover picks its action by
flat Monte-Carlo — expand every one of the 36 actions for every game (both collapse branches) and run S
fused random playouts from every child, all in ONE launch (qttt_expand_rollout), average, take the best.
One search sweep touches G*36*2 children and G*36*2*S playouts without leaving the GPU — the
"env as rollout backend" use BASELINE.json's config 5 has in mind (G*36 = 65 536 at G = 1820).

    python examples/flat_mc_selfplay.py [--games 1024] [--sims 16]

Player 1 (X) searches, player 2 (O) plays the uniform-legal random policy; prints P1's score.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qtttgym_amd import VecEnv  # noqa: E402
from qtttgym_amd import recommended_env  # noqa: E402
recommended_env(apply=True)   # HIP_FORCE_DEV_KERNARG=1 etc., before the first HIP call
from qtttgym_amd.actions import action36_to_pairs  # noqa: E402  (ind2move for a whole batch, mcts.py:339-343)


def search_actions(env, sims, sweep):
    """Best action36 per game for the side to move: ONE launch per sweep (qttt_expand_rollout = expand + `sims`
    playouts from each child, mcts.py:166-176,233-267)."""
    G, dev = env.num_envs, env.device
    # one row per (game, action): every state lined up 36 times (the packed planes are indexed, nothing is unpacked)
    rep = env.take(torch.arange(G, device=dev).repeat_interleave(36), seed=env.seed + 1000 + sweep)
    act = torch.arange(36, dtype=torch.uint8, device=dev).repeat(G)
    out = rep.expand_rollout(act, n_sims=sims, step_idx0=100 + 32 * sweep * sims)
    nch = out["n_children"].to(torch.float32)                       # 0 illegal, 1, or 2 (collapse)
    # value_sum is signed for the player to move AT THE LEAF (mcts.py:174) — the mover's opponent; a child that
    # does not exist contributes 0; both collapse branches are equally likely
    value = -out["value_sum"].to(torch.float32).sum(dim=1) / sims / nch.clamp(min=1)
    value = torch.where(nch > 0, value, torch.full_like(value, -2.0))
    return value.view(G, 36).argmax(dim=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--games", type=int, default=1024)
    ap.add_argument("--sims", type=int, default=16)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    G = args.games
    env = VecEnv(G, seed=args.seed)
    finished = torch.zeros(G, dtype=torch.bool, device=env.device)
    for ply in range(9):
        if ply % 2 == 0:
            a36 = search_actions(env, args.sims, ply)
            actions = action36_to_pairs(a36)
        else:
            actions = env.sample_actions()
        actions = torch.where(finished[:, None], torch.full_like(actions, 255), actions)   # freeze finished games
        _, term = env.step_raw(actions.contiguous())
        finished |= term
    info = env.node_info(python_key=False)
    w = info["winner"]
    p1, p2, none = int((w == 1).sum()), int((w == 0).sum()), int((w == -1).sum())
    print("games %d  sims/child %d :  P1 (flat MC) wins %d (%.1f %%), P2 (random) wins %d, no winner %d"
          % (G, args.sims, p1, 100.0 * p1 / G, p2, none))
    return p1 / G


if __name__ == "__main__":
    main()
