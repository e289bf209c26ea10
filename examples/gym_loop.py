#!/usr/bin/env python3
"""The reference's gym loop (`obs, r, terminated, truncated, info = env.step(action)`, env.py:34-53)
for N boards at once: a policy that reads the observation tensors on the GPU and answers with an
action per board, auto-reset on, episode statistics accumulated on the device.

    python examples/gym_loop.py [--boards 262144] [--steps 200] [--graph | --fresh]

--graph: one whole agent step (policy reading the observation, env.step, the statistics) is captured in a hipGraph
and replayed: possible because the step index lives on the device (VecEnv.use_device_step_counter), so every
replay draws fresh collapse bits, and because no call of the library allocates, synchronises or queries.

The policy here: play the uniform-legal random move, except take the centre square (4) together
with the first other empty square whenever the centre is still empty — just enough to show a
policy that depends on `obs["classical"]`.

Aliasing: the loop consumes each observation at once, so it asks for the zero-copy form
(`copy_obs=False`: the returned tensors are the environment's own buffers and the NEXT step overwrites
them).  A caller that keeps observations — a replay buffer storing (obs, next_obs) — uses the default
`env.step(action)`, which returns fresh tensors every call (one allocation from torch's caching allocator, carved into the
eight tensors), still from ONE kernel: `--fresh` runs the loop that way.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qtttgym_amd import VecEnv  # noqa: E402
from qtttgym_amd import recommended_env  # noqa: E402
recommended_env(apply=True)   # HIP_FORCE_DEV_KERNARG=1 etc., before the first HIP call


def policy(env, obs):
    a = env.sample_actions()                                   # uniform over legal pairs, u8[N, 2]
    classical = obs["classical"]                               # i8[N, 9], -1 = not classical yet
    empty = classical < 0
    centre_free = empty[:, 4]
    others = empty.clone()
    others[:, 4] = False
    first_other = others.to(torch.uint8).argmax(dim=1).to(torch.uint8)
    use = centre_free & others.any(dim=1)
    four = torch.full_like(first_other, 4)
    return torch.stack((torch.where(use, four, a[:, 0]), torch.where(use, first_other, a[:, 1])), dim=1)   # no host sync


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=262144)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--fresh", action="store_true", help="the default env.step(): fresh output tensors every call")
    args = ap.parse_args()
    if args.graph and args.fresh:
        ap.error("--graph replays ONE captured step: it needs the in-place form")
    env = VecEnv(args.boards, seed=1, auto_reset=True)
    obs, _ = env.reset(copy_obs=False)                          # the environment's own buffers: every step refreshes them
    episodes = torch.zeros((), dtype=torch.int64, device=env.device)
    lines = torch.zeros((), dtype=torch.int64, device=env.device)

    state = {"obs": obs}

    def agent_step():
        if args.fresh:                                         # the gym loop as written against the reference
            state["obs"], reward, terminated, truncated, info = env.step(policy(env, state["obs"]))
        else:
            # obs are the environment's own buffers: the step overwrites them in place (copy_obs=False)
            _, reward, terminated, truncated, info = env.step(policy(env, obs), copy_obs=False)
        episodes.add_(terminated.sum())
        lines.add_((reward != 0).sum())                        # env.py:49: -1.0 iff somebody holds a line

    if args.graph:
        env.use_device_step_counter()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(10):                                 # warm the allocator and the kernels
                agent_step()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side):
                agent_step()
        torch.cuda.current_stream().wait_stream(side)
        run = g.replay
    else:
        for _ in range(10):
            agent_step()
        run = agent_step
    episodes.zero_(); lines.zero_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("%d boards x %d steps in %.3f s = %.3g env steps/s (policy included); %d episodes finished, %d with a line"
          % (args.boards, args.steps, dt, args.boards * args.steps / dt, int(episodes), int(lines)))


if __name__ == "__main__":
    main()
