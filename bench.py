#!/usr/bin/env python3
"""bench.py — vectorised env.step()/s at batch = 1 048 576 boards per MI355X (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--boards B]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one launch of the fused qttt_step kernel over the whole batch of B boards of this
rank (Env.step for every board, auto-reset throughput mode).  Actions are pre-recorded (an
untimed pass of policy kernel + step kernel), the boards are reset, and the timed region
replays the recorded actions, so it contains env.step and nothing else, with inputs resident in
HBM.  Boards are independent: each rank owns B boards (global ids rank*B..), no data-path
collective; one RCCL all_reduce of episode counters after the timed region.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process RCCL on this pool needs dmabuf IPC (the image exports this; keep it if a launcher drops it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

STATE_BYTES = 20          # packed state per board
ALGO_BYTES_PER_STEP = 2 * STATE_BYTES + 2 + 4 + 1   # state r+w, action, reward f32, terminated
HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
PMC_SUMMARY = os.path.join(ROOT, "profiles", "pmc_traffic.json")


def pmc_traffic_per_launch(boards):
    """HBM bytes per step launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, FETCH_SIZE doubled per the gfx950 correction in
    MI355X_MICROARCH.md §HBM).  None if no summary for this batch size is committed."""
    try:
        with open(PMC_SUMMARY) as f:
            d = json.load(f)
        e = d.get(str(boards))
        return None if e is None else float(e["hbm_bytes_per_launch"])
    except (OSError, ValueError, KeyError):
        return None


def cpu_baseline(actions_host, seed, budget_s=12.0):
    """Times the CPU oracle (oracle/qttt_oracle.c, a scalar C port of the reference algorithm)
    on this box's host cores, on a bounded sample of the same workload: the first recorded steps
    of a slice of the boards, one slice per thread."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    import oracle
    T, n = actions_host.shape[0], actions_host.shape[1]
    cores = max(1, min(os.cpu_count() or 1, 16))   # the GPU box's CPU share for one GPU
    per = n // cores
    boards = [oracle.OracleBoards(per) for _ in range(cores)]

    def work(k):
        ob = boards[k]
        lo = k * per
        done = 0
        t_end = time.perf_counter() + budget_s
        for t in range(T):
            ob.step(np.ascontiguousarray(actions_host[t, lo:lo + per]), None, seed, t, lo, True)
            done += per
            if time.perf_counter() > t_end:
                break
        return done
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        total = sum(ex.map(work, range(cores)))
    dt = time.perf_counter() - t0
    py = python_interpreter_line(actions_host, seed)
    return {"value": total / dt, "unit": "steps/s", "cores": cores, "kind": "port",
            "python_interpreter_steps_per_s": py,
            "sample": "%d boards x %d recorded steps of the same workload (uniform-legal policy, "
                      "auto-reset), %d threads x %d boards, %.1f s" % (per * cores, total // (per * cores), cores, per, dt)}


def python_interpreter_line(actions_host, seed, budget_s=2.0):
    """The like-for-like interpreter-speed line: oracle/py_env.py, a pure-Python
    single-board restatement with the reference's own data structures, one core, ~2 s."""
    import oracle
    from oracle.py_env import PyEnv
    T = actions_host.shape[0]
    n_boards = actions_host.shape[1]
    done, b = 0, 0
    t_end = time.perf_counter() + budget_s
    t0 = time.perf_counter()
    while time.perf_counter() < t_end:
        env = PyEnv()
        for t in range(min(T, 64)):
            a0, a1 = int(actions_host[t, b, 0]), int(actions_host[t, b, 1])
            _, term = env.step(a0, a1, oracle.collapse_bit(seed, b, t))
            done += 1
            if term:
                env.reset()          # auto-reset, like the workload
        b = (b + 1) % n_boards
    return done / (time.perf_counter() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--boards", type=int, default=1 << 20, help="boards per GPU")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", choices=["replay", "policy", "random"], default="replay",
                    help="replay: timed region is env.step only (default, the metric); "
                         "policy: policy kernel + env.step per step; "
                         "random: policy and env.step fused in one kernel per step")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from qtttgym_amd import VecEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # one rank per GPU; QTTT_DIST_BACKEND=gloo lets several ranks rehearse on one GPU (plumbing only)
    backend = os.environ.get("QTTT_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    B, K, W = args.boards, args.steps, args.warmup
    T = K + W
    env = VecEnv(B, device=dev, seed=args.seed, auto_reset=True, board_offset=rank * B)

    # ---- untimed: record the action stream of the uniform-legal policy ------------------
    actions = torch.empty((T, B, 2), dtype=torch.uint8, device=dev)
    term_count = torch.zeros((), dtype=torch.int64, device=dev)
    win_count = torch.zeros((), dtype=torch.int64, device=dev)
    for t in range(T):
        env.sample_actions(out=actions[t])
        r, tm = env.step_raw(actions[t])
        term_count += tm.sum()
        win_count += (r != 0).sum()
    torch.cuda.synchronize(dev)
    final_state = env.state.clone()

    def barrier():
        if world > 1:
            if backend == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.barrier()

    # ---- timed: replay --------------------------------------------------------------
    env.reset()
    def one_step():
        if args.mode == "policy":
            env.step_raw(env.sample_actions())
        else:
            env.step_random()

    if args.mode == "replay":
        env.step_many(actions[:W])
    else:
        for t in range(W):
            one_step()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    ev0.record()
    if args.mode == "replay":
        env.step_many(actions[W:])
    else:
        for t in range(K):
            one_step()
    ev1.record()
    torch.cuda.synchronize(dev)
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    ev_ms = ev0.elapsed_time(ev1)
    replay_ok = bool(torch.equal(env.state, final_state))

    if world > 1:
        tt = torch.tensor([elapsed, ev_ms], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, ev_ms = float(tt[0]), float(tt[1])
        # episode counters: the only exchange in the design, once per run, off the timed path
        cnt = torch.stack([term_count, win_count]).to(coll_dev)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
        term_count, win_count = cnt[0], cnt[1]

    if rank == 0:
        total_steps = B * K * world
        value = total_steps / elapsed
        launch_s = ev_ms * 1e-3 / K
        achieved = ALGO_BYTES_PER_STEP * B / launch_s / 1e9
        out = {
            "metric": "env_steps_per_sec", "value": value, "unit": "steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": elapsed * 1e3 / K, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "synthetic",
            "config": {"workload": "%d boards per GPU, uniform-legal random policy, auto-reset, "
                                   "%s" % (B, "recorded actions replayed (env.step only in the timed region)"
                                           if args.mode == "replay" else ("policy kernel + env.step per step" if args.mode == "policy"
                                                                    else "policy + env.step fused in one kernel per step")),
                       "boards_per_gpu": B, "state_bytes_per_board": STATE_BYTES,
                       "parallelism": "shard%d" % world, "mode": args.mode,
                       "replay_matches_recording": replay_ok,
                       "episodes_finished": int(term_count), "steps_with_line": int(win_count)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": pmc_traffic_per_launch(B),
                         "kernel": "step_kernel<2,false,true>", "launch_us": launch_s * 1e6,
                         "algorithmic_bytes_per_launch": ALGO_BYTES_PER_STEP * B},
        }
        if not args.no_cpu_baseline:
            # bounded sample: the first <=256 recorded steps of every board of rank 0
            t_cpu = min(T, 256)
            out["cpu_baseline"] = cpu_baseline(actions[:t_cpu].cpu().numpy(), args.seed)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if not replay_ok:
        raise SystemExit("replay diverged from the recording")


if __name__ == "__main__":
    main()
