#!/usr/bin/env python3
"""bench.py — vectorised env.step()/s at batch = 1 048 576 boards per MI355X (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--boards B | --total-boards N]
                    [--mode replay|gym|policy|random|random-fused] [--no-legs]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one launch of the fused qttt_step kernel over the whole batch of B boards of this
rank (Env.step for every board, auto-reset throughput mode).  Actions are pre-recorded (an
untimed pass of policy kernel + step kernel), the boards are reset, and a timed region replays
K recorded steps, so it contains env.step and nothing else, with inputs resident in HBM.

Clock.  A region of K launches is short (K = 20 -> 0.16 ms), so the region (W untimed warm-up
steps + K timed steps, bracketed by barrier + torch.cuda.synchronize() on both sides) is repeated
R times (R chosen so that the timed parts add up to >= ~50 ms) and the MEDIAN region is reported.
Each region is timed twice: by HIP events recorded on the launch stream right around the K timed
launches (`value`, `ms_per_step` and `roofline` all come from this one clock) and by the host's
perf_counter over the whole bracket (`host_wall_ms_per_step` = bracket time / (W + K) launches,
reported beside it; it adds the host's synchronise latency, a fixed ~20-60 us per region).

Multi-GPU.  Boards are independent: each rank owns B boards (global ids rank*B..), no data-path
collective.  The ranks' bookkeeping (the clock's barrier and MAX, rank count, per-rank timings, episode
counters) is a few CPU scalars over gloo; the one real exchange, the gather of per-board returns after
the timed regions, is device tensors over RCCL and runs under a deadline (QTTT_BENCH_GATHER_TIMEOUT,
120 s): failing or hanging, it costs the line its `returns_gather` entry, never the value.  `--gpus N` with no
WORLD_SIZE in the environment starts its own N ranks (one process per GPU) before anything touches
the GPU; under torchrun the ranks are taken from the environment.  `ranks_seen` is an all_reduce
of ones over the process group; the line is refused unless n_gpus == ranks_seen.
`--total-boards 2097152 --gpus 8` is BASELINE config 4 as stated (2 097 152 boards sharded over 8 GPUs with
dist.shard_range: STRONG scaling, the JSON says so); `--boards B` is per GPU (weak scaling, the default).

Legs.  At N = 1 the same JSON object carries `legs`: the other BASELINE configurations and modes measured
in the same process on the same clock (HIP events, median region) — env.step replay at 4 096 / 262 144 /
16 777 216 boards (the last cannot live in the 256 MB Infinity Cache), `gym` and `random` at 1 M, the fused
random-policy multi-step kernel (qttt_step_random_many) at 4 096 / 262 144 / 1 M, BASELINE config 5's
unit (one MCTS rollout below 65 536 selected nodes = qttt_expand_rollout: expand + the children's bookkeeping +
playouts from each child, ONE launch; the three-launch composition beside it), and the kernels beside the step one
by one at 1 M boards (observe, export, turn, check_win, node_info with the native / the CPython key, expand with
either, rollout, encode).  Each with us per launch, algorithmic bytes, frac.

Commands.  The headline (and what the driver's SCALE run measures, WEAK scaling: 1 048 576 boards per GPU, so N
GPUs are N independent loops and the >= 7x of BASELINE.json holds by construction):
    python bench.py --gpus N --steps K --warmup W
BASELINE config 4 as stated (2 097 152 boards over 8 GPUs, STRONG scaling: 262 144 boards per GPU is one partial
occupancy round per launch, predicted 3.9 - 4.1x launch-per-step and 5.4 - 5.7x fused, DESIGN.md §8 — the >= 7x claim
does NOT apply to it):
    python bench.py --gpus 8 --total-boards 2097152 --mode random-fused
QTTT_BENCH_NO_GATHER=1 skips the optional returns gather (the only collective that moves per-board data).
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# multi-process RCCL on this pool needs dmabuf IPC (the image exports this; keep it if a launcher drops it)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# kernel arguments placed in device memory: ROCm 7.2's default on this part, made explicit because it is worth 1.2 us
# per launch (tools/stepbench, 1 M boards: 7.13 us with it, 8.32 without; 262 144 boards: 3.80 / 4.77) — set before
# anything initialises HIP
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0   # same guide: what a float4 copy kernel reaches on this part
PMC_SUMMARY = os.path.join("profiles", "pmc_traffic.json")
TARGET_TIMED_S = 0.05     # the repeated regions add up to at least this much device time
MAX_REGIONS = 400


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--boards", type=int, default=1 << 20, help="boards per GPU (weak scaling)")
    ap.add_argument("--total-boards", type=int, default=0,
                    help="boards over ALL GPUs, sharded with dist.shard_range (strong scaling; "
                         "2097152 with --gpus 8 = BASELINE config 4); overrides --boards")
    ap.add_argument("--no-legs", action="store_true", help="skip the extra legs (other configs / modes) at N = 1")
    ap.add_argument("--fused-steps", type=int, default=64, help="steps per launch of --mode random-fused (<= 64)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--regions", type=int, default=0, help="repeat count of the K-step timed region (0 = auto)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=12.0, help="seconds of CPU work for cpu_baseline")
    ap.add_argument("--mode", choices=["replay", "gym", "gym-default", "policy", "random", "random-fused"], default="replay",
                    help="replay: timed region is env.step only (default, the metric); "
                         "gym: env.step returning the observation too (step + obs fused in one kernel), into the "
                         "environment's own buffers (VecEnv.step_observe_raw, the zero-copy form); "
                         "gym-default: the same kernel through the DEFAULT VecEnv.step(actions): fresh output tensors every call; "
                         "policy: policy kernel + env.step per step; "
                         "random: policy and env.step fused in one kernel per step; "
                         "random-fused: the same, --fused-steps steps per launch with the boards in registers "
                         "(qttt_step_random_many, every step's action / reward / terminated kept)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------- self-launch
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args):
    """--gpus N > 1 without a launcher: start N ranks of this script, one per GPU, and exit with
    the worst of their codes.  The parent never touches the GPU (no torch import here)."""
    n = args.gpus
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), QTTT_BENCH_SELF_LAUNCHED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    deadline = time.time() + float(os.environ.get("QTTT_BENCH_TIMEOUT", "1500"))
    pending = list(procs)
    while pending:
        for p in list(pending):
            code = p.poll()
            if code is not None:
                pending.remove(p)
                if code != 0:
                    rc = rc or code
        if rc != 0 or time.time() > deadline:       # one rank failed: the others would hang in a barrier
            for p in pending:
                p.terminate()
            for p in pending:
                try:
                    p.wait(20)
                except subprocess.TimeoutExpired:
                    p.kill()
            rc = rc or 124
            break
        time.sleep(0.05)
    return rc


# ---------------------------------------------------------------------------- roofline inputs
def pmc_traffic_per_launch(boards, state_bytes, suffix=""):
    """HBM bytes per step launch from the committed rocprofv3 --pmc passes (FETCH_SIZE and
    WRITE_SIZE collected in separate runs, FETCH_SIZE doubled per the gfx950 correction in
    MI355X_MICROARCH.md §HBM).  Not measured in this run (a process cannot read the PMC counters of its own
    kernels without the profiler): `traffic_source` names the file, and the second value returned says whether the
    passes were made on the step-kernel sources this run was built from (tools/pmc_summary.py's fingerprint).
    (None, None) if no summary for this batch size / state layout is committed."""
    try:
        with open(os.path.join(ROOT, PMC_SUMMARY)) as f:
            d = json.load(f)
        e = d.get("%d@%dB%s" % (boards, state_bytes, suffix))
        if e is None or int(e.get("state_bytes_per_board", 20)) != state_bytes:
            return None, None
        import hashlib
        h = hashlib.sha256()
        for name in ("qttt_state.h", "qttt_step_core.h", "qttt_observation.h", "qttt_step_kernels.h"):
            with open(os.path.join(ROOT, "qtttgym_amd", "csrc", name), "rb") as src:
                h.update(src.read())
        return float(e["hbm_bytes_per_launch"]), e.get("step_sources_sha256") == h.hexdigest()[:16]
    except (OSError, ValueError, KeyError):
        return None, None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return None


def cgroup_cpu_max():
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                return f.read().strip()
        except OSError:
            continue
    return None


ALLOWED_CPUS = None      # the affinity mask the process started with (run() records it before binding to the GPU's cores)


def cpu_baseline(actions_host, seed, budget_s):
    """Times the CPU oracle (oracle/qttt_oracle.c, a scalar C port of the reference algorithm) on this box's host
    cores, on a bounded sample of the same workload: the recorded steps of a slice of the boards, one slice per thread,
    replayed (reset + replay, one C call per replay: qo_replay_batch) until the pass's share of the budget of wall time
    has been spent on every thread.  Three passes, as SURVEY.md §8(d) asks ("1 thread and all host cores", with the
    box's nproc and CPU model beside them): ONE thread (`threads1`), the threads of this GPU's CPU share (`value`,
    `cores` — a one-GPU box of the pool gives a job 16 of the host's cores), and ALL the host's hardware threads
    (`threads_all`, `threads_all_cores` = os.cpu_count())."""
    import numpy as np
    from concurrent.futures import ThreadPoolExecutor
    import oracle
    actions_host = np.ascontiguousarray(actions_host)
    T, n = actions_host.shape[0], actions_host.shape[1]
    nproc = os.cpu_count() or 1
    share = max(1, min(nproc, 16))                 # the GPU box's CPU share for one GPU
    base = actions_host.ctypes.data

    def run_pass(threads, budget, per=None):
        per = per or n // threads
        # one C call replays `chunk` steps of the slice: ~100 K board-steps, a few ms — so that the clock is looked at often
        # enough even when the threads outnumber the cores the box lets this job use
        chunk = max(1, min(T, 100000 // max(per, 1)))

        def work(k):
            ob = oracle.OracleBoards(per)
            scratch = (np.empty(per, dtype=np.float32), np.empty(per, dtype=np.uint8))
            done = 0
            t_end = time.perf_counter() + budget
            while True:
                ob.reset()
                for t0 in range(0, T, chunk):
                    tc = min(chunk, T - t0)
                    ob.replay(base + 2 * (t0 * n + k * per), n, tc, seed, t0, k * per, True, scratch)
                    done += per * tc
                    if time.perf_counter() > t_end:
                        return done
        t0 = time.perf_counter()
        if threads == 1:
            total = work(0)
        else:
            with ThreadPoolExecutor(threads) as ex:
                total = sum(ex.map(work, range(threads)))
        dt = time.perf_counter() - t0
        return total / dt, dt, per, total
    v1, dt1, per1, _ = run_pass(1, budget_s * 0.12, per=n // share)     # one of the share pass's slices
    v, dt, per, total = run_pass(share, budget_s * 0.5)
    # the all-cores pass runs on every CPU the process MAY use: the binding to the GPU's own cores (bind_cpu) is lifted
    # for it and put back afterwards (threads inherit the mask they are started under)
    bound = os.sched_getaffinity(0)
    try:
        if ALLOWED_CPUS and ALLOWED_CPUS != bound:
            os.sched_setaffinity(0, ALLOWED_CPUS)
        allowed = len(os.sched_getaffinity(0))
        va, dta, pera, _ = run_pass(nproc, budget_s * 0.2) if nproc != share else (v, dt, per, total)
    finally:
        os.sched_setaffinity(0, bound)
    py = python_interpreter_line(actions_host, seed)
    return {"value": v, "unit": "steps/s", "cores": share, "kind": "port",
            "threads1": v1, "threads_all": va, "threads_all_cores": nproc, "nproc": nproc, "cpu_model": cpu_model(),
            # what the box lets this job use: the CPUs in its affinity mask and the container's CPU quota ("max" = none;
            # "1600000 100000" = 16 cores' worth) — with a quota below nproc the all-cores pass is 256 threads sharing
            # that quota and reads LOWER than the 16-thread pass
            "cpus_allowed": allowed, "cgroup_cpu_max": cgroup_cpu_max(),
            "python_interpreter_steps_per_s": py,
            "sample": "%d boards x the first %d recorded steps of the same workload (uniform-legal policy, "
                      "auto-reset), replayed from reset %.1f times, %d threads x %d boards, %.1f s of wall time; "
                      "threads1: one slice of %d boards on one thread, %.1f s; threads_all: %d threads x %d boards, %.1f s"
                      % (per * share, T, total / float(per * share * T), share, per, dt, per1, dt1, nproc, pera, dta)}


def python_interpreter_line(actions_host, seed, budget_s=2.0):
    """The like-for-like interpreter-speed line: oracle/py_env.py, a pure-Python
    single-board restatement with the reference's own data structures returning what Env.step returns (the
    observation dict too, env.py:46,68-85), one core, ~2 s.  Actions and collapse bits are turned into Python ints
    BEFORE the clock starts (the reference's caller holds Python ints too): what is timed is the step calls and the
    auto-reset, the figure tools/facade_latency.py prints as interpreter_Env.step_us."""
    import oracle
    from oracle.py_env import PyEnv
    T = min(actions_host.shape[0], 64)
    n_boards = min(actions_host.shape[1], 512)
    plan = [[(int(actions_host[t, b, 0]), int(actions_host[t, b, 1]), int(oracle.collapse_bit(seed, b, t))) for t in range(T)]
            for b in range(n_boards)]
    done, b = 0, 0
    t_end = time.perf_counter() + budget_s
    t0 = time.perf_counter()
    while time.perf_counter() < t_end:
        env = PyEnv()
        step = env.step_full
        for a0, a1, bit in plan[b]:
            if step(a0, a1, bit)[2]:
                env.reset()          # auto-reset, like the workload
        done += T
        b = b + 1 if b + 1 < n_boards else 0
    return done / (time.perf_counter() - t0)


def median(xs):
    s = sorted(xs)
    m = len(s) // 2
    return s[m] if len(s) & 1 else 0.5 * (s[m - 1] + s[m])


GRAPH_LAUNCHES = 100


def time_calls(torch, dev, fn, K, regions):
    """us per call of `fn` (a VecEnv method with reused buffers), two ways, both by HIP events on the launch stream:
    eager — K Python calls per region: below ~6 us per kernel this is the HOST's rate (ctypes + hipLaunchKernel), not
    the kernel's; graph — the same K launches captured once in a hipGraph and replayed: the device's own back-to-back
    rate, which is what a roofline fraction may be held against.  Returns (graph us list, eager us list)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    eager = []
    for _ in range(regions):
        for _ in range(3):
            fn()
        e0.record()
        for _ in range(K):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        eager.append(e0.elapsed_time(e1) * 1e3 / K)
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    G = max(K, GRAPH_LAUNCHES)                      # a replay costs ~10 - 20 us of its own: spread it over many launches
    with torch.cuda.stream(side):
        fn()
        with torch.cuda.graph(graph, stream=side):
            for _ in range(G):
                fn()
    torch.cuda.current_stream(dev).wait_stream(side)
    torch.cuda.synchronize(dev)
    dev_paced = []
    for _ in range(regions):
        graph.replay()
        e0.record()
        graph.replay()
        e1.record()
        torch.cuda.synchronize(dev)
        dev_paced.append(e0.elapsed_time(e1) * 1e3 / G)
    return dev_paced, eager


# ---------------------------------------------------------------------------- one workload, one clock
# wave64 VALU instructions per ns the chip can issue at best: 1 024 SIMDs x one instruction per 1.03 ns, the measured
# rate of the FAST class (plain logic, add / sub, right shifts; profiles/r02/valu_rates.txt).  Everything else (left
# shifts, bfe, compares, selects, multiplies, anything with an SGPR operand) issues at 1.7x that, so a real mix tops out
# well below 1.0 of this peak.
VALU_PEAK_GINST = 1024 / 1.03


def kernel_label(mode, bpl, blk, boards=None):
    """the kernel's name as rocprofv3 prints it (template arguments: BLOCK, BPL, HAS_BITS, AUTO_RESET, SAMPLE, OBS, DEVSTEP /
    BLOCK, AUTO_RESET, RETURNS, KEEP — qttt_step_kernels.h; KEEP below 262 144 boards with every output kept), so that the line can be held against profiles/*/kernel_stats*.csv"""
    if mode == "random-fused":
        return "step_random_fused_kernel<256, true, false, %s>" % ("true" if 0 < int(boards or 0) < 262144 else "false")
    if mode == "random":
        return "step_kernel<%d, %d, false, true, true, false, false>" % (blk, bpl)
    step = "step_kernel<%d, %d, false, true, false, %s, false>" % (blk, bpl, "true" if mode in ("gym", "gym-default") else "false")
    return ("sample_actions_kernel + " + step) if mode == "policy" else step


WHAT = {"replay": "recorded actions replayed (env.step only in the timed region)",
        "gym": "recorded actions replayed, env.step returning the observation (step + obs in one kernel) into the environment's own buffers: VecEnv.step_observe_raw",
        "gym-default": "recorded actions replayed through the default VecEnv.step(actions): (obs, reward, terminated, truncated, info) as fresh tensors every call, one kernel",
        "policy": "policy kernel + env.step per step",
        "random": "policy + env.step fused in one kernel per step",
        "random-fused": "policy + env.step, %d steps per launch with the boards in registers, every step's "
                        "action / reward / terminated kept"}


class Workload:
    """B boards on one device, uniform-legal policy, auto-reset: records the action stream (untimed), then
    times regions of W warm-up + EXACTLY K steps with HIP events on the launch stream."""

    def __init__(self, torch, dev, B, K, W, seed, board_offset, mode, fused_T):
        from qtttgym_amd import VecEnv, _native
        self.torch, self.dev, self.B, self.K, self.W, self.mode, self.fused_T = torch, dev, B, K, W, mode, fused_T
        T = K + W
        self.env = env = VecEnv(B, device=dev, seed=seed, auto_reset=True, board_offset=board_offset)
        self.state_bytes = int(_native.lib().qttt_state_bytes(64)) // 64
        gym = mode in ("gym", "gym-default")
        # algorithmic bytes per board-step: state r+w, action, reward f32, terminated (+ the 30-byte
        # observation of env.py:68-85 in gym mode: classical 9, q_p1 10+1, q_p2 8+1, turn 1).  The fused
        # multi-step form keeps the boards in registers: 7 B of outputs per step + the state once per launch
        self.algo_bytes = 2 * self.state_bytes + 2 + 4 + 1 + (30 if gym else 0)
        if mode == "random-fused":
            self.algo_bytes = 7.0 + 2.0 * self.state_bytes / fused_T
        self.shape = _native.step_launch_shape(B, 0, gym)
        # ---- untimed: record the action stream of the uniform-legal policy
        self.actions = actions = torch.empty((T, B, 2), dtype=torch.uint8, device=dev)
        self.term_count = torch.zeros((), dtype=torch.int64, device=dev)
        self.win_count = torch.zeros((), dtype=torch.int64, device=dev)
        for t in range(T):
            env.sample_actions(out=actions[t])
            r, tm = env.step_raw(actions[t])
            if t >= W:
                self.term_count += tm.sum()
                self.win_count += (r != 0).sum()
        torch.cuda.synchronize(dev)
        self.final_state = env.state.clone()
        if mode == "random-fused":
            Tf = min(fused_T, K)
            self.f_act = torch.empty((Tf, B, 2), dtype=torch.uint8, device=dev)
            self.f_rew = torch.empty((Tf, B), dtype=torch.float32, device=dev)
            self.f_term = torch.empty((Tf, B), dtype=torch.bool, device=dev)
        self.ev0 = torch.cuda.Event(enable_timing=True)
        self.ev1 = torch.cuda.Event(enable_timing=True)

    def preroll(self):
        """back to the recorded state after W steps (the contract's W untimed warm-up steps)"""
        self.env.reset_raw()
        if self.W:
            self.env.step_many(self.actions[:self.W])

    def timed_steps(self):
        env, K, W, mode = self.env, self.K, self.W, self.mode
        if mode == "replay":
            env.step_many(self.actions[W:])
        elif mode == "gym":
            for t in range(K):
                env.step_observe_raw(self.actions[W + t])
        elif mode == "gym-default":
            for t in range(K):                                    # a gym loop: the names are rebound every step
                obs, reward, terminated, truncated, info = env.step(self.actions[W + t])
        elif mode == "policy":
            for t in range(K):
                env.step_raw(env.sample_actions())
        elif mode == "random":
            for t in range(K):
                env.step_random()
        else:                                                     # random-fused: exactly K steps, <= fused_T per launch
            done = 0
            while done < K:
                t = min(self.fused_T, K - done)
                env.step_random_many(t, actions_out=self.f_act[:t], reward=self.f_rew[:t], terminated=self.f_term[:t])
                done += t

    def region(self, barrier):
        """One bracket = barrier + synchronize, W untimed steps, EXACTLY K timed steps, synchronize +
        barrier.  The W warm-up launches are enqueued right in front of the timed ones without a
        host synchronise in between: the stream is then still busy when the K timed launches are
        queued, so the HIP events around them read the K kernels back to back — what rocprofv3's
        kernel trace shows for the same dispatches — and not the host's first-launch latency onto
        an idle stream (a fixed ~5-15 us per region, 3-10 % at K = 20)."""
        torch = self.torch
        torch.cuda.synchronize(self.dev)
        barrier()
        t0 = time.perf_counter()
        self.preroll()
        self.ev0.record()
        self.timed_steps()
        self.ev1.record()
        torch.cuda.synchronize(self.dev)
        t1 = time.perf_counter()
        barrier()
        return self.ev0.elapsed_time(self.ev1) * 1e-3, (t1 - t0) * self.K / float(self.K + self.W)

    def measure(self, barrier, all_max, regions=0, target_s=TARGET_TIMED_S, min_regions=5, max_regions=MAX_REGIONS):
        pilot_ev, _ = self.region(barrier)                        # also the first-touch / clock ramp pass
        pilot_ev = all_max(pilot_ev)
        R = regions if regions > 0 else int(min(max_regions, max(min_regions, math.ceil(target_s / max(pilot_ev, 1e-6)))))
        ev_s, wall_s = [], []
        for _ in range(R):
            e, w = self.region(barrier)
            ev_s.append(e)
            wall_s.append(w)
        self.replay_ok = bool(self.torch.equal(self.env.state, self.final_state))
        # this rank's own numbers (reported per rank in the N > 1 line) ...
        self.rank_median_s, self.rank_best_s, self.rank_wall_s = median(ev_s), min(ev_s), median(wall_s)
        # ... and the MAX over ranks of each rank's median region: what `value` is made from
        return all_max(self.rank_median_s), all_max(self.rank_best_s), all_max(self.rank_wall_s), R


def run_legs(torch, dev, args):
    """The other BASELINE configurations and modes, same process, same clock (N = 1 only)."""
    legs = []
    no_barrier = lambda: None
    ident = lambda x: float(x)

    def step_leg(name, B, mode, K, W, fused_T=64, **kw):
        w = Workload(torch, dev, B, K, W, args.seed, 0, mode, fused_T)
        ev, ev_min, _, R = w.measure(no_barrier, ident, target_s=0.02, max_regions=100, **kw)
        us = ev / K * 1e6
        leg = {"name": name, "boards": B, "mode": mode, "kernel": kernel_label(mode, *w.shape, boards=B), "steps": K, "warmup": W,
               "regions": R, "us_per_step": us, "best_region_us_per_step": ev_min / K * 1e6, "steps_per_s": B * K / ev,
               "algorithmic_bytes_per_board_step": w.algo_bytes, "achieved_GBps": w.algo_bytes * B / (ev / K) / 1e9,
               "frac": w.algo_bytes * B / (ev / K) / 1e9 / HBM_PEAK_GBS, "bound": "hbm",
               "replay_matches_recording": w.replay_ok}
        if mode == "random-fused":
            # VALU-bound (the boards never leave the registers): SQ_INSTS_VALU per board-step (a committed PMC
            # figure, not measured in this run) against the chip's best wave-instruction issue rate
            leg.update(bound="valu", steps_per_launch=min(fused_T, K), us_per_launch=us * min(fused_T, K),
                       valu_insts_per_board_step=FUSED_VALU_PER_STEP, valu_source="profiles/r05/pmc_sq_fused_summary.csv",
                       valu_peak_ginst_per_s=VALU_PEAK_GINST,
                       valu_frac=None if FUSED_VALU_PER_STEP is None else
                       FUSED_VALU_PER_STEP * (B / 64.0) / (ev / K * 1e9) / VALU_PEAK_GINST,
                       issue_ns_per_wave_ply=FUSED_ISSUE_NS_PER_WAVE_PLY, issue_source="tools/isa_budget.py --loop",
                       issue_frac=FUSED_ISSUE_NS_PER_WAVE_PLY * (B / 64.0 / 1024.0) / (ev / K * 1e9))
        else:
            leg["us_per_launch"] = us
        legs.append(leg)
        del w
        torch.cuda.empty_cache()

    step_leg("config2_4096_boards", 4096, "replay", 200, 20)
    step_leg("config3_262144_boards", 262144, "replay", 200, 20)
    step_leg("beyond_infinity_cache_16777216_boards", 16777216, "replay", 10, 2, min_regions=5)
    step_leg("gym_1048576_boards", 1 << 20, "gym", 100, 10)
    legs.append(gym_default_leg(torch, dev, args))
    step_leg("random_1048576_boards", 1 << 20, "random", 100, 10)
    step_leg("random_fused_1048576_boards", 1 << 20, "random-fused", 128, 10)
    step_leg("random_fused_262144_boards", 262144, "random-fused", 128, 10)
    step_leg("random_fused_4096_boards", 4096, "random-fused", 128, 10)
    legs.append(config5_leg(torch, dev, args))
    legs.extend(row_legs(torch, dev, args))
    return legs


def gym_default_leg(torch, dev, args, n=1 << 20, K=100, W=10):
    """The call BASELINE.json's north_star names — `obs, reward, terminated, truncated, info = env.step(actions)`, the
    DEFAULT VecEnv.step() (env.py:34-53): fresh output tensors every call, written by the one fused kernel itself.
    Three clocks, all HIP events on the launch stream: the region clock of every step leg (W launches queued ahead, K
    timed Python calls: the slower of host and device), the same calls captured in a hipGraph and replayed
    (device-paced: the kernel's own rate), and K eager calls onto an idle stream."""
    w = Workload(torch, dev, n, K, W, args.seed, 0, "gym-default", 64)
    ev, ev_min, _, R = w.measure(lambda: None, float, target_s=0.02, max_regions=100)
    us = ev / K * 1e6
    env, a = w.env, w.actions[W]
    dev_paced, eager = time_calls(torch, dev, lambda: env.step(a), 50, 8)
    dp = median(dev_paced)
    # the same loop with VecEnv(output_pool=4): output sets the caller has dropped are re-used instead of allocated
    from qtttgym_amd import VecEnv, vec_env
    pooled = VecEnv(n, device=dev, seed=args.seed, auto_reset=True, output_pool=4)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pooled_us = []
    for _ in range(8):
        for t in range(W):
            out = pooled.step(w.actions[t])
        e0.record()
        for t in range(W, W + K):
            out = pooled.step(w.actions[t])
        e1.record()
        torch.cuda.synchronize(dev)
        pooled_us.append(e0.elapsed_time(e1) * 1e3 / K)
    del pooled, out
    leg = {"name": "gym_default_1048576_boards", "boards": n, "mode": "gym-default", "kernel": kernel_label("gym-default", *w.shape, boards=n),
           "call": "VecEnv.step(actions) -> (obs, reward, terminated, truncated, info), fresh tensors every call (copy_obs=True, output_pool=0: the defaults)",
           "steps": K, "warmup": W, "regions": R,
           "us_per_step": us, "best_region_us_per_step": ev_min / K * 1e6,
           "device_paced_us_per_step": dp, "us_per_python_call_idle_stream": median(eager),
           "timing": "us_per_step: region clock (W launches queued ahead of K timed Python calls); device_paced: hipGraph of %d "
                     "default step() calls replayed; us_per_python_call_idle_stream: K eager calls, events around them" % GRAPH_LAUNCHES,
           "steps_per_s": n * K / ev, "outputs": "one fresh allocation per call (torch's caching allocator), carved into the eight tensors by "
           + ("qtttgym_amd/_fastviews.so" if vec_env._fastviews is not None else "Python-level torch calls (_fastviews.so not built)"),
           "us_per_step_with_output_pool_4": median(pooled_us),
           "algorithmic_bytes_per_board_step": w.algo_bytes, "achieved_GBps": w.algo_bytes * n / (dp * 1e-6) / 1e9,
           "frac": w.algo_bytes * n / (dp * 1e-6) / 1e9 / HBM_PEAK_GBS, "frac_region_clock": w.algo_bytes * n / (ev / K) / 1e9 / HBM_PEAK_GBS,
           "bound": "hbm", "replay_matches_recording": w.replay_ok}
    del w
    torch.cuda.empty_cache()
    return leg


def row_legs(torch, dev, args, n=1 << 20, K=20, regions=5):
    """The kernels beside the step at 1 048 576 boards (SURVEY §8(f) rows and the cold paths), through VecEnv with
    reused buffers, each timed on its own: us per launch, algorithmic bytes per board, fraction of the HBM spec."""
    from qtttgym_amd import VecEnv
    env = VecEnv(n, device=dev, seed=args.seed)
    for _ in range(5):                                         # mid-game boards
        env.step_raw(env.sample_actions())
    sb = env.state.numel() // ((n + 63) // 64 * 64)
    act = torch.randint(0, 36, (n,), dtype=torch.uint8, device=dev)
    obs, ex, ni, cw, tn = env.observ(), env.export_boards(), env.node_info(python_key=False), env.check_win(), env.turn()
    ni_py, sk = env.node_info(python_key=True), env.state_keys()
    xp, xp_py, ro, enc = env.expand(act, python_key=False), env.expand(act, python_key=True), env.rollout(), env.encode()
    env_in = VecEnv(n, device=dev, seed=args.seed)             # the target of the import row
    env_rs = VecEnv(n, device=dev, seed=args.seed)             # the target of the reset row
    rows = [
        # Env.reset with its observation (env.py:55-57), one launch of seven fills: 16 + 30 bytes written per board
        ("reset_observe", "reset_observe_kernel", lambda: env_rs.reset(copy_obs=False), sb + 30, "hbm (write)"),
        ("observe", "observe_kernel", lambda: env.observ(), sb + 30, "hbm"),
        ("export", "export_kernel", lambda: env.export_boards(out=ex), sb + 37, "hbm"),
        ("import", "import_kernel", lambda: env_in.import_boards(ex["moves"], ex["n_moves"], ex["board"], ex["qmask"], ex["n_q"]),
         sb + 37, "valu (tree rooting) + hbm"),
        ("turn", "export_kernel (n_moves only)", lambda: env.turn(out=tn), sb // 2 + 1, "launch"),
        ("check_win", "check_win_kernel", lambda: env.check_win(out=cw), sb // 2 + 2, "launch"),
        # winner + terminal + legal + the native position key (state_key): what a device-side search asks for
        ("node_info", "node_info_kernel<1024, false>", lambda: env.node_info(out=ni), sb + 18, "hbm"),
        ("state_keys", "node_info_kernel<1024, false> (state_key alone)", lambda: env.state_keys(out=sk), sb + 8, "hbm"),
        # the same + CPython's hash(tuple(board) + tuple(moves)) bit for bit, for host-side dicts of reference code
        ("node_info_python_key", "node_info_kernel<1024, true>", lambda: env.node_info(out=ni_py), sb + 26, "valu (CPython tuple hash)"),
        ("expand", "expand_kernel<1024, false>", lambda: env.expand(act, out=xp), sb + 1 + 2 * sb + 1 + 2 * 18, "hbm + valu (a step with two path reversals)"),
        ("expand_python_key", "expand_kernel<1024, true>", lambda: env.expand(act, out=xp_py), sb + 1 + 2 * sb + 1 + 2 * 26,
         "valu (the same + two tuple hashes)"),
        ("rollout", "rollout_kernel", lambda: env.rollout(out=ro), sb + 2, "valu (playouts to the end from boards five plies deep)"),
        ("encode", "encode_kernel", lambda: env.encode(out=enc), sb + 720 + 36, "hbm (write)"),
    ]
    out = []
    for name, kernel, fn, algo, bound in rows:
        us, us_eager = time_calls(torch, dev, fn, K, regions)
        u = median(us)
        out.append({"name": "row_%s_1048576_boards" % name, "boards": n, "mode": "row", "kernel": kernel, "steps": K,
                    "regions": regions, "us_per_launch": u, "best_region_us_per_launch": min(us),
                    "timing": "hipGraph of %d launches replayed (device-paced); us_per_python_call = the same calls eager" % max(K, GRAPH_LAUNCHES),
                    "us_per_python_call": median(us_eager),
                    "algorithmic_bytes_per_board": algo, "achieved_GBps": algo * n / (u * 1e-6) / 1e9,
                    "frac": algo * n / (u * 1e-6) / 1e9 / HBM_PEAK_GBS, "bound": bound})
    return out


# SQ_INSTS_VALU per board-step of step_random_fused_kernel<256, true, false, false>: 139 538 125 per dispatch of 1 048 576
# boards x 64 steps = 8 516.7 per wave = 133.1 per ply (profiles/r05/pmc_sq_fused_summary.csv; rocprofv3 --pmc, its own
# pass; round 4: 148.3); SQ_INSTS_SALU 30.4 executed per wave and ply (the plies' keys travel as a kernel argument)
FUSED_VALU_PER_STEP = 133.1
# issue time of that instruction mix per wave and ply (tools/isa_budget.py --loop, profiles/r05/isa_budget_random_fused.txt:
# 65.0 of the 133.1 in the fast class at 1.03 ns per instruction per SIMD, 68.1 in the slow class at 1.75 — a literal
# operand does not make a logic instruction slow, an SGPR operand does: profiles/r02/valu_rates.txt): what a SIMD needs
# per resident wave and ply when it never idles
FUSED_ISSUE_NS_PER_WAVE_PLY = 186.1


def config5_leg(torch, dev, args, n=65536, K=50):
    """BASELINE config 5's unit: one MCTS rollout below each of 65 536 selected nodes (mcts.py:166-176: select ->
    _expand_child -> num_simulations x _simulate FROM THE LEAF -> the value for _backpropogate; :210-221,233-267) =
    qttt_expand_rollout, ONE launch: the expansion, both children's bookkeeping (winner, terminal, legal mask, native
    position key) and the playouts from each child, through VecEnv with reused buffers.  Beside it the same unit as
    three launches (expand, then rollout_many on each child buffer) — what round 3 had, now playing out the children."""
    from qtttgym_amd import VecEnv
    env = VecEnv(n, device=dev, seed=args.seed)
    for _ in range(4):
        env.step_raw(env.sample_actions())
    act = torch.randint(0, 36, (n,), dtype=torch.uint8, device=dev)
    x1, x10 = env.expand_rollout(act, 1), env.expand_rollout(act, 10)
    ex = env.expand(act, python_key=False)
    r1 = (ex["child0"].rollout_many(1), ex["child1"].rollout_many(1))
    r10 = (ex["child0"].rollout_many(10), ex["child1"].rollout_many(10))
    def three(S, r):
        env.expand(act, out=ex)
        ex["child0"].rollout_many(S, step_idx0=0, out=r[0])
        ex["child1"].rollout_many(S, step_idx0=16 * S, out=r[1])
    us1, us1_eager = time_calls(torch, dev, lambda: env.expand_rollout(act, 1, out=x1), K, 8)
    us10, us10_eager = time_calls(torch, dev, lambda: env.expand_rollout(act, 10, out=x10), K, 8)
    us1_3, us1_3_eager = time_calls(torch, dev, lambda: three(1, r1), K, 8)
    us10_3, us10_3_eager = time_calls(torch, dev, lambda: three(10, r10), K, 8)
    sb = env.state.numel() // ((n + 63) // 64 * 64)
    # state + action in; two children, n_children, per-child winner / terminal / legal / state_key and value_sum out
    algo = sb + 1 + 2 * sb + 1 + 2 * (1 + 1 + 8 + 8 + 4)
    u = median(us1)
    return {"name": "config5_expand_rollout_65536_pairs", "boards": n, "mode": "mcts-unit",
            "kernel": "expand_rollout_kernel<256, false> (one playout per child: a lane per (pair, simulation, child)); "
                      "expand_rollout_jobs_kernel<256, false> (ten: the children that exist dealt to the lanes)",
            "steps": K, "regions": len(us1),
            "us_per_unit": u, "best_region_us_per_unit": min(us1), "expansions_per_s": n / (u * 1e-6),
            "us_per_unit_with_10_playouts_per_leaf": median(us10),
            "playouts_per_s_10_per_leaf": 10 * float((x10["n_children"].to(torch.int64)).sum()) / (median(us10) * 1e-6),
            "children_per_pair": float(x10["n_children"].to(torch.float32).mean()),
            "us_per_unit_as_three_launches": median(us1_3), "us_per_unit_as_three_launches_10_playouts": median(us10_3),
            "timing": "hipGraph of %d units replayed (device-paced); *_python_call = the same calls eager" % max(K, GRAPH_LAUNCHES),
            "us_per_unit_python_call": median(us1_eager), "us_per_unit_with_10_playouts_python_call": median(us10_eager),
            "us_per_unit_as_three_launches_python_calls": median(us1_3_eager),
            "us_per_unit_as_three_launches_10_playouts_python_calls": median(us10_3_eager),
            "algorithmic_bytes_per_board_unit": algo, "achieved_GBps": algo * n / (u * 1e-6) / 1e9,
            "frac": algo * n / (u * 1e-6) / 1e9 / HBM_PEAK_GBS,
            "bound": "launch + valu (one launch; one lane per (pair, simulation, child))"}


# ---------------------------------------------------------------------------- one rank
CPU_AFFINITY = None


def bind_cpu(n_dev):
    """Bind this rank to the cores local to its GPU (sysfs + sched_setaffinity only: nothing here touches HIP; n_dev comes
    from torch.cuda.device_count(), which does not initialise the GPU).  QTTT_BENCH_NO_BIND=1 skips it.  The result goes
    into config.cpu_affinity."""
    global CPU_AFFINITY
    from qtttgym_amd.affinity import bind_to_gpu
    if os.environ.get("QTTT_BENCH_NO_BIND") == "1":
        CPU_AFFINITY = {"bound": False, "reason": "QTTT_BENCH_NO_BIND=1"}
        return
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # (several gloo ranks rehearsing on one card all use device local_rank % n_dev)
    CPU_AFFINITY = bind_to_gpu(local_rank % max(n_dev, 1), expected_devices=n_dev)


def run(args):
    import torch
    import torch.distributed as dist
    from qtttgym_amd import _native
    from qtttgym_amd.dist import shard_range

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    # one rank per GPU; QTTT_DIST_BACKEND=gloo lets several ranks rehearse on one GPU (plumbing only)
    backend = os.environ.get("QTTT_DIST_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if n_dev < 1:
        raise SystemExit("no HIP device visible; bench.py has no CPU path")
    global ALLOWED_CPUS
    ALLOWED_CPUS = os.sched_getaffinity(0)
    bind_cpu(n_dev)                  # before anything initialises HIP (device_count() does not)
    affinity = CPU_AFFINITY
    if backend == "nccl" and world > n_dev:
        raise SystemExit("--gpus %d needs %d GPUs, %d visible (QTTT_DIST_BACKEND=gloo rehearses several "
                         "ranks on one GPU)" % (world, world, n_dev))
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if not affinity.get("bound") and os.environ.get("QTTT_BENCH_NO_BIND") != "1":
        # the topology was not readable before the runtime came up (the containers of this pool): ask the runtime for
        # the device's PCI address now and bind the launching thread to that device's cores
        try:
            from qtttgym_amd.affinity import bind_to_pci
            pr = torch.cuda.get_device_properties(dev)
            first = affinity.get("reason")
            affinity = bind_to_pci("%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
            affinity["before_runtime"] = first
        except Exception as e:                                   # noqa: BLE001 — never take the run down
            affinity = dict(affinity, pci_error="%s: %s" % (type(e).__name__, e))
    # QTTT_DIST_FORCE=1: build the process group and run every collective of the N > 1 path with one
    # rank too (a one-GPU box can then exercise the RCCL branch; tests/test_bench_contract_gpu.py)
    use_dist = world > 1 or os.environ.get("QTTT_DIST_FORCE") == "1"
    # Two planes.  CONTROL = the clock's barrier and MAX, the rank count, the per-rank timings, the episode counters, the
    # agreement flags: a few CPU scalars over gloo on the loopback interface.  DATA = the one real exchange of the path, the
    # gather of per-board returns: device tensors over RCCL (xGMI), brought up lazily by its first collective — inside the
    # watchdog below, so that an RCCL that fails OR HANGS on a node this code has never seen costs the line its
    # `returns_gather` entry and nothing else.  QTTT_BENCH_CONTROL=nccl puts the control plane on RCCL too (the layout
    # of rounds 1 - 4: eager communicator, device tensors everywhere).
    control = "nccl" if (backend == "nccl" and os.environ.get("QTTT_BENCH_CONTROL") == "nccl") else "gloo"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ["MASTER_ADDR"] in ("127.0.0.1", "localhost"):
            os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node by contract; the hostname may not resolve
        if world == 1 and "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port())
        if backend == "nccl" and control == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        elif backend == "nccl":
            dist.init_process_group("cpu:gloo,cuda:nccl", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    cpu = torch.device("cpu")
    coll_dev = dev if control == "nccl" else cpu                    # control plane
    data_dev = dev if backend == "nccl" else cpu                    # the returns gather

    def barrier():
        if use_dist:
            if control == "nccl":
                dist.barrier(device_ids=[dev_index])
            else:
                dist.all_reduce(torch.zeros(1, dtype=torch.int32))  # a CPU all_reduce over gloo is a barrier

    def all_max(x):
        if not use_dist:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t[0])

    # how many ranks the process group really has (RCCL's own count, not the environment's)
    ranks_seen = 1
    if use_dist:
        ones = torch.ones((), dtype=torch.int64, device=coll_dev)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        ranks_seen = int(ones)

    K, W = args.steps, args.warmup
    strong = args.total_boards > 0
    if strong:                                   # BASELINE config 4's shape: a fixed total, sharded
        lo, hi = shard_range(args.total_boards, rank, world)
        B, offset, total = hi - lo, lo, args.total_boards
    else:
        B, offset, total = args.boards, rank * args.boards, args.boards * world
    # Everything of this run is enqueued on ONE created stream, made the current one (the library launches on the caller's
    # current stream; the HIP events of the clock are recorded on it): the legacy default stream costs 0.1 us per launch
    # more at K = 20 (7.21 - 7.24 against 7.07 - 7.12 us, 7.05 - 7.23 against 7.00 - 7.16 at K = 1000; same box, alternating,
    # profiles/r05/bench_stream_ab.txt).  QTTT_BENCH_DEFAULT_STREAM=1 keeps the default stream.
    own_stream = os.environ.get("QTTT_BENCH_DEFAULT_STREAM") != "1"
    if own_stream:
        torch.cuda.set_stream(torch.cuda.Stream(device=dev))
    wl = Workload(torch, dev, B, K, W, args.seed, offset, args.mode, args.fused_steps)
    env, state_bytes, algo_bytes = wl.env, wl.state_bytes, wl.algo_bytes
    ev_med, ev_min, wall_med, R = wl.measure(barrier, all_max, regions=args.regions)
    replay_ok = wl.replay_ok
    term_count, win_count = wl.term_count, wl.win_count
    from qtttgym_amd.dist import gather_rank_values, agree
    # what each rank measured by itself (value / ms_per_step above are the MAX over ranks of the medians): the first
    # thing to look at when N GPUs give less than N x.  One all_gather of three doubles, the same kind of collective as
    # the all_max inside measure(); a failure leaves None in the line, never takes the value with it.
    per_rank = None
    try:
        rows = gather_rank_values([wl.rank_median_s, wl.rank_best_s, wl.rank_wall_s], coll_dev if use_dist else None)
        per_rank = {"ms_per_step": [r[0] / K * 1e3 for r in rows], "best_region_ms_per_step": [r[1] / K * 1e3 for r in rows],
                    "host_wall_ms_per_step": [r[2] * 1e3 / K for r in rows]}
    except Exception as e:                                       # noqa: BLE001
        sys.stderr.write("bench.py: per-rank timing all_gather failed on rank %d: %r\n" % (rank, e))

    gather = None
    hung = False
    if use_dist:
        # Everything below is OPTIONAL bookkeeping off the timed path (the step has no collective): a failure of
        # any of it — e.g. the very first RCCL gather on a node this code has never seen — is reported inside the
        # line and never takes the scaling value down with it.
        # episode counters: once per run
        # Each optional collective is entered by ALL ranks or by none: a rank that fails while PREPARING one says so in a
        # one-flag all_reduce(MIN) first (dist.agree), so the others do not block in a gather it never joins.  (A rank
        # that dies INSIDE a collective cannot be helped from here: bench.py's launcher deadline ends the job.)
        try:
            cnt = torch.stack([term_count, win_count]).to(coll_dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            term_count, win_count = cnt[0], cnt[1]
        except Exception as e:                                   # noqa: BLE001
            sys.stderr.write("bench.py: episode-counter all_reduce failed on rank %d: %r\n" % (rank, e))
        # and a gather of per-board returns (4 B per board) to rank 0, timed on its own (RCCL over xGMI on a
        # multi-GPU node; never part of a step).  QTTT_BENCH_NO_GATHER=1 skips it.
        if os.environ.get("QTTT_BENCH_NO_GATHER") == "1":
            gather = {"skipped": "QTTT_BENCH_NO_GATHER=1"}
        else:
            ret, prep_error = None, None
            try:
                from qtttgym_amd.dist import gather_returns
                fail = os.environ.get("QTTT_BENCH_FAIL_GATHER")          # test hooks (tests/test_bench_contract_gpu.py, tests/test_bench_line_gpu.py)
                if fail == "1" or (fail is not None and fail.startswith("rank") and int(fail[4:]) == rank):
                    raise RuntimeError("QTTT_BENCH_FAIL_GATHER=%s: injected failure of the returns gather" % fail)
                ret = env._reward.clone().to(data_dev)
            except Exception as e:                               # noqa: BLE001
                prep_error = e
                sys.stderr.write("bench.py: returns gather could not be prepared on rank %d: %r\n" % (rank, e))

            def exchange():
                if not agree(prep_error is None, coll_dev):
                    raise RuntimeError("skipped on every rank: %s" % ("this rank failed to prepare it: %s: %s"
                                       % (type(prep_error).__name__, prep_error) if prep_error is not None else
                                       "another rank failed to prepare it"))
                hang = os.environ.get("QTTT_BENCH_HANG_GATHER")             # test hook: this rank never joins the exchange
                if hang is not None and hang.startswith("rank") and int(hang[4:]) == rank:
                    time.sleep(3600)
                t_up = time.perf_counter()
                if backend == "nccl":                                # the first device collective brings RCCL up
                    dist.all_reduce(torch.ones(1, dtype=torch.int32, device=dev))
                    torch.cuda.synchronize(dev)
                up_ms = (time.perf_counter() - t_up) * 1e3
                gather_returns(ret, dst=0)                           # untimed: connection set-up of the gather
                torch.cuda.synchronize(dev)
                barrier()
                tg0 = time.perf_counter()
                gathered = gather_returns(ret, dst=0)
                torch.cuda.synchronize(dev)
                tg = all_max(time.perf_counter() - tg0)
                return {"ms": tg * 1e3, "bytes_per_rank": 4 * B, "backend": backend, "bring_up_ms": up_ms,
                        "boards_gathered": None if gathered is None else int(gathered.numel())}

            # the exchange runs under a deadline: a collective that never returns must not take the measured value with it
            import threading
            limit = float(os.environ.get("QTTT_BENCH_GATHER_TIMEOUT", "120"))
            box = {}

            def guarded():
                try:
                    box["value"] = exchange()
                except BaseException as e:                           # noqa: BLE001
                    box["error"] = e
            th = threading.Thread(target=guarded, daemon=True)
            th.start()
            th.join(limit)
            if th.is_alive():
                hung = True
                sys.stderr.write("bench.py: returns gather still running after %.0f s on rank %d: reported as timed out; "
                                 "the process group is abandoned\n" % (limit, rank))
                gather = {"error": "TimeoutError: no answer within %.0f s (QTTT_BENCH_GATHER_TIMEOUT)" % limit, "backend": backend}
            elif "error" in box:
                e = box["error"]
                sys.stderr.write("bench.py: returns gather failed on rank %d: %r\n" % (rank, e))
                gather = {"error": "%s: %s" % (type(e).__name__, e), "backend": backend}
            else:
                gather = box["value"]

    rc = 0
    if rank == 0:
        if ranks_seen != args.gpus:
            print("bench.py: refusing to report n_gpus=%d: the process group has %d ranks" % (args.gpus, ranks_seen),
                  file=sys.stderr)
            rc = 3
        else:
            launch_s = ev_med / K                                          # per env step (= per launch unless fused)
            value = total * K / ev_med
            achieved = algo_bytes * B / launch_s / 1e9
            bpl, blk = wl.shape                                            # the library's own choice for this batch
            gym = args.mode in ("gym", "gym-default")
            traffic, traffic_fresh = (pmc_traffic_per_launch(B, state_bytes) if args.mode == "replay" else
                                      pmc_traffic_per_launch(B, state_bytes, "+" + args.mode) if gym else (None, None))
            what = WHAT[args.mode] % args.fused_steps if args.mode == "random-fused" else WHAT[args.mode]
            out = {
                "metric": "env_steps_per_sec", "value": value, "unit": "steps/s", "n_gpus": world,
                "ranks_seen": ranks_seen, "steps": K, "warmup": W, "ms_per_step": launch_s * 1e3,
                "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "u64",
                "data": "synthetic",
                "clock": "HIP events on the launch stream around the K launches; median of %d regions "
                         "(max over ranks); value, ms_per_step and roofline all use it" % R,
                "regions": R, "host_wall_ms_per_step": wall_med * 1e3 / K, "returns_gather": gather,
                "best_region_ms_per_step": ev_min * 1e3 / K,
                # each rank's own median / best region and host wall time per step, in rank order, and who set the value
                "per_rank_ms_per_step": None if per_rank is None else per_rank["ms_per_step"],
                "per_rank_best_region_ms_per_step": None if per_rank is None else per_rank["best_region_ms_per_step"],
                "per_rank_host_wall_ms_per_step": None if per_rank is None else per_rank["host_wall_ms_per_step"],
                "slowest_rank": None if per_rank is None else max(range(len(per_rank["ms_per_step"])), key=lambda r: per_rank["ms_per_step"][r]),
                "rank_spread": None if per_rank is None else max(per_rank["ms_per_step"]) / max(min(per_rank["ms_per_step"]), 1e-12),
                "config": {"workload": "%d boards %s, uniform-legal random policy, auto-reset, %s"
                                       % (total if strong else B, "sharded over %d GPUs" % world if strong else "per GPU", what),
                           "boards_per_gpu": B, "boards_total": total, "state_bytes_per_board": state_bytes,
                           "parallelism": "shard%d" % world, "mode": args.mode,
                           "dist_backend": backend if use_dist else None,
                           "control_backend": control if use_dist else None,
                           "self_launched": bool(os.environ.get("QTTT_BENCH_SELF_LAUNCHED")),
                           "hip_force_dev_kernarg": os.environ.get("HIP_FORCE_DEV_KERNARG"),
                           "stream": "created, current" if own_stream else "legacy default",
                           "cpu_affinity": affinity,                      # rank 0's (every rank binds to its own GPU's cores)
                           "board_offset_last_rank": shard_range(total, world - 1, world)[0] if strong else (world - 1) * B,
                           "replay_matches_recording": replay_ok,
                           "episodes_finished": int(term_count), "steps_with_line": int(win_count)},
                "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                             "traffic_source": None if traffic is None else PMC_SUMMARY,
                             "traffic_measured_on_this_build": traffic_fresh,
                             "kernel": kernel_label(args.mode, bpl, blk, boards=B), "launch_us": launch_s * 1e6,
                             "algorithmic_bytes_per_board_step": algo_bytes,
                             "algorithmic_bytes_per_launch": algo_bytes * B,
                             # the guide's measured float4-copy rate: what a kernel that moves only its algorithmic
                             # bytes can reach on this part
                             "achievable_peak": HBM_ACHIEVABLE_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                             # 1 048 576 boards x 16 B of state live in the 256 MB Infinity Cache: `achieved` above is fabric
                             # traffic, not HBM traffic.  The same kernel with a working set that cannot be cached
                             # (the 16 M-board leg, filled in below when the legs run):
                             "state_resident_in_infinity_cache": bool(B * algo_bytes <= 256 << 20),
                             # scalars (a driver that keeps only scalar fields of this object still sees them)
                             "beyond_cache_boards": None, "beyond_cache_launch_us": None, "beyond_cache_frac": None,
                             "beyond_cache_frac_of_achievable": None,
                             "beyond_cache": None},
            }
            actions = wl.actions
            del wl
            if world == 1 and not args.no_legs:
                out["legs"] = run_legs(torch, dev, args)
                big = [l for l in out["legs"] if l["name"].startswith("beyond_infinity_cache")]
                if big:
                    out["roofline"]["beyond_cache"] = {
                        "boards": big[0]["boards"], "launch_us": big[0]["us_per_step"], "achieved": big[0]["achieved_GBps"],
                        "frac": big[0]["frac"], "frac_of_achievable": big[0]["achieved_GBps"] / HBM_ACHIEVABLE_GBS,
                        "kernel": big[0]["kernel"], "working_set_MB": big[0]["boards"] * state_bytes / 1e6}
                    out["roofline"].update(beyond_cache_boards=big[0]["boards"], beyond_cache_launch_us=big[0]["us_per_step"],
                                           beyond_cache_frac=big[0]["frac"],
                                           beyond_cache_frac_of_achievable=big[0]["achieved_GBps"] / HBM_ACHIEVABLE_GBS)
            if not args.no_cpu_baseline and world == 1:                 # rank 0 at N = 1 only
                # bounded sample: the first <=256 recorded steps of every board of rank 0, ~14 s of CPU in all
                t_cpu = min(K + W, 256)
                out["cpu_baseline"] = cpu_baseline(actions[:t_cpu].cpu().numpy(), args.seed, args.cpu_budget)
            if "legs" in out:
                # LAST in the object, so that a stored tail of the line still carries them: BASELINE configs 2 / 3 / 5 and
                # the beyond-cache fraction as plain scalars ("report both")
                L = {l["name"]: l for l in out["legs"]}
                g = lambda name, key: (L[name].get(key) if name in L else None)
                out["configs"] = {
                    "config1_us": launch_s * 1e6, "config1_frac": achieved / HBM_PEAK_GBS,
                    "config2_us": g("config2_4096_boards", "us_per_step"), "config2_frac": g("config2_4096_boards", "frac"),
                    "config3_us": g("config3_262144_boards", "us_per_step"), "config3_frac": g("config3_262144_boards", "frac"),
                    "config3_fused_us": g("random_fused_262144_boards", "us_per_step"),
                    "gym_us": g("gym_1048576_boards", "us_per_step"),
                    "gym_default_us": g("gym_default_1048576_boards", "device_paced_us_per_step"),
                    "gym_default_eager_us": g("gym_default_1048576_boards", "us_per_step"),
                    "config5_us": g("config5_expand_rollout_65536_pairs", "us_per_unit"),
                    "config5_frac": g("config5_expand_rollout_65536_pairs", "frac"),
                    "beyond_cache_us": g("beyond_infinity_cache_16777216_boards", "us_per_step"),
                    "beyond_cache_frac": g("beyond_infinity_cache_16777216_boards", "frac")}
            print(json.dumps(out), flush=True)
    if hung:
        # a collective of the exchange is still blocked in its thread: no further collective, no destroy — leave at once
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(rc if replay_ok else 1)
    if use_dist:
        barrier()
        dist.destroy_process_group()
    if not replay_ok:
        raise SystemExit("replay diverged from the recording")
    return rc


def ensure_built():
    """A fresh checkout has no libqttt_hip.so (built artefacts are not in the history): build it once, under a file
    lock so that the N ranks of a launcher do not compile at the same time.  No GPU is touched here."""
    lib = os.path.join(ROOT, "qtttgym_amd", "libqttt_hip.so")
    if os.path.exists(lib):
        return
    import fcntl
    with open(os.path.join(ROOT, "qtttgym_amd", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not os.path.exists(lib):
                import __graft_entry__ as g
                g.build_hip()
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def main():
    args = parse_args()
    ensure_built()
    if args.gpus < 1 or args.steps < 1 or args.warmup < 0 or args.boards < 1 or args.total_boards < 0 \
            or not 1 <= args.fused_steps <= 64 or (args.total_boards and args.total_boards < args.gpus):
        raise SystemExit("need --gpus >= 1, --steps >= 1, --warmup >= 0, --boards >= 1, --total-boards >= --gpus, "
                         "1 <= --fused-steps <= 64 (a fused launch holds at most 64 plies)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))          # before anything imports torch or touches the GPU
    sys.exit(run(args))


if __name__ == "__main__":
    main()
