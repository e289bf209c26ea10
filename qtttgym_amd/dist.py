"""Multi-GPU sharding of independent boards.

Boards never interact, so N boards over G ranks is G contiguous shards and **no collective in
step()**.  The global board id (shard offset + local index) keys the collapse-bit / policy hash,
so results do not depend on G.  The only exchange is optional and off the step path: a sum of
episode counters (RCCL all_reduce, a few int64) or a gather of per-board returns (<= 4 B/board)
once per episode.  One process per GPU, `torch.distributed` backend "nccl" (= RCCL over xGMI);
the same helpers run on "gloo" for the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Contiguous shard [lo, hi) of rank; sizes differ by at most one board."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    base, rem = divmod(int(n_total), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).  Returns
    (rank, local_rank, world).  No-op for world == 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, local_rank, world


def make_sharded_env(n_total, rank, world, device, seed=0, auto_reset=False):
    """This rank's shard of an n_total-board environment."""
    from .vec_env import VecEnv
    lo, hi = shard_range(n_total, rank, world)
    return VecEnv(hi - lo, device=device, seed=seed, auto_reset=auto_reset, board_offset=lo)


class EpisodeCounters:
    """Running int64 counters of finished episodes, kept on the device of the step outputs.

    [0] episodes finished, [1] finished with a completed line (reward -1.0, env.py:49),
    [2] finished without one (board full), [3] steps taken."""

    def __init__(self, device):
        self.c = torch.zeros(4, dtype=torch.int64, device=device)

    def update(self, reward, terminated):
        term = terminated.to(torch.int64)
        line = (reward != 0).to(torch.int64)      # -1.0 -> 1, -0.0 -> 0
        self.c[0] += term.sum()
        self.c[1] += (term * line).sum()
        self.c[2] += (term * (1 - line)).sum()
        self.c[3] += reward.numel()

    def all_reduce(self):
        """Whole-job totals: one all_reduce(sum) of 4 int64 (RCCL on GPUs, gloo on CPU)."""
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(self.c, op=dist.ReduceOp.SUM)
        return self.c


def gather_returns(local_returns, dst=0):
    """Gathers per-board returns (any dtype, shape [n_local]) to rank `dst` in board order.
    Shards may differ by one board, so they are padded to the largest shard."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return local_returns
    world, rank = dist.get_world_size(), dist.get_rank()
    n_local = torch.tensor([local_returns.numel()], dtype=torch.int64, device=local_returns.device)
    sizes = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(sizes, n_local)
    sizes = [int(s) for s in sizes]
    pad = max(sizes)
    buf = torch.zeros(pad, dtype=local_returns.dtype, device=local_returns.device)
    buf[:local_returns.numel()] = local_returns
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, out, dst=dst)
    if rank != dst:
        return None
    return torch.cat([o[:s] for o, s in zip(out, sizes)])


def gather_rank_values(values, device=None):
    """all_gather of a few floats per rank (timings): returns [[rank 0's values], [rank 1's], ...] on every rank.
    One tiny collective, same shape on every rank; a single-rank job returns [values]."""
    vals = [float(v) for v in values]
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return [vals]
    t = torch.tensor(vals, dtype=torch.float64, device=device if device is not None else "cpu")
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[float(x) for x in o.cpu()] for o in out]


def agree(ok, device=None):
    """True iff EVERY rank passes ok=True (one all_reduce(MIN) of a flag).  Lets the ranks skip an optional collective
    together when one of them failed while preparing it — instead of the others blocking in it."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return bool(ok)
    t = torch.tensor([1 if ok else 0], dtype=torch.int64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t[0]))
