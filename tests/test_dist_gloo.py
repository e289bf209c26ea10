"""N>1 path on CPU: world_size-2 gloo.  The sharding arithmetic, the global-board-id keying and
the two collectives (counter all_reduce, returns gather) are exercised with the ORACLE standing in
for the HIP stepper (tests only — the product has no CPU stepper)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

N_TOTAL = 4099          # odd on purpose: ragged shards
STEPS = 12
SEED = 77


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run_shard(lo, hi):
    import oracle
    from qtttgym_amd.dist import EpisodeCounters
    n = hi - lo
    ob = oracle.OracleBoards(n)
    counters = EpisodeCounters("cpu")
    returns = torch.zeros(n, dtype=torch.float32)
    for t in range(STEPS):
        a = ob.sample_actions(SEED, t, lo, True)
        r, tm = ob.step(a, None, SEED, t, lo, True)
        r, tm = torch.from_numpy(r.copy()), torch.from_numpy(tm.copy()).bool()
        counters.update(r, tm)
        returns += r
    return ob, counters, returns


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from qtttgym_amd.dist import init_from_env, shard_range, gather_returns
    r, _, w = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    lo, hi = shard_range(N_TOTAL, rank, world)
    ob, counters, returns = _run_shard(lo, hi)
    total = counters.all_reduce().clone()
    gathered = gather_returns(returns, dst=0)
    q.put((rank, lo, hi, total.tolist(), None if gathered is None else gathered.numpy(),
           ob.board.copy(), ob.n_moves.copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_partitions():
    from qtttgym_amd.dist import shard_range
    for n, w in [(10, 3), (4099, 2), (8, 8), (5, 8), (2097152, 8)]:
        edges = [shard_range(n, r, w) for r in range(w)]
        assert edges[0][0] == 0 and edges[-1][1] == n
        assert all(edges[i][1] == edges[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in edges]
        assert max(sizes) - min(sizes) <= 1
    assert shard_range(2097152, 3, 8) == (3 * 262144, 4 * 262144)   # BASELINE config 4


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    results.sort(key=lambda x: x[0])
    # single-process run over all boards
    ob, counters, returns = _run_shard(0, N_TOTAL)
    for rank, lo, hi, total, gathered, board, n_moves in results:
        assert total == counters.c.tolist()                 # all_reduce == whole-job counters
        assert np.array_equal(board, ob.board[lo:hi])       # shard k == slice of the full run
        assert np.array_equal(n_moves, ob.n_moves[lo:hi])
    assert np.array_equal(results[0][4].view(np.uint32), returns.numpy().view(np.uint32))
    assert results[1][4] is None
    assert counters.c[0] > 0 and counters.c[3] == N_TOTAL * STEPS
