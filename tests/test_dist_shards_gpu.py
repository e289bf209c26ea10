"""N > 1 on the HIP path: two ranks sharing this box's one card (rendezvous over gloo — RCCL refuses two
ranks on one GPU) each step THEIR shard with the fused random-policy kernel; the per-board returns come back through
dist.gather_returns and equal the boards of one single-process run."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DIST_TOTAL, DIST_T, DIST_SEED = 262145, 40, 12            # odd: the shards differ by one board


def _dist_worker(rank, world, port, q):
    import os
    import sys
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist
    from qtttgym_amd.dist import init_from_env, make_sharded_env, gather_returns, EpisodeCounters, shard_range
    init_from_env(backend="gloo")
    env = make_sharded_env(DIST_TOTAL, rank, world, "cuda:0", seed=DIST_SEED, auto_reset=True)
    n = env.num_envs
    assert (env.board_offset, env.board_offset + n) == shard_range(DIST_TOTAL, rank, world)
    r = torch.empty((DIST_T, n), dtype=torch.float32, device="cuda")
    tm = torch.empty((DIST_T, n), dtype=torch.bool, device="cuda")
    ret = torch.zeros(n, dtype=torch.float32, device="cuda")
    env.step_random_many(DIST_T, reward=r, terminated=tm, returns=ret)    # the kernel's own per-board returns
    counters = EpisodeCounters("cuda")
    for t in range(DIST_T):
        counters.update(r[t], tm[t])
    counters.c = counters.c.cpu()                          # gloo reduces host tensors
    total = counters.all_reduce().clone()
    gathered = gather_returns(ret.cpu(), dst=0)
    q.put((rank, total.tolist(), None if gathered is None else gathered.numpy(), env.turn().cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_random_fused_shards_equal_the_single_run_through_returns_gather():
    import socket
    import torch.multiprocessing as mp
    from qtttgym_amd import VecEnv
    from qtttgym_amd.dist import shard_range, EpisodeCounters
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dist_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted((q.get(timeout=400) for _ in range(world)), key=lambda x: x[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    # the single run over all the boards, in this process
    env = VecEnv(DIST_TOTAL, seed=DIST_SEED, auto_reset=True)
    r = torch.empty((DIST_T, DIST_TOTAL), dtype=torch.float32, device="cuda")
    tm = torch.empty((DIST_T, DIST_TOTAL), dtype=torch.bool, device="cuda")
    env.step_random_many(DIST_T, reward=r, terminated=tm)
    counters = EpisodeCounters("cuda")
    for t in range(DIST_T):
        counters.update(r[t], tm[t])
    want_returns = r.sum(dim=0).cpu().numpy()
    turn = env.turn().cpu().numpy()
    assert np.array_equal(results[0][2].view(np.uint32), want_returns.view(np.uint32))     # gathered on rank 0, board order
    assert results[1][2] is None
    for rank, total, _, shard_turn in results:
        lo, hi = shard_range(DIST_TOTAL, rank, world)
        assert total == counters.c.cpu().tolist()                                          # all_reduce == whole-job counters
        assert np.array_equal(shard_turn, turn[lo:hi])                                     # shard == slice of the single run
    assert counters.c[0] > DIST_TOTAL and counters.c[3] == DIST_TOTAL * DIST_T
