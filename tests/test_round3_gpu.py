"""Round 3 on the GPU, through the C ABI:
  * qttt_step_random_many — T steps of the in-kernel uniform-legal policy in one launch — against the
    oracle's sample_actions/step loop (the policy -> step loop of mcts.py:185-198 over env.py:34-53),
  * the launch shape carried per call (QTTT_FLAG_SHAPE) from two host threads at once,
  * qttt_export through LDS tiles with every output nullable (Env.turn = n_moves alone, env.py:65-66),
  * two boards per lane in qttt_node_info / paired children in qttt_expand at ragged sizes,
  * the out= forms of the rows a search loop calls, VecEnv.step's copies and non-contiguous inputs."""
import threading

import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _np(t):
    return t.cpu().numpy()


def _assert_same_as_oracle(env, ob, tag="", sel=slice(None)):
    ex = {k: _np(v) for k, v in env.export_boards().items()}
    assert np.array_equal(ex["board"][sel], ob.board[sel]), tag
    assert np.array_equal(ex["n_moves"][sel], ob.n_moves[sel]), tag
    assert np.array_equal(ex["moves"][sel], ob.moves[sel]), tag
    assert np.array_equal(ex["n_q"][sel], ob.n_q[sel]), tag
    assert np.array_equal(ex["qmask"].view(np.uint16)[sel], ob.qmask[sel]), tag


def _oracle_random_steps(n, T, seed, off, auto_reset, step_idx0=0):
    ob = oracle.OracleBoards(n)
    acts = np.empty((T, n, 2), dtype=np.uint8)
    rew = np.empty((T, n), dtype=np.uint32)
    term = np.empty((T, n), dtype=np.uint8)
    for t in range(T):
        acts[t] = ob.sample_actions(seed, step_idx0 + t, off, auto_reset)
        r, tm = ob.step(acts[t], None, seed, step_idx0 + t, off, auto_reset)
        rew[t], term[t] = r.view(np.uint32), tm
    return ob, acts, rew, term


@pytest.mark.parametrize("n,T", [(1, 1), (1, 9), (1, 64), (4096, 1), (4096, 9), (4096, 64), (262144, 1), (262144, 9),
                                 (262144, 64), (1048577, 1), (1048577, 9), (1048577, 64)])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_step_random_many_every_output_kept_vs_oracle(n, T, auto_reset):
    from qtttgym_amd import VecEnv
    if n >= 262144 and T == 64 and not auto_reset:
        pytest.skip("64 plies without auto-reset are 55 noops on finished boards: covered at 4 096 boards; keeps the suite short")
    seed, off = 4242 + n + T, 3 * n
    ob, acts, rew, term = _oracle_random_steps(n, T, seed, off, auto_reset)
    env = VecEnv(n, seed=seed, auto_reset=auto_reset, board_offset=off)
    a = torch.empty((T, n, 2), dtype=torch.uint8, device="cuda")
    r = torch.empty((T, n), dtype=torch.float32, device="cuda")
    tm = torch.empty((T, n), dtype=torch.bool, device="cuda")
    got_r, got_t = env.step_random_many(T, actions_out=a, reward=r, terminated=tm)
    assert got_r is r and got_t is tm and env.step_idx == T
    assert np.array_equal(_np(a), acts)
    assert np.array_equal(_np(r).view(np.uint32), rew)
    assert np.array_equal(_np(tm).astype(np.uint8), term)
    _assert_same_as_oracle(env, ob, (n, T, auto_reset))


@pytest.mark.parametrize("n,T,off", [(4099, 13, 0), (5000, 9, (1 << 32) - 2500), (70000, 20, (1 << 40) + 5)])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_step_random_many_last_only_in_two_chunks_and_across_2_pow_32(n, T, off, auto_reset):
    """Only the last step's outputs are written (out_stride 0); two launches of T1 + T2 steps continue the
    step counter; board ids cross 2^32 inside the batch."""
    from qtttgym_amd import VecEnv
    seed = 99
    T1 = T // 2
    ob, acts, rew, term = _oracle_random_steps(n, T, seed, off, auto_reset)
    env = VecEnv(n, seed=seed, auto_reset=auto_reset, board_offset=off)
    last_a = torch.zeros((n, 2), dtype=torch.uint8, device="cuda")
    env.step_random_many(T1)
    r, tm = env.step_random_many(T - T1, actions_out=last_a)
    assert r is env._reward and env.step_idx == T
    assert np.array_equal(_np(last_a), acts[-1])
    assert np.array_equal(_np(r).view(np.uint32), rew[-1])
    assert np.array_equal(_np(tm).astype(np.uint8), term[-1])
    _assert_same_as_oracle(env, ob, (n, T, off))


def test_step_random_many_equals_step_random_launch_by_launch():
    """The fused form is bit-identical to T calls of step_random (policy + step fused in one kernel per step)."""
    from qtttgym_amd import VecEnv
    n, T, seed = 300001, 24, 5
    a, b = VecEnv(n, seed=seed, auto_reset=True), VecEnv(n, seed=seed, auto_reset=True)
    ra = torch.empty((T, n), dtype=torch.float32, device="cuda")
    ta = torch.empty((T, n), dtype=torch.bool, device="cuda")
    aa = torch.empty((T, n, 2), dtype=torch.uint8, device="cuda")
    a.step_random_many(T, actions_out=aa, reward=ra, terminated=ta)
    act = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
    for t in range(T):
        r, tm = b.step_random(actions_out=act)
        assert torch.equal(act, aa[t]), t
        assert torch.equal(r.view(torch.int32), ra[t].view(torch.int32)) and torch.equal(tm, ta[t]), t
    assert torch.equal(a.state, b.state)


def test_step_random_many_argument_errors():
    from qtttgym_amd import VecEnv, _native
    n = 256
    env = VecEnv(n)
    L = _native.lib()
    s = torch.cuda.current_stream().cuda_stream
    r = torch.empty(n, dtype=torch.float32, device="cuda")
    tm = torch.empty(n, dtype=torch.bool, device="cuda")
    st = env.state.data_ptr()
    assert L.qttt_step_random_many(st, 1, 0, 0, 0, None, r.data_ptr(), tm.data_ptr(), 0, None, n, 0, s) == 0      # no steps
    assert L.qttt_step_random_many(st, 1, 0, 0, 0, None, r.data_ptr(), tm.data_ptr(), 0, None, 0, 5, s) == 0      # no boards
    assert L.qttt_step_random_many(None, 1, 0, 0, 0, None, r.data_ptr(), tm.data_ptr(), 0, None, n, 5, s) == -1
    assert L.qttt_step_random_many(st, 1, 0, 0, 0, None, r.data_ptr(), None, 0, None, n, 5, s) == -1              # reward without terminated
    assert L.qttt_step_random_many(st, 1, 0, -1, 0, None, r.data_ptr(), tm.data_ptr(), 0, None, n, 5, s) == -2
    assert L.qttt_step_random_many(st, 1, 0, 0, 0, None, r.data_ptr(), tm.data_ptr(), -1, None, n, 5, s) == -2
    assert L.qttt_step_random_many(st, 1, 0, 0, 0, None, r.data_ptr() + 2, tm.data_ptr(), 0, None, n, 5, s) == -3
    assert L.qttt_step_random_many(st, 1, 0, 0, 0, None, None, None, 0, None, n, 5, s) == 0                      # state only
    with pytest.raises(ValueError):
        env.step_random_many(3, reward=torch.empty((3, n), dtype=torch.float32, device="cuda"))
    with pytest.raises(ValueError):
        env.step_random_many(3, actions_out=torch.empty((3, n, 2), dtype=torch.uint8, device="cuda"))


def test_two_host_threads_with_different_launch_shapes_vs_oracle():
    """the library is re-entrant.  Two host threads step two environments at once, each
    with its own forced launch shape carried in the calls' flags (QTTT_FLAG_SHAPE), each on its own
    stream; both bit-exact against the oracle."""
    from qtttgym_amd import VecEnv, _native
    n, T = 200001, 16
    shapes = [(1, 256), (2, 1024)]
    seeds = [31, 32]
    envs = [VecEnv(n, seed=seeds[k], auto_reset=True, launch_shape=shapes[k]) for k in range(2)]
    for k in range(2):
        assert _native.step_launch_shape(n, envs[k]._flags()) == shapes[k]
    streams = [torch.cuda.Stream() for _ in range(2)]
    got = [None, None]
    errs = []

    def work(k):
        try:
            with torch.cuda.stream(streams[k]):
                env = envs[k]
                acts, rew, term = [], [], []
                for t in range(T):
                    a = env.sample_actions()
                    r, tm = env.step_raw(a)
                    acts.append(a.clone()); rew.append(r.clone()); term.append(tm.clone())
                streams[k].synchronize()
                got[k] = (torch.stack(acts), torch.stack(rew), torch.stack(term))
        except Exception as e:                               # noqa: BLE001
            errs.append(e)

    torch.cuda.synchronize()                              # the environments' resets ran on the default stream
    th = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    torch.cuda.synchronize()
    for k in range(2):
        ob, acts, rew, term = _oracle_random_steps(n, T, seeds[k], 0, True)
        assert np.array_equal(_np(got[k][0]), acts), k
        assert np.array_equal(_np(got[k][1]).view(np.uint32), rew), k
        assert np.array_equal(_np(got[k][2]).astype(np.uint8), term), k
        _assert_same_as_oracle(envs[k], ob, k)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 1000, 65536 + 7])
def test_export_tiles_every_subset_and_misaligned_views(n):
    """qttt_export writes through LDS tiles; any subset of its outputs may be asked for, and outputs that are
    views offset by one board (every alignment phase of the tile copy) give the same rows."""
    from qtttgym_amd import VecEnv
    seed = 17 + n
    env = VecEnv(n, seed=seed)
    ob = oracle.OracleBoards(n)
    for t in range(8):
        a = ob.sample_actions(seed, t, 0, False)
        env.step_raw(torch.from_numpy(a).cuda())
        ob.step(a, None, seed, t, 0, False)
    want = {"moves": ob.moves, "n_moves": ob.n_moves, "board": ob.board, "qmask": ob.qmask.view(np.int16), "n_q": ob.n_q}
    L, s = env._lib, torch.cuda.current_stream().cuda_stream
    spec = dict((k, (dt, shp)) for k, dt, shp in VecEnv._EXPORT_SPEC)
    order = ["moves", "n_moves", "board", "qmask", "n_q"]
    for mask in range(1, 32):
        bufs = {k: torch.full((n + 1,) + spec[k][1], 77, dtype=spec[k][0], device="cuda") for k in order}
        sel = [k for j, k in enumerate(order) if mask >> j & 1]
        # odd masks write whole tensors, even ones views that start one board in
        first = 0 if mask & 1 else 1
        ptrs = [bufs[k][first:].data_ptr() if k in sel else None for k in order]
        assert L.qttt_export(env.state.data_ptr(), *ptrs, n, s) == 0
        for k in order:
            got = _np(bufs[k])
            if k in sel:
                assert np.array_equal(got[first:first + n], want[k]), (mask, k)
                assert (got[:first] == 77).all() and (got[first + n:] == 77).all(), (mask, k)
            else:
                assert (got == 77).all(), (mask, k)
    assert np.array_equal(_np(env.turn()), ob.n_moves)
    buf = torch.zeros(n, dtype=torch.uint8, device="cuda")
    assert env.turn(out=buf) is buf and np.array_equal(_np(buf), ob.n_moves)
    ex = env.export_boards()
    for v in ex.values():
        v.zero_()
    assert env.export_boards(out=ex) is ex
    for k in order:
        assert np.array_equal(_np(ex[k]), want[k]), k
    with pytest.raises(ValueError):
        env.export_boards(out={**ex, "board": ex["board"][:, :8]})


def test_export_at_the_end_of_the_game_nine_moves_and_autofill():
    """Boards played to the end: nine real moves (round 8's x kept in the `last x` field), implicit
    autofill moves, finished boards — every depth in one batch."""
    from qtttgym_amd import VecEnv
    n, seed = 20000, 8
    env = VecEnv(n, seed=seed)
    ob = oracle.OracleBoards(n)
    for t in range(11):
        a = ob.sample_actions(seed, t, 0, False)
        env.step_raw(torch.from_numpy(a).cuda())
        ob.step(a, None, seed, t, 0, False)
        _assert_same_as_oracle(env, ob, t)
    nm = ob.n_moves
    assert (nm == 9).sum() > n // 2
    auto = (ob.moves[np.arange(n), np.minimum(nm, 9) - 1, 0] == ob.moves[np.arange(n), np.minimum(nm, 9) - 1, 1])
    assert auto.sum() > 100 and (~auto).sum() > 100          # both kinds of ninth move are present


@pytest.mark.parametrize("n", [1, 2, 3, 127, 128, 129, 4097, 100001])
def test_node_info_pairs_and_expand_pairs_at_ragged_sizes(n):
    """Two boards per lane in node_info (the last board of an odd batch alone), outputs as offset views
    (scalar-store path), and the paired children of expand, against the oracle."""
    from qtttgym_amd import VecEnv
    seed = 23
    env = VecEnv(n, seed=seed)
    ob = oracle.OracleBoards(n)
    rng = np.random.default_rng(n)
    depth = rng.integers(0, 10, size=n)
    for t in range(9):                                    # boards frozen at random depths 0..9
        a = ob.sample_actions(seed, t, 0, False)
        a[depth <= t] = 0                                  # (0,0): a noop
        env.step_raw(torch.from_numpy(a).cuda())
        ob.step(a, None, seed, t, 0, False)
    w, tm, lg, ky = oracle.node_info(ob) if n <= 5000 else (None,) * 4
    info = env.node_info()
    if w is not None:
        assert np.array_equal(_np(info["winner"]), w) and np.array_equal(_np(info["terminal"]).astype(np.uint8), tm)
        assert np.array_equal(_np(info["legal"]).view(np.uint64), lg) and np.array_equal(_np(info["key"]), ky)
    # offset views: one element in (misaligned for the vector stores) -> same values
    big = {"winner": torch.zeros(n + 1, dtype=torch.int8, device="cuda"), "terminal": torch.zeros(n + 1, dtype=torch.bool, device="cuda"),
           "legal": torch.zeros(n + 1, dtype=torch.int64, device="cuda"), "key": torch.zeros(n + 1, dtype=torch.int64, device="cuda")}
    view = {k: v[1:] for k, v in big.items()}
    env.node_info(out=view)
    for k in big:
        assert torch.equal(view[k], info[k]), k
        assert int(big[k][0]) == 0, k
    if n <= 5000:
        act = rng.integers(0, 40, size=n).astype(np.uint8)      # 36..39: not an action
        nch, kids, ow, ot, ol, ok = oracle.expand(ob, act)
        out = env.expand(torch.from_numpy(act).cuda())
        assert np.array_equal(_np(out["n_children"]), nch)
        assert np.array_equal(_np(out["winner"]), ow) and np.array_equal(_np(out["terminal"]).astype(np.uint8), ot)
        assert np.array_equal(_np(out["legal"]).view(np.uint64), ol) and np.array_equal(_np(out["key"]), ok)
        for c in range(2):                                   # child c is meaningful where c < n_children (include/qttt.h)
            _assert_same_as_oracle(out["child%d" % c], kids[c], (n, c), sel=nch > c)
        # out= : the same buffers, overwritten
        keep = {k: (v.state.data_ptr() if k.startswith("child") else v.data_ptr()) for k, v in out.items()}
        for k, v in out.items():
            (v.state if k.startswith("child") else v).zero_()
        again = env.expand(torch.from_numpy(act).cuda(), out=out)
        assert again is out
        assert keep == {k: (v.state.data_ptr() if k.startswith("child") else v.data_ptr()) for k, v in out.items()}
        assert np.array_equal(_np(out["key"]), ok) and np.array_equal(_np(out["n_children"]), nch)
        for c in range(2):
            _assert_same_as_oracle(out["child%d" % c], kids[c], (n, c, "out="), sel=nch > c)


def test_expand_wants_aligned_rows():
    from qtttgym_amd import VecEnv
    n = 64
    env = VecEnv(n)
    L, s = env._lib, torch.cuda.current_stream().cuda_stream
    a = torch.zeros(n, dtype=torch.uint8, device="cuda")
    c0, c1 = torch.empty_like(env.state), torch.empty_like(env.state)
    nch = torch.empty(n, dtype=torch.uint8, device="cuda")
    w = torch.empty(2 * n + 8, dtype=torch.int8, device="cuda")
    tm = torch.empty(2 * n + 8, dtype=torch.uint8, device="cuda")
    lg = torch.empty(2 * n + 2, dtype=torch.int64, device="cuda")
    ky = torch.empty(2 * n + 2, dtype=torch.int64, device="cuda")
    sk = torch.empty(2 * n + 2, dtype=torch.int64, device="cuda")
    ok = lambda wp, tp, lp, kp, sp=sk.data_ptr(): L.qttt_expand(env.state.data_ptr(), a.data_ptr(), c0.data_ptr(), c1.data_ptr(),
                                                                nch.data_ptr(), wp, tp, lp, kp, sp, n, s)
    assert ok(w.data_ptr(), tm.data_ptr(), lg.data_ptr(), ky.data_ptr()) == 0
    assert ok(w.data_ptr() + 1, tm.data_ptr(), lg.data_ptr(), ky.data_ptr()) == -3
    assert ok(w.data_ptr(), tm.data_ptr(), lg.data_ptr() + 8, ky.data_ptr()) == -3
    assert ok(w.data_ptr(), tm.data_ptr(), lg.data_ptr(), ky.data_ptr() + 8) == -3
    assert ok(w.data_ptr(), tm.data_ptr(), lg.data_ptr(), ky.data_ptr(), sk.data_ptr() + 8) == -3
    assert ok(None, None, None, None, None) == 0                      # every per-child row is nullable


def test_rollout_and_encode_out_reuse():
    from qtttgym_amd import VecEnv
    n = 5000
    env = VecEnv(n, seed=3)
    for _ in range(3):
        env.step_raw(env.sample_actions())
    res, pl, fin = env.rollout(return_final=True)
    planes = lambda st: st.view(torch.int64).view(2, -1)[:, :n]      # P | Q planes without the padding to 64 boards
    ref = (res.clone(), pl.clone(), planes(fin.state).clone())
    res.zero_(); pl.zero_(); fin.state.zero_()
    out = env.rollout(return_final=True, out=(res, pl, fin))
    assert out[0] is res and out[2] is fin
    assert torch.equal(res, ref[0]) and torch.equal(pl, ref[1]) and torch.equal(planes(fin.state), ref[2])
    r2 = env.rollout(out=(res, pl))
    assert r2[0] is res and torch.equal(res, ref[0])
    vec, mask = env.encode()
    v0, m0 = vec.clone(), mask.clone()
    vec.zero_(); mask.zero_()
    v1, m1 = env.encode(out=(vec, mask))
    assert v1 is vec and torch.equal(vec, v0) and torch.equal(mask, m0)
    v2 = env.encode(with_mask=False, out=vec)
    assert v2 is vec and torch.equal(vec, v0)
    with pytest.raises(ValueError):
        env.encode(out=(vec[:-1], mask))
    with pytest.raises(ValueError):
        env.rollout(out=(res[:-1], pl))


def test_step_returns_copies_by_default_and_takes_non_contiguous_inputs(golden):
    """VecEnv.step() hands out fresh tensors (a caller that keeps (obs, next_obs) pairs must not see
    them alias); copy_obs=False is the zero-copy form.  Transposed / strided uint8 device tensors are accepted."""
    from qtttgym_amd import VecEnv
    acts, bits = golden["actions"], golden["bits"]
    E = bits.shape[0]
    env, raw = VecEnv(E), VecEnv(E)
    a0 = torch.from_numpy(acts[:, 0].copy()).cuda()
    a1 = torch.from_numpy(acts[:, 1].copy()).cuda()
    obs0, r0, t0, _, _ = env.step(a0.t().contiguous().t(), torch.from_numpy(bits[:, 0].copy()).cuda())   # (2,N).t(): not contiguous
    keep = {k: v.clone() for k, v in obs0.items()}
    wide = torch.zeros((E, 2), dtype=torch.uint8, device="cuda")
    wide[:, 0] = torch.from_numpy(bits[:, 1].copy()).cuda()
    obs1, r1, t1, _, _ = env.step(a1, wide[:, 0])                                                          # strided bits
    for k in obs0:
        assert obs0[k].data_ptr() != obs1[k].data_ptr() and torch.equal(obs0[k], keep[k]), k
    assert r0.data_ptr() != r1.data_ptr()
    assert np.array_equal(_np(obs1["classical"]), golden["board"][:, 1])
    assert np.array_equal(_np(obs0["classical"]), golden["board"][:, 0])
    o_a, _, _, _, _ = raw.step(a0, torch.from_numpy(bits[:, 0].copy()).cuda(), copy_obs=False)
    o_b, _, _, _, _ = raw.step(a1, torch.from_numpy(bits[:, 1].copy()).cuda(), copy_obs=False)
    assert all(o_a[k].data_ptr() == o_b[k].data_ptr() for k in o_a)                                      # the env's own buffers
    assert np.array_equal(_np(o_b["classical"]), golden["board"][:, 1])


# ---------------------------------------------------------------------------------------------------
# N > 1 on the HIP path: two ranks (both on this box's one card, rendezvous over gloo — RCCL refuses two
# ranks on one GPU) each step THEIR shard with the fused random-policy kernel; the per-board returns come
# back through dist.gather_returns and must equal boards [0, 2B) of one single-process run.
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


# ---------------------------------------------------------------------------------------------------
# hipGraph of step launches with the step index on the device (VecEnv.capture, qttt_env.step_counter)
@pytest.mark.parametrize("n,off", [(4096, 0), (70001, 0), (4099, (1 << 32) - 2000)])
def test_captured_random_steps_replay_with_a_fresh_step_index_every_time(n, off):
    from qtttgym_amd import VecEnv
    T, R, seed = 9, 3, 21
    ob, acts, rew, term = _oracle_random_steps(n, T * R + 2, seed, off, True)
    env = VecEnv(n, seed=seed, auto_reset=True, board_offset=off)      # (ids crossing 2^32: two launch segments per node)
    a = torch.zeros((T, n, 2), dtype=torch.uint8, device="cuda")
    r = torch.zeros((T, n), dtype=torch.float32, device="cuda")
    tm = torch.zeros((T, n), dtype=torch.bool, device="cuda")
    g = env.capture(T, "random", actions_out=a, reward=r, terminated=tm)
    assert env.step_idx == 0 and not bool(a.any())                       # capturing ran nothing
    for k in range(R):
        g.replay()
        torch.cuda.synchronize()
        sl = slice(k * T, (k + 1) * T)
        assert np.array_equal(_np(a), acts[sl]), k
        assert np.array_equal(_np(r).view(np.uint32), rew[sl]), k
        assert np.array_equal(_np(tm).astype(np.uint8), term[sl]), k
        assert env.step_idx == (k + 1) * T
    # eager calls keep working on the same environment (they advance the device counter too)
    act = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
    r1, t1 = env.step_random(actions_out=act)
    assert np.array_equal(_np(act), acts[T * R]) and np.array_equal(_np(r1).view(np.uint32), rew[T * R])
    r2, t2 = env.step_raw(env.sample_actions())
    assert np.array_equal(_np(r2).view(np.uint32), rew[T * R + 1]) and np.array_equal(_np(t2).astype(np.uint8), term[T * R + 1])
    assert env.step_idx == T * R + 2
    _assert_same_as_oracle(env, ob)
    sd = env.state_dict()
    assert sd["step_idx"] == T * R + 2
    env.reset()
    assert env.step_idx == 0
    g.replay()                                                            # from a fresh reset: the first T steps again
    torch.cuda.synchronize()
    assert np.array_equal(_np(a), acts[:T]) and np.array_equal(_np(r).view(np.uint32), rew[:T])


def test_captured_step_and_observe_modes_read_the_callers_action_buffer():
    """The agent-loop shape: a graph of ONE step launch (with the observation) replayed every step, the action
    buffer refilled in between; collapse bits from the device-side step counter."""
    from qtttgym_amd import VecEnv
    n, seed, steps = 5000, 33, 14
    ob, acts, rew, term = _oracle_random_steps(n, steps, seed, 0, True)
    env = VecEnv(n, seed=seed, auto_reset=True)
    a = torch.zeros((1, n, 2), dtype=torch.uint8, device="cuda")
    g = env.capture(1, "observe", actions=a)
    ref = oracle.OracleBoards(n)
    for t in range(steps):
        a[0].copy_(torch.from_numpy(acts[t]))
        r, tm = g.replay()
        ref.step(acts[t], None, seed, t, 0, True)
        assert np.array_equal(_np(r).view(np.uint32), rew[t]) and np.array_equal(_np(tm).astype(np.uint8), term[t]), t
        cl = ref.observe()[0]
        assert np.array_equal(_np(env._obs["classical"]), cl), t
    _assert_same_as_oracle(env, ob)
    env2 = VecEnv(n, seed=seed, auto_reset=True)
    a2 = torch.from_numpy(acts[:6].copy()).cuda()
    g2 = env2.capture(6, "step", actions=a2)
    g2.replay()
    a2.copy_(torch.from_numpy(acts[6:12].copy()))
    r, tm = g2.replay()
    torch.cuda.synchronize()
    assert np.array_equal(_np(r).view(np.uint32), rew[11]) and env2.step_idx == 12
    with pytest.raises(ValueError):
        env2.capture(2, "step")
    with pytest.raises(ValueError):
        env2.capture(2, "random", actions=a2[:2])


def test_a_whole_agent_step_is_graph_capturable_with_the_device_step_counter():
    """The caller's own graph (not VecEnv.capture): policy kernel + torch ops + step with observation captured
    once, replayed; equals the eager loop of another environment (and so, transitively, the oracle)."""
    from qtttgym_amd import VecEnv
    n, seed, steps = 3000, 44, 12
    ref = VecEnv(n, seed=seed, auto_reset=True)
    env = VecEnv(n, seed=seed, auto_reset=True)
    env.use_device_step_counter()
    env.observ()
    total = torch.zeros((), dtype=torch.int64, device="cuda")

    def agent_step(e, acc):
        a = e.sample_actions()
        a = torch.where(e._obs["turn"][:, None] > 1, torch.zeros_like(a), a)      # (a policy that reads the observation)
        _, r, tm = e.step_observe_raw(a)
        acc.add_(tm.sum())

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        warm = VecEnv(n, seed=seed, auto_reset=True)
        warm.use_device_step_counter(); warm.observ()
        agent_step(warm, torch.zeros_like(total))                                  # kernels resident before the capture
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            agent_step(env, total)
    torch.cuda.current_stream().wait_stream(side)
    want = torch.zeros_like(total)
    ref.observ()
    for t in range(steps):
        g.replay()
        agent_step(ref, want)
        torch.cuda.synchronize()
        assert torch.equal(env.state, ref.state), t
        assert torch.equal(env._obs["classical"], ref._obs["classical"]), t
    assert int(total) == int(want) > 0 and env.step_idx == steps == ref.step_idx


def test_checkpoint_round_trip_with_the_device_step_counter_and_python_inputs():
    """state_dict / load_state_dict carry the step index whether it lives on the host or on the device; step() takes
    Python lists / numpy arrays / wide integer tensors (out-of-range values are noops, env.py:41)."""
    from qtttgym_amd import VecEnv
    n, seed = 2000, 3
    a = VecEnv(n, seed=seed, auto_reset=True)
    a.step_random_many(5)
    a.use_device_step_counter()
    a.step_random()
    sd = a.state_dict()
    assert sd["step_idx"] == 6
    b = VecEnv(n, seed=0)                                   # host-side counter, other seed: everything comes from the dict
    b.load_state_dict(sd)
    assert b.step_idx == 6 and b.seed == seed and b.auto_reset
    ra, ta = a.step_random()
    rb, tb = b.step_random()
    assert torch.equal(ra.view(torch.int32), rb.view(torch.int32)) and torch.equal(ta, tb) and torch.equal(a.state, b.state)
    c = VecEnv(n, seed=0)
    c.use_device_step_counter()
    c.load_state_dict(sd)                                   # device-side counter on the receiving end
    c.step_random()
    assert c.step_idx == 7 and torch.equal(c.state, b.state)
    # Python-side inputs of step(): list of pairs, numpy int64 with junk, both equal the uint8 tensor path
    e1, e2, e3 = VecEnv(4), VecEnv(4), VecEnv(4)
    acts = [[0, 1], [9, 3], [-1, 2], [300, 4]]               # legal, out of range, negative, > 255: three noops
    o1, r1, t1, _, _ = e1.step(acts)
    o2, r2, t2, _, _ = e2.step(np.asarray(acts, dtype=np.int64))
    o3, r3, t3, _, _ = e3.step(torch.tensor([[0, 1], [255, 3], [255, 2], [255, 4]], dtype=torch.uint8, device="cuda"))
    for k in o1:
        assert torch.equal(o1[k], o2[k]) and torch.equal(o1[k], o3[k]), k
    assert _np(e1.turn()).tolist() == [1, 0, 0, 0]


def test_take_lines_boards_up_without_unpacking_them():
    from qtttgym_amd import VecEnv
    n = 1000
    env = VecEnv(n, seed=8)
    for _ in range(5):
        env.step_raw(env.sample_actions())
    idx = torch.tensor([3, 3, 999, 0, 3, 500], device="cuda")
    sub = env.take(idx)
    ex, sx = env.export_boards(), sub.export_boards()
    for k in ex:
        assert torch.equal(sx[k], ex[k][idx]), k
    assert sub.num_envs == 6 and sub.seed == env.seed
    rep = env.take(torch.arange(n, device="cuda").repeat_interleave(36))
    assert rep.num_envs == 36 * n
    out = rep.expand(torch.arange(36, dtype=torch.uint8, device="cuda").repeat(n))
    legal = env.node_info()["legal"]
    bits = ((legal[:, None] >> torch.arange(36, device="cuda")[None, :]) & 1).bool().reshape(-1)
    assert torch.equal(out["n_children"] > 0, bits)                      # an action has children iff node_info calls it legal
    assert env.take(torch.empty(0, dtype=torch.int64, device="cuda")).num_envs == 0


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 100003])
def test_import_tiles_round_trip_misaligned_views_and_garbage(n):
    """qttt_import (LDS tiles, nibble-parallel unpack, round-order insertion): export -> import -> export is the
    identity at every depth incl. finished games, from whole tensors and from views offset by one board (every
    alignment phase); the imported boards then step exactly like the originals; arbitrary bytes neither fault nor hang."""
    from qtttgym_amd import VecEnv
    seed = 5 + n
    env = VecEnv(n, seed=seed)
    rng = np.random.default_rng(n)
    depth = torch.from_numpy(rng.integers(0, 11, n).astype(np.uint8)).cuda()
    a = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
    for t in range(10):
        env.sample_actions(out=a)
        a[depth <= t] = 0
        env.step_raw(a)
    ex = env.export_boards()
    L, s = env._lib, torch.cuda.current_stream().cuda_stream
    order = ("moves", "n_moves", "board", "qmask", "n_q")
    for first in (0, 1):
        bufs = {}
        for k in order:
            big = torch.zeros((n + 1,) + tuple(ex[k].shape[1:]), dtype=ex[k].dtype, device="cuda")
            big[first:first + n] = ex[k]
            bufs[k] = big[first:first + n]
        other = VecEnv(n, seed=seed)
        assert L.qttt_import(other.state.data_ptr(), *[bufs[k].data_ptr() for k in order], n, s) == 0
        back = other.export_boards()
        for k in order:
            assert torch.equal(back[k], ex[k]), (first, k)
        assert torch.equal(other.check_win()[0], env.check_win()[0])
        ia, ib = other.node_info(), env.node_info()
        for k in ia:
            assert torch.equal(ia[k], ib[k]), (first, k)
        # the imported boards continue like the originals (same seed / step index / ids), attribute for attribute
        cont = VecEnv.from_state(env.state.clone(), n, seed=seed)
        cont.step_idx = other.step_idx = 50
        for _ in range(4):
            act = cont.sample_actions()
            assert torch.equal(other.sample_actions(), act)
            r1, t1 = cont.step_raw(act)
            r2, t2 = other.step_raw(act)
            assert torch.equal(r1.view(torch.int32), r2.view(torch.int32)) and torch.equal(t1, t2)
        e1, e2 = cont.export_boards(), other.export_boards()
        for k in order:
            assert torch.equal(e1[k], e2[k]), (first, k, "after steps")
    junk = VecEnv(n)
    g = lambda shape, dt: torch.from_numpy(rng.integers(0, 256, size=shape, dtype=np.uint8)).cuda().view(dt)
    junk.import_boards(g((n, 9, 2), torch.uint8), g((n,), torch.uint8), g((n, 9), torch.int8),
                       g((n, 4, 2), torch.uint8).view(torch.int16).reshape(n, 4), g((n,), torch.uint8))
    for _ in range(3):
        junk.step_raw(junk.sample_actions())
    jx = junk.export_boards()
    torch.cuda.synchronize()                                             # garbage in, garbage out — but no fault and no hang
    assert jx["n_moves"].shape == (n,) and int(jx["n_moves"].max()) <= 15


@pytest.mark.parametrize("auto_reset", [False, True])
def test_step_random_many_accumulates_per_board_returns(auto_reset):
    """returns[i] += the sum of board i's rewards over the launch (env.py:49: -1.0 / -0.0 per ply): the per-board
    episode returns SURVEY §8(e) lets a multi-GPU run gather, produced without keeping a single per-ply output."""
    from qtttgym_amd import VecEnv
    n, seed = 50001, 13
    a = VecEnv(n, seed=seed, auto_reset=auto_reset)
    b = VecEnv(n, seed=seed, auto_reset=auto_reset)
    ret = torch.full((n,), 2.0, dtype=torch.float32, device="cuda")     # accumulated onto what is there
    total = torch.full((n,), 2.0, dtype=torch.float32, device="cuda")
    for T in (1, 9, 40):
        a.step_random_many(T, returns=ret)
        r = torch.empty((T, n), dtype=torch.float32, device="cuda")
        tm = torch.empty((T, n), dtype=torch.bool, device="cuda")
        b.step_random_many(T, reward=r, terminated=tm)
        total += r.sum(dim=0)
        assert torch.equal(ret, total), T
        assert torch.equal(a.state, b.state)
    assert float(ret.min()) < 2.0 - (3.0 if auto_reset else 0.5)
    with pytest.raises(ValueError):
        a.step_random_many(3, returns=ret[:-1])
    L, s = a._lib, torch.cuda.current_stream().cuda_stream
    assert L.qttt_step_random_many(a.state.data_ptr(), 1, 0, 0, 0, None, None, None, 0, ret.data_ptr() + 2, n, 3, s) == -3


def test_large_batch_shapes_of_export_and_node_info_on_ragged_offset_views():
    """Above 384 K boards export runs two boards per lane and node_info / expand 1024-thread workgroups: an odd batch
    size, outputs as views offset by one board (every alignment phase, the scalar-store paths), against the
    whole-tensor results of the same kernels (which the oracle / the torch restatements pin elsewhere)."""
    from qtttgym_amd import VecEnv
    n = 400001
    env = VecEnv(n, seed=21)
    depth = (torch.arange(n, device="cuda") * 2654435761 % 11).to(torch.uint8)
    a = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
    for t in range(10):
        env.sample_actions(out=a)
        a[depth <= t] = 0
        env.step_raw(a)
    ex = env.export_boards()
    small = VecEnv.from_state(env.take(torch.arange(1000, device="cuda")).state, 1000)     # the same boards through the small-batch shape
    sx = small.export_boards()
    for k in ex:
        assert torch.equal(ex[k][:1000], sx[k]), k
    L, s = env._lib, torch.cuda.current_stream().cuda_stream
    spec = dict((k, (dt, shp)) for k, dt, shp in VecEnv._EXPORT_SPEC)
    order = ["moves", "n_moves", "board", "qmask", "n_q"]
    bufs = {k: torch.full((n + 1,) + spec[k][1], 77, dtype=spec[k][0], device="cuda") for k in order}
    assert L.qttt_export(env.state.data_ptr(), *[bufs[k][1:].data_ptr() for k in order], n, s) == 0
    for k in order:
        assert torch.equal(bufs[k][1:], ex[k]), k
        assert bool((bufs[k][0] == 77).all()), k
    info = env.node_info()
    big = {"winner": torch.zeros(n + 1, dtype=torch.int8, device="cuda"), "terminal": torch.zeros(n + 1, dtype=torch.bool, device="cuda"),
           "legal": torch.zeros(n + 1, dtype=torch.int64, device="cuda"), "key": torch.zeros(n + 1, dtype=torch.int64, device="cuda")}
    env.node_info(out={k: v[1:] for k, v in big.items()})
    si = small.node_info()
    for k in big:
        assert torch.equal(big[k][1:], info[k]) and int(big[k][0]) == 0, k
        assert torch.equal(info[k][:1000], si[k]), k
    act = torch.randint(0, 36, (n,), dtype=torch.uint8, device="cuda")
    out, so = env.expand(act), small.expand(act[:1000].contiguous())
    for k in ("n_children", "winner", "terminal", "legal", "key"):
        assert torch.equal(out[k][:1000], so[k]), k
    planes = lambda st, m: st.view(torch.int64).view(2, -1)[:, :m]
    assert torch.equal(planes(out["child0"].state, n)[:, :1000], planes(so["child0"].state, 1000))
