"""The step beyond one launch per call, through the C ABI against the oracle and against the launch-by-launch forms:
qttt_step_random_many / qttt_step_many(FUSED) (boards in registers, <= 64 plies per launch, longer runs split), the
policy-in-the-step kernel in every launch shape, two host threads with different shapes, hipGraph capture with the
device-side step counter (VecEnv.capture, a whole agent step), checkpoint round trips."""
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


# ---------------------------------------------------------------------------------------- policy in the step kernel
@pytest.mark.parametrize("n", [4096, 500 * 1024, 600 * 1024 + 3, 1 << 20, (1 << 21) + 64])
@pytest.mark.parametrize("auto_reset", [True, False])
def test_step_random_equals_policy_kernel_plus_step_in_every_shape(n, auto_reset):
    """qttt_step_random (policy through the full n-th-empty-square table, trusted step under auto-reset) against
    qttt_sample_actions + qttt_step, in each region of the launch-shape table."""
    from qtttgym_amd import VecEnv
    a_env = VecEnv(n, seed=77, auto_reset=auto_reset, board_offset=(1 << 32) - n // 2)   # ids cross 2^32
    b_env = VecEnv(n, seed=77, auto_reset=auto_reset, board_offset=(1 << 32) - n // 2)
    played = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
    for t in range(12):
        act = b_env.sample_actions()
        rb, tb = b_env.step_raw(act)
        ra, ta = a_env.step_random(actions_out=played)
        assert torch.equal(played, act), t
        assert torch.equal(ra.view(torch.int32), rb.view(torch.int32)) and torch.equal(ta, tb), t
    assert torch.equal(a_env.state, b_env.state)


# ---------------------------------------------------------------------------------------- fused runs of more than one launch
@pytest.mark.parametrize("T", [65, 130, 200])
@pytest.mark.parametrize("auto_reset", [True, False])
@pytest.mark.parametrize("off", [12345, (1 << 32) - 2000])     # inside one 2^32 block of ids (the FUSED replay kernel) / across
def test_fused_runs_longer_than_the_64_plies_of_one_launch(T, auto_reset, off):
    """The fused kernels take at most 64 plies per launch (their launch keys travel as a kernel argument); the library
    splits a longer run.  Same results as the launch-by-launch forms (which the oracle tests pin): every ply's outputs,
    last-ply-only outputs, the accumulated returns, the state — for qttt_step_random_many and qttt_step_many(FUSED),
    the latter with hashed and with explicit collapse bits."""
    import torch
    from qtttgym_amd import VecEnv
    n, seed = 5003, 77 + T
    mk = lambda: VecEnv(n, seed=seed, auto_reset=auto_reset, board_offset=off)
    # launch by launch: the recording
    rec = mk()
    acts = torch.empty((T, n, 2), dtype=torch.uint8, device="cuda")
    rew = torch.empty((T, n), dtype=torch.float32, device="cuda")
    term = torch.empty((T, n), dtype=torch.bool, device="cuda")
    for t in range(T):
        r, tm = rec.step_random(actions_out=acts[t])
        rew[t], term[t] = r, tm
    # qttt_step_random_many, every ply kept + returns
    a = mk()
    aa, ra, ta = torch.zeros_like(acts), torch.zeros_like(rew), torch.zeros_like(term)
    ret = torch.zeros(n, dtype=torch.float32, device="cuda")
    a.step_random_many(T, actions_out=aa, reward=ra, terminated=ta, returns=ret)
    assert torch.equal(aa, acts) and torch.equal(ra.view(torch.int32), rew.view(torch.int32)) and torch.equal(ta, term)
    assert torch.equal(a.state, rec.state) and a.step_idx == T
    assert torch.equal(ret, rew.sum(dim=0))
    # last ply only
    b = mk()
    last = torch.zeros((n, 2), dtype=torch.uint8, device="cuda")
    r, tm = b.step_random_many(T, actions_out=last)
    assert torch.equal(last, acts[-1]) and torch.equal(r.view(torch.int32), rew[-1].view(torch.int32)) and torch.equal(tm, term[-1])
    assert torch.equal(b.state, rec.state)
    # qttt_step_many(FUSED) on the recorded actions, hashed bits: the same boards again
    c = mk()
    rc, tc = torch.zeros_like(rew), torch.zeros_like(term)
    c.step_many(acts, reward=rc, terminated=tc, fused=True)
    assert torch.equal(rc.view(torch.int32), rew.view(torch.int32)) and torch.equal(tc, term) and torch.equal(c.state, rec.state)
    d = mk()
    r, tm = d.step_many(acts, fused=True)
    assert torch.equal(r.view(torch.int32), rew[-1].view(torch.int32)) and torch.equal(tm, term[-1]) and torch.equal(d.state, rec.state)
    # ... and with explicit bits against its own launch-by-launch form
    bits = torch.randint(0, 2, (T, n), dtype=torch.uint8, device="cuda")
    e, f = mk(), mk()
    re_, te = torch.zeros_like(rew), torch.zeros_like(term)
    e.step_many(acts, bits, reward=re_, terminated=te, fused=True)
    rf, tf = torch.zeros_like(rew), torch.zeros_like(term)
    f.step_many(acts, bits, reward=rf, terminated=tf, fused=False)
    assert torch.equal(re_.view(torch.int32), rf.view(torch.int32)) and torch.equal(te, tf) and torch.equal(e.state, f.state)
