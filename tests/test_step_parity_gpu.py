"""T2: the HIP path (through the C ABI, via qtttgym_amd.VecEnv) against
  (a) the golden traces recorded from the unmodified reference, and
  (b) the C oracle on seeded inputs at BASELINE.json's batch sizes.
Bit-exact: integer state, reward compared as IEEE bits (-0.0 matters, env.py:49)."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _np(t):
    return t.cpu().numpy()


def assert_same_as_oracle(env, ob, tag=""):
    ex = {k: _np(v) for k, v in env.export_boards().items()}
    assert np.array_equal(ex["board"], ob.board), tag
    assert np.array_equal(ex["n_moves"], ob.n_moves), tag
    assert np.array_equal(ex["moves"], ob.moves), tag
    assert np.array_equal(ex["n_q"], ob.n_q), tag
    assert np.array_equal(ex["qmask"].view(np.uint16), ob.qmask), tag


def _same(got, want, what, t):
    """array_equal with a report that says where and how much, should it ever fail."""
    got, want = np.asarray(got), np.asarray(want)
    if got.shape == want.shape and np.array_equal(got, want):
        return
    bad = np.argwhere(got != want) if got.shape == want.shape else None
    raise AssertionError("%s differs at step %s: shapes %s/%s, %s mismatches, first at %s (got %s want %s)" % (
        what, t, got.shape, want.shape, None if bad is None else len(bad),
        None if bad is None or not len(bad) else bad[0].tolist(),
        None if bad is None or not len(bad) else got[tuple(bad[0])],
        None if bad is None or not len(bad) else want[tuple(bad[0])]))


def test_golden_traces_through_hip(golden):
    from qtttgym_amd import VecEnv
    acts, bits = golden["actions"], golden["bits"]
    E, T = bits.shape
    env = VecEnv(E)
    for t in range(T):
        obs, reward, term, trunc, info = env.step(torch.from_numpy(acts[:, t].copy()),
                                                  torch.from_numpy(bits[:, t].copy()))
        assert info == {} and not bool(trunc.any())
        ex = {k: _np(v) for k, v in env.export_boards().items()}
        _same(ex["board"], golden["board"][:, t], "board", t)
        _same(ex["moves"], golden["moves"][:, t], "moves", t)
        _same(ex["n_moves"], golden["n_moves"][:, t], "n_moves", t)
        _same(ex["qmask"].view(np.uint16), golden["qmask"][:, t], "qmask", t)
        _same(ex["n_q"], golden["n_q"][:, t], "n_q", t)
        want = golden["reward"][:, t].astype(np.float32).view(np.uint32)
        _same(_np(reward).view(np.uint32), want, "reward bits", t)
        _same(_np(term).astype(np.uint8), golden["terminated"][:, t], "terminated", t)
        _same(_np(obs["classical"]), golden["board"][:, t], "obs.classical", t)
        _same(_np(obs["q_states_p1"]), golden["q_p1"][:, t], "obs.q_states_p1", t)
        _same(_np(obs["q_states_p1_len"]), golden["q_p1_len"][:, t], "obs.q_states_p1_len", t)
        _same(_np(obs["q_states_p2"]), golden["q_p2"][:, t], "obs.q_states_p2", t)
        _same(_np(obs["q_states_p2_len"]), golden["q_p2_len"][:, t], "obs.q_states_p2_len", t)
        _same(_np(obs["turn"]), golden["turn"][:, t], "obs.turn", t)
        p1, p2 = env.check_win()
        _same(_np(p1), golden["p1_round"][:, t], "p1_round", t)
        _same(_np(p2), golden["p2_round"][:, t], "p2_round", t)


@pytest.mark.parametrize("n", [1, 63, 64, 65, 257, 4096, 262144])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_hip_vs_oracle_uniform_policy_hash_bits(n, auto_reset):
    """Policy kernel + step kernel with in-kernel hash bits vs the oracle, 14 steps
    (episodes last <= 9 steps, so with auto_reset every board restarts at least once)."""
    from qtttgym_amd import VecEnv
    seed, off = 1234 + n, 7 * n
    env = VecEnv(n, seed=seed, auto_reset=auto_reset, board_offset=off)
    ob = oracle.OracleBoards(n)
    for t in range(14):
        a_hip = env.sample_actions()
        a_or = ob.sample_actions(seed, t, off, auto_reset)
        assert np.array_equal(_np(a_hip), a_or), t
        reward, term = env.step_raw(a_hip)
        r_or, t_or = ob.step(a_or, None, seed, t, off, auto_reset)
        assert np.array_equal(_np(reward).view(np.uint32), r_or.view(np.uint32)), t
        assert np.array_equal(_np(term).astype(np.uint8), t_or), t
        assert_same_as_oracle(env, ob, t)


def test_hip_vs_oracle_random_garbage_actions():
    """Arbitrary bytes as actions (mostly noops) + explicit bits."""
    from qtttgym_amd import VecEnv
    n = 8192
    rng = np.random.default_rng(5)
    env = VecEnv(n)
    ob = oracle.OracleBoards(n)
    for t in range(40):
        hi = 256 if t % 3 == 0 else 10
        acts = rng.integers(0, hi, size=(n, 2), dtype=np.uint8)
        bits = rng.integers(0, 256, size=n, dtype=np.uint8)   # only bit 0 counts
        reward, term = env.step_raw(torch.from_numpy(acts).cuda(), torch.from_numpy(bits).cuda())
        r_or, t_or = ob.step(acts, bits)
        assert np.array_equal(_np(reward).view(np.uint32), r_or.view(np.uint32)), t
        assert np.array_equal(_np(term).astype(np.uint8), t_or), t
        assert_same_as_oracle(env, ob, t)


def test_full_size_one_million_boards_vs_oracle():
    """BASELINE metric size: 1 048 576 boards, auto-reset throughput mode, 12 steps."""
    from qtttgym_amd import VecEnv
    n = 1 << 20
    env = VecEnv(n, seed=3, auto_reset=True)
    ob = oracle.OracleBoards(n)
    n_term = 0
    for t in range(12):
        a = env.sample_actions()
        reward, term = env.step_raw(a)
        a_np = _np(a)
        r_or, t_or = ob.step(a_np, None, 3, t, 0, True)
        assert np.array_equal(_np(term).astype(np.uint8), t_or), t
        assert np.array_equal(_np(reward).view(np.uint32), r_or.view(np.uint32)), t
        n_term += int(t_or.sum())
    assert_same_as_oracle(env, ob)
    assert n_term > n  # every board finished at least one episode on average


@pytest.mark.parametrize("pre_steps", [5, 8, 9])
def test_export_import_round_trip_continues_identically(pre_steps):
    """Board attributes out and back in (qttt_export -> qttt_import) at mid-game and at the end of
    the game (nine real moves, implicit autofill, finished boards), then both continue identically."""
    from qtttgym_amd import VecEnv
    n = 4096
    env = VecEnv(n, seed=11)
    for t in range(pre_steps):
        env.step_raw(env.sample_actions())
    ex = env.export_boards()
    env2 = VecEnv(n, seed=11)
    env2.import_boards(ex["moves"], ex["n_moves"], ex["board"], ex["qmask"], ex["n_q"])
    env2.step_idx = env.step_idx
    ex2 = env2.export_boards()
    for k in ex:
        assert torch.equal(ex[k], ex2[k]), k
    for t in range(6):
        a = env.sample_actions()
        a2 = env2.sample_actions()
        assert torch.equal(a, a2)
        r1, t1 = env.step_raw(a)
        r1, t1 = r1.clone(), t1.clone()
        r2, t2 = env2.step_raw(a2)
        assert torch.equal(r1.view(torch.int32), r2.view(torch.int32)) and torch.equal(t1, t2)
        e1, e2 = env.export_boards(), env2.export_boards()
        for k in e1:
            assert torch.equal(e1[k], e2[k]), (t, k)


def test_shard_equals_slice_of_single_run():
    """T4: shard k of a G-way run == boards [kN/G,(k+1)N/G) of the 1-GPU run (global board ids
    key the hash)."""
    from qtttgym_amd import VecEnv
    n, G = 8192, 4
    full = VecEnv(n, seed=21, auto_reset=True)
    shards = [VecEnv(n // G, seed=21, auto_reset=True, board_offset=k * (n // G)) for k in range(G)]
    for t in range(12):
        a = full.sample_actions()
        r, tm = full.step_raw(a)
        for k, sh in enumerate(shards):
            sl = slice(k * (n // G), (k + 1) * (n // G))
            ak = sh.sample_actions()
            assert torch.equal(ak, a[sl])
            rk, tk = sh.step_raw(ak)
            assert torch.equal(rk.view(torch.int32), r[sl].view(torch.int32))
            assert torch.equal(tk, tm[sl])
    ef = full.export_boards()
    for k, sh in enumerate(shards):
        es = sh.export_boards()
        sl = slice(k * (n // G), (k + 1) * (n // G))
        for key in ef:
            assert torch.equal(es[key], ef[key][sl]), key


def test_board_ids_across_the_2_pow_32_boundary():
    """The hash folds the 64-bit global board id; a batch whose ids cross a multiple of 2^32 is
    cut there inside qttt_step (ragged cut: 1001 boards before the boundary)."""
    from qtttgym_amd import VecEnv
    n, off, seed = 4096, (1 << 32) - 1001, 5
    env = VecEnv(n, seed=seed, auto_reset=True, board_offset=off)
    ob = oracle.OracleBoards(n)
    for t in range(12):
        a = env.sample_actions()
        a_or = ob.sample_actions(seed, t, off, True)
        assert np.array_equal(_np(a), a_or), t
        reward, term = env.step_raw(a)
        r_or, t_or = ob.step(a_or, None, seed, t, off, True)
        assert np.array_equal(_np(reward).view(np.uint32), r_or.view(np.uint32)), t
        assert np.array_equal(_np(term).astype(np.uint8), t_or), t
    assert_same_as_oracle(env, ob)


def test_empty_batch_and_argument_errors():
    from qtttgym_amd import VecEnv, _native
    env = VecEnv(0)
    r, t = env.step_raw(torch.empty((0, 2), dtype=torch.uint8, device="cuda"))
    assert r.numel() == 0 and t.numel() == 0
    L = _native.lib()
    assert L.qttt_reset(None, 5, None) == -1
    assert L.qttt_reset(None, -1, None) == -2
    assert L.qttt_state_bytes(1 << 20) == 16 << 20
    env = VecEnv(4)
    with pytest.raises(ValueError):
        env.step_raw(torch.zeros((4, 2), dtype=torch.int64, device="cuda"))
    with pytest.raises(_native.QtttNativeError):
        VecEnv(4, device="cpu")


def test_distribution_sanity_uniform_policy():
    """T3: statistics of the uniform-legal policy at scale vs SURVEY.md §8d (measured on the
    reference: mean episode length 8.30, collapse on 22.3 % of steps, autofill in 32.4 % of
    episodes; outcomes P1-only 52.8 / both 22.2 / none 12.8 / P2-only 12.2 %)."""
    from qtttgym_amd import VecEnv
    n = 1 << 18
    env = VecEnv(n, seed=9, auto_reset=False)
    alive = torch.ones(n, dtype=torch.bool, device="cuda")
    length = torch.zeros(n, dtype=torch.int32, device="cuda")
    for t in range(9):
        a = env.sample_actions()
        r, term = env.step_raw(a)
        length += alive.to(torch.int32)
        alive &= ~term
    assert not bool(alive.any())
    mean_len = float(length.float().mean())
    assert abs(mean_len - 8.30) < 0.05, mean_len


def test_steps_are_hip_graph_capturable():
    """qttt_step never allocates, syncs or queries (include/qttt.h conventions), so a rollout loop
    can be captured into a HIP graph on the caller's stream and replayed: same results as eager."""
    from qtttgym_amd import VecEnv
    n, T, seed = 4096, 12, 31
    rec = VecEnv(n, seed=seed, auto_reset=True)
    actions = torch.empty((T, n, 2), dtype=torch.uint8, device="cuda")
    for t in range(T):
        rec.sample_actions(out=actions[t])
        rec.step_raw(actions[t])
    want = rec.state.clone()
    env = VecEnv(n, seed=seed, auto_reset=True)
    reward = torch.empty((T, n), dtype=torch.float32, device="cuda")
    term = torch.empty((T, n), dtype=torch.bool, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        env.step_many(actions[:1])            # warm the stream / kernels outside capture
        env.reset()
        with torch.cuda.graph(g, stream=side):
            env.step_many(actions, reward=reward, terminated=term)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(2):                        # replay twice from a fresh reset: identical both times
        env.reset()
        reward.zero_()
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(env.state, want)
    assert int(term.sum()) > 0 and bool((reward.view(torch.int32) < 0).all())


def test_two_million_boards_counters_vs_oracle():
    """BASELINE config 4's total (2 097 152 boards, here on one GPU): episode counters and the
    final reference-visible state against the oracle."""
    from qtttgym_amd import VecEnv
    from qtttgym_amd.dist import EpisodeCounters
    n, seed = 1 << 21, 8
    env = VecEnv(n, seed=seed, auto_reset=True)
    ob = oracle.OracleBoards(n)
    cnt = EpisodeCounters("cuda")
    fin = lines = 0
    for t in range(10):
        a = env.sample_actions()
        r, tm = env.step_raw(a)
        cnt.update(r, tm)
        r_o, t_o = ob.step(_np(a), None, seed, t, 0, True)
        fin += int(t_o.sum())
        lines += int(((r_o != 0) & (t_o != 0)).sum())
    c = cnt.all_reduce().tolist()
    assert c == [fin, lines, fin - lines, 10 * n]
    assert_same_as_oracle(env, ob)


def test_misaligned_caller_buffers_fall_back_to_narrower_accesses():
    """The vector width of the step kernel follows the alignment of the caller's pointers
    (include/qttt.h): views offset by one board (2 / 4 / 1 bytes) must give the same results."""
    from qtttgym_amd import VecEnv, _native
    n, seed = 1000, 3
    env = VecEnv(n, seed=seed)
    ref = VecEnv(n, seed=seed)
    L = _native.lib()
    big_a = torch.zeros((n + 1, 2), dtype=torch.uint8, device="cuda")
    big_r = torch.zeros(n + 1, dtype=torch.float32, device="cuda")
    big_t = torch.zeros(n + 1, dtype=torch.bool, device="cuda")
    big_b = torch.zeros(n + 1, dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for t in range(10):
        a = ref.sample_actions()
        bits = torch.randint(0, 2, (n,), dtype=torch.uint8, device="cuda")
        r_ref, t_ref = ref.step_raw(a, bits)
        big_a[1:] = a
        big_b[1:] = bits
        rc = L.qttt_step(env.state.data_ptr(), big_a[1:].data_ptr(), big_b[1:].data_ptr(), seed, t, 0, 0,
                         big_r[1:].data_ptr(), big_t[1:].data_ptr(), n, s)
        assert rc == 0
        assert torch.equal(big_r[1:].view(torch.int32), r_ref.view(torch.int32))
        assert torch.equal(big_t[1:], t_ref)
        assert torch.equal(env.state, ref.state)
    # an odd actions address cannot be read as u16 pairs at all
    odd = torch.zeros(2 * n + 1, dtype=torch.uint8, device="cuda")
    assert L.qttt_step(env.state.data_ptr(), odd[1:].data_ptr(), None, 0, 0, 0, 0, big_r.data_ptr(),
                       big_t.data_ptr(), n, s) == -3


@pytest.mark.parametrize("n", [1000, 65536])
@pytest.mark.parametrize("with_bits", [False, True])
def test_fused_step_many_equals_step_by_step(n, with_bits):
    """QTTT_FLAG_FUSED: T steps in one launch == T launches, every per-step output included."""
    from qtttgym_amd import VecEnv
    T, seed = 14, 12
    rec = VecEnv(n, seed=seed, auto_reset=True, board_offset=5 * n)
    actions = torch.empty((T, n, 2), dtype=torch.uint8, device="cuda")
    bits = torch.randint(0, 2, (T, n), dtype=torch.uint8, device="cuda") if with_bits else None
    r_ref = torch.empty((T, n), dtype=torch.float32, device="cuda")
    t_ref = torch.empty((T, n), dtype=torch.bool, device="cuda")
    for t in range(T):
        rec.sample_actions(out=actions[t])
        r, tm = rec.step_raw(actions[t], None if bits is None else bits[t])
        r_ref[t], t_ref[t] = r, tm
    env = VecEnv(n, seed=seed, auto_reset=True, board_offset=5 * n)
    r_f = torch.zeros((T, n), dtype=torch.float32, device="cuda")
    t_f = torch.zeros((T, n), dtype=torch.bool, device="cuda")
    env.step_many(actions, bits, reward=r_f, terminated=t_f, fused=True)
    assert torch.equal(env.state, rec.state)
    assert torch.equal(r_f.view(torch.int32), r_ref.view(torch.int32)) and torch.equal(t_f, t_ref)
    env2 = VecEnv(n, seed=seed, auto_reset=True, board_offset=5 * n)
    r_last, t_last = env2.step_many(actions, bits, fused=True)          # last step's outputs only
    assert torch.equal(r_last.view(torch.int32), r_ref[-1].view(torch.int32)) and torch.equal(t_last, t_ref[-1])
    assert torch.equal(env2.state, rec.state)


@pytest.mark.parametrize("n", [1001, 65536])
@pytest.mark.parametrize("auto_reset", [False, True])
def test_step_random_equals_policy_then_step(n, auto_reset):
    """qttt_step_random (policy + step in one kernel) == qttt_sample_actions + qttt_step, and both
    == the oracle."""
    from qtttgym_amd import VecEnv
    seed, off = 41, 3 * n
    a_env = VecEnv(n, seed=seed, auto_reset=auto_reset, board_offset=off)
    b_env = VecEnv(n, seed=seed, auto_reset=auto_reset, board_offset=off)
    ob = oracle.OracleBoards(n)
    played = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
    for t in range(13):
        acts = a_env.sample_actions()
        r1, t1 = a_env.step_raw(acts)
        r1, t1 = r1.clone(), t1.clone()
        r2, t2 = b_env.step_random(played if t % 2 == 0 else None)
        if t % 2 == 0:
            assert torch.equal(played, acts), t
        assert torch.equal(r1.view(torch.int32), r2.view(torch.int32)) and torch.equal(t1, t2), t
        assert torch.equal(a_env.state, b_env.state), t
        a_or = ob.sample_actions(seed, t, off, auto_reset)
        ob.step(a_or, None, seed, t, off, auto_reset)
    assert_same_as_oracle(b_env, ob)


def test_wave_per_board_study_kernel_gives_the_same_results():
    """The mapping-study kernel (one wavefront per board; tools/libqttt_study.so, outside the
    product library) is the same function."""
    import ctypes
    import os
    from qtttgym_amd import VecEnv
    import __graft_entry__ as g
    n, seed = 4099, 6
    ref = VecEnv(n, seed=seed, auto_reset=True)
    env = VecEnv(n, seed=seed, auto_reset=True)
    L = ctypes.CDLL(g.build_study())
    vp = ctypes.c_void_p
    L.qttt_step_wave_per_board.argtypes = [vp, vp, vp, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int64, ctypes.c_uint32,
                                           vp, vp, ctypes.c_int64, vp]
    r = torch.empty(n, dtype=torch.float32, device="cuda")
    tm = torch.empty(n, dtype=torch.bool, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for t in range(12):
        a = ref.sample_actions()
        r_ref, t_ref = ref.step_raw(a)
        rc = L.qttt_step_wave_per_board(env.state.data_ptr(), a.data_ptr(), None, seed, t, 0, 1,
                                        r.data_ptr(), tm.data_ptr(), n, s)
        assert rc == 0
        assert torch.equal(r.view(torch.int32), r_ref.view(torch.int32)) and torch.equal(tm, t_ref)
        assert torch.equal(env.state, ref.state)
