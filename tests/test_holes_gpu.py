"""Round-2 additions: every exported launch shape and host-side state path against the oracle
(`-m gpu`, through the C ABI): the boards-per-lane knob, explicit bits + auto-reset, BASELINE
config 4's real shape (8 shards of 2 097 152 boards), checkpoint/restore, and the fused
step + observation kernel (qttt_step_observe) against qttt_step + qttt_observe and the oracle."""
import numpy as np
import pytest
import torch

import oracle

pytestmark = pytest.mark.gpu


def _np(t):
    return t.cpu().numpy()


def _assert_same_as_oracle(env, ob, tag=""):
    ex = {k: _np(v) for k, v in env.export_boards().items()}
    assert np.array_equal(ex["board"], ob.board), tag
    assert np.array_equal(ex["n_moves"], ob.n_moves), tag
    assert np.array_equal(ex["moves"], ob.moves), tag
    assert np.array_equal(ex["n_q"], ob.n_q), tag
    assert np.array_equal(ex["qmask"].view(np.uint16), ob.qmask), tag


def _assert_obs_equals_oracle(obs, ob, tag=""):
    classical, q1, l1, q2, l2, turn = ob.observe()
    assert np.array_equal(_np(obs["classical"]), classical), tag
    assert np.array_equal(_np(obs["q_states_p1"]), q1), tag
    assert np.array_equal(_np(obs["q_states_p1_len"]), l1), tag
    assert np.array_equal(_np(obs["q_states_p2"]), q2), tag
    assert np.array_equal(_np(obs["q_states_p2_len"]), l2), tag
    assert np.array_equal(_np(obs["turn"]), turn), tag


@pytest.fixture
def tuning():
    """qttt_set_tuning is process-wide: always put the default (shape chosen by batch size) back."""
    from qtttgym_amd import _native
    L = _native.lib()
    yield L.qttt_set_tuning
    assert L.qttt_set_tuning(0, 0) == 0


@pytest.mark.parametrize("bpl,blk", [(1, 0), (2, 0), (4, 0), (1, 256), (2, 256), (1, 1024), (2, 1024), (0, 0)])
@pytest.mark.parametrize("n", [1, 3, 64, 65, 257, 4099, 262144])
def test_every_launch_shape_vs_oracle(tuning, bpl, blk, n):
    """qttt_set_tuning(boards per lane 1|2|4, workgroup size 256|512|1024; 0 = by batch size) is exported
    ABI: each launch shape (plus its ragged tail of n mod bpl boards) is the same function as the
    oracle, with and without auto-reset."""
    from qtttgym_amd import VecEnv, _native
    assert tuning(bpl, blk) == 0
    assert tuning(3, 0) != 0 and tuning(2, 128) != 0 and tuning(-1, 0) != 0
    shape = _native.step_launch_shape(n)
    if bpl:
        auto_blk = 256 if n <= 448 * 1024 else 1024
        assert shape == (bpl, 512 if bpl == 4 else (blk or auto_blk))
    else:
        assert shape == ((1, 256) if n <= 448 * 1024 else (2, 1024))
    for auto_reset in (False, True):
        seed, off = 77 + bpl, 11 * n
        env = VecEnv(n, seed=seed, auto_reset=auto_reset, board_offset=off)
        ob = oracle.OracleBoards(n)
        for t in range(12 if n < 100000 else 10):
            a = env.sample_actions()
            a_or = ob.sample_actions(seed, t, off, auto_reset)
            assert np.array_equal(_np(a), a_or), (t, auto_reset)
            reward, term = env.step_raw(a)
            r_or, t_or = ob.step(a_or, None, seed, t, off, auto_reset)
            assert np.array_equal(_np(reward).view(np.uint32), r_or.view(np.uint32)), (t, auto_reset)
            assert np.array_equal(_np(term).astype(np.uint8), t_or), (t, auto_reset)
        _assert_same_as_oracle(env, ob, (bpl, n, auto_reset))


@pytest.mark.parametrize("n,shape", [(450001, (1, 256)), (520001, (1, 1024)), (600001, (2, 512)),
                                     (1048577, (2, 1024)), (1600003, (2, 256))])
def test_launch_shape_chosen_by_batch_size_vs_oracle(n, shape):
    """The library picks boards per lane and workgroup size from the batch size: every
    region of that table, odd batch sizes, against the oracle."""
    from qtttgym_amd import VecEnv, _native
    assert _native.step_launch_shape(n) == shape
    seed, off = 19, 5 * n
    env = VecEnv(n, seed=seed, auto_reset=True, board_offset=off)
    ob = oracle.OracleBoards(n)
    for t in range(7):
        a = env.sample_actions()
        a_or = ob.sample_actions(seed, t, off, True)
        assert np.array_equal(_np(a), a_or), t
        reward, term = env.step_raw(a)
        r_or, t_or = ob.step(a_or, None, seed, t, off, True)
        assert np.array_equal(_np(reward).view(np.uint32), r_or.view(np.uint32)), t
        assert np.array_equal(_np(term).astype(np.uint8), t_or), t
    _assert_same_as_oracle(env, ob, (n, shape))


@pytest.mark.parametrize("n", [1001, 65536])
def test_explicit_bits_with_auto_reset_vs_oracle(n):
    """step_kernel<., HAS_BITS = true, AUTO_RESET = true>: explicit collapse bits in throughput
    mode, against OracleBoards.step(bits, auto_reset=True)."""
    from qtttgym_amd import VecEnv
    seed, off = 5, 3 * n
    rng = np.random.default_rng(n)
    env = VecEnv(n, seed=seed, auto_reset=True, board_offset=off)
    ob = oracle.OracleBoards(n)
    n_term = 0
    for t in range(24):
        a = env.sample_actions()
        a_or = ob.sample_actions(seed, t, off, True)
        assert np.array_equal(_np(a), a_or), t
        bits = rng.integers(0, 256, size=n, dtype=np.uint8)          # only bit 0 counts
        reward, term = env.step_raw(a, torch.from_numpy(bits).cuda())
        r_or, t_or = ob.step(a_or, bits, seed, t, off, True)
        assert np.array_equal(_np(reward).view(np.uint32), r_or.view(np.uint32)), t
        assert np.array_equal(_np(term).astype(np.uint8), t_or), t
        n_term += int(t_or.sum())
    assert n_term > 2 * n                                             # every board restarted twice on average
    _assert_same_as_oracle(env, ob)


def test_config4_eight_shards_of_two_million_boards_equal_the_single_run():
    """BASELINE config 4's actual shape: 2 097 152 boards as 8 shards of 262 144
    (dist.make_sharded_env), here run one after another on the one GPU, == the single 2 M run."""
    from qtttgym_amd import VecEnv
    from qtttgym_amd.dist import make_sharded_env, shard_range, EpisodeCounters
    n, G, seed, T = 2_097_152, 8, 19, 10
    full = VecEnv(n, seed=seed, auto_reset=True)
    cf = EpisodeCounters("cuda")
    r_full = torch.empty((T, n), dtype=torch.float32, device="cuda")
    t_full = torch.empty((T, n), dtype=torch.bool, device="cuda")
    for t in range(T):
        r, tm = full.step_random()
        cf.update(r, tm)
        r_full[t], t_full[t] = r, tm
    ef = full.export_boards()
    total = torch.zeros(4, dtype=torch.int64, device="cuda")
    for rank in range(G):
        lo, hi = shard_range(n, rank, G)
        assert (lo, hi) == (rank * 262144, (rank + 1) * 262144)
        sh = make_sharded_env(n, rank, G, "cuda", seed=seed, auto_reset=True)
        assert sh.num_envs == 262144 and sh.board_offset == lo
        cs = EpisodeCounters("cuda")
        for t in range(T):
            r, tm = sh.step_random()
            cs.update(r, tm)
            assert torch.equal(r.view(torch.int32), r_full[t, lo:hi].view(torch.int32)), (rank, t)
            assert torch.equal(tm, t_full[t, lo:hi]), (rank, t)
        es = sh.export_boards()
        for key in ef:
            assert torch.equal(es[key], ef[key][lo:hi]), (rank, key)
        total += cs.c
    assert total.tolist() == cf.c.tolist() and int(total[0]) > n       # sum of the shards' counters


def test_state_dict_round_trip_continues_identically():
    """Checkpoint / resume: state_dict -> load_state_dict into a fresh environment,
    then both continue bit-identically (hash bits depend on seed, step_idx, board_offset)."""
    from qtttgym_amd import VecEnv
    n = 5000
    env = VecEnv(n, seed=23, auto_reset=True, board_offset=12345)
    for t in range(7):
        env.step_random()
    sd = env.state_dict()
    # the checkpoint is a copy, not a view: advancing env must not change it
    snap = sd["state"].clone()
    env.step_random()
    assert torch.equal(sd["state"], snap)
    env2 = VecEnv(n, seed=0, auto_reset=False)
    env2.load_state_dict(sd)
    assert (env2.seed, env2.step_idx, env2.board_offset, env2.auto_reset) == (23, 7, 12345, True)
    env.load_state_dict(sd)                                            # rewind the original too
    for t in range(12):
        a1, a2 = env.sample_actions(), env2.sample_actions()
        assert torch.equal(a1, a2), t
        r1, t1 = env.step_raw(a1)
        r1, t1 = r1.clone(), t1.clone()
        r2, t2 = env2.step_raw(a2)
        assert torch.equal(r1.view(torch.int32), r2.view(torch.int32)) and torch.equal(t1, t2), t
        assert torch.equal(env.state, env2.state), t
    with pytest.raises(ValueError):
        VecEnv(n + 64).load_state_dict(sd)


@pytest.mark.parametrize("bpl", [1, 2, 4])
@pytest.mark.parametrize("n", [1, 2, 63, 1001, 4099, 70001])
def test_fused_step_observe_equals_step_then_observe_and_the_oracle(tuning, bpl, n):
    """qttt_step_observe (one kernel) == qttt_step + qttt_observe (two kernels) == the oracle's
    Env._observation, at ragged sizes, for every launch shape, with and without auto-reset."""
    from qtttgym_amd import VecEnv
    assert tuning(bpl, 0) == 0
    for auto_reset in (False, True):
        seed, off = 31 + bpl, 5 * n
        fused = VecEnv(n, seed=seed, auto_reset=auto_reset, board_offset=off)
        split = VecEnv(n, seed=seed, auto_reset=auto_reset, board_offset=off)
        ob = oracle.OracleBoards(n)
        for t in range(12):
            a = split.sample_actions()
            a_or = ob.sample_actions(seed, t, off, auto_reset)
            r_s, t_s = split.step_raw(a)
            r_s, t_s = r_s.clone(), t_s.clone()
            obs_s = {k: v.clone() for k, v in split.observ().items()}
            obs_f, r_f, t_f = fused.step_observe_raw(a)
            assert torch.equal(r_f.view(torch.int32), r_s.view(torch.int32)) and torch.equal(t_f, t_s), t
            assert torch.equal(fused.state, split.state), t
            for k in obs_s:
                assert torch.equal(obs_f[k], obs_s[k]), (t, k)
            r_or, t_or = ob.step(a_or, None, seed, t, off, auto_reset)
            assert np.array_equal(_np(r_f).view(np.uint32), r_or.view(np.uint32)), t
            if n <= 5000:                                               # the oracle's observe() is a Python loop
                _assert_obs_equals_oracle(obs_f, ob, (t, auto_reset))


def test_fused_step_observe_with_explicit_bits_and_offset_views():
    """Explicit bits through the fused kernel, with every output a view offset by one board (the
    tiles keep each output's own alignment phase), against the two-kernel path."""
    from qtttgym_amd import VecEnv, _native
    n, seed = 3001, 4
    L = _native.lib()
    ref = VecEnv(n, seed=seed)
    env = VecEnv(n, seed=seed)
    big = {"classical": torch.zeros((n + 1, 9), dtype=torch.int8, device="cuda"),
           "q_states_p1": torch.zeros((n + 1, 5, 2), dtype=torch.uint8, device="cuda"),
           "q_states_p1_len": torch.zeros(n + 1, dtype=torch.uint8, device="cuda"),
           "q_states_p2": torch.zeros((n + 1, 4, 2), dtype=torch.uint8, device="cuda"),
           "q_states_p2_len": torch.zeros(n + 1, dtype=torch.uint8, device="cuda"),
           "turn": torch.zeros(n + 1, dtype=torch.uint8, device="cuda")}
    r = torch.empty(n, dtype=torch.float32, device="cuda")
    tm = torch.empty(n, dtype=torch.bool, device="cuda")
    s = torch.cuda.current_stream().cuda_stream
    for t in range(10):
        a = ref.sample_actions()
        bits = torch.randint(0, 2, (n,), dtype=torch.uint8, device="cuda")
        r_ref, t_ref = ref.step_raw(a, bits)
        obs_ref = ref.observ()
        rc = L.qttt_step_observe(env.state.data_ptr(), a.data_ptr(), bits.data_ptr(), seed, t, 0, 0,
                                 r.data_ptr(), tm.data_ptr(), big["classical"][1:].data_ptr(),
                                 big["q_states_p1"][1:].data_ptr(), big["q_states_p1_len"][1:].data_ptr(),
                                 big["q_states_p2"][1:].data_ptr(), big["q_states_p2_len"][1:].data_ptr(),
                                 big["turn"][1:].data_ptr(), n, s)
        assert rc == 0
        assert torch.equal(r.view(torch.int32), r_ref.view(torch.int32)) and torch.equal(tm, t_ref)
        for k in big:
            assert torch.equal(big[k][1:], obs_ref[k]), (t, k)
            assert int(big[k][0].abs().sum()) == 0, k                  # the board before the view is untouched
    # q_p1 rows are written with 2-byte, q_p2 rows with 8-byte LDS stores: odd bases are refused
    odd = torch.zeros(10 * n + 1, dtype=torch.uint8, device="cuda")
    assert L.qttt_step_observe(env.state.data_ptr(), a.data_ptr(), None, seed, 0, 0, 0, r.data_ptr(), tm.data_ptr(),
                               big["classical"].data_ptr(), odd[1:].data_ptr(), big["q_states_p1_len"].data_ptr(),
                               big["q_states_p2"].data_ptr(), big["q_states_p2_len"].data_ptr(),
                               big["turn"].data_ptr(), n, s) == -3


def test_export_into_offset_views_equals_plain_export():
    """qttt_export into a caller's views offset by one board (odd byte addresses for the 18-, 9- and
    1-byte rows) gives the same rows as into fresh tensors and touches nothing in front of the view,
    at a ragged size, mid-game and at the end of the game."""
    from qtttgym_amd import VecEnv, _native
    n = 1003
    L = _native.lib()
    env = VecEnv(n, seed=14)
    s = torch.cuda.current_stream().cuda_stream
    for t in range(9):
        env.step_random()
        want = env.export_boards()
        big = {"moves": torch.zeros((n + 1, 9, 2), dtype=torch.uint8, device="cuda"),
               "n_moves": torch.zeros(n + 1, dtype=torch.uint8, device="cuda"),
               "board": torch.zeros((n + 1, 9), dtype=torch.int8, device="cuda"),
               "qmask": torch.zeros((n + 1, 4), dtype=torch.int16, device="cuda"),
               "n_q": torch.zeros(n + 1, dtype=torch.uint8, device="cuda")}
        rc = L.qttt_export(env.state.data_ptr(), big["moves"][1:].data_ptr(), big["n_moves"][1:].data_ptr(),
                           big["board"][1:].data_ptr(), big["qmask"][1:].data_ptr(), big["n_q"][1:].data_ptr(), n, s)
        assert rc == 0
        for k in want:
            assert torch.equal(big[k][1:], want[k]), (t, k)
            assert int(big[k][0].to(torch.int64).abs().sum()) == 0, k


def test_ragged_batch_on_the_1024_thread_workgroup_path():
    """Batches of >= 1 M boards run with 1024-thread workgroups; a size that leaves the last workgroup
    with three lane-groups and one odd board (its own 1-board-per-lane launch) must still be the
    oracle's function — step alone, and step + observation against the two-kernel path."""
    from qtttgym_amd import VecEnv
    n, seed, off = (1 << 20) + 7, 61, 1234567
    env = VecEnv(n, seed=seed, auto_reset=True, board_offset=off)
    fused = VecEnv(n, seed=seed, auto_reset=True, board_offset=off)
    ob = oracle.OracleBoards(n)
    for t in range(10):
        a = env.sample_actions()
        a_or = ob.sample_actions(seed, t, off, True)
        assert np.array_equal(_np(a), a_or), t
        r, tm = env.step_raw(a)
        r_or, t_or = ob.step(a_or, None, seed, t, off, True)
        assert np.array_equal(_np(r).view(np.uint32), r_or.view(np.uint32)), t
        assert np.array_equal(_np(tm).astype(np.uint8), t_or), t
        obs_f, r_f, t_f = fused.step_observe_raw(a)
        assert torch.equal(r_f.view(torch.int32), r.view(torch.int32)) and torch.equal(t_f, tm), t
        assert torch.equal(fused.state, env.state), t
        obs_s = env.observ()
        for k in obs_s:
            assert torch.equal(obs_f[k], obs_s[k]), (t, k)
    _assert_same_as_oracle(env, ob)
