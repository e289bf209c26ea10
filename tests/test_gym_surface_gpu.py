"""The default gym surface of VecEnv — `obs, reward, terminated, truncated, info = env.step(actions)` and
`obs, info = env.reset()` as the reference's Env returns them (env.py:34-57): new objects every call (env.py:46,68-85
builds fresh lists), ONE kernel per call, nothing copied afterwards.  What is checked here:
  * bit-exact against the reference's recorded traces while the caller keeps EVERY step's tensors (none is ever
    written again), while it keeps (obs, next_obs) pairs, and while it rebinds — with the default (one fresh allocation
    from torch's caching allocator per call, carved by csrc/fastviews.cpp) and with VecEnv(output_pool=N) (sets the
    caller has dropped are re-used: then two sets alternate and nothing is allocated);
  * a default step() / reset() enqueues exactly one kernel and nothing else (the call captured into a hipGraph, the
    graph's nodes listed by the HIP runtime: tests/hip_graph_nodes.py);
  * with the pool: anything that can still see a set's memory (a view, a detached alias, a DLPack capsule, the dict)
    keeps it from being handed out again; another stream gets another set."""
import numpy as np
import pytest
import torch

from hip_graph_nodes import kernels_enqueued

pytestmark = pytest.mark.gpu

OBS = (("classical", "board"), ("q_states_p1", "q_p1"), ("q_states_p1_len", "q_p1_len"), ("q_states_p2", "q_p2"),
       ("q_states_p2_len", "q_p2_len"), ("turn", "turn"))


def _np(t):
    return t.cpu().numpy()


def _check_step(golden, t, obs, reward, term):
    for k, g in OBS:
        assert np.array_equal(_np(obs[k]), golden[g][:, t]), (k, t)
    assert np.array_equal(_np(reward).view(np.uint32), golden["reward"][:, t].astype(np.float32).view(np.uint32)), t
    assert np.array_equal(_np(term).astype(np.uint8), golden["terminated"][:, t]), t


def _inputs(golden, t):
    return torch.from_numpy(golden["actions"][:, t].copy()).cuda(), torch.from_numpy(golden["bits"][:, t].copy()).cuda()


def test_every_step_kept_and_checked_at_the_end(golden):
    """A caller that keeps every observation of the episode (a replay buffer of references): all T steps' tensors are
    compared with the reference's recording AFTER the last step — none was overwritten on the way."""
    from qtttgym_amd import VecEnv
    E, T = golden["bits"].shape
    env = VecEnv(E)
    obs0, info = env.reset()
    assert info == {}
    kept = []
    for t in range(T):
        kept.append(env.step(*_inputs(golden, t)))
    torch.cuda.synchronize()
    for t, (obs, reward, term, trunc, info) in enumerate(kept):
        _check_step(golden, t, obs, reward, term)
        assert info == {} and not bool(trunc.any())
    # the reset's observation too: the empty board (env.py:55-57,68-85)
    assert bool((obs0["classical"] == -1).all()) and int(obs0["turn"].sum()) == 0
    assert int(obs0["q_states_p1_len"].sum()) == 0 and int(obs0["q_states_p2_len"].sum()) == 0
    assert bool((obs0["q_states_p1"] == 255).all()) and bool((obs0["q_states_p2"] == 255).all())
    ptrs = {o["classical"].data_ptr() for o, *_ in kept} | {obs0["classical"].data_ptr()}
    assert len(ptrs) == T + 1                                           # T + 1 different allocations


@pytest.mark.parametrize("pool", [0, 4])
def test_step_t_is_intact_after_step_t_plus_1_and_a_rebinding_caller_allocates_nothing(golden, pool):
    """(obs, next_obs) pairs: step t's tensors are unchanged after step t + 1 (and t + 2).  A caller that rebinds its
    names every step: with output_pool=4 it is served from two or three output sets in turn, not one allocator call per
    step; with the default (a fresh allocation per call) the allocator recycles the blocks the caller dropped — the
    memory in use does not grow."""
    from qtttgym_amd import VecEnv, vec_env
    E, T = golden["bits"].shape
    env = VecEnv(E, output_pool=pool)
    import os
    built = os.path.exists(os.path.join(os.path.dirname(os.path.abspath(vec_env.__file__)), "_fastviews.so"))
    assert (vec_env._fastviews is not None) is built  # qtttgym_amd/_fastviews.so is optional: in use iff built
    obs, _ = env.reset()
    seen = set()
    prev = None
    for t in range(T):
        nxt = env.step(*_inputs(golden, t))
        if prev is not None:
            _check_step(golden, t - 1, prev[0], prev[1], prev[2])       # step t-1 after step t was enqueued
        _check_step(golden, t, nxt[0], nxt[1], nxt[2])
        prev = nxt                                                       # the set of step t-1 is released HERE
        seen.add(nxt[1].data_ptr())
    a, b = _inputs(golden, 0)
    stats = torch.cuda.memory_stats()
    calls, in_use = stats["allocation.all.allocated"], stats["allocated_bytes.all.current"]
    for t in range(50):
        obs, reward, term, trunc, info = env.step(a, b)
    stats = torch.cuda.memory_stats()
    if pool:
        assert len(seen) <= 3 and len(env._pool) <= 3, (len(seen), len(env._pool))
        assert stats["allocation.all.allocated"] == calls                     # not one allocator call in 50 steps
    else:
        assert env._pool == [] and stats["allocation.all.allocated"] == calls + 50    # one allocation per step, exactly
        assert stats["allocated_bytes.all.current"] <= in_use + 2 * (34 * E + 8 * 512)    # ... and at most two alive


def test_default_step_and_reset_enqueue_exactly_one_kernel():
    """the default step() used to be the fused kernel + eight clone kernels.  One call captured into a
    hipGraph holds exactly ONE node, a kernel (no memset, no copy) — with hashed and with explicit collapse bits, at a small and at the
    headline batch size; reset() likewise (qttt_reset_observe); the zero-copy forms too."""
    from qtttgym_amd import VecEnv
    for n, pool in ((4096, 0), (1 << 20, 0), (4096, 3)):
        env = VecEnv(n, seed=3, auto_reset=True, output_pool=pool)
        a = env.sample_actions()
        bits = torch.zeros(n, dtype=torch.uint8, device="cuda")
        assert kernels_enqueued(lambda: env.step(a)) == (1, 1), n
        assert kernels_enqueued(lambda: env.step(a, bits)) == (1, 1), n
        assert kernels_enqueued(lambda: env.reset()) == (1, 1), n
        assert kernels_enqueued(lambda: env.step(a, copy_obs=False)) == (1, 1), n
        assert kernels_enqueued(lambda: env.reset(copy_obs=False)) == (1, 1), n
        assert kernels_enqueued(lambda: env.step_raw(a)) == (1, 1), n
        # the counter works: what the default step() was up to round 5 — the kernel and eight copies (a captured
        # .clone() is a device-to-device copy node)
        def old_default():
            obs, r, tm = env.step_observe_raw(a)
            return {k: v.clone() for k, v in obs.items()}, r.clone(), tm.clone()
        assert kernels_enqueued(old_default) == (1, 9), n


def test_anything_that_can_see_a_set_keeps_it_from_being_reused():
    from qtttgym_amd import VecEnv
    n = 2048
    env = VecEnv(n, seed=5, auto_reset=True, output_pool=2)
    a = env.sample_actions()

    def step_keeping(what):
        """one step whose outputs are dropped except for `what(outputs)`; returns (kept object, its expected bytes,
        the reward pointer of the set)"""
        obs, r, tm, _, _ = env.step(a)
        k = what(obs, r, tm)
        return k, k.clone() if torch.is_tensor(k) else None, r.data_ptr()

    holders = {
        "a view": lambda obs, r, tm: obs["classical"][5:9],
        "a detached alias": lambda obs, r, tm: obs["q_states_p1"].detach(),
        "a reshaped alias": lambda obs, r, tm: r.view(torch.int32),
        "the tensor itself": lambda obs, r, tm: tm,
    }
    for name, what in holders.items():
        kept, want, ptr = step_keeping(what)
        ptrs = set()
        for _ in range(6):                               # more steps than the pool has sets
            o, r, tm, _, _ = env.step(env.sample_actions())
            ptrs.add(r.data_ptr())
            del o, r, tm
        assert ptr not in ptrs, name                     # the held set was never handed out again
        assert torch.equal(kept, want), name
        del kept
    # a DLPack capsule holds the TensorImpl, not a Python reference
    obs, r, tm, _, _ = env.step(a)
    cap, ptr = torch.utils.dlpack.to_dlpack(obs["turn"]), r.data_ptr()
    del obs, r, tm
    for _ in range(6):
        o, r, tm, _, _ = env.step(a)
        assert r.data_ptr() != ptr
        del o, r, tm
    del cap
    # released: the pool goes back to alternating between its sets
    ptrs = set()
    for _ in range(8):
        o, r, tm, _, _ = env.step(a)
        ptrs.add(r.data_ptr())
    assert len(ptrs) <= 3


def test_another_stream_gets_another_set_and_output_pool_zero_always_allocates(golden):
    from qtttgym_amd import VecEnv
    E = golden["bits"].shape[0]
    env = VecEnv(E, output_pool=4)
    a, b = _inputs(golden, 0)
    o, r, tm, _, _ = env.step(a, b)
    p0 = r.data_ptr()
    del o, r, tm
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        o, r, tm, _, _ = env.step(*_inputs(golden, 1))
        assert r.data_ptr() != p0                        # the free set was last written on the other stream
        side.synchronize()
        _check_step(golden, 1, o, r, tm)
    torch.cuda.synchronize()
    env0 = VecEnv(E, output_pool=0)
    ptrs, kept = set(), []
    for t in range(4):
        out = env0.step(*_inputs(golden, t))
        _check_step(golden, t, out[0], out[1], out[2])
        kept.append(out)
        ptrs.add(out[1].data_ptr())
    assert len(ptrs) == 4 and env0._pool == []


def test_default_step_with_a_device_step_counter_and_inside_a_graph():
    """use_device_step_counter() (graph mode) reaches the default step() too; default steps captured into a hipGraph
    write into the graph's own pool (never into a pooled set) and replay bit-identically to eager steps."""
    from qtttgym_amd import VecEnv
    n, T = 4096, 6
    ref, env = VecEnv(n, seed=9, auto_reset=True), VecEnv(n, seed=9, auto_reset=True, output_pool=2)
    acts = torch.empty((T, n, 2), dtype=torch.uint8, device="cuda")
    want = []
    for t in range(T):
        ref.sample_actions(out=acts[t])
        o, r, tm, _, _ = ref.step(acts[t])
        want.append((o, r, tm))
    env.use_device_step_counter()
    o, r, tm, _, _ = env.step(acts[0])                   # eager, counter on the device
    assert torch.equal(r.view(torch.int32), want[0][1].view(torch.int32)) and torch.equal(o["classical"], want[0][0]["classical"])
    env.reset()
    env.use_device_step_counter()
    pool_ptrs = {s.t[0].data_ptr() for s in env._pool}
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    got = []
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            for t in range(T):
                o, r, tm, _, _ = env.step(acts[t])
                got.append((o, r, tm))
            # (a graph of steps ends by advancing the counter, as VecEnv.capture does; here one replay only)
    torch.cuda.current_stream().wait_stream(side)
    assert not ({g[1].data_ptr() for g in got} & pool_ptrs)
    env.reset_raw()
    graph.replay()
    torch.cuda.synchronize()
    for t in range(T):
        for k in want[t][0]:
            assert torch.equal(got[t][0][k], want[t][0][k]), (t, k)
        assert torch.equal(got[t][1].view(torch.int32), want[t][1].view(torch.int32)) and torch.equal(got[t][2], want[t][2])
    assert torch.equal(env.state, ref.state)


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


def test_zero_boards_and_one_board():
    """the edges of the batch size: an empty environment steps and resets without a launch; one board is one board"""
    from qtttgym_amd import VecEnv
    e0 = VecEnv(0)
    obs, info = e0.reset()
    assert obs["classical"].shape == (0, 9) and info == {}
    o, r, tm, tr, info = e0.step(torch.empty((0, 2), dtype=torch.uint8, device="cuda"))
    assert r.shape == (0,) and tm.shape == (0,) and o["q_states_p1"].shape == (0, 5, 2) and o["turn"].shape == (0,)
    e1 = VecEnv(1)
    obs, _ = e1.reset()
    assert obs["classical"].tolist() == [[-1] * 9] and obs["turn"].tolist() == [0]
    o, r, tm, tr, info = e1.step(torch.tensor([[0, 1]], dtype=torch.uint8, device="cuda"))
    assert o["q_states_p1"][0, 0].tolist() == [0, 1] and int(o["q_states_p1_len"]) == 1 and int(o["turn"]) == 1     # K1, SURVEY Appendix A
    o2, r2, tm2, _, _ = e1.step(torch.tensor([[1, 0]], dtype=torch.uint8, device="cuda"), torch.tensor([0], dtype=torch.uint8, device="cuda"))
    assert o2["classical"][0].tolist() == [1, 0, -1, -1, -1, -1, -1, -1, -1] and int(o2["turn"]) == 0
    assert o["classical"][0].tolist() == [-1] * 9                                      # step 1's tensors are still step 1's
    assert r2.view(torch.int32).item() == -(2 ** 31) and not bool(tm2)                 # -0.0 (env.py:49)


@pytest.mark.parametrize("n", [1, 2, 7, 63, 64, 65, 1000, 4097, 65537, 300001])
def test_reset_observe_writes_exactly_its_buffers_at_any_alignment(n):
    """include/qttt.h qttt_reset_observe (Env.reset with its observation, env.py:55-57,68-85): seven fills in one launch,
    16-byte pieces + the unaligned head / tail of a caller's odd pointer bytewise.  Every output at every byte offset
    0..3 and 13 into a guard-filled buffer: the values are the empty board's, the guard bytes around them untouched;
    the state equals qttt_reset's."""
    from qtttgym_amd import VecEnv, _native
    L = _native.lib()
    env, ref = VecEnv(n, seed=1), VecEnv(n, seed=1)
    for _ in range(3):
        env.step_raw(env.sample_actions())
    ref.reset_raw()
    sizes = {"classical": 9 * n, "q_p1": 10 * n, "q_p1_len": n, "q_p2": 8 * n, "q_p2_len": n, "turn": n}
    want = {"classical": 255, "q_p1": 255, "q_p1_len": 0, "q_p2": 255, "q_p2_len": 0, "turn": 0}
    GUARD, FILL = 64, 0x5A
    for shift in (0, 1, 2, 3, 13):
        bufs = {k: torch.full((GUARD + shift + b + GUARD,), FILL, dtype=torch.uint8, device="cuda") for k, b in sizes.items()}
        ptr = {k: bufs[k].data_ptr() + GUARD + shift for k in sizes}
        rc = L.qttt_reset_observe(env.state.data_ptr(), ptr["classical"], ptr["q_p1"], ptr["q_p1_len"], ptr["q_p2"],
                                  ptr["q_p2_len"], ptr["turn"], n, torch.cuda.current_stream().cuda_stream)
        assert rc == 0
        torch.cuda.synchronize()
        for k, b in sizes.items():
            t = bufs[k]
            lo = GUARD + shift
            assert bool((t[:lo] == FILL).all()) and bool((t[lo + b:] == FILL).all()), (k, shift, "guard")
            assert bool((t[lo:lo + b] == want[k]).all()), (k, shift)
        assert torch.equal(env.state, ref.state)
        for _ in range(2):                                   # dirty the state again for the next alignment
            env.step_raw(env.sample_actions())
    assert L.qttt_reset_observe(None, 1, 1, 1, 1, 1, 1, n, None) == -1 and L.qttt_reset_observe(env.state.data_ptr(), 1, 1, 1, 1, 1, 1, -1, None) == -2
    assert L.qttt_reset_observe(None, None, None, None, None, None, None, 0, None) == 0
