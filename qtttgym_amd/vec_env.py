"""VecEnv — N independent Quantum Tic-Tac-Toe boards advanced per call on one MI355X.

Same method names as the reference's gymnasium-style `Env` (env.py:15-85): reset / step /
observ / turn / action_space / observation_space, returning tensors of leading dimension N.
All arithmetic happens in libqttt_hip.so (include/qttt.h); torch is used for device memory
and streams only.
"""
import ctypes
import sys

import torch

from . import _native
from .spaces import reference_action_space, reference_observation_space


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
if _raw_stream is None:                                   # older torch: the documented, slower way
    def _raw_stream(index):
        return torch.cuda.current_stream(index).cuda_stream


def _ptr(t):
    return 0 if t is None else t.data_ptr()


def _layout(n):
    """Byte offsets of the eight tensors a default step() returns inside their one allocation (reward f32[n], terminated
    bool[n], q_states_p1 u8[n,5,2], q_states_p1_len u8[n], q_states_p2 u8[n,4,2], q_states_p2_len u8[n], classical i8[n,9],
    turn u8[n]; every one starts on a 512-byte boundary) and, last, the allocation's size.  csrc/fastviews.cpp has the
    same arithmetic (checked by tests/test_fastviews_cpu.py)."""
    offs, total = [], 0
    for b in (4 * n, n, 10 * n, n, 8 * n, n, 9 * n, n):
        offs.append(total)
        total += (b + 511) // 512 * 512
    return tuple(offs) + (total,)


def _carve_py(n, dev):
    """One allocation from torch's caching allocator and the eight tensors as views of it: (tensors, address)."""
    o = _layout(n)
    buf = torch.empty(o[8], dtype=torch.uint8, device=dev)
    cut = lambda k, shape, stride: torch.as_strided(buf, shape, stride, o[k])
    return ([cut(0, (4 * n,), (1,)).view(torch.float32), cut(1, (n,), (1,)).view(torch.bool),
             cut(2, (n, 5, 2), (10, 2, 1)), cut(3, (n,), (1,)), cut(4, (n, 4, 2), (8, 2, 1)), cut(5, (n,), (1,)),
             cut(6, (n, 9), (9, 1)).view(torch.int8), cut(7, (n,), (1,))], buf.data_ptr())


# The same in C++ when qtttgym_amd/_fastviews.so is built (csrc/fastviews.cpp, __graft_entry__.build()): ~2 us of host
# time per call instead of ~12 for the twelve torch calls above — what lets the default step() take a FRESH allocation
# every call and still stay ahead of the kernel at 1 M boards.
try:
    from . import _fastviews
    _carve = _fastviews.carve
except ImportError:
    _fastviews = None
    _carve = _carve_py

# Re-using an output set (VecEnv(output_pool=N), opt-in) needs two counters torch keeps but does not document (the
# storage's and the TensorImpl's use counts); without either, the pool is off and every call allocates
_storage_use_count = getattr(torch._C, "_storage_Use_Count", None)
if not hasattr(torch.Tensor, "_use_count"):
    _storage_use_count = None
_is_capturing = getattr(torch._C, "_cuda_isCurrentStreamCapturing", None) or torch.cuda.is_current_stream_capturing


class _OutputSet:
    """VecEnv(output_pool=N > 0): one step's outputs kept for RE-USE — for loops whose host time per step matters more
    than plain allocator semantics (a pooled step() costs ~2 us of host time less than a fresh allocation; without
    _fastviews.so ~10 us less).

    A set is handed out again only when nothing outside the environment can still see it, which `free()` checks
    exactly: no Python reference to any of its eight tensors beyond the environment's own (sys.getrefcount), no C++
    reference to their TensorImpls (autograd, DLPack: Tensor._use_count) and no other tensor on their storage (views,
    .detach(), .data: the storage's use count) — and only on the stream that wrote it last.  A caller that rebinds
    `obs, reward, terminated, ... = env.step(a)` every step alternates between two sets with no allocation at all; a
    caller that keeps every observation gets a new allocation every step.  What the pool does NOT see: work queued on
    ANOTHER stream that reads a tensor the caller has already dropped — Tensor.record_stream protects an allocation of
    the caching allocator, not a pooled set.  That is why the pool is opt-in and the default step() allocates."""
    __slots__ = ("t", "base", "stream", "_st", "_base")

    def __init__(self, env):
        t, self.base = _carve(env.num_envs, env.device)
        self.t = tuple(t)
        del t
        self.stream = None
        self._st = self.t[0].untyped_storage() if _storage_use_count is not None else None
        self._base = self._probe() if self._st is not None else None

    def _probe(self):
        rc, m = sys.getrefcount, 0
        for t in self.t:
            c = rc(t) + (t._use_count() << 20)
            if c > m:
                m = c
        return m, _storage_use_count(self._st._cdata)

    def free(self):
        return self._st is not None and self._probe() == self._base


def _check_out(t, dtype, shape, dev, what):
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous() or t.device != dev:
        raise ValueError("%s must be a contiguous %s device tensor of shape %s" % (what, dtype, tuple(shape)))
    return t


class VecEnv:
    """Thread-safety: the C library may be called from any number of host threads at once; ONE VecEnv
    (its state, its output buffers and its qttt_env record) belongs to one thread at a time, exactly
    like the reference's mutable Env (env.py:15)."""

    def __init__(self, num_envs, device="cuda", seed=0, auto_reset=False, board_offset=0, launch_shape=None,
                 output_pool=0):
        self.num_envs = int(num_envs)
        if self.num_envs < 0:
            raise ValueError("num_envs must be >= 0")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _native.QtttNativeError(
                "VecEnv runs on an MI355X through libqttt_hip.so only (device=%r); there is no "
                "CPU path" % (device,))
        if not torch.cuda.is_available():
            raise _native.QtttNativeError("no HIP device visible (torch.cuda.is_available() is False)")
        if self.device.index is None:                     # pin "cuda" to the device that is current now
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._lib = _native.lib()
        self.seed = int(seed)
        self.auto_reset = bool(auto_reset)
        self.board_offset = int(board_offset)     # global index of board 0 (multi-GPU shards)
        self._step_host, self._ctr = 0, None      # the step index: a host int, or (graph mode) a device u32
        # (boards per lane, workgroup size) of this environment's step launches, carried by every call's
        # flags (QTTT_FLAG_SHAPE); None = the library picks it from the batch size.  Never changes results.
        self._shape_flags = _native.flag_shape(*launch_shape) if launch_shape else 0
        self.action_space = reference_action_space()
        self.observation_space = reference_observation_space()
        n = self.num_envs
        with torch.cuda.device(self.device):
            self.state = torch.empty(int(self._lib.qttt_state_bytes(n)), dtype=torch.uint8, device=self.device)
            self._reward = torch.empty(n, dtype=torch.float32, device=self.device)
            self._terminated = torch.empty(n, dtype=torch.bool, device=self.device)
            self._truncated = torch.zeros(n, dtype=torch.bool, device=self.device)  # env.py:52
        self._obs = None
        # default step() / reset(): a fresh allocation per call (0, the default), or at most this many output sets kept
        # for re-use (_OutputSet)
        self._pool, self._pool_i, self._pool_max = [], 0, max(0, int(output_pool))
        self._bind_outputs()
        self.reset_raw()

    # ------------------------------------------------------------------ the step index
    @property
    def step_idx(self):
        """Steps taken since reset: keys the collapse-bit / policy hash.  A host int — or, once
        use_device_step_counter() / capture() was called, a device-side u32 (reading it then synchronises)."""
        return self._step_host + (0 if self._ctr is None else int(self._ctr))

    @step_idx.setter
    def step_idx(self, v):
        if self._ctr is None:
            self._step_host = int(v)
        else:
            self._step_host = 0
            self._ctr.fill_(int(v))

    def use_device_step_counter(self):
        """Moves the step index into a device-side u32 that the step kernels read when they RUN
        (qttt_env.step_counter): launches captured in a hipGraph then use a fresh index on every replay.
        Eager calls keep working; each of them advances the counter with one extra one-lane launch, and the
        step launches that read the counter run in one shape (one board per lane, 256-thread workgroups): this is the
        mode for small, launch-bound batches — at 1 M boards the ordinary path is ~20 % faster."""
        if self._ctr is None:
            with torch.cuda.device(self.device):
                self._ctr = torch.tensor(self._step_host, dtype=torch.int32, device=self.device)
            self._step_host = 0
            self._rec.step_counter = self._ctr.data_ptr()
        return self._ctr

    def _advance(self, k):
        if self._ctr is None:
            self._step_host += k
        else:
            _native.check(self._launch(self._lib.qttt_counter_add, self._ctr.data_ptr(), k, self._stream()), "qttt_counter_add")

    # ------------------------------------------------------------------ helpers
    def _bind_outputs(self):
        """Addresses of the environment's own output buffers, looked up once (the per-step calls are
        host-bound below ~500 K boards: every data_ptr() and context switch saved is throughput)."""
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._p_reward, self._p_term = self._reward.data_ptr(), self._terminated.data_ptr()
        self._p_obs = None
        # include/qttt.h struct qttt_env: the per-step calls then pass 6 arguments instead of 11 - 17
        self._rec = _native.EnvRecord(state=self.state.data_ptr(), n=self.num_envs, reward=self._p_reward,
                                      terminated=self._p_term)
        self._rec_ref = ctypes.byref(self._rec)
        self._env_step = self._lib.qttt_env_step
        # the record a default step() / reset() points at ITS output tensors (read by the library during the call only)
        self._out_offs = _layout(self.num_envs)
        self._rec_out = _native.EnvRecord(n=self.num_envs)
        self._rec_out_ref = ctypes.byref(self._rec_out)

    def _record(self):
        """The qttt_env record with the fields a caller may have changed since the last step."""
        r = self._rec
        r.state = self.state.data_ptr()
        r.board_offset, r.seed, r.flags = self.board_offset, self.seed, self._flags()
        return self._rec_ref

    def _launch(self, fn, *args):
        """Calls into the library with this environment's device current."""
        if torch.cuda.current_device() == self._dev_index:
            return fn(*args)
        with torch.cuda.device(self.device):
            return fn(*args)

    def _stream(self):
        """The caller's current stream on this environment's device, as a raw hipStream_t
        (torch.cuda.current_stream(dev).cuda_stream costs ~3 us per call, the raw lookup ~0.1)."""
        return _raw_stream(self._dev_index)

    def _flags(self):
        return (_native.FLAG_AUTO_RESET if self.auto_reset else 0) | self._shape_flags

    def _as_actions(self, actions):
        if not torch.is_tensor(actions):
            actions = torch.as_tensor(actions)
        if actions.shape != (self.num_envs, 2):
            raise ValueError("actions must have shape (%d, 2), got %s" % (self.num_envs, tuple(actions.shape)))
        if actions.dtype != torch.uint8:
            # anything outside 0..8 is a noop in the reference (IndexError swallowed at env.py:41);
            # map it to 255 before narrowing so that e.g. 256 does not wrap to square 0
            a = actions.to(torch.int64)
            actions = torch.where((a < 0) | (a > 255), torch.full_like(a, 255), a).to(torch.uint8)
        return actions.to(self.device).contiguous()

    # what gym-style callers touch besides reset / step (gymnasium.vector's names for the per-board spaces; the reference's
    # Env declares the per-board spaces, env.py:19-25, and those are what action_space / observation_space hold here)
    @property
    def single_action_space(self):
        return self.action_space

    @property
    def single_observation_space(self):
        return self.observation_space

    @property
    def unwrapped(self):
        return self

    def close(self):
        """Drops the pooled output sets (the state and the tensors a caller holds stay valid)."""
        self._pool, self._pool_i = [], 0

    def synchronize(self):
        """torch.cuda.synchronize(device) for a process that also uses the single-board façades (Board, Env): their
        mailbox wave (include/qttt.h, qttt_board_op_host) is asked to leave first, so the device-wide wait has nothing
        of this library to wait for (otherwise: up to the wave's idle window, 20 us by default)."""
        self._lib.qttt_board_mailbox_retire(1)
        torch.cuda.synchronize(self.device)

    # ------------------------------------------------------------------ gym surface
    def reset_raw(self, seed=None):
        """Fresh boards without building the observation (one launch of zero stores on the stream)."""
        if seed is not None:
            self.seed = int(seed)
        self.step_idx = 0                         # (the device counter's fill is ordered on the stream like the memset)
        with torch.cuda.device(self.device):
            _native.check(self._lib.qttt_reset(self.state.data_ptr(), self.num_envs, self._stream()),
                          "qttt_reset")

    def _fresh_outputs(self):
        """The eight tensors of one default step() / reset() — reward, terminated, q_states_p1, q_states_p1_len,
        q_states_p2, q_states_p2_len, classical, turn — which nobody else holds, and self._rec_out pointing at them.
        Default: ONE new allocation from torch's caching allocator per call, carved into the eight views (csrc/fastviews.cpp
        when built): plain allocator semantics, nothing is ever re-used behind the caller's back.  With output_pool=N: a
        pooled set that is free (_OutputSet), else a new one.  Inside a hipGraph capture every call allocates (from the
        graph's private pool, which lives as long as the graph): a captured launch keeps writing where it was captured."""
        if self._pool_max and not _is_capturing():
            stream = self._stream()
            pool, i = self._pool, self._pool_i
            k = len(pool)
            s = None
            for _ in range(k):
                i = i + 1 if i + 1 < k else 0
                c = pool[i]
                # (re-use only on the stream that wrote it last: the same rule as torch's caching allocator)
                if c.stream == stream and c.free():
                    s = c
                    break
            if s is None:
                s = _OutputSet(self)
                s.stream = stream
                if k < self._pool_max:
                    pool.append(s)
                    i = k
                else:                               # every set is still held by the caller: forget the oldest one
                    i = self._pool_i + 1 if self._pool_i + 1 < k else 0
                    pool[i] = s
            self._pool_i = i
            t, base = s.t, s.base
        else:
            t, base = _carve(self.num_envs, self.device)
        r, o = self._rec_out, self._out_offs
        r.reward, r.terminated, r.q_p1, r.q_p1_len = base, base + o[1], base + o[2], base + o[3]
        r.q_p2, r.q_p2_len, r.classical, r.turn = base + o[4], base + o[5], base + o[6], base + o[7]
        return t

    def reset(self, *, seed=None, options=None, copy_obs=True):
        """env.py:55-57: fresh boards and their observation, ONE kernel (qttt_reset_observe); `seed`/`options`
        accepted and ignored like the reference, except that an int `seed` re-keys the collapse-bit hash (the
        reference has no per-env RNG).  The observation is made of fresh tensors (written by the kernel itself, never
        copied) unless copy_obs=False (then: the environment's own buffers, which the next copy_obs=False step() /
        observ() overwrites)."""
        if seed is not None:
            self.seed = int(seed)
        self.step_idx = 0
        if copy_obs:
            t = self._fresh_outputs()
            r = self._rec_out
            obs = {"q_states_p1": t[2], "q_states_p1_len": t[3], "q_states_p2": t[4], "q_states_p2_len": t[5],
                   "classical": t[6], "turn": t[7]}
        else:
            obs = self._obs_buffers()
            r = self._rec
        rc = self._launch(self._lib.qttt_reset_observe, self.state.data_ptr(), r.classical, r.q_p1, r.q_p1_len, r.q_p2,
                          r.q_p2_len, r.turn, self.num_envs, self._stream())
        if rc:
            _native.check(rc, "qttt_reset_observe")
        return obs, {}

    def step_raw(self, actions, bits=None):
        """The hot path alone: one fused kernel launch, no observation unpack.
        actions u8[N,2] on the device; bits u8[N] (explicit collapse bits, parity mode) or None
        (bit = counter hash of (seed, board_offset+i, step_idx)).
        Returns (reward f32[N], terminated bool[N]) — buffers reused across calls."""
        n = self.num_envs
        if actions.dtype != torch.uint8 or not actions.is_contiguous() or actions.device != self.state.device \
                or actions.numel() != 2 * n:
            raise ValueError("step_raw wants a contiguous uint8 device tensor of shape (N, 2)")
        if bits is not None and (bits.dtype != torch.uint8 or not bits.is_contiguous()
                                 or bits.device != self.state.device or bits.numel() != n):
            raise ValueError("bits must be a contiguous uint8 device tensor of shape (N,)")
        rc = self._launch(self._env_step, self._record(), actions.data_ptr(), _ptr(bits), self._step_host,
                          _native.ENV_STEP, self._stream())
        if rc:
            _native.check(rc, "qttt_step")
        self._advance(1)
        return self._reward, self._terminated

    def step_random(self, actions_out=None):
        """One step under the synthetic uniform-legal policy, policy and step fused in one kernel
        (== sample_actions() followed by step_raw()).  Returns (reward, terminated); the actions
        played are written to `actions_out` (u8[N,2]) if given."""
        n = self.num_envs
        if actions_out is not None and (actions_out.dtype != torch.uint8 or not actions_out.is_contiguous()
                                        or actions_out.numel() != 2 * n or actions_out.device != self.state.device):
            raise ValueError("actions_out must be a contiguous uint8 device tensor of shape (N, 2)")
        rc = self._launch(self._env_step, self._record(), _ptr(actions_out), None, self._step_host,
                          _native.ENV_STEP_RANDOM, self._stream())
        if rc:
            _native.check(rc, "qttt_step_random")
        self._advance(1)
        return self._reward, self._terminated

    def step_many(self, actions, bits=None, reward=None, terminated=None, fused=False):
        """T consecutive steps from pre-recorded device tensors actions u8[T,N,2] (bits u8[T,N]),
        enqueued from C with no per-step host work.  With `reward`/`terminated` of shape [T,N]
        every step's outputs are kept; otherwise only the last step's (returned).
        fused=True runs the T steps with the boards in registers, one launch per 64 steps (same results)."""
        n = self.num_envs
        T = int(actions.shape[0])
        if actions.dtype != torch.uint8 or not actions.is_contiguous() or tuple(actions.shape) != (T, n, 2) \
                or actions.device != self.state.device:
            raise ValueError("step_many wants a contiguous uint8 device tensor of shape (T, N, 2)")
        if bits is not None and (bits.dtype != torch.uint8 or not bits.is_contiguous()
                                 or tuple(bits.shape) != (T, n) or bits.device != self.state.device):
            raise ValueError("bits must be a contiguous uint8 device tensor of shape (T, N)")
        stride = 0
        r, tm = self._reward, self._terminated
        if reward is not None or terminated is not None:
            if reward is None or terminated is None or tuple(reward.shape) != (T, n) \
                    or tuple(terminated.shape) != (T, n) or reward.dtype != torch.float32 \
                    or terminated.dtype != torch.bool or not reward.is_contiguous() \
                    or not terminated.is_contiguous():
                raise ValueError("reward f32[T,N] and terminated bool[T,N] must be given together")
            r, tm, stride = reward, terminated, n
        with torch.cuda.device(self.device):
            rc = self._lib.qttt_step_many(self.state.data_ptr(), actions.data_ptr(), _ptr(bits), self.seed,
                                          self.step_idx, self.board_offset,
                                          self._flags() | (_native.FLAG_FUSED if fused else 0), r.data_ptr(),
                                          tm.data_ptr(), stride, n, T, self._stream())
        _native.check(rc, "qttt_step_many")
        self._advance(T)
        return r, tm

    def step(self, actions, bits=None, verbose=False, copy_obs=True):
        """env.py:34-53 for N boards: (obs, reward, terminated, truncated, info) from ONE kernel launch and nothing
        else (the step kernel writes the observation from the registers it holds, straight into the tensors returned).
        copy_obs=True (default): the returned observation, reward and terminated are tensors nobody else holds — as
        the reference's Env.step builds new lists every call (env.py:46,68-85), a gym caller may keep (obs, next_obs)
        pairs, or every observation of an episode: one fresh allocation per call, carved into the eight tensors
        (_fresh_outputs; VecEnv(output_pool=N) re-uses sets the caller has dropped instead).  The eight tensors of one
        call share that one allocation (34 bytes per board + padding): keeping — or torch.save-ing — any one of them
        keeps all of it, so a replay buffer that stores one field for long should store `.clone()`s.  copy_obs=False returns the environment's own buffers, overwritten by the next such step() /
        observ() — = step_observe_raw."""
        dev = self.state.device
        if not (torch.is_tensor(actions) and actions.dtype == torch.uint8 and actions.device == dev
                and actions.shape == (self.num_envs, 2) and actions.is_contiguous()):
            actions = self._as_actions(actions)
        if bits is not None and not (torch.is_tensor(bits) and bits.dtype == torch.uint8 and bits.device == dev
                                     and bits.is_contiguous()):
            bits = torch.as_tensor(bits).to(torch.uint8).to(self.device).contiguous()
        if not copy_obs:
            obs, reward, terminated = self.step_observe_raw(actions, bits)
            return obs, reward, terminated, self._truncated, {}
        if bits is not None and bits.numel() != self.num_envs:
            raise ValueError("bits must be a contiguous uint8 device tensor of shape (N,)")
        t = self._fresh_outputs()
        r = self._rec_out
        r.state = self.state.data_ptr()
        r.board_offset, r.seed, r.flags, r.step_counter = self.board_offset, self.seed, self._flags(), self._rec.step_counter
        rc = self._launch(self._env_step, self._rec_out_ref, actions.data_ptr(), _ptr(bits), self._step_host,
                          _native.ENV_STEP_OBSERVE, self._stream())
        if rc:
            _native.check(rc, "qttt_step_observe")
        self._advance(1)
        return ({"q_states_p1": t[2], "q_states_p1_len": t[3], "q_states_p2": t[4], "q_states_p2_len": t[5],
                 "classical": t[6], "turn": t[7]}, t[0], t[1], self._truncated, {})

    def _obs_buffers(self):
        """The observation tensors (env.py:19-25,68-85), allocated once per environment."""
        if self._obs is None:
            n, dev = self.num_envs, self.device
            with torch.cuda.device(dev):
                self._obs = {
                    "q_states_p1": torch.empty((n, 5, 2), dtype=torch.uint8, device=dev),
                    "q_states_p1_len": torch.empty(n, dtype=torch.uint8, device=dev),
                    "q_states_p2": torch.empty((n, 4, 2), dtype=torch.uint8, device=dev),
                    "q_states_p2_len": torch.empty(n, dtype=torch.uint8, device=dev),
                    "classical": torch.empty((n, 9), dtype=torch.int8, device=dev),
                    "turn": torch.empty(n, dtype=torch.uint8, device=dev),
                }
            o = self._obs
            self._p_obs = tuple(o[k].data_ptr() for k in ("classical", "q_states_p1", "q_states_p1_len",
                                                          "q_states_p2", "q_states_p2_len", "turn"))
            r = self._rec
            r.classical, r.q_p1, r.q_p1_len, r.q_p2, r.q_p2_len, r.turn = self._p_obs
        return self._obs

    def step_observe_raw(self, actions, bits=None):
        """Env.step for N boards including the observation (env.py:46), one fused kernel:
        qttt_step_observe.  Same argument rules as step_raw.  Returns (obs dict, reward, terminated),
        all buffers owned by the environment and reused across calls."""
        n = self.num_envs
        if actions.dtype != torch.uint8 or not actions.is_contiguous() or actions.device != self.state.device \
                or actions.numel() != 2 * n:
            raise ValueError("step_observe_raw wants a contiguous uint8 device tensor of shape (N, 2)")
        if bits is not None and (bits.dtype != torch.uint8 or not bits.is_contiguous()
                                 or bits.device != self.state.device or bits.numel() != n):
            raise ValueError("bits must be a contiguous uint8 device tensor of shape (N,)")
        o = self._obs_buffers()
        rc = self._launch(self._env_step, self._record(), actions.data_ptr(), _ptr(bits), self._step_host,
                          _native.ENV_STEP_OBSERVE, self._stream())
        if rc:
            _native.check(rc, "qttt_step_observe")
        self._advance(1)
        return o, self._reward, self._terminated

    def observ(self):
        """env.py:62-63,68-85 as tensors: q_states_p{1,2} u8[N,5|4,2] (255 pad) with *_len,
        classical i8[N,9], turn u8[N].  The tensors are the environment's own buffers (allocated
        once), overwritten by the next observ()/step()."""
        n = self.num_envs
        obs = self._obs_buffers()
        with torch.cuda.device(self.device):
            rc = self._lib.qttt_observe(self.state.data_ptr(), obs["classical"].data_ptr(),
                                        obs["q_states_p1"].data_ptr(), obs["q_states_p1_len"].data_ptr(),
                                        obs["q_states_p2"].data_ptr(), obs["q_states_p2_len"].data_ptr(),
                                        obs["turn"].data_ptr(), n, self._stream())
        _native.check(rc, "qttt_observe")
        return obs

    def turn(self, out=None):
        """env.py:65-66: len(moves) per board (counts the autofill move), u8[N]: qttt_export asked for
        n_moves alone (8 bytes read and 1 written per board).  `out` = a tensor to overwrite."""
        n = self.num_envs
        if out is None:
            with torch.cuda.device(self.device):
                out = torch.empty(n, dtype=torch.uint8, device=self.device)
        else:
            _check_out(out, torch.uint8, (n,), self.state.device, "out")
        rc = self._launch(self._lib.qttt_export, self.state.data_ptr(), None, out.data_ptr(), None, None, None, n,
                          self._stream())
        _native.check(rc, "qttt_export")
        return out

    def render(self, index=0):
        from .board import Board, displayBoard
        displayBoard(Board.from_export(self.export_boards(), index))

    # ------------------------------------------------------------------ Board-level access
    def check_win(self, out=None):
        """board.py:71-115 per board: (p1_round i8[N], p2_round i8[N]).  `out` = a pair returned by an
        earlier call, to be overwritten instead of allocating (half the cost of the call at 1 M boards)."""
        n = self.num_envs
        if out is None:
            with torch.cuda.device(self.device):
                out = (torch.empty(n, dtype=torch.int8, device=self.device),
                       torch.empty(n, dtype=torch.int8, device=self.device))
        p1, p2 = out
        for t in (p1, p2):
            if t.dtype != torch.int8 or t.numel() != n or not t.is_contiguous() or t.device != self.state.device:
                raise ValueError("out must be two contiguous int8 device tensors of N elements")
        rc = self._launch(self._lib.qttt_check_win, self.state.data_ptr(), p1.data_ptr(), p2.data_ptr(), n,
                          self._stream())
        _native.check(rc, "qttt_check_win")
        return p1, p2

    _EXPORT_SPEC = (("moves", torch.uint8, (9, 2)), ("n_moves", torch.uint8, ()), ("board", torch.int8, (9,)),
                    ("qmask", torch.int16, (4,)), ("n_q", torch.uint8, ()))

    def export_boards(self, out=None):
        """Board.moves / .board / .qstructs (board.py:4-6) as tensors.  `out` = the dict of an earlier
        call, to be overwritten instead of allocating."""
        n, dev = self.num_envs, self.device
        if out is None:
            with torch.cuda.device(dev):
                out = {k: torch.empty((n,) + shp, dtype=dt, device=dev) for k, dt, shp in self._EXPORT_SPEC}
        else:
            for k, dt, shp in self._EXPORT_SPEC:
                _check_out(out[k], dt, (n,) + shp, self.state.device, "out[%r]" % k)
        rc = self._launch(self._lib.qttt_export, self.state.data_ptr(), out["moves"].data_ptr(),
                          out["n_moves"].data_ptr(), out["board"].data_ptr(), out["qmask"].data_ptr(),
                          out["n_q"].data_ptr(), n, self._stream())
        _native.check(rc, "qttt_export")
        return out

    def import_boards(self, moves, n_moves, board, qmask, n_q):
        n = self.num_envs
        dev = self.device

        def prep(t, dtype, shape):
            t = torch.as_tensor(t).to(dtype).to(dev).contiguous()
            if tuple(t.shape) != shape:
                raise ValueError("expected shape %s, got %s" % (shape, tuple(t.shape)))
            return t
        moves = prep(moves, torch.uint8, (n, 9, 2))
        n_moves = prep(n_moves, torch.uint8, (n,))
        board = prep(board, torch.int8, (n, 9))
        qmask = prep(qmask, torch.int16, (n, 4))
        n_q = prep(n_q, torch.uint8, (n,))
        with torch.cuda.device(dev):
            rc = self._lib.qttt_import(self.state.data_ptr(), moves.data_ptr(), n_moves.data_ptr(),
                                       board.data_ptr(), qmask.data_ptr(), n_q.data_ptr(), n, self._stream())
        _native.check(rc, "qttt_import")

    def sample_actions(self, out=None):
        """Uniform-legal synthetic policy for the *next* step.  `out` u8[N,2] to overwrite."""
        n = self.num_envs
        if out is None:
            with torch.cuda.device(self.device):
                out = torch.empty((n, 2), dtype=torch.uint8, device=self.device)
        elif out.dtype != torch.uint8 or out.numel() != 2 * n or not out.is_contiguous() or out.device != self.state.device:
            raise ValueError("out must be a contiguous uint8 device tensor of shape (N, 2)")
        # through the qttt_env record: with a device-side step counter the call stays capturable in a hipGraph
        rc = self._launch(self._env_step, self._record(), out.data_ptr(), None, self._step_host, _native.ENV_SAMPLE,
                          self._stream())
        _native.check(rc, "qttt_sample_actions")
        return out

    def step_random_many(self, n_steps, actions_out=None, reward=None, terminated=None, returns=None):
        """n_steps steps under the in-kernel uniform-legal policy with the boards in registers, ONE launch per
        64 plies (qttt_step_random_many) == n_steps calls of step_random().  With reward f32[T,N] +
        terminated bool[T,N] (and optionally actions_out u8[T,N,2]) every step's outputs are kept; without
        them only the last step's are written (to the environment's own reward / terminated buffers, which
        are returned; actions_out u8[N,2] optional).  returns f32[N] (optional) is ACCUMULATED: += the sum of each
        board's rewards over these plies (env.py:49) — the per-board episode returns, with no per-ply output kept."""
        n, T, dev = self.num_envs, int(n_steps), self.state.device
        keep = reward is not None or terminated is not None
        stride = 0
        r, tm = self._reward, self._terminated
        if keep:
            if reward is None or terminated is None:
                raise ValueError("reward f32[T,N] and terminated bool[T,N] must be given together")
            r, tm = _check_out(reward, torch.float32, (T, n), dev, "reward"), _check_out(terminated, torch.bool, (T, n), dev, "terminated")
            stride = n
        if actions_out is not None:        # one out_stride for all outputs: [T,N,2] with reward/terminated [T,N], else [N,2]
            _check_out(actions_out, torch.uint8, (T, n, 2) if keep else (n, 2), dev, "actions_out")
        if returns is not None:
            _check_out(returns, torch.float32, (n,), dev, "returns")
        rc = self._launch(self._lib.qttt_step_random_many, self.state.data_ptr(), self.seed, self.step_idx,
                          self.board_offset, self._flags(), _ptr(actions_out), r.data_ptr(), tm.data_ptr(), stride,
                          _ptr(returns), n, T, self._stream())
        _native.check(rc, "qttt_step_random_many")
        self._advance(T)
        return r, tm

    # ------------------------------------------------------------------ MCTS-side rows
    @classmethod
    def from_state(cls, state, num_envs, seed=0, auto_reset=False, board_offset=0):
        """Wraps an existing packed state tensor (e.g. a child buffer written by expand()).

        The tensor must hold states this library wrote (reset, step, expand, import_boards ...).  Two things are relied
        on and NOT re-checked: the cached qstructs and rooted forest are consistent with the moves, and the done bit
        (bit 63 of plane P) is set iff the board has a completed line or at least eight classical squares — the step
        only ever SETS that bit (a line stays a line), and with auto_reset the in-kernel policy trusts "not done" to
        mean "at least two empty squares".  A hand-made tensor with a wrong done bit is stepped without faults (the
        kernels keep everything in registers, every loop is bounded) but gives states the reference never reaches;
        to bring boards from attributes use import_boards(), which computes the bit."""
        env = cls.__new__(cls)
        env.num_envs = int(num_envs)
        env.device = state.device
        env._lib = _native.lib()
        if state.dtype != torch.uint8 or state.numel() != env._lib.qttt_state_bytes(env.num_envs):
            raise ValueError("state must be a uint8 tensor of qttt_state_bytes(num_envs) bytes")
        env.seed, env.auto_reset, env.board_offset = int(seed), bool(auto_reset), int(board_offset)
        env._step_host, env._ctr = 0, None
        env._shape_flags = 0
        env.action_space = reference_action_space()
        env.observation_space = reference_observation_space()
        env.state = state
        n = env.num_envs
        with torch.cuda.device(env.device):
            env._reward = torch.empty(n, dtype=torch.float32, device=env.device)
            env._terminated = torch.empty(n, dtype=torch.bool, device=env.device)
            env._truncated = torch.zeros(n, dtype=torch.bool, device=env.device)
        env._obs = None
        env._pool, env._pool_i, env._pool_max = [], 0, 0
        env._bind_outputs()
        return env

    def take(self, index, seed=None, auto_reset=None, board_offset=None):
        """A new VecEnv holding boards self[index] (any int64 index tensor; repeats allowed): the batched form of
        `copy.deepcopy(node)` + attribute assignment in MCTS._step (mcts.py:236-241) — e.g.
        `env.take(torch.arange(N).repeat_interleave(36))` lines every leaf up 36 times for one expand() over all
        its actions.  Pure indexing of the two packed planes; nothing is unpacked."""
        idx = torch.as_tensor(index, device=self.state.device).to(torch.int64).reshape(-1)
        m = int(idx.numel())
        lib = self._lib
        planes = self.state.view(torch.int64).view(2, -1)
        with torch.cuda.device(self.device):
            st = torch.zeros(int(lib.qttt_state_bytes(m)), dtype=torch.uint8, device=self.device)
        if m:
            st.view(torch.int64).view(2, -1)[:, :m] = planes[:, idx]
        return VecEnv.from_state(st, m, seed=self.seed if seed is None else seed,
                                 auto_reset=self.auto_reset if auto_reset is None else auto_reset,
                                 board_offset=self.board_offset if board_offset is None else board_offset)

    def node_info(self, out=None, python_key=True):
        """GameState bookkeeping per board (mcts.py:20-27,52-65,93-94): winner i8 (1/0/-1 = True/
        False/None), terminal bool, legal int64 (bit a = action a legal), state_key int64 (the native 64-bit
        position key: equal <=> equal (board, moves); qttt_state_key of the packed words) and — with
        python_key — key int64 = Python's hash(tuple(board)+tuple(moves)), for host-side dicts built by
        reference code (three quarters of the kernel's work: a device-side search passes python_key=False).
        `out` = the dict of an earlier call, to be overwritten: only the entries it holds are computed."""
        n, dev = self.num_envs, self.device
        spec = (("winner", torch.int8), ("terminal", torch.bool), ("legal", torch.int64), ("state_key", torch.int64),
                ("key", torch.int64))
        if out is None:
            with torch.cuda.device(dev):
                out = {k: torch.empty(n, dtype=dt, device=dev) for k, dt in spec if k != "key" or python_key}
        else:                                       # every entry is optional: only what the dict holds is computed
            for k, dt in spec:
                t = out.get(k)
                if t is not None and (t.dtype != dt or t.numel() != n or not t.is_contiguous() or t.device != self.state.device):
                    raise ValueError("out[%r] must be a contiguous %s device tensor of N elements" % (k, dt))
        rc = self._launch(self._lib.qttt_node_info, self.state.data_ptr(), _ptr(out.get("winner")),
                          _ptr(out.get("terminal")), _ptr(out.get("legal")), _ptr(out.get("key")),
                          _ptr(out.get("state_key")), n, self._stream())
        _native.check(rc, "qttt_node_info")
        return out

    def state_keys(self, out=None):
        """The native position keys alone (int64[N]): 16 bytes read and 8 written per board."""
        n = self.num_envs
        if out is None:
            with torch.cuda.device(self.device):
                out = torch.empty(n, dtype=torch.int64, device=self.device)
        else:
            _check_out(out, torch.int64, (n,), self.state.device, "out")
        rc = self._launch(self._lib.qttt_node_info, self.state.data_ptr(), None, None, None, None, out.data_ptr(), n,
                          self._stream())
        _native.check(rc, "qttt_node_info")
        return out

    _EXPAND_ROWS = (("n_children", torch.uint8, ()), ("winner", torch.int8, (2,)), ("terminal", torch.bool, (2,)),
                    ("legal", torch.int64, (2,)), ("state_key", torch.int64, (2,)))

    def _expand_out(self, out, python_key, extra=()):
        """The output dict of expand / expand_rollout: allocated, or the one of an earlier call checked."""
        n, dev = self.num_envs, self.device
        if out is None:
            with torch.cuda.device(dev):
                mk = lambda: VecEnv.from_state(torch.empty_like(self.state), n, seed=self.seed, board_offset=self.board_offset)
                out = {"child0": mk(), "child1": mk()}
                for k, dt, shp in self._EXPAND_ROWS + tuple(extra):
                    out[k] = torch.empty((n,) + shp, dtype=dt, device=dev)
                if python_key:
                    out["key"] = torch.empty((n, 2), dtype=torch.int64, device=dev)
        else:                                       # the per-child rows are optional: only what the dict holds is computed
            sd = self.state.device
            for c in ("child0", "child1"):
                if c not in out:
                    raise ValueError("out[%r] is required" % c)
                if out[c].num_envs != n or out[c].state.device != sd:
                    raise ValueError("out[%r] must be a VecEnv of N boards on this device" % c)
            required = tuple(k for k, _, _ in extra) + (("key",) if python_key else ())
            for k, dt, shp in self._EXPAND_ROWS + tuple(extra) + (("key", torch.int64, (2,)),):
                if k in required and k not in out:
                    # e.g. the dict of expand() handed to expand_rollout(), or python_key=True with a dict made without it
                    raise ValueError("out[%r] is required" % k)
                if k in out:
                    _check_out(out[k], dt, (n,) + shp, sd, "out[%r]" % k)
        return out

    def _as_action36(self, action36):
        a = action36
        if not (torch.is_tensor(a) and a.dtype == torch.uint8 and a.device == self.state.device and a.is_contiguous()):
            a = torch.as_tensor(a).to(torch.uint8).to(self.device).contiguous()
        if a.shape != (self.num_envs,):
            raise ValueError("action36 must have shape (%d,)" % self.num_envs)
        return a

    def expand(self, action36, out=None, python_key=None):
        """MCTS._step (mcts.py:233-267) for every board: action36 u8[N] (ind2move index).
        Returns dict(child0, child1 = VecEnv over the child states, n_children u8[N],
        winner i8[N,2], terminal bool[N,2], legal int64[N,2], state_key int64[N,2] and — with python_key —
        key int64[N,2] = Python's hash of each child, see node_info).
        `out` = the dict of an earlier call: its child states and tensors are overwritten (a search loop
        then allocates nothing per expansion); its "key" entry decides python_key (python_key=True with a dict
        that has no "key" entry raises ValueError).  python_key=None: True for a fresh dict, the dict's choice otherwise."""
        n = self.num_envs
        a = self._as_action36(action36)
        out = self._expand_out(out, (out is None) if python_key is None else bool(python_key))
        rc = self._launch(self._lib.qttt_expand, self.state.data_ptr(), a.data_ptr(), out["child0"].state.data_ptr(),
                          out["child1"].state.data_ptr(), _ptr(out.get("n_children")), _ptr(out.get("winner")),
                          _ptr(out.get("terminal")), _ptr(out.get("legal")), _ptr(out.get("key")),
                          _ptr(out.get("state_key")), n, self._stream())
        _native.check(rc, "qttt_expand")
        return out

    def expand_rollout(self, action36, n_sims=1, step_idx0=None, out=None, python_key=None, with_result=False):
        """One MCTS._rollout below the selected node in ONE launch (mcts.py:166-176,210-221,233-267): expand() plus
        n_sims random playouts from EACH child.  Returns expand()'s dict with two more entries:
        value_sum int32[N,2] = the sum over a child's playouts of `r if leaf.turn else -r` (mcts.py:174; divide by
        n_sims for the value _backpropogate receives) and — with_result — result int8[N,2,n_sims], every playout's
        MCTS._reward.  Bit-identical to expand() followed by child0.rollout_many(n_sims, step_idx0) and
        child1.rollout_many(n_sims, step_idx0 + 16 * n_sims).  `out` = the dict of an earlier call with the same
        n_sims, overwritten."""
        n, S = self.num_envs, int(n_sims)
        if not 1 <= S <= _native.EXPAND_ROLLOUT_MAX_SIMS:
            raise ValueError("n_sims must be in 1..%d" % _native.EXPAND_ROLLOUT_MAX_SIMS)
        if step_idx0 is None:
            step_idx0 = self.step_idx
        a = self._as_action36(action36)
        with_result = with_result or (out is not None and "result" in out)
        extra = (("value_sum", torch.int32, (2,)),) + ((("result", torch.int8, (2, S)),) if with_result else ())
        out = self._expand_out(out, bool(python_key), extra)       # None = False for a fresh dict, the dict's choice otherwise
        rc = self._launch(self._lib.qttt_expand_rollout, self.state.data_ptr(), a.data_ptr(),
                          out["child0"].state.data_ptr(), out["child1"].state.data_ptr(), _ptr(out.get("n_children")),
                          _ptr(out.get("winner")), _ptr(out.get("terminal")), _ptr(out.get("legal")),
                          _ptr(out.get("key")), _ptr(out.get("state_key")), self.seed, int(step_idx0), self.board_offset,
                          S, out["value_sum"].data_ptr(), _ptr(out.get("result")), n, self._stream())
        _native.check(rc, "qttt_expand_rollout")
        return out

    def rollout(self, step_idx0=None, return_final=False, out=None):
        """MCTS._simulate (mcts.py:185-198) under uniform priors: one fused random playout per
        board, boards unchanged.  Returns (result i8[N] in {+1,-1,0}, plies u8[N][, final VecEnv]).
        `out` = the tuple of an earlier call with the same return_final, to be overwritten."""
        n, dev = self.num_envs, self.device
        if step_idx0 is None:
            step_idx0 = self.step_idx
        if out is None:
            with torch.cuda.device(dev):
                result = torch.empty(n, dtype=torch.int8, device=dev)
                plies = torch.empty(n, dtype=torch.uint8, device=dev)
                final = (VecEnv.from_state(torch.empty_like(self.state), n, seed=self.seed, board_offset=self.board_offset)
                         if return_final else None)
        else:
            result, plies = out[0], out[1]
            final = out[2] if return_final else None
            _check_out(result, torch.int8, (n,), self.state.device, "out[0]")
            _check_out(plies, torch.uint8, (n,), self.state.device, "out[1]")
            if return_final and (final.num_envs != n or final.state.device != self.state.device):
                raise ValueError("out[2] must be a VecEnv of N boards on this device")
        rc = self._launch(self._lib.qttt_rollout, self.state.data_ptr(), self.seed, int(step_idx0), self.board_offset,
                          result.data_ptr(), plies.data_ptr(), None if final is None else final.state.data_ptr(), n,
                          self._stream())
        _native.check(rc, "qttt_rollout")
        return (result, plies, final) if return_final else (result, plies)

    def rollout_many(self, n_sims, step_idx0=None, with_plies=False, out=None):
        """MCTS._rollout's simulation loop (mcts.py:170-176: num_simulations playouts from each leaf) in ONE launch,
        one lane per (board, simulation): result i8[N, n_sims] (+1 / -1 / 0 per MCTS._reward), optionally plies
        u8[N, n_sims].  Column s equals rollout(step_idx0 + s * 16).  `out` = what an earlier call returned."""
        n, S, dev = self.num_envs, int(n_sims), self.device
        if S < 1:
            raise ValueError("n_sims must be >= 1")
        if step_idx0 is None:
            step_idx0 = self.step_idx
        if out is None:
            with torch.cuda.device(dev):
                result = torch.empty((n, S), dtype=torch.int8, device=dev)
                plies = torch.empty((n, S), dtype=torch.uint8, device=dev) if with_plies else None
        else:
            result, plies = (out if with_plies else (out, None))
            _check_out(result, torch.int8, (n, S), self.state.device, "out result")
            if with_plies:
                _check_out(plies, torch.uint8, (n, S), self.state.device, "out plies")
        rc = self._launch(self._lib.qttt_rollout_many, self.state.data_ptr(), self.seed, int(step_idx0), self.board_offset,
                          S, result.data_ptr(), _ptr(plies), n, self._stream())
        _native.check(rc, "qttt_rollout_many")
        return (result, plies) if with_plies else result

    def encode(self, with_mask=True, out=None):
        """GameState.to_vector (mcts.py:67-85) as f32[N,18,10] and action_mask (mcts.py:87-91) as
        bool[N,36], without leaving the GPU.  `out` = what an earlier call returned, to be overwritten."""
        n, dev = self.num_envs, self.device
        if out is None:
            with torch.cuda.device(dev):
                vec = torch.empty((n, 18, 10), dtype=torch.float32, device=dev)
                mask = torch.empty((n, 36), dtype=torch.bool, device=dev) if with_mask else None
        else:
            vec, mask = (out if with_mask else (out, None))
            _check_out(vec, torch.float32, (n, 18, 10), self.state.device, "out vec")
            if with_mask:
                _check_out(mask, torch.bool, (n, 36), self.state.device, "out mask")
        rc = self._launch(self._lib.qttt_encode, self.state.data_ptr(), vec.data_ptr(), _ptr(mask), n, self._stream())
        _native.check(rc, "qttt_encode")
        return (vec, mask) if with_mask else vec

    # ------------------------------------------------------------------ hipGraph of T step launches
    def capture(self, n_steps, mode="random", actions=None, bits=None, actions_out=None, reward=None, terminated=None):
        """Captures n_steps step LAUNCHES into one hipGraph and returns it (`.replay()`): for loops on small
        batches, where one launch per step is host-bound (4-5 us per call against a ~3 us kernel at <= 262 144
        boards) and the policy must see the state every step, so the fused multi-step kernels do not apply.
        mode "random": step_random() x n_steps (actions_out u8[T,N,2] optional);
        mode "step" / "observe": step_raw / step_observe_raw reading actions u8[T,N,2] (bits u8[T,N] optional) —
        the caller's buffers, refilled between replays (e.g. by a policy network captured in its own graph).
        reward f32[T,N] + terminated bool[T,N] keep every step's outputs; without them the environment's own
        buffers hold the last step's.  The step index lives on the device from here on
        (use_device_step_counter), so every replay draws fresh collapse bits; the graph ends by advancing it."""
        T, n, dev = int(n_steps), self.num_envs, self.state.device
        if mode not in ("random", "step", "observe") or T < 1:
            raise ValueError("mode must be 'random', 'step' or 'observe' and n_steps >= 1")
        if mode == "random":
            if actions is not None or bits is not None:
                raise ValueError("mode 'random' draws its own actions (actions_out receives them)")
            if actions_out is not None:
                _check_out(actions_out, torch.uint8, (T, n, 2), dev, "actions_out")
        else:
            if actions is None or actions_out is not None:
                raise ValueError("mode %r reads actions u8[T,N,2]" % mode)
            _check_out(actions, torch.uint8, (T, n, 2), dev, "actions")
            if bits is not None:
                _check_out(bits, torch.uint8, (T, n), dev, "bits")
        if (reward is None) != (terminated is None):
            raise ValueError("reward f32[T,N] and terminated bool[T,N] must be given together")
        if reward is not None:
            _check_out(reward, torch.float32, (T, n), dev, "reward")
            _check_out(terminated, torch.bool, (T, n), dev, "terminated")
        if mode == "observe":
            self._obs_buffers()
        ctr = self.use_device_step_counter()
        self._record()
        code = {"random": _native.ENV_STEP_RANDOM, "step": _native.ENV_STEP, "observe": _native.ENV_STEP_OBSERVE}[mode]
        recs = []
        for t in range(T):                     # one record per node: its own slice of the per-step outputs
            r = _native.EnvRecord.from_buffer_copy(self._rec)
            if reward is not None:
                r.reward, r.terminated = reward[t].data_ptr(), terminated[t].data_ptr()
            recs.append(r)
        a_src = actions_out if mode == "random" else actions
        # the kernels' code objects must be resident before the capture: one eager launch of the same entry on a
        # scratch copy of the state, then everything it touched is put back
        keep = (self.state.clone(), int(ctr), self._reward.clone(), self._terminated.clone())
        a0 = None if a_src is None else a_src[0].clone()
        r0 = None if reward is None else (reward[0].clone(), terminated[0].clone())
        obs0 = None if mode != "observe" else {k: v.clone() for k, v in self._obs.items()}
        self._check(self._launch(self._env_step, ctypes.byref(recs[0]), _ptr(None if a_src is None else a_src[0]),
                                 _ptr(None if bits is None else bits[0]), 0, code, self._stream()))
        self._check(self._launch(self._lib.qttt_counter_add, ctr.data_ptr(), 0, self._stream()))
        self.state.copy_(keep[0]); ctr.fill_(keep[1]); self._reward.copy_(keep[2]); self._terminated.copy_(keep[3])
        if a0 is not None:
            a_src[0].copy_(a0)
        if r0 is not None:
            reward[0].copy_(r0[0]); terminated[0].copy_(r0[1])
        if obs0 is not None:
            for k, v in obs0.items():
                self._obs[k].copy_(v)
        graph = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                st = self._stream()
                for t in range(T):
                    self._check(self._launch(self._env_step, ctypes.byref(recs[t]), _ptr(None if a_src is None else a_src[t]),
                                             _ptr(None if bits is None else bits[t]), t, code, st))
                self._check(self._launch(self._lib.qttt_counter_add, ctr.data_ptr(), T, st))
        torch.cuda.current_stream(self.device).wait_stream(side)
        return StepGraph(self, graph, T, mode, recs, (actions, bits, actions_out, reward, terminated))

    @staticmethod
    def _check(rc):
        if rc:
            _native.check(rc, "qttt_env_step (graph capture)")

    # ------------------------------------------------------------------ checkpointing
    def state_dict(self):
        return {"state": self.state.clone(), "seed": self.seed, "step_idx": self.step_idx,
                "board_offset": self.board_offset, "auto_reset": self.auto_reset}

    def load_state_dict(self, sd):
        if sd["state"].numel() != self.state.numel():
            raise ValueError("state size mismatch")
        self.state.copy_(sd["state"])
        self.seed, self.step_idx = int(sd["seed"]), int(sd["step_idx"])
        self.board_offset, self.auto_reset = int(sd["board_offset"]), bool(sd["auto_reset"])


class StepGraph:
    """A hipGraph of T step launches of one VecEnv (VecEnv.capture).  replay() enqueues all of them with ONE
    host call on the current stream; reward / terminated / the observation are where capture() was told to
    put them (or the environment's own buffers, holding the last step's)."""

    def __init__(self, env, graph, n_steps, mode, records, buffers):
        self.env, self.graph, self.n_steps, self.mode = env, graph, n_steps, mode
        self._keep = (records, buffers)            # the nodes hold raw pointers into these

    def replay(self):
        self.graph.replay()
        return self.env._reward, self.env._terminated
