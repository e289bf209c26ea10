"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

ctypes/numpy front end to ``libqttt_oracle.so`` (oracle/qttt_oracle.c), the CPU restatement
of the reference's ``Env.step`` path.  Importers allowed: ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py``.  The product
package ``qtttgym_amd`` must never import this module
(tests/test_abi_and_host.py::test_product_never_imports_the_oracle).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libqttt_oracle.so")

# mirrors `qo_board` in qttt_oracle.h (sizeof == 48)
BOARD_DTYPE = np.dtype([
    ("n_moves", np.int32),
    ("moves", np.int8, (9, 2)),
    ("board", np.int8, (9,)),
    ("_pad0", np.uint8, (1,)),
    ("n_q", np.int32),
    ("q", np.uint16, (5,)),
    ("_pad1", np.uint8, (2,)),
], align=False)
assert BOARD_DTYPE.itemsize == 48, BOARD_DTYPE.itemsize


def build(force=False):
    src = os.path.join(_HERE, "qttt_oracle.c")
    hdr = os.path.join(_HERE, "qttt_oracle.h")
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(src), os.path.getmtime(hdr))):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libqttt_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        vp, i32, i64, u64, u32 = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64,
                                  ctypes.c_uint64, ctypes.c_uint32)
        L.qo_init.argtypes = [vp]
        L.qo_make_move.argtypes = [vp, i32, i32, i32, vp]
        L.qo_make_move.restype = i32
        L.qo_check_win.argtypes = [vp, vp, vp]
        L.qo_step.argtypes = [vp, i32, i32, i32, vp, vp, vp]
        L.qo_step.restype = i32
        L.qo_observe.argtypes = [vp] * 7
        L.qo_reset_batch.argtypes = [vp, i64]
        L.qo_step_batch.argtypes = [vp, i64, vp, vp, u64, u32, i64, i32, vp, vp]
        L.qo_replay_batch.argtypes = [vp, i64, vp, i64, i32, u64, u32, i64, i32, vp, vp]
        L.qo_terminated.argtypes = [vp]
        L.qo_terminated.restype = i32
        L.qo_hash.argtypes = [u64, u64, u32]
        L.qo_hash.restype = u64
        L.qo_collapse_bit.argtypes = [u64, u64, u32]
        L.qo_collapse_bit.restype = i32
        L.qo_sample_action.argtypes = [vp, u64, u64, u32, vp]
        L.qo_sample_actions_batch.argtypes = [vp, i64, u64, u32, i64, i32, vp]
        L.qo_expand.argtypes = [vp, i32, vp, vp, vp, vp]
        L.qo_expand.restype = i32
        L.qo_ind2move.argtypes = [i32, vp, vp]
        L.qo_update_winner.argtypes = [vp, vp, vp]
        L.qo_legal_mask.argtypes = [vp]
        L.qo_legal_mask.restype = u64
        L.qo_to_vector.argtypes = [vp, vp]
        L.qo_pyhash.argtypes = [vp]
        L.qo_pyhash.restype = i64
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


class OracleBoards:
    """N reference-shaped boards advanced by the C oracle."""

    def __init__(self, n):
        self.n = int(n)
        self.b = np.zeros(self.n, dtype=BOARD_DTYPE)
        self.reset()

    def reset(self):
        lib().qo_reset_batch(_ptr(self.b), self.n)

    def copy(self):
        o = OracleBoards.__new__(OracleBoards)
        o.n = self.n
        o.b = self.b.copy()
        return o

    def step(self, actions, bits=None, seed=0, step_idx=0, board_offset=0, auto_reset=False):
        actions = np.ascontiguousarray(actions, dtype=np.uint8).reshape(self.n, 2)
        reward = np.empty(self.n, dtype=np.float32)
        term = np.empty(self.n, dtype=np.uint8)
        bp = None
        if bits is not None:
            bits = np.ascontiguousarray(bits, dtype=np.uint8).reshape(self.n)
            bp = _ptr(bits)
        lib().qo_step_batch(_ptr(self.b), self.n, _ptr(actions), bp, int(seed), int(step_idx),
                            int(board_offset), int(bool(auto_reset)), _ptr(reward), _ptr(term))
        return reward, term

    def replay(self, actions_base_ptr, stride, n_steps, seed=0, step_idx0=0, board_offset=0, auto_reset=False, scratch=None):
        """n_steps steps from a recorded action stream u8[T, stride, 2] whose slice for these boards starts at the
        address actions_base_ptr (hashed collapse bits): ONE C call, for bench.py's cpu_baseline."""
        if scratch is None:
            scratch = (np.empty(self.n, dtype=np.float32), np.empty(self.n, dtype=np.uint8))
        lib().qo_replay_batch(_ptr(self.b), self.n, ctypes.c_void_p(actions_base_ptr), int(stride), int(n_steps), int(seed),
                              int(step_idx0), int(board_offset), int(bool(auto_reset)), _ptr(scratch[0]), _ptr(scratch[1]))
        return scratch

    def sample_actions(self, seed, step_idx, board_offset=0, auto_reset=False):
        actions = np.empty((self.n, 2), dtype=np.uint8)
        lib().qo_sample_actions_batch(_ptr(self.b), self.n, int(seed), int(step_idx),
                                      int(board_offset), int(bool(auto_reset)), _ptr(actions))
        return actions

    # ---- reference-visible views -------------------------------------------------
    @property
    def board(self):
        return self.b["board"]

    @property
    def n_moves(self):
        return self.b["n_moves"].astype(np.uint8)

    @property
    def moves(self):
        """u8[n,9,2] padded with 255 (the golden-fixture convention)."""
        m = self.b["moves"].astype(np.int16)
        idx = np.arange(9)[None, :, None]
        m = np.where(idx < self.b["n_moves"][:, None, None], m, 255)
        return m.astype(np.uint8)

    @property
    def qmask(self):
        q = self.b["q"][:, :4].copy()
        idx = np.arange(4)[None, :]
        q[idx >= self.b["n_q"][:, None]] = 0
        return q

    @property
    def n_q(self):
        return self.b["n_q"].astype(np.uint8)

    def check_win(self):
        p1 = np.empty(self.n, dtype=np.int8)
        p2 = np.empty(self.n, dtype=np.int8)
        a, c = ctypes.c_int(), ctypes.c_int()
        L = lib()
        base = self.b.ctypes.data
        for i in range(self.n):
            L.qo_check_win(base + 48 * i, ctypes.byref(a), ctypes.byref(c))
            p1[i], p2[i] = a.value, c.value
        return p1, p2

    def observe(self):
        n = self.n
        classical = np.empty((n, 9), dtype=np.int8)
        q1 = np.full((n, 5, 2), 255, dtype=np.uint8)
        q2 = np.full((n, 4, 2), 255, dtype=np.uint8)
        l1 = np.empty(n, dtype=np.uint8)
        l2 = np.empty(n, dtype=np.uint8)
        turn = np.empty(n, dtype=np.uint8)
        a, c, t = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
        L = lib()
        base = self.b.ctypes.data
        for i in range(n):
            L.qo_observe(base + 48 * i, classical[i].ctypes.data, q1[i].ctypes.data,
                         ctypes.byref(a), q2[i].ctypes.data, ctypes.byref(c), ctypes.byref(t))
            l1[i], l2[i], turn[i] = a.value, c.value, t.value
        return classical, q1, l1, q2, l2, turn

    def terminated(self):
        L = lib()
        base = self.b.ctypes.data
        return np.array([L.qo_terminated(base + 48 * i) for i in range(self.n)], dtype=np.uint8)


def boards_from_arrays(board, moves, n_moves, qmask, n_q):
    """OracleBoards from reference-shaped arrays (the golden / export conventions)."""
    n = len(n_moves)
    ob = OracleBoards(n)
    ob.b["board"] = np.asarray(board, dtype=np.int8)
    mv = np.asarray(moves).astype(np.int16)
    mv[mv == 255] = -1
    ob.b["moves"] = mv.astype(np.int8)
    ob.b["n_moves"] = np.asarray(n_moves, dtype=np.int32)
    ob.b["q"][:, :4] = np.asarray(qmask, dtype=np.uint16)
    ob.b["n_q"] = np.asarray(n_q, dtype=np.int32)
    return ob


def expand(ob, action36):
    """qo_expand over a batch -> (n_children, child OracleBoards x2, winner[n,2], terminal[n,2],
    legal u64[n,2], key i64[n,2])."""
    n = ob.n
    kids = [OracleBoards(n), OracleBoards(n)]
    nch = np.zeros(n, dtype=np.uint8)
    winner = np.full((n, 2), -1, dtype=np.int8)
    terminal = np.zeros((n, 2), dtype=np.uint8)
    legal = np.zeros((n, 2), dtype=np.uint64)
    key = np.zeros((n, 2), dtype=np.int64)
    L = lib()
    child = np.zeros(2, dtype=BOARD_DTYPE)
    w = (ctypes.c_int * 2)()
    t = (ctypes.c_int * 2)()
    lm = (ctypes.c_uint64 * 2)()
    base = ob.b.ctypes.data
    for i in range(n):
        k = L.qo_expand(base + 48 * i, int(action36[i]), _ptr(child), w, t, lm)
        nch[i] = k
        for c in range(k):
            kids[c].b[i] = child[c]
            winner[i, c], terminal[i, c], legal[i, c] = w[c], t[c], lm[c]
            key[i, c] = L.qo_pyhash(child[c:c + 1].ctypes.data)
        for c in range(k, 2):
            kids[c].b[i] = ob.b[i]
    return nch, kids, winner, terminal, legal, key


def node_info(ob):
    n = ob.n
    winner = np.empty(n, dtype=np.int8)
    terminal = np.empty(n, dtype=np.uint8)
    legal = np.empty(n, dtype=np.uint64)
    key = np.empty(n, dtype=np.int64)
    L = lib()
    w, t = ctypes.c_int(), ctypes.c_int()
    base = ob.b.ctypes.data
    for i in range(n):
        L.qo_update_winner(base + 48 * i, ctypes.byref(w), ctypes.byref(t))
        winner[i], terminal[i] = w.value, t.value
        legal[i] = L.qo_legal_mask(base + 48 * i)
        key[i] = L.qo_pyhash(base + 48 * i)
    return winner, terminal, legal, key


def to_vector(ob):
    out = np.empty((ob.n, 18, 10), dtype=np.float64)
    L = lib()
    base = ob.b.ctypes.data
    for i in range(ob.n):
        L.qo_to_vector(base + 48 * i, out[i].ctypes.data)
    return out


def rollout(ob, seed, step_idx0, board_offset=0):
    """MCTS._simulate under uniform priors, restated as the sample/step loop it is: ply p uses
    the hash of (seed, id, step_idx0+p).  Returns (result i8[n] per MCTS._reward, plies u8[n],
    final OracleBoards)."""
    n = ob.n
    cur = ob.copy()
    plies = np.zeros(n, dtype=np.uint8)
    L = lib()
    for i in range(n):
        p = cur.b[i:i + 1]
        ptr = p.ctypes.data
        act = np.zeros(2, dtype=np.uint8)
        r, tm, cons = ctypes.c_double(), ctypes.c_int(), ctypes.c_int()
        k = 0
        while k < 9 and not L.qo_terminated(ptr) and bin(int(L.qo_legal_mask(ptr))).count("1") > 0:
            L.qo_sample_action(ptr, int(seed), int(board_offset + i), int(step_idx0 + k), _ptr(act))
            bit = L.qo_collapse_bit(int(seed), int(board_offset + i), int(step_idx0 + k))
            L.qo_step(ptr, int(act[0]), int(act[1]), bit, ctypes.byref(r), ctypes.byref(tm), ctypes.byref(cons))
            k += 1
        plies[i] = k
    winner, _, _, _ = node_info(cur)
    result = np.where(winner < 0, 0, np.where(winner > 0, 1, -1)).astype(np.int8)
    return result, plies, cur


def collapse_bit(seed, board_id, step_idx):
    return lib().qo_collapse_bit(int(seed), int(board_id), int(step_idx))


def hash64(seed, board_id, step_idx):
    return lib().qo_hash(int(seed), int(board_id), int(step_idx))


def ind2move(a):
    lo, hi = ctypes.c_int(), ctypes.c_int()
    lib().qo_ind2move(int(a), ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value
