"""`Board`, `QEvalClassic`, `displayBoard` — the reference's L1 names (qtttgym/board.py,
qeval.py, display.py) backed by the HIP library.

`Board` keeps the reference's *attributes* (`.moves`, `.board`, `.qstructs`, `.qeval`) as plain
Python objects because L3 callers subclass it and assign them directly (mcts.py:11-17,241).
`make_move` / `update_qstructs` / `check_win` write those attributes into one 64-byte record in
pinned host memory, run qttt_board_op (import -> the same step function the batch kernels use ->
export + check_win, ONE kernel launch reading and writing the pinned records directly) and take the
attributes back from the out record.  It is a compatibility surface, not a fast path (a launch and
a synchronise per call): batch work belongs in `VecEnv`, or — for reference-shaped objects —
in `Board.make_moves`, which ships n records through one launch.

Thread-safety: the pinned staging records are shared by every Board of the process and guarded by a lock;
a Board object itself is a plain mutable Python object, as in the reference.
"""
import ctypes
import os
import random
import threading

import torch

from . import _native
from .vec_env import _raw_stream

try:                                              # optional host-side accelerator (csrc/fastboard.c); never a compute path
    from . import _fastboard
except ImportError:
    _fastboard = None
_current_device = getattr(torch._C, "_cuda_getDevice", torch.cuda.current_device)


class QEvalClassic:
    """qeval.py:4-51.  The collapse itself is resolved inside the HIP kernel; the only thing left
    on the host is the reference's single random draw (qeval.py:35), made with the same call on
    the same global `random` stream so that `random.seed(s)` reproduces the reference's games."""

    def choose(self, lo, hi):
        return random.choice((lo, hi))

    def eval(self, entangled_moves):
        """Stand-alone use, same contract as qeval.py:5: k moves (lo, hi, round) of one cyclic
        component, closing move last -> the square each move collapses onto."""
        k = len(entangled_moves)
        if k == 0:
            return []
        lo, hi = entangled_moves[-1][0], entangled_moves[-1][1]
        b = Board(self)
        b.moves = [(m[0], m[1], i) for i, m in enumerate(entangled_moves[:-1])]
        squares = set()
        for m in entangled_moves:
            squares.add(m[0])
            squares.add(m[1])
        b.qstructs = [squares] if k > 1 else []
        b._make_move_device(lo, hi, 1 if self.choose(lo, hi) == hi else 0, autofill=False)
        out = [-1] * k
        for sq, r in enumerate(b.board):
            if 0 <= r < k:
                out[r] = sq
        return out


def _device_eval(qeval):
    """True iff the collapse may run in the HIP kernel: the evaluator's `eval` IS QEvalClassic.eval (the class itself or a
    subclass that does not override it).  Anything else — an unrelated class, or a QEvalClassic subclass with its own
    `eval` — is the reference's plug point (board.py:2,7,51) and decides the collapse on the host."""
    return getattr(type(qeval), "eval", None) is QEvalClassic.eval


class _Staging:
    """One 64-byte record in and one out (include/qttt.h: qttt_board_op), both in pinned host memory
    that the kernel reads and writes directly: a Board call is one kernel launch and a poll of the out
    record's completion stamp (qttt_board_op_host; it synchronises the stream only for more than 256
    records or when the poll has not ended after ~2 ms) — no host-to-device or device-to-host copy calls."""

    def __init__(self):
        if not torch.cuda.is_available():
            raise _native.QtttNativeError("no HIP device visible (torch.cuda.is_available() is False); "
                                          "Board runs its rules in libqttt_hip.so, there is no CPU path")
        self.lib = _native.lib()
        self.lock = threading.Lock()
        self.op_sync = self.lib.qttt_board_op_host         # pinned records: poll the stamp, no stream synchronise
        self.device = torch.device("cuda", torch.cuda.current_device())
        self._alloc(64)
        # The bookkeeping around the call (pack / adopt below) in C when qtttgym_amd/_fastboard.so is built
        # (csrc/fastboard.c, __graft_entry__.build()): same attributes, same aliasing, ~4 us less per call.  It works on
        # the FIRST record of the pinned buffers, so it is re-initialised whenever they are re-allocated.
        self.fast = None
        if _fastboard is not None and os.environ.get("QTTT_NO_FASTBOARD") != "1":
            self._fn_addr = ctypes.cast(self.lib.qttt_board_op_host, ctypes.c_void_p).value
            _fastboard.init(self._fn_addr, self.p_in, self.p_out)
            self.fast = _fastboard.board_op

    def _alloc(self, n_records):
        nb = _native.BOARD_RECORD_BYTES * n_records
        self.t_in = torch.zeros(nb, dtype=torch.uint8).pin_memory()
        self.t_out = torch.zeros(nb, dtype=torch.uint8).pin_memory()
        self.a_in = self.t_in.numpy()          # plain memory views of the pinned buffers
        self.a_out = self.t_out.numpy()
        self.m_in = memoryview(self.a_in)
        self.p_in, self.p_out = self.t_in.data_ptr(), self.t_out.data_ptr()
        if getattr(self, "fast", None) is not None:
            _fastboard.init(self._fn_addr, self.p_in, self.p_out)

    @staticmethod
    def pack(board, op, lo=0, hi=0, bit=0, drop_last_move=False):
        """board's attributes + the move as the first 41 bytes of a qttt_board_op record."""
        moves = board.moves[:-1] if drop_last_move else board.moves
        if len(moves) > 9:
            moves = moves[:9]
        n = len(moves)
        qs = board.qstructs
        nq = min(len(qs), 4)
        masks = [0, 0, 0, 0]
        for k in range(nq):
            mask = 0
            for x in qs[k]:
                mask |= 1 << int(x)
            masks[k] = mask & 0xFFFF
        return (bytes([v & 255 for m in moves for v in (m[0], m[1])]) + _PAD18[2 * n:] + bytes([n])
                + bytes([x & 255 for x in board.board[:9]])
                + bytes([nq, op, masks[0] & 255, masks[0] >> 8, masks[1] & 255, masks[1] >> 8,
                         masks[2] & 255, masks[2] >> 8, masks[3] & 255, masks[3] >> 8,
                         lo & 255, hi & 255, bit & 255]))

    def run(self, board, op, lo=0, hi=0, bit=0, drop_last_move=False):
        """Ships board's attributes + the move, runs qttt_board_op, returns the out record (bytes)."""
        rec = self.pack(board, op, lo, hi, bit, drop_last_move)
        with self.lock:                                    # one staging record per process: one caller at a time
            # the record is built as one bytes object and lands in the pinned buffer with one copy
            self.m_in[0:41] = rec
            self._launch(1)
            return self.a_out[:_native.BOARD_RECORD_BYTES].tobytes()

    def run_many(self, records):
        """n packed records (41 bytes each) -> n out records (64 bytes each), ONE launch and one synchronise."""
        n = len(records)
        nb = _native.BOARD_RECORD_BYTES
        with self.lock:
            if n * nb > self.t_in.numel():
                self._alloc(max(n, 2 * (self.t_in.numel() // nb)))
            self.m_in[0:n * nb] = b"".join(r + _PAD23 for r in records)
            self._launch(n)
            out = self.a_out[:n * nb].tobytes()
        return [out[i * nb:(i + 1) * nb] for i in range(n)]

    def _launch(self, n):
        # launch + wait for the out records' stamps in one call, on the caller's current stream (raw handle)
        if torch.cuda.current_device() == self.device.index:
            rc = self.op_sync(self.p_in, self.p_out, n, _raw_stream(self.device.index))
        else:
            with torch.cuda.device(self.device):
                rc = self.op_sync(self.p_in, self.p_out, n, _raw_stream(self.device.index))
        if rc:
            _native.check(rc, "qttt_board_op_host")


_PAD18 = b"\xff" * 18
_PAD23 = b"\x00" * 23                                                              # 41 packed bytes -> a 64-byte record
_I8 = tuple(x - 256 if x > 127 else x for x in range(256))                           # u8 -> i8
_SQUARES = tuple(tuple(v for v in range(9) if m >> v & 1) for m in range(512))       # mask -> squares

_staging = None


def _stage():
    """The staging records shared by every Board façade in the process."""
    global _staging
    if _staging is None:
        _staging = _Staging()
    return _staging


class Board:
    def __init__(self, qevaluator):
        self.moves = []                                   # board.py:4  [(lo, hi, round)]
        self.board = [-1, -1, -1, -1, -1, -1, -1, -1, -1]  # board.py:5
        self.qstructs = []                                # board.py:6  [set of squares]
        self.qeval = qevaluator                           # board.py:7

    # ------------------------------------------------------------------ device round trip
    def _adopt(self, o):
        """Takes the attributes back from an out record (bytes)."""
        r = o
        n = min(r[18], 9)
        # All three attributes are updated IN PLACE, as the reference does (board.py:19,25 append to .moves,
        # :53-54 write into .board, :56-69 pop / assign / append on .qstructs): a caller that took `b.moves`,
        # `b.board` or `b.qstructs` before the call sees the move afterwards (env.py:71,82 aliases .board).
        self.moves[:] = zip(r[0:2 * n:2], r[1:2 * n:2], range(n))
        self.board[:] = [_I8[x] for x in r[19:28]]
        new = [set(_SQUARES[(r[30 + 2 * k] | r[31 + 2 * k] << 8) & 511]) for k in range(min(r[28], 4))]
        old = self.qstructs
        if old:
            # the set OBJECTS too: an untouched component stays the object it was (board.py:56,61 only pop
            # the others' neighbour), "add to sets" grows its set in place (board.py:68-69); a union is a
            # new set (board.py:60), as in the reference
            spare = list(old)
            for k, s in enumerate(new):
                for t in spare:
                    if t == s:
                        new[k] = t
                        spare.remove(t)
                        break
            if len(new) == len(old):
                for k, s in enumerate(new):
                    if not any(s is t for t in old):
                        grown = [t for t in spare if t < s]
                        if len(grown) == 1:
                            grown[0].update(s)
                            new[k] = grown[0]
                            spare.remove(grown[0])
        self.qstructs[:] = new
        # the out record carries check_win of the new state (a function of .board alone,
        # board.py:71-115): remembered, keyed by the board it belongs to
        self._win = (tuple(self.board), _I8[r[49]], _I8[r[50]])

    @staticmethod
    def _i8(x):
        x = int(x)
        return x - 256 if x > 127 else x

    def _make_move_device(self, lo, hi, bit, autofill=True, drop_last_move=False):
        op = _native.OP_MAKE_MOVE if autofill else _native.OP_UPDATE_QSTRUCTS
        st = _stage()
        if st.fast is not None and _current_device() == st.device.index:
            with st.lock:                                          # the one staging record: one caller at a time
                rc = st.fast(self, op, lo, hi, bit, drop_last_move, _raw_stream(st.device.index))
            if rc == 0:
                return
            if rc != -100:                                         # -100: an attribute of an unusual type, the Python path below
                _native.check(rc, "qttt_board_op_host")
        self._adopt(st.run(self, op, lo, hi, bit, drop_last_move))

    @classmethod
    def from_export(cls, ex, index=0, qevaluator=None):
        b = cls(qevaluator or QEvalClassic())
        n = int(ex["n_moves"][index])
        b.moves = [(int(ex["moves"][index, i, 0]), int(ex["moves"][index, i, 1]), i) for i in range(n)]
        b.board = [int(x) for x in ex["board"][index]]
        b.qstructs = [set(s for s in range(9) if int(ex["qmask"][index, i]) >> s & 1)
                      for i in range(int(ex["n_q"][index]))]
        return b

    # ------------------------------------------------------------------ reference API
    def _validate(self, move):
        """board.py:10-18 + 28-42: raises what the reference raises (before any mutation), else returns
        (lo, hi, closes_a_cycle, index of lo's component)."""
        if move[0] == move[1]:
            raise Exception("Move in same square not allowed when not necessary")
        if self.board[move[0]] != -1 or self.board[move[1]] != -1:   # IndexError for >8, as the list does
            raise Exception("Move in classical square not allowed")
        if move[0] < 0 or move[1] < 0:
            # the reference would alias a negative index to square 9+i and store the negative
            # number in .moves; that is outside the action space (env.py:19) and the u8 ABI
            raise IndexError("negative square index is outside the action space")
        lo, hi = (move[0], move[1]) if move[0] < move[1] else (move[1], move[0])
        # board.py:28-42: does the move close a cycle?  Host-side only to decide whether the
        # reference would consume a random draw (qeval.py:35); the device recomputes it.
        m0, m1 = -1, -2
        for i, s in enumerate(self.qstructs):
            if lo in s:
                m0 = i
                break
        for j, s in enumerate(self.qstructs):
            if hi in s:
                m1 = j
                break
        return lo, hi, m0 == m1, m0

    def make_move(self, move):
        """board.py:9-25, same exceptions with the same messages, raised before any mutation."""
        lo, hi, cycle, m0 = self._validate(move)
        if cycle and not _device_eval(self.qeval):
            return self._make_move_custom_eval(lo, hi, m0)
        bit = 0
        if cycle:
            bit = 1 if self.qeval.choose(lo, hi) == hi else 0
        self._make_move_device(lo, hi, bit)

    @staticmethod
    def make_moves(boards, moves, bits=None):
        """n (board, move) pairs in ONE qttt_board_op launch and one synchronise: what an `_expand_child`-style
        loop over a node's actions (mcts.py:210-221,233-267: copy the parent, make_move, 36 times) costs as
        one round trip instead of 36.  Each board is mutated exactly as board.make_move(move) would.
        bits: optional per-pair collapse choices (0 -> the closing move lands on min(a,b), 1 -> on max(a,b));
        without them every cycle draws from its board's evaluator in order, as a loop of make_move would.
        Returns a list with None where the move was made and the exception make_move would have raised
        where it was not (that board is untouched)."""
        n = len(boards)
        if len(moves) != n or (bits is not None and len(bits) != n):
            raise ValueError("boards, moves (and bits) must have the same length")
        result = [None] * n
        recs, idx = [], []
        for i in range(n):
            b = boards[i]
            try:
                lo, hi, cycle, m0 = b._validate(moves[i])
            except Exception as e:                               # noqa: BLE001 — the reference raises bare Exception
                if isinstance(e, _native.QtttNativeError):
                    raise
                result[i] = e
                continue
            if cycle and not _device_eval(b.qeval):
                b._make_move_custom_eval(lo, hi, m0)
                continue
            bit = 0
            if cycle:
                bit = (int(bits[i]) & 1) if bits is not None else (1 if b.qeval.choose(lo, hi) == hi else 0)
            recs.append(_Staging.pack(b, _native.OP_MAKE_MOVE, lo, hi, bit))
            idx.append(i)
        if recs:
            for i, o in zip(idx, _stage().run_many(recs)):
                boards[i]._adopt(o)
        return result

    def _make_move_custom_eval(self, lo, hi, m0, autofill=True):
        """Plug point board.py:2,7,51: a caller-supplied evaluator decides the collapse.  Its
        answer is applied verbatim (board.py:53-56), then the autofill (board.py:22-25)."""
        self.moves.append((lo, hi, len(self.moves)))
        comp = self.qstructs[m0]
        zipped = [(i, m) for i, m in enumerate(self.moves) if m[0] in comp]
        outcomes = self.qeval.eval([m for _, m in zipped])
        for (r, _), o in zip(zipped, outcomes):
            self.board[o] = r
        self.qstructs.pop(m0)
        if autofill and self.board.count(-1) == 1:
            idx = self.board.index(-1)
            self.board[idx] = len(self.moves)
            self.moves.append((idx, idx, len(self.moves)))

    def update_qstructs(self, move):
        """board.py:27-69 on its own: entangle `move` with the components, or — if it closes a cycle —
        collapse the component (self.board gets the rounds, the component leaves self.qstructs).
        The reference calls it from make_move right after `self.moves.append(...)` (board.py:19-20)
        and its collapse branch reads the move back from self.moves, so the contract is the same
        here: `move` must be the last entry of self.moves.  No validation, no autofill (those are
        make_move's, board.py:10-15,22-25).  Same device path as make_move."""
        lo, hi = (move[0], move[1]) if move[0] < move[1] else (move[1], move[0])
        if not self.moves or (self.moves[-1][0], self.moves[-1][1]) != (lo, hi):
            raise ValueError("update_qstructs(move): move must be the last entry of self.moves "
                             "(board.py:19-20 appends it before the call)")
        m0, m1 = -1, -2
        for i, s in enumerate(self.qstructs):
            if lo in s:
                m0 = i
                break
        for j, s in enumerate(self.qstructs):
            if hi in s:
                m1 = j
                break
        if m0 == m1 and not _device_eval(self.qeval):
            self.moves.pop()                                       # _make_move_custom_eval appends it again
            return self._make_move_custom_eval(lo, hi, m0, autofill=False)
        bit = 0
        if m0 == m1:
            bit = 1 if self.qeval.choose(lo, hi) == hi else 0
        self._make_move_device(lo, hi, bit, autofill=False, drop_last_move=True)

    def check_win(self):
        """board.py:71-115 -> (p1_round, p2_round), -1 = no line."""
        key = tuple(self.board)
        w = getattr(self, "_win", None)
        if w is None or w[0] != key:                                   # .board was assigned by the caller
            o = _stage().run(self, _native.OP_CHECK_WIN)
            w = self._win = (key, self._i8(o[49]), self._i8(o[50]))
        return w[1], w[2]


def displayBoard(board):
    """display.py:4-32: ASCII 3x3-of-3x3 rendering (host-side pretty print, no compute)."""
    cells = [[" "] * 9 for _ in range(9)]
    for i, m in enumerate(board.moves):
        cells[m[0]][i] = str(i)
        cells[m[1]][i] = str(i)
    for sq, r in enumerate(board.board):
        if r >= 0:
            mark = "x" if r % 2 == 0 else "o"
            cells[sq] = [mark if j % 2 == r % 2 else " " for j in range(9)]
            cells[sq][4] = str(r)
    bar = "+---+---+---+\n"
    out = ""
    for big_row in range(3):
        out += bar
        for k in range(3):
            for big_col in range(3):
                out += "|" + "".join(cells[big_row * 3 + big_col][k * 3:k * 3 + 3])
            out += "|\n"
    out += bar
    print(out)
