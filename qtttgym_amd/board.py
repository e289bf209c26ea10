"""`Board`, `QEvalClassic`, `displayBoard` — the reference's L1 names (qtttgym/board.py,
qeval.py, display.py) backed by the HIP library.

`Board` keeps the reference's *attributes* (`.moves`, `.board`, `.qstructs`, `.qeval`) as plain
Python objects because L3 callers subclass it and assign them directly (mcts.py:11-17,241).
`make_move` / `check_win` ship those attributes to the device (qttt_import), run the same fused
kernel `VecEnv` uses on a 1-board batch (qttt_step / qttt_check_win) and read the result back
(qttt_export).  It is a compatibility surface, not a fast path: batch work belongs in `VecEnv`.
"""
import random

import torch

from . import _native


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


_scratch = None


def _scratch_env():
    """One 1-board device batch shared by every Board façade in the process."""
    global _scratch
    if _scratch is None:
        from .vec_env import VecEnv
        _scratch = VecEnv(1, device="cuda")
    return _scratch


class Board:
    def __init__(self, qevaluator):
        self.moves = []                                   # board.py:4  [(lo, hi, round)]
        self.board = [-1, -1, -1, -1, -1, -1, -1, -1, -1]  # board.py:5
        self.qstructs = []                                # board.py:6  [set of squares]
        self.qeval = qevaluator                           # board.py:7

    # ------------------------------------------------------------------ device round trip
    def _upload(self, env):
        moves = torch.full((1, 9, 2), 255, dtype=torch.uint8)
        for i, m in enumerate(self.moves[:9]):
            moves[0, i, 0], moves[0, i, 1] = int(m[0]), int(m[1])
        qmask = torch.zeros((1, 4), dtype=torch.int16)
        for i, s in enumerate(self.qstructs[:4]):
            qmask[0, i] = sum(1 << int(x) for x in s)
        env.import_boards(moves, torch.tensor([len(self.moves)], dtype=torch.uint8),
                          torch.tensor([self.board], dtype=torch.int8), qmask,
                          torch.tensor([len(self.qstructs)], dtype=torch.uint8))

    def _download(self, env):
        ex = {k: v.cpu() for k, v in env.export_boards().items()}
        n = int(ex["n_moves"][0])
        self.moves = [(int(ex["moves"][0, i, 0]), int(ex["moves"][0, i, 1]), i) for i in range(n)]
        self.board[:] = [int(x) for x in ex["board"][0]]   # in place: env.py:71,82 aliasing
        self.qstructs = [set(s for s in range(9) if int(ex["qmask"][0, i]) >> s & 1)
                         for i in range(int(ex["n_q"][0]))]

    def _make_move_device(self, lo, hi, bit, autofill=True):
        env = _scratch_env()
        self._upload(env)
        act = torch.tensor([[lo, hi]], dtype=torch.uint8, device=env.device)
        bits = torch.tensor([bit], dtype=torch.uint8, device=env.device)
        env.step_raw(act, bits)
        self._download(env)
        if not autofill and len(self.moves) and self.moves[-1][0] == self.moves[-1][1]:
            # QEvalClassic.eval() on its own never autofills (that is board.py:22-25's job)
            sq = self.moves[-1][0]
            self.moves.pop()
            self.board[sq] = -1

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
    def make_move(self, move):
        """board.py:9-25, same exceptions with the same messages, raised before any mutation."""
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
        if m0 == m1 and not isinstance(self.qeval, QEvalClassic):
            return self._make_move_custom_eval(lo, hi, m0)
        bit = 0
        if m0 == m1:
            bit = 1 if self.qeval.choose(lo, hi) == hi else 0
        self._make_move_device(lo, hi, bit)

    def _make_move_custom_eval(self, lo, hi, m0):
        """Plug point board.py:2,7,51: a caller-supplied evaluator decides the collapse.  Its
        answer is applied verbatim (board.py:53-56), then the autofill (board.py:22-25)."""
        self.moves.append((lo, hi, len(self.moves)))
        comp = self.qstructs[m0]
        zipped = [(i, m) for i, m in enumerate(self.moves) if m[0] in comp]
        outcomes = self.qeval.eval([m for _, m in zipped])
        for (r, _), o in zip(zipped, outcomes):
            self.board[o] = r
        self.qstructs.pop(m0)
        if self.board.count(-1) == 1:
            idx = self.board.index(-1)
            self.board[idx] = len(self.moves)
            self.moves.append((idx, idx, len(self.moves)))

    def update_qstructs(self, move):
        raise NotImplementedError(
            "update_qstructs (board.py:27-69) is fused into the qttt_step kernel; call make_move")

    def check_win(self):
        """board.py:71-115 -> (p1_round, p2_round), -1 = no line."""
        env = _scratch_env()
        self._upload(env)
        p1, p2 = env.check_win()
        return int(p1[0]), int(p2[0])


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
