"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

Pure-Python single-board restatement of the reference's Env.step path (lists and sets, like the
reference), for small cases and for the like-for-like "interpreter speed" line beside the GPU number
(SURVEY.md §8d).  Each method cites the reference lines it follows; pinned by the golden traces
(tests/test_oracle_golden.py)."""


class PyBoard:
    def __init__(self):                                   # board.py:2-7
        self.moves = []
        self.board = [-1] * 9
        self.qstructs = []

    def make_move(self, a, b, bit):                       # board.py:9-25; returns True if a bit was used
        if a == b:
            raise Exception("Move in same square not allowed when not necessary")
        if self.board[a] != -1 or self.board[b] != -1:    # IndexError for > 8, like the list
            raise Exception("Move in classical square not allowed")
        if a < 0 or b < 0:
            raise IndexError("negative square index is outside the action space")
        lo, hi = (a, b) if a < b else (b, a)
        self.moves.append((lo, hi, len(self.moves)))
        used = self._update_qstructs(lo, hi, bit)
        if self.board.count(-1) == 1:                     # board.py:22-25
            idx = self.board.index(-1)
            self.board[idx] = len(self.moves)
            self.moves.append((idx, idx, len(self.moves)))
        return used

    def _update_qstructs(self, lo, hi, bit):              # board.py:27-69
        m0 = next((i for i, s in enumerate(self.qstructs) if lo in s), -1)
        m1 = next((i for i, s in enumerate(self.qstructs) if hi in s), -2)
        if m0 == m1:
            comp = self.qstructs[m0]
            ent = [m for m in self.moves if m[0] in comp]
            for m, sq in zip(ent, self._eval(ent, bit)):
                self.board[sq] = m[2]
            self.qstructs.pop(m1)
            return True
        if m0 >= 0 and m1 >= 0:
            self.qstructs[m0] = self.qstructs[m0] | self.qstructs[m1]
            self.qstructs.pop(m1)
        else:
            i = max(m0, m1)
            if i < 0:
                self.qstructs.append(set())
                i = len(self.qstructs) - 1
            self.qstructs[i].update((lo, hi))
        return False

    @staticmethod
    def _eval(ent, bit):                                  # qeval.py:5-51
        out = [-1] * len(ent)
        inx = {m: i for i, m in enumerate(ent)}
        rutor = [set() for _ in range(9)]
        for m in ent:
            rutor[m[0]].add(m)
            rutor[m[1]].add(m)
        for i0 in range(9):                               # :23-31 leaf peel
            i = i0
            while len(rutor[i]) == 1:
                m = rutor[i].pop()
                nxt = m[1] if i == m[0] else m[0]
                out[inx[m]] = i
                rutor[nxt].remove(m)
                i = nxt
        last = ent[-1]                                    # :35-49 forced walk round the cycle
        out[-1] = last[1] if bit else last[0]
        r_start, r = last[0], last[1]
        fell = r == out[-1]
        rutor[r].remove(last)
        while r != r_start:
            m = rutor[r].pop()
            res = m[0] if fell ^ (m[0] == r) else m[1]
            out[inx[m]] = res
            r = m[0] if m[1] == r else m[1]
            rutor[r].remove(m)
            fell = r == res
        return out

    def check_win(self):                                  # board.py:71-115
        mark = [0 if m < 0 else (m % 2) * 2 - 1 for m in self.board]
        p1 = p2 = 10
        for line in ((0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (2, 4, 6), (0, 4, 8)):
            s = sum(mark[i] for i in line)
            if s == -3:
                p1 = min(p1, max(self.board[i] for i in line))
            elif s == 3:
                p2 = min(p2, max(self.board[i] for i in line))
        return (p1 if p1 < 10 else -1), (p2 if p2 < 10 else -1)


class PyEnv:
    def __init__(self):
        self.b = PyBoard()

    def reset(self):                                      # env.py:55-57
        self.b = PyBoard()

    def step(self, a, b, bit):                            # env.py:34-53 without the observation
        try:
            self.b.make_move(a, b, bit)
        except Exception:
            pass
        p1, p2 = self.b.check_win()
        r = (-1 ** 0) * float(p1 > 0 or p2 > 0)           # env.py:49: -(1 ** cur_player) * float(win)
        return r, (p1 > 0 or p2 > 0) or len(self.b.moves) > 8

    def observation(self):                                # env.py:68-85
        q1, q2 = [], []
        board = self.b.board
        for i, m in enumerate(self.b.moves):
            if i not in board:                            # a move is collapsed iff its round is on the board (:72-74)
                (q2 if i % 2 else q1).append((m[0], m[1]))
        return {"q_states_p1": q1, "q_states_p2": q2, "classical": board, "turn": len(self.b.moves) % 2}

    def step_full(self, a, b, bit):
        """What the reference's Env.step returns (env.py:34-53), observation included: the like-for-like interpreter
        line beside the GPU numbers (bench.py cpu_baseline.python_interpreter_steps_per_s, tools/facade_latency.py)."""
        cur_player = len(self.b.moves) % 2                # env.py:35
        try:
            self.b.make_move(a, b, bit)
        except Exception:
            pass
        obs = self.observation()                          # env.py:46
        p1, p2 = self.b.check_win()
        r = (-1 ** cur_player) * float(p1 > 0 or p2 > 0)  # env.py:49, verbatim precedence
        return obs, r, (p1 > 0 or p2 > 0) or len(self.b.moves) > 8, False, {}
