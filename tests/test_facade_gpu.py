"""The reference's single-board surface (`Env`, `Board`, `QEvalClassic`) on top of the HIP
library, checked against the golden traces: same Python types, same aliasing, same exceptions."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class Bits:
    """Stands in for the `random` module inside qtttgym_amd.board (the one draw of qeval.py:35)."""

    def __init__(self):
        self.bit = 0
        self.calls = 0

    def choice(self, seq):
        self.calls += 1
        return seq[self.bit]


@pytest.fixture()
def bits(monkeypatch):
    import qtttgym_amd.board as board_mod
    b = Bits()
    monkeypatch.setattr(board_mod, "random", b)
    return b


def test_env_matches_golden_python_types(golden, bits):
    from qtttgym_amd import Env
    kinds = list(golden["kind"])
    picks = [i for i, k in enumerate(kinds) if k.startswith("K")]
    picks += [i for i, k in enumerate(kinds) if k == "uniform"][:40]
    picks += [i for i, k in enumerate(kinds) if k == "adversarial"][:40]
    T = golden["bits"].shape[1]
    for e in picks:
        env = Env()
        obs, info = env.reset()
        assert info == {} and obs["classical"] == [-1] * 9 and obs["turn"] == 0
        classical_alias = obs["classical"]
        for t in range(T):
            bits.bit = int(golden["bits"][e, t])
            calls = bits.calls
            a, b = (int(x) for x in golden["actions"][e, t])
            obs, r, term, trunc, info = env.step((a, b))
            assert isinstance(r, float) and isinstance(term, bool) and trunc is False and info == {}
            assert bits.calls - calls == int(golden["consumed"][e, t])
            assert obs["classical"] is classical_alias                 # env.py:71,82 aliasing
            assert obs["classical"] == golden["board"][e, t].tolist()
            n1, n2 = int(golden["q_p1_len"][e, t]), int(golden["q_p2_len"][e, t])
            assert obs["q_states_p1"] == [tuple(x) for x in golden["q_p1"][e, t, :n1].tolist()]
            assert obs["q_states_p2"] == [tuple(x) for x in golden["q_p2"][e, t, :n2].tolist()]
            assert obs["turn"] == int(golden["turn"][e, t])
            assert np.float64(r).view(np.uint64) == golden["reward"][e, t].view(np.uint64)   # -0.0
            assert term == bool(golden["terminated"][e, t])
            assert env.turn() == int(golden["n_moves"][e, t])
            gb = env._gameboard
            nm = int(golden["n_moves"][e, t])
            assert gb.moves == [(int(golden["moves"][e, t, i, 0]), int(golden["moves"][e, t, i, 1]), i)
                                for i in range(nm)]
            nq = int(golden["n_q"][e, t])
            assert gb.qstructs == [set(s for s in range(9) if int(golden["qmask"][e, t, i]) >> s & 1)
                                   for i in range(nq)]
            assert gb.check_win() == (int(golden["p1_round"][e, t]), int(golden["p2_round"][e, t]))


def test_board_exceptions_and_no_mutation(bits):
    from qtttgym_amd import Board, QEvalClassic
    b = Board(QEvalClassic())
    with pytest.raises(Exception, match="Move in same square not allowed when not necessary"):
        b.make_move((4, 4))
    b.make_move((1, 0))
    assert b.moves == [(0, 1, 0)] and b.qstructs == [{0, 1}]
    bits.bit = 1
    b.make_move((0, 1))
    assert b.board[:2] == [0, 1] and b.qstructs == []
    snapshot = (list(b.board), list(b.moves), list(b.qstructs))
    with pytest.raises(Exception, match="Move in classical square not allowed"):
        b.make_move((0, 5))
    with pytest.raises(IndexError):
        b.make_move((9, 5))
    assert (b.board, b.moves, b.qstructs) == snapshot


def test_l3_style_subclass_assigning_attributes(bits):
    """mcts.py:9-17,235-242: subclass Board, assign .board/.moves/.qstructs, then make_move."""
    from qtttgym_amd import Board, QEvalClassic

    class GameState(Board):
        def __init__(self, board, moves):
            Board.__init__(self, QEvalClassic())
            self.board = board
            self.moves = moves

    parent = GameState([-1] * 9, [])
    for mv in [(4, 6), (1, 2), (1, 8), (0, 2)]:           # K7 prefix
        parent.make_move(mv)
    child = GameState(parent.board.copy(), parent.moves.copy())
    child.qstructs = [set(s) for s in parent.qstructs]
    bits.bit = 1
    child.make_move((4, 6))
    assert child.board == [-1, -1, -1, -1, 0, -1, 4, -1, -1]
    assert parent.board == [-1] * 9                          # parent untouched
    child.make_move((0, 2))
    assert child.board == [3, 1, 5, -1, 0, -1, 4, -1, 2]
    assert child.check_win() == (-1, 5)


def test_qevalclassic_eval_standalone(bits):
    from qtttgym_amd import QEvalClassic
    q = QEvalClassic()
    bits.bit = 0
    assert q.eval([(0, 1, 0), (0, 1, 1)]) == [1, 0]          # K1
    bits.bit = 1
    assert q.eval([(0, 1, 0), (0, 1, 1)]) == [0, 1]          # K2
    # K3: 3-cycle 1-2-3 with tail 0-1, closing move (1,3) lands on 3
    assert q.eval([(0, 1, 0), (1, 2, 1), (2, 3, 2), (1, 3, 3)]) == [0, 1, 2, 3]


def test_custom_evaluator_plug_point():
    """board.py:2,7,51: any object with .eval(list) decides the collapse."""
    from qtttgym_amd import Board

    class FirstSquare:
        def __init__(self):
            self.seen = None

        def eval(self, entangled):
            self.seen = list(entangled)
            return [1, 0]

    ev = FirstSquare()
    b = Board(ev)
    b.make_move((0, 1))
    b.make_move((0, 1))
    assert ev.seen == [(0, 1, 0), (0, 1, 1)]
    assert b.board[:2] == [1, 0] and b.qstructs == []

    # a SUBCLASS of QEvalClassic that overrides .eval is the same plug point (the rule is "whose eval is it", not
    # isinstance): its answer is applied verbatim, on make_move, make_moves and update_qstructs alike
    from qtttgym_amd import QEvalClassic

    class Contrary(QEvalClassic):
        def __init__(self):
            self.calls = 0

        def eval(self, entangled):
            self.calls += 1
            return [m[1] if i % 2 == 0 else m[0] for i, m in enumerate(entangled)]     # move 0 on its hi, move 1 on its lo

    ev2 = Contrary()
    b2 = Board(ev2)
    b2.make_move((3, 5))
    b2.make_move((5, 3))
    assert ev2.calls == 1 and b2.board[3] == 1 and b2.board[5] == 0 and b2.qstructs == []
    b3 = Board(ev2)
    b3.make_move((0, 8))
    assert Board.make_moves([b3], [(0, 8)]) == [None] and ev2.calls == 2 and b3.board[8] == 0 and b3.board[0] == 1
    b4 = Board(ev2)
    b4.make_move((1, 2))
    b4.moves.append((1, 2, 1))
    b4.update_qstructs((1, 2))
    assert ev2.calls == 3 and b4.board[2] == 0 and b4.board[1] == 1

    # ... while a subclass that only overrides the DRAW (qeval.py:35) stays on the device path
    class AlwaysHi(QEvalClassic):
        def choose(self, lo, hi):
            return hi

    b5 = Board(AlwaysHi())
    b5.make_move((4, 6))
    b5.make_move((4, 6))
    assert b5.board[6] == 1 and b5.board[4] == 0


def test_vecenv_render_and_turn(capsys):
    from qtttgym_amd import VecEnv
    import torch
    env = VecEnv(3)
    env.step(torch.tensor([[0, 1], [2, 2], [4, 8]], dtype=torch.uint8))
    assert env.turn().tolist() == [1, 0, 1]
    env.render(2)
    assert "+---+---+---+" in capsys.readouterr().out


def test_update_qstructs_is_callable_and_is_make_move_minus_append_and_autofill(golden, bits):
    """board.py:27-69 as a public method of the Board duck type: replaying the golden episodes with
    the reference's own make_move body (validate, normalise, append — board.py:10-19 — then
    update_qstructs, then the autofill of board.py:22-25) written out here on the host gives the
    reference's states."""
    from qtttgym_amd import Board, QEvalClassic
    kinds = list(golden["kind"])
    picks = [i for i, k in enumerate(kinds) if k.startswith("K")] + [i for i, k in enumerate(kinds) if k == "uniform"][:25]
    T = golden["bits"].shape[1]
    for e in picks:
        b = Board(QEvalClassic())
        for t in range(T):
            bits.bit = int(golden["bits"][e, t])
            a, c = (int(x) for x in golden["actions"][e, t])
            ok = a != c and a < 9 and c < 9 and b.board[a] == -1 and b.board[c] == -1   # board.py:10-15
            if ok:
                lo, hi = min(a, c), max(a, c)
                b.moves.append((lo, hi, len(b.moves)))              # board.py:19
                b.update_qstructs((lo, hi))                         # board.py:20
                if b.board.count(-1) == 1:                          # board.py:22-25
                    idx = b.board.index(-1)
                    b.board[idx] = len(b.moves)
                    b.moves.append((idx, idx, len(b.moves)))
            nm = int(golden["n_moves"][e, t])
            assert b.board == golden["board"][e, t].tolist(), (e, t)
            assert b.moves == [(int(golden["moves"][e, t, i, 0]), int(golden["moves"][e, t, i, 1]), i) for i in range(nm)]
            nq = int(golden["n_q"][e, t])
            assert b.qstructs == [set(s for s in range(9) if int(golden["qmask"][e, t, i]) >> s & 1) for i in range(nq)]
    b = Board(QEvalClassic())
    with pytest.raises(ValueError):
        b.update_qstructs((0, 1))                                   # not appended to .moves first


def test_board_call_is_one_launch_without_copies(bits):
    """The façade's round trip: attributes -> one pinned 64-byte record -> qttt_board_op -> one
    pinned record back.  Checked through the C ABI on hand-written records, incl. the op codes."""
    import torch
    from qtttgym_amd import _native
    L = _native.lib()
    rec_in = torch.zeros(64, dtype=torch.uint8).pin_memory()
    rec_out = torch.zeros(64, dtype=torch.uint8).pin_memory()
    a = rec_in.numpy()
    a[0:18] = 255
    a[0:2] = (0, 1)                                                  # moves = [(0,1,0)]
    a[18] = 1
    a[19:28] = 255                                                   # board = [-1]*9
    a[28] = 1
    a[30] = 0b11                                                     # qstructs = [{0,1}]
    a[38], a[39], a[40] = 1, 0, 1                                    # move (1,0), bit 1 -> lands on hi = 1
    s = torch.cuda.current_stream()
    for op in (_native.OP_MAKE_MOVE, _native.OP_UPDATE_QSTRUCTS):
        a[29] = op
        assert L.qttt_board_op(rec_in.data_ptr(), rec_out.data_ptr(), 1, s.cuda_stream) == 0
        s.synchronize()
        o = rec_out.numpy()
        assert o[18] == 2 and list(o[0:4]) == [0, 1, 0, 1] and o[41] == 0
        assert list(o[19:21]) == [0, 1] and all(x == 255 for x in o[21:28]) and o[28] == 0   # K2
        assert o[48] == 0 and o[49] == 255 and o[50] == 255 and bytes(o[44:48]) == b"\x00\x00\x00\x80"
    a[29] = _native.OP_CHECK_WIN
    assert L.qttt_board_op(rec_in.data_ptr(), rec_out.data_ptr(), 1, s.cuda_stream) == 0
    s.synchronize()
    assert rec_out.numpy()[18] == 1 and rec_out.numpy()[28] == 1     # nothing moved
    a[29], a[38], a[39] = _native.OP_MAKE_MOVE, 4, 4                 # same square: rejected, state unchanged
    assert L.qttt_board_op(rec_in.data_ptr(), rec_out.data_ptr(), 1, s.cuda_stream) == 0
    s.synchronize()
    assert rec_out.numpy()[41] == 1 and rec_out.numpy()[18] == 1
    assert L.qttt_board_op(None, rec_out.data_ptr(), 1, None) == -1 and L.qttt_board_op(None, None, 0, None) == 0


def test_board_op_batches_records(golden):
    """qttt_board_op takes n records per launch: a whole golden step for 300 boards at once, every
    board imported from the reference's attributes before the move and compared after it."""
    import torch
    from qtttgym_amd import _native
    L = _native.lib()
    g = golden
    E = 300
    rin = torch.zeros((E, 64), dtype=torch.uint8).pin_memory()
    rout = torch.zeros((E, 64), dtype=torch.uint8).pin_memory()
    s = torch.cuda.current_stream()
    T = g["bits"].shape[1]
    for t in range(1, T):
        a = rin.numpy()
        a[:, 0:18] = g["moves"][:E, t - 1].reshape(E, 18)
        a[:, 18] = g["n_moves"][:E, t - 1]
        a[:, 19:28] = g["board"][:E, t - 1].view(np.uint8)
        a[:, 28] = g["n_q"][:E, t - 1]
        a[:, 29] = _native.OP_MAKE_MOVE
        a[:, 30:38] = g["qmask"][:E, t - 1].astype("<u2").view(np.uint8).reshape(E, 8)
        a[:, 38:40] = g["actions"][:E, t]
        a[:, 40] = g["bits"][:E, t]
        assert L.qttt_board_op(rin.data_ptr(), rout.data_ptr(), E, s.cuda_stream) == 0
        s.synchronize()
        o = rout.numpy()
        nm = g["n_moves"][:E, t]
        assert np.array_equal(o[:, 18], nm), t
        assert np.array_equal(o[:, 19:28].view(np.int8), g["board"][:E, t]), t
        assert np.array_equal(o[:, 0:18].reshape(E, 9, 2), g["moves"][:E, t]), t
        assert np.array_equal(o[:, 28], g["n_q"][:E, t]), t
        assert np.array_equal(o[:, 30:38].copy().view("<u2").reshape(E, 4), g["qmask"][:E, t]), t
        assert np.array_equal(o[:, 41], (g["n_moves"][:E, t] == g["n_moves"][:E, t - 1]).astype(np.uint8)), t   # rejected <=> nothing appended
        assert np.array_equal(o[:, 44:48].copy().view("<u4")[:, 0], g["reward"][:E, t].astype(np.float32).view(np.uint32)), t
        assert np.array_equal(o[:, 48], g["terminated"][:E, t]), t
        assert np.array_equal(o[:, 49].view(np.int8), g["p1_round"][:E, t]) and np.array_equal(o[:, 50].view(np.int8), g["p2_round"][:E, t]), t


# ---------------------------------------------------------------------------------------------------
# The reference's MCTS expansion through the façade (mcts.py:9-17, 233-267): a GameState(Board) subclass
# gets its parent's attributes assigned, make_move is called with the bit source, and the result is held
# against the children the reference's own mcts.py produced (tests/golden/expand_traces.npz).
IND2MOVE = [(i, j) for i in range(9) for j in range(i + 1, 9)]          # mcts.py:339-343


def _game_state_class():
    from qtttgym_amd import Board, QEvalClassic

    class GameState(Board):                                             # mcts.py:9-17
        def __init__(self, board, moves, qstructs):
            Board.__init__(self, QEvalClassic())
            self.board = board
            self.moves = moves
            self.qstructs = qstructs
    return GameState


def _parent_attrs(gx, p):
    nm, nq = int(gx["p_n_moves"][p]), int(gx["p_n_q"][p])
    board = [int(x) for x in gx["p_board"][p]]
    moves = [(int(gx["p_moves"][p, i, 0]), int(gx["p_moves"][p, i, 1]), i) for i in range(nm)]
    qstructs = [set(s for s in range(9) if int(gx["p_qmask"][p, k]) >> s & 1) for k in range(nq)]
    return board, moves, qstructs


def _assert_child(gs, gx, k, c):
    nm, nq = int(gx["c_n_moves"][k, c]), int(gx["c_n_q"][k, c])
    assert gs.board == [int(x) for x in gx["c_board"][k, c]], (k, c)
    assert gs.moves == [(int(gx["c_moves"][k, c, i, 0]), int(gx["c_moves"][k, c, i, 1]), i) for i in range(nm)], (k, c)
    assert gs.qstructs == [set(s for s in range(9) if int(gx["c_qmask"][k, c, q]) >> s & 1) for q in range(nq)], (k, c)


@pytest.fixture(scope="module")
def gx():
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with np.load(os.path.join(root, "tests", "golden", "expand_traces.npz")) as d:
        return {k: d[k] for k in d.files}


def test_expand_traces_through_a_gamestate_subclass_one_make_move_at_a_time(gx, bits):
    GameState = _game_state_class()
    picks = range(0, len(gx["action"]), 3)                              # every third pair: 3 120 expansions
    for k in picks:
        p, a, nch = int(gx["parent"][k]), int(gx["action"][k]), int(gx["n_children"][k])
        for c in range(max(nch, 1)):
            gs = GameState(*_parent_attrs(gx, p))
            bits.bit = c                                                # child c = the closing move on min / max
            calls = bits.calls
            if nch == 0:
                with pytest.raises(Exception):
                    gs.make_move(IND2MOVE[a])
                assert (gs.board, gs.moves, gs.qstructs) == _parent_attrs(gx, p)
            else:
                gs.make_move(IND2MOVE[a])
                assert bits.calls - calls == (1 if nch == 2 else 0)    # one draw per collapse (qeval.py:35)
                _assert_child(gs, gx, k, c)


def test_expand_traces_through_board_make_moves_in_one_launch(gx):
    """All 9 360 (parent, action) pairs x both collapse choices as TWO batched calls of Board.make_moves."""
    from qtttgym_amd import Board
    GameState = _game_state_class()
    n = len(gx["action"])
    moves = [IND2MOVE[int(a)] for a in gx["action"]]
    for c in range(2):
        kids = [GameState(*_parent_attrs(gx, int(gx["parent"][k]))) for k in range(n)]
        res = Board.make_moves(kids, moves, bits=[c] * n)
        for k in range(n):
            nch = int(gx["n_children"][k])
            if nch == 0:
                assert isinstance(res[k], Exception) and str(res[k]) == "Move in classical square not allowed", k
                assert (kids[k].board, kids[k].moves, kids[k].qstructs) == _parent_attrs(gx, int(gx["parent"][k]))
            else:
                assert res[k] is None, (k, res[k])
                _assert_child(kids[k], gx, k, c if nch == 2 else 0)


def test_make_moves_draws_like_a_loop_of_make_move_and_reports_each_exception(bits):
    from qtttgym_amd import Board, QEvalClassic
    mk = lambda: Board(QEvalClassic())
    a, b, c, d = mk(), mk(), mk(), mk()
    for x in (a, b):
        x.make_move((0, 1))
    bits.bit = 1
    calls = bits.calls
    res = Board.make_moves([a, b, c, d], [(1, 0), (0, 2), (4, 4), (3, 9)])
    assert bits.calls - calls == 1                                      # only a's move closes a cycle
    assert res[0] is None and a.board[:2] == [0, 1] and a.moves == [(0, 1, 0), (0, 1, 1)] and a.qstructs == []
    assert res[1] is None and b.moves == [(0, 1, 0), (0, 2, 1)] and b.qstructs == [{0, 1, 2}]
    assert str(res[2]) == "Move in same square not allowed when not necessary" and c.moves == []
    assert isinstance(res[3], IndexError) and d.moves == []
    assert a.check_win() == (-1, -1)
    with pytest.raises(ValueError):
        Board.make_moves([a], [(2, 3), (4, 5)])
    assert Board.make_moves([], []) == []


def test_the_binding_stub_of_integration_md_runs_as_written():
    """INTEGRATION.md §2 shows the ctypes binding a maintainer of the reference would add: the code block is taken
    from the document, pointed at the in-tree library and run against the first golden episodes."""
    import os
    import re
    import torch
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# qtttgym/_hip\.py.*?)```", doc, flags=re.S).group(1)
    code = code.replace('"libqttt_hip.so"', repr(os.path.join(root, "qtttgym_amd", "libqttt_hip.so")))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    with np.load(os.path.join(root, "tests", "golden", "step_traces.npz")) as d:
        acts, bits, board_last = d["actions"], d["bits"], d["board"]
        rew, term = d["reward"], d["terminated"]
    E, T = bits.shape
    bb = ns["BatchedBoards"](E)
    for t in range(T):
        r, tm = bb.step(torch.from_numpy(acts[:, t].copy()).cuda(), torch.from_numpy(bits[:, t].copy()).cuda())
        assert np.array_equal(r.cpu().numpy().view(np.uint32), rew[:, t].astype(np.float32).view(np.uint32)), t
        assert np.array_equal(tm.cpu().numpy().astype(np.uint8), term[:, t]), t
