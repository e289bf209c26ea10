"""Live fuzz of the C oracle against the UNMODIFIED reference imported from /root/reference
(build container only; skipped on the GPU box where the reference does not exist).  Fresh
random episodes every run-independent seed list, far more than the committed fixtures hold."""
import random

import numpy as np
import pytest

import oracle
from ref_shim import load_reference, reference_available

pytestmark = pytest.mark.skipif(not reference_available(), reason="reference not present")


def _play(qtttgym, src, rng, n_eps, T, adversarial):
    acts = np.zeros((n_eps, T, 2), dtype=np.uint8)
    bits = np.zeros((n_eps, T), dtype=np.uint8)
    exp = {k: [] for k in ("board", "n_moves", "moves", "qmask", "n_q", "reward", "term", "p1", "p2")}
    for e in range(n_eps):
        env = qtttgym.Env()
        env.reset()
        rows = {k: [] for k in exp}
        for t in range(T):
            board = env._gameboard.board
            empty = [i for i in range(9) if board[i] == -1]
            if adversarial and rng.random() < 0.4 or len(empty) < 2:
                a, b = rng.randrange(0, 12), rng.randrange(0, 12)
            else:
                a, b = rng.sample(empty, 2)
            bit = rng.getrandbits(1)
            src.bit = bit
            obs, r, term, trunc, info = env.step((a, b))
            acts[e, t] = (a, b)
            bits[e, t] = bit
            gb = env._gameboard
            rows["board"].append(list(gb.board))
            rows["n_moves"].append(len(gb.moves))
            mv = [[255, 255]] * 9
            for i, m in enumerate(gb.moves):
                mv[i] = [m[0], m[1]]
            rows["moves"].append(mv)
            qm = [0] * 4
            for i, s in enumerate(gb.qstructs):
                qm[i] = sum(1 << x for x in s)
            rows["qmask"].append(qm)
            rows["n_q"].append(len(gb.qstructs))
            rows["reward"].append(r)
            rows["term"].append(bool(term))
            p1, p2 = gb.check_win()
            rows["p1"].append(p1)
            rows["p2"].append(p2)
        for k in exp:
            exp[k].append(rows[k])
    return acts, bits, {k: np.array(v) for k, v in exp.items()}


@pytest.mark.parametrize("adversarial", [False, True])
def test_oracle_vs_reference_live(adversarial):
    qtttgym, src = load_reference()
    rng = random.Random(9001 + adversarial)
    n_eps, T = 3000, 12
    acts, bits, exp = _play(qtttgym, src, rng, n_eps, T, adversarial)
    ob = oracle.OracleBoards(n_eps)
    for t in range(T):
        reward, term = ob.step(acts[:, t], bits[:, t])
        assert np.array_equal(ob.board, exp["board"][:, t].astype(np.int8))
        assert np.array_equal(ob.n_moves, exp["n_moves"][:, t].astype(np.uint8))
        assert np.array_equal(ob.moves, exp["moves"][:, t].astype(np.uint8))
        assert np.array_equal(ob.qmask, exp["qmask"][:, t].astype(np.uint16))
        assert np.array_equal(ob.n_q, exp["n_q"][:, t].astype(np.uint8))
        assert np.array_equal(reward.view(np.uint32),
                              exp["reward"][:, t].astype(np.float32).view(np.uint32))
        assert np.array_equal(term, exp["term"][:, t].astype(np.uint8))
        p1, p2 = ob.check_win()
        assert np.array_equal(p1, exp["p1"][:, t].astype(np.int8))
        assert np.array_equal(p2, exp["p2"][:, t].astype(np.int8))


def test_reference_draws_one_bit_per_collapse_only():
    """random.choice is called exactly once per collapse with (lo, hi)."""
    qtttgym, src = load_reference()
    rng = random.Random(5)
    for _ in range(300):
        env = qtttgym.Env()
        env.reset()
        for t in range(10):
            board = env._gameboard.board
            empty = [i for i in range(9) if board[i] == -1]
            if len(empty) < 2:
                break
            before = list(board)
            calls = src.calls
            env.step(tuple(rng.sample(empty, 2)))
            collapsed = before != env._gameboard.board
            assert src.calls - calls == (1 if collapsed else 0)


def test_host_side_mirrors_against_the_reference_live(capsys):
    """The pieces of the host mirror that are plain Python (no device work) against the imported reference:
    `displayBoard` prints what display.py:4-32 prints for the same attributes, `Env._reward` is env.py:87-112 for every
    check_win pair, and `ind2move` / `move2ind` (qtttgym_amd.actions) are mcts.py:339-350's tables."""
    import sys
    from ref_shim import REFERENCE_ROOT
    qtttgym, src = load_reference()
    from qtttgym_amd.board import displayBoard
    from qtttgym_amd.actions import ind2move, move2ind
    rng = random.Random(77)

    class Attrs:                                  # the duck type both renderers read: .moves and .board
        pass
    for _ in range(200):
        gb = qtttgym.Board(qtttgym.QEvalClassic())
        for _ in range(rng.randrange(0, 10)):
            empty = [i for i in range(9) if gb.board[i] == -1]
            if len(empty) < 2:
                break
            src.bit = rng.getrandbits(1)
            gb.make_move(tuple(rng.sample(empty, 2)))
        qtttgym.displayBoard(gb)
        want = capsys.readouterr().out
        mine = Attrs()
        mine.moves, mine.board = list(gb.moves), list(gb.board)
        displayBoard(mine)
        assert capsys.readouterr().out == want
    # Env._reward (env.py:87-112, a helper step() does not call): same value for every check_win pair
    from qtttgym_amd.env import Env as MyEnv

    class FixedBoard:
        def __init__(self, pair):
            self.pair = pair

        def check_win(self):
            return self.pair
    ref_env = qtttgym.Env()
    for p1 in range(-1, 10):
        for p2 in range(-1, 10):
            mine_env, ref_env._gameboard = Attrs(), FixedBoard((p1, p2))
            mine_env._gameboard, mine_env._reward_map = FixedBoard((p1, p2)), dict(ref_env._reward_map)
            got, want = MyEnv._reward(mine_env), ref_env._reward()
            assert type(got) is type(want) and got == want, (p1, p2, got, want)
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import mcts as ref_mcts
    for a in range(36):
        assert tuple(ind2move(a)) == tuple(ref_mcts.ind2move(a))
        i, j = ref_mcts.ind2move(a)
        assert move2ind(i, j) == ref_mcts.move2ind(i, j) == a and move2ind(j, i) == ref_mcts.move2ind(j, i)


def test_oracle_expand_equals_the_reference_step_exhaustively_to_depth_three():
    """Exhaustive link of the pin chain on the reference's side: from the empty board, every action at every position to
    depth 2 is run through the reference's own MCTS._step (unmodified mcts.py) — (1 + 36 + 1 332) x 36 = 49 284 calls —
    and through the oracle's qo_expand; the children (both collapse branches), their winner / terminal / legal
    actions and Python hash are the same, position by position.  (tests/test_rows_tiles_and_keys_gpu.py then holds the HIP side
    against the oracle over every position to depth 4.)"""
    import sys
    from ref_shim import REFERENCE_ROOT
    qtttgym, _ = load_reference()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import mcts as ref_mcts

    class Toggle:                                         # _step re-runs make_move until the other branch shows up
        def __init__(self):
            self.bit = 0

        def choice(self, seq):
            out = seq[self.bit]
            self.bit ^= 1
            return out
    tog = Toggle()
    saved = qtttgym.qeval.random
    qtttgym.qeval.random = tog
    try:
        strat = ref_mcts.MCTS(rollouts=1, num_simulations=1)
        GS = ref_mcts.MCTS.GameState
        frontier = [GS([-1] * 9, [], True, None, False)]
        ob_frontier = oracle.OracleBoards(1)
        wcode = {True: 1, False: 0, None: -1}
        calls = 0
        for depth in range(3):
            n = len(frontier)
            ob_rep = oracle.OracleBoards(n * 36)
            ob_rep.b[:] = np.repeat(ob_frontier.b, 36)
            nch, kids, winner, terminal, legal, key = oracle.expand(ob_rep, np.tile(np.arange(36, dtype=np.uint8), n))
            nxt, order = [], []
            for i, gs in enumerate(frontier):
                for a in range(36):
                    lo, hi = ref_mcts.ind2move(a)
                    tog.bit = 0
                    try:
                        ch = strat._step(gs, a)
                    except Exception:
                        ch = []
                    calls += 1
                    row = i * 36 + a
                    assert len(ch) == nch[row], (depth, i, a)
                    if len(ch) == 2:
                        r = len(gs.moves)
                        ch.sort(key=lambda k: 0 if k.board[lo] == r else 1)
                    for c, k in enumerate(ch):
                        assert list(kids[c].board[row]) == list(k.board)
                        assert [tuple(m[:2]) for m in k.moves] == [tuple(int(x) for x in kids[c].b["moves"][row][j])
                                                                   for j in range(len(k.moves))]
                        assert wcode[k.winner] == winner[row, c] and bool(k.terminal) == bool(terminal[row, c])
                        assert hash(k) == key[row, c]
                        assert sum(1 << x for x in k.actions) == int(legal[row, c])
                        nxt.append(k)
                        order.append((c, row))
            # the oracle's frontier in the same (generation) order as `nxt`
            ob_next = oracle.OracleBoards(len(nxt))
            for j, (c, row) in enumerate(order):
                ob_next.b[j] = kids[c].b[row]
            frontier, ob_frontier = nxt, ob_next
        assert calls == 1 * 36 + 36 * 36 + 1332 * 36 and len(frontier) == 49896
    finally:
        qtttgym.qeval.random = saved


def test_oracle_step_equals_the_reference_env_step_exhaustively_to_depth_two():
    """Every position reachable in <= 2 plies (1 + 36 + 1 332) x every action of {0..9}^2 plus two with a square of 255 x
    both collapse bits = 279 276 calls of the reference's own Env.step (same-square, classical-square and IndexError
    noops included) against qo_step: board, moves, qstructs, reward bits (-0.0 / -1.0), terminated, check_win."""
    import copy
    qtttgym, src = load_reference()
    positions = []

    def grow(gb, depth):
        positions.append(gb)
        if depth == 2:
            return
        for a in range(9):
            for b in range(a + 1, 9):
                if gb.board[a] != -1 or gb.board[b] != -1:
                    continue
                for bit in (0, 1):
                    src.bit = bit
                    calls = src.calls
                    k = copy.deepcopy(gb)
                    k.make_move((a, b))
                    if src.calls == calls and bit == 1:
                        continue                          # no collapse: the bit was not consumed, one child only
                    grow(k, depth + 1)
    grow(qtttgym.Board(qtttgym.QEvalClassic()), 0)
    assert len(positions) == 1 + 36 + 1332
    n = len(positions)
    mv = np.full((n, 9, 2), 255, dtype=np.uint8)
    bd = np.zeros((n, 9), dtype=np.int8)
    qm = np.zeros((n, 4), dtype=np.uint16)
    nm = np.zeros(n, dtype=np.uint8)
    nq = np.zeros(n, dtype=np.uint8)
    for i, gb in enumerate(positions):
        for t, m in enumerate(gb.moves):
            mv[i, t] = m[:2]
        bd[i], nm[i], nq[i] = gb.board, len(gb.moves), len(gb.qstructs)
        for k, s in enumerate(gb.qstructs):
            qm[i, k] = sum(1 << x for x in s)
    ob0 = oracle.boards_from_arrays(bd, mv, nm, qm, nq)
    actions = [(a, b) for a in range(10) for b in range(10)] + [(255, 0), (3, 255)]
    env = qtttgym.Env()
    for a, b in actions:
        for bit in (0, 1):
            ob = ob0.copy()
            r_o, t_o = ob.step(np.tile(np.array([a, b], dtype=np.uint8), (n, 1)), np.full(n, bit, dtype=np.uint8))
            p1_o, p2_o = ob.check_win()
            for i, gb in enumerate(positions):
                g = qtttgym.Board(gb.qeval)                # (a hand copy: deepcopy is most of this test's time)
                g.board, g.moves, g.qstructs = list(gb.board), list(gb.moves), [set(q) for q in gb.qstructs]
                env._gameboard = g
                src.bit = bit
                _, r, term, trunc, _ = env.step((a, b))
                g = env._gameboard
                assert list(ob.board[i]) == list(g.board), (a, b, bit, i)
                assert int(ob.n_moves[i]) == len(g.moves), (a, b, bit, i)
                assert [tuple(int(x) for x in ob.b["moves"][i][j]) for j in range(len(g.moves))] == [tuple(m[:2]) for m in g.moves]
                assert [int(x) for x in ob.qmask[i][:len(g.qstructs)]] == [sum(1 << x for x in s) for s in g.qstructs], (a, b, bit, i)
                assert np.float32(r).view(np.uint32) == r_o[i].view(np.uint32) and bool(term) == bool(t_o[i]) and trunc is False
                assert g.check_win() == (int(p1_o[i]), int(p2_o[i]))
