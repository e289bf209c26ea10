"""Size-independent properties of the reference's state (board.py:2-69, 71-115; env.py:49,51), checked with plain
torch tensor arithmetic on EVERY board of batches far beyond what the oracle replays in seconds: 16 777 216 boards
(the working set that does not fit the Infinity Cache) after random plies at mixed depths.  None of this goes
through oracle/ — the checks are restated from the reference's invariants:
  * a collapsed move's round stands on exactly one of its two squares; an un-collapsed move touches no classical
    square; rounds on the board are distinct (board.py:19, 44-56);
  * qstructs are disjoint, cover exactly the squares of the un-collapsed moves, and each is a TREE of them
    (#moves = #squares - 1: a cycle collapses at once, board.py:42);
  * the autofill: nine classical squares iff moves ends with (idx, idx, 8) or nine real moves (board.py:22-25);
  * check_win recomputed from .board (board.py:71-115), reward = -1.0 iff a line, terminated = line or
    len(moves) > 8 (env.py:49, 51), Env.turn = len(moves) (env.py:65-66)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

LINES = [(0, 1, 2), (3, 4, 5), (6, 7, 8), (0, 3, 6), (1, 4, 7), (2, 5, 8), (2, 4, 6), (0, 4, 8)]   # board.py:85-110


def check_win_torch(board):
    """board.py:71-115 for i8[N,9] -> (p1_round, p2_round) i8[N], by the reference's definition."""
    b = board.to(torch.int16)
    mark = torch.where(b < 0, torch.zeros_like(b), torch.where(b % 2 == 0, -torch.ones_like(b), torch.ones_like(b)))
    p1 = torch.full((board.shape[0],), 10, dtype=torch.int16, device=board.device)
    p2 = p1.clone()
    for ln in LINES:
        idx = list(ln)
        ssum = mark[:, idx].sum(dim=1)
        mx = b[:, idx].max(dim=1).values
        p1 = torch.where(ssum == -3, torch.minimum(p1, mx), p1)
        p2 = torch.where(ssum == 3, torch.minimum(p2, mx), p2)
    p1 = torch.where(p1 >= 10, -torch.ones_like(p1), p1)
    p2 = torch.where(p2 >= 10, -torch.ones_like(p2), p2)
    return p1.to(torch.int8), p2.to(torch.int8)


def check_batch(env, reward, terminated):
    n = env.num_envs
    ex = env.export_boards()
    board, moves, nm = ex["board"].to(torch.int16), ex["moves"].to(torch.int16), ex["n_moves"].to(torch.int16)
    qmask, nq = ex["qmask"].to(torch.int32) & 0xFFFF, ex["n_q"].to(torch.int16)
    sq = torch.arange(9, device=board.device)
    rounds = torch.arange(9, device=board.device, dtype=torch.int16)
    used = rounds[None, :] < nm[:, None]                                      # [N,9] move t exists
    lo, hi = moves[:, :, 0], moves[:, :, 1]
    assert bool(((lo == 255) == ~used).all()) and bool(((hi == 255) == ~used).all())   # 255-padding exactly beyond n_moves
    assert bool((nm <= 9).all()) and bool((lo[used] <= hi[used]).all()) and bool((hi[used] <= 8).all())
    classical = board >= 0
    ncl = classical.sum(dim=1)
    # rounds on the board: distinct, each the round of an existing move, standing on one of that move's squares
    onehot = (board[:, :, None] == rounds[None, None, :])                     # [N, square, round]
    per_round = onehot.sum(dim=1)                                             # how many squares carry round t
    assert bool((per_round <= 1).all()) and bool((per_round.bool() <= used).all())
    collapsed = per_round.bool()                                              # move t has collapsed
    landing = (onehot * sq.to(torch.int16)[None, :, None]).sum(dim=1)        # its square
    lo_c, hi_c = lo.clamp(max=8), hi.clamp(max=8)
    assert bool(((landing == lo_c) | (landing == hi_c))[collapsed].all())
    # un-collapsed moves touch no classical square; autofill moves (lo == hi) are always collapsed and come last
    live = used & ~collapsed
    cl_lo = torch.gather(classical, 1, lo_c.long())
    cl_hi = torch.gather(classical, 1, hi_c.long())
    assert not bool((live & (cl_lo | cl_hi)).any())
    auto = used & (lo == hi)
    assert bool((auto <= collapsed).all()) and bool((auto.sum(dim=1) <= 1).all())
    assert bool((auto <= (rounds[None, :] == (nm[:, None] - 1))).all())
    assert bool(((ncl == 9) == ((nm == 9))).all())                           # nine classical <=> nine moves (autofill incl.)
    assert not bool((ncl == 8).any()) and not bool((ncl == 1).any())          # eight is autofilled at once; one cannot happen
    # qstructs: disjoint, exactly the squares of the live moves, every one a tree of them
    bits = ((qmask[:, :, None] >> sq[None, None, :]) & 1).to(torch.int16)     # [N, slot, square]
    assert bool((bits.sum(dim=1) <= 1).all())
    slot_used = torch.arange(4, device=board.device)[None, :] < nq[:, None]
    assert bool(((qmask != 0) == slot_used).all())                           # compact, in list order
    live_sq = torch.zeros_like(classical)
    for t in range(9):
        m = live[:, t]
        live_sq |= m[:, None] & ((sq[None, :] == lo_c[:, t:t + 1]) | (sq[None, :] == hi_c[:, t:t + 1]))
    assert bool((bits.sum(dim=1).bool() == live_sq).all())
    lo_slot = torch.gather(bits, 2, lo_c.long()[:, None, :].expand(-1, 4, -1))   # [N, slot, move]: is lo in the slot
    hi_slot = torch.gather(bits, 2, hi_c.long()[:, None, :].expand(-1, 4, -1))
    assert bool(((lo_slot == hi_slot) | ~live[:, None, :]).all())             # both ends of a live move share a slot
    edges = (lo_slot.bool() & live[:, None, :]).sum(dim=2)
    assert bool(((edges == bits.sum(dim=2) - 1) | ~slot_used).all())          # a tree: #moves = #squares - 1
    # check_win, reward, terminated, turn
    p1, p2 = env.check_win()
    w1, w2 = check_win_torch(ex["board"])
    assert torch.equal(p1, w1) and torch.equal(p2, w2)
    line = (w1 > 0) | (w2 > 0)
    minus_one = torch.tensor(-1082130432, dtype=torch.int32, device=board.device)     # bits of -1.0f
    minus_zero = torch.tensor(-2147483648, dtype=torch.int32, device=board.device)    # bits of -0.0f (env.py:49)
    assert torch.equal(reward.view(torch.int32), torch.where(line, minus_one, minus_zero))
    assert torch.equal(terminated, line | (nm > 8))
    assert torch.equal(env.turn().to(torch.int16), nm)
    return int(line.sum()), int((nm == 9).sum()), int(live.sum())


@pytest.mark.parametrize("n", [1 << 20, 1 << 24])
def test_reference_invariants_hold_on_every_board(n):
    from qtttgym_amd import VecEnv
    env = VecEnv(n, seed=n, auto_reset=False)
    # boards frozen at mixed depths: ply t is played by the boards whose depth allows it (the others get a noop)
    depth = (torch.arange(n, device="cuda") * 2654435761 % 11).to(torch.uint8)        # 0..10 plies
    a = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
    r = tm = None
    for t in range(10):
        env.sample_actions(out=a)
        a[depth <= t] = 0                                                            # (0, 0): a noop (board.py:10-12)
        r, tm = env.step_raw(a)
    with_line, full, live = check_batch(env, r, tm)
    assert with_line > n // 10 and full > n // 50 and live > n // 4                  # every regime is present in bulk
    del depth, a
    # and after a long auto-reset run of the fused random-policy kernel (every board mid-episode somewhere)
    env2 = VecEnv(n, seed=n + 1, auto_reset=True)
    r2, t2 = env2.step_random_many(37)
    check_batch(env2, r2, t2)


# ---------------------------------------------------------------------------------------------------
# The MCTS-side rows at full size, against restatements in torch of what the reference computes (no oracle):
# GameState.update_winner / actions / __hash__ (mcts.py:20-27, 52-65, 93-94; CPython's tuplehash in wrapping
# int64 arithmetic), and qttt_expand against qttt_step on copies of the parents with the collapse bit forced.
_M64 = (1 << 64) - 1


def _s64(x):
    x &= _M64
    return x - (1 << 64) if x >> 63 else x


_P1, _P2, _P5 = _s64(11400714785074694791), _s64(14029467366897019727), _s64(2870177450012600261)


def _lane(acc, h):
    acc = acc + h * _P2
    acc = (acc << 31) | ((acc >> 33) & 0x7FFFFFFF)                      # rotl 31 (>> is arithmetic: mask the sign copies)
    return acc * _P1


def _fin(acc, length):
    acc = acc + (length ^ (_P5 ^ 3527539))
    return torch.where(acc == -1, torch.full_like(acc, 1546275796), acc)


def python_hash_torch(board, moves, n_moves):
    """hash(tuple(board) + tuple(moves)) of CPython >= 3.8 for i8[N,9] boards and (lo, hi, t) move tuples."""
    n = board.shape[0]
    acc = torch.full((n,), _P5, dtype=torch.int64, device=board.device)
    b = board.to(torch.int64)
    for v in range(9):
        acc = _lane(acc, torch.where(b[:, v] == -1, torch.full_like(acc, -2), b[:, v]))        # hash(-1) == -2
    mv = moves.to(torch.int64)
    for t in range(9):
        inner = torch.full((n,), _P5, dtype=torch.int64, device=board.device)
        for x in (mv[:, t, 0], mv[:, t, 1], torch.full_like(acc, t)):
            inner = _lane(inner, x)
        inner = _fin(inner, 3)
        acc = torch.where(n_moves.to(torch.int64) > t, _lane(acc, inner), acc)
    return _fin(acc, 9 + n_moves.to(torch.int64))


def node_info_torch(ex):
    board = ex["board"]
    p1, p2 = check_win_torch(board)
    both, only1, only2 = (p1 > 0) & (p2 > 0), (p1 > 0) & (p2 < 0), (p1 < 0) & (p2 > 0)
    winner = torch.full_like(p1, -1)
    winner = torch.where(both, (p1 < p2).to(torch.int8), winner)                               # mcts.py:54-56
    winner = torch.where(only1, torch.ones_like(winner), winner)
    winner = torch.where(only2, torch.zeros_like(winner), winner)
    terminal = (ex["n_moves"] == 9) | both | only1 | only2                                      # mcts.py:65
    empty = board < 0
    legal = torch.zeros(board.shape[0], dtype=torch.int64, device=board.device)
    a = 0
    for i in range(9):
        for j in range(i + 1, 9):                                                               # mcts.py:20-27 in ind2move order
            legal |= (empty[:, i] & empty[:, j]).to(torch.int64) << a
            a += 1
    return winner, terminal, legal, python_hash_torch(board, ex["moves"], ex["n_moves"])


def test_node_info_and_expand_at_one_million_boards_against_torch_restatements():
    from qtttgym_amd import VecEnv
    from qtttgym_amd.actions import action36_to_pairs
    n = 1 << 20
    env = VecEnv(n, seed=77)
    depth = (torch.arange(n, device="cuda") * 2654435761 % 11).to(torch.uint8)
    a = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
    for t in range(10):
        env.sample_actions(out=a)
        a[depth <= t] = 0
        env.step_raw(a)
    info = env.node_info()
    w, tm, lg, ky = node_info_torch(env.export_boards())
    assert torch.equal(info["winner"], w) and torch.equal(info["terminal"], tm)
    assert torch.equal(info["legal"], lg) and torch.equal(info["key"], ky)
    assert int((w >= 0).sum()) > n // 10 and int((ky < 0).sum()) > n // 4                      # signed keys, decided games present
    # expand == two steps of a copy with the collapse bit forced, for every (state, action) pair
    act = torch.randint(0, 40, (n,), dtype=torch.uint8, device="cuda")                         # 36..39: not an action
    out = env.expand(act)
    nch = out["n_children"]
    pairs = action36_to_pairs(act)
    planes = lambda st: st.view(torch.int64).view(2, -1)[:, :n]
    for bit in range(2):
        cp = VecEnv.from_state(env.state.clone(), n)
        cp.step_raw(pairs, torch.full((n,), bit, dtype=torch.uint8, device="cuda"))
        child = out["child%d" % bit]
        sel = nch > bit
        assert torch.equal(planes(cp.state)[:, sel], planes(child.state)[:, sel]), bit
        if bit == 0:                                                                            # no children: copies of the parent
            assert torch.equal(planes(env.state)[:, nch == 0], planes(child.state)[:, nch == 0])
            moved = planes(cp.state)[0] != planes(env.state)[0]
            assert torch.equal(moved, nch > 0)                                                  # n_children == 0 <=> make_move raised
        cw, ct, cl, ck = node_info_torch(child.export_boards())
        assert torch.equal(out["winner"][:, bit][sel], cw[sel]) and torch.equal(out["terminal"][:, bit][sel], ct[sel])
        assert torch.equal(out["legal"][:, bit][sel], cl[sel]) and torch.equal(out["key"][:, bit][sel], ck[sel])
        assert bool((out["key"][:, bit][~sel] == 0).all()) and bool((out["winner"][:, bit][~sel] == -1).all())
    two = nch == 2                                                                              # a collapse: the children differ
    assert int(two.sum()) > n // 20
    assert bool((planes(out["child0"].state)[0][two] != planes(out["child1"].state)[0][two]).all())


def observation_torch(ex):
    """Env._observation (env.py:68-85) from Board.moves / .board: the un-collapsed moves (round not on the board) of
    even / odd round, in move order, as 255-padded (lo, hi) pairs + lengths; classical = the board; turn = len % 2."""
    board, moves, nm = ex["board"], ex["moves"], ex["n_moves"].to(torch.int64)
    n, dev = board.shape[0], board.device
    rounds = torch.arange(9, device=dev)
    used = rounds[None, :] < nm[:, None]
    on_board = (board.to(torch.int64)[:, :, None] == rounds[None, None, :]).any(dim=1)          # env.py:72-74
    live = used & ~on_board
    out = []
    for parity, width in ((0, 5), (1, 4)):
        sel = live & ((rounds % 2) == parity)[None, :]
        pos = torch.zeros_like(sel, dtype=torch.int64)                                          # rank among the selected, move order
        run = torch.zeros(n, dtype=torch.int64, device=dev)                                     # (a column loop: torch's cumsum refuses 16 M rows)
        for t in range(9):
            pos[:, t] = run
            run = run + sel[:, t].to(torch.int64)
        q = torch.full((n, width + 1, 2), 255, dtype=torch.uint8, device=dev)                   # one spare row swallows the rest
        idx = torch.where(sel, pos, torch.full_like(pos, width))
        q.scatter_(1, idx[:, :, None].expand(-1, -1, 2), moves)
        out += [q[:, :width].contiguous(), sel.sum(dim=1).to(torch.uint8)]
    return out[0], out[1], out[2], out[3], board, (nm % 2).to(torch.uint8)


def to_vector_torch(ex):
    """GameState.to_vector (mcts.py:67-85) -> f32[N,18,10] and action_mask (mcts.py:87-91) -> bool[N,36]."""
    board, moves, nm = ex["board"].to(torch.int64), ex["moves"].to(torch.int64), ex["n_moves"].to(torch.int64)
    n, dev = board.shape[0], board.device
    vec = torch.zeros((n, 18, 10), dtype=torch.float32, device=dev)
    col = torch.where(board < 0, torch.full_like(board, 9), board)                              # board[i] == -1 indexes column 9
    vec[:, :9].scatter_(2, col[:, :, None], 1.0)
    third = torch.tensor(1.0 / 3.0, dtype=torch.float64).to(torch.float32).item()               # 1/math.sqrt(9), rounded to f32
    bidx = torch.arange(n, device=dev)
    for t in range(9):
        m = nm > t
        for e in range(2):
            s = moves[:, t, e].clamp(max=8)
            cur = vec[bidx, 9 + s, t]
            vec[bidx, 9 + s, t] = torch.where(m, torch.full_like(cur, third), cur)
    qs = ex["qmask"].to(torch.int64) & 0xFFFF
    in_q = ((qs[:, :, None] >> torch.arange(9, device=dev)[None, None, :]) & 1).any(dim=1)
    vec[:, 9:, 9] = torch.where(in_q, vec[:, 9:, 9], torch.ones_like(vec[:, 9:, 9]))
    empty = board < 0
    mask = torch.stack([empty[:, i] & empty[:, j] for i in range(9) for j in range(i + 1, 9)], dim=1)
    return vec, mask


@pytest.mark.parametrize("n", [1 << 20, 1 << 24])
def test_observation_of_the_fused_step_on_every_board(n):
    """a4 at full size: the observation VecEnv.step() returns (written by the step kernel from its registers) and
    qttt_observe, against env.py:68-85 restated on the exported Board attributes."""
    from qtttgym_amd import VecEnv
    env = VecEnv(n, seed=5, auto_reset=True)
    env.step_random_many(23)                                                                    # every board somewhere mid-episode
    obs, r, tm = env.step_observe_raw(env.sample_actions())
    want = observation_torch(env.export_boards())
    names = ("q_states_p1", "q_states_p1_len", "q_states_p2", "q_states_p2_len", "classical", "turn")
    for k, w in zip(names, want):
        assert torch.equal(obs[k], w), k
    again = {k: v.clone() for k, v in obs.items()}
    env.observ()                                                                                # the stand-alone kernel, same buffers
    for k in names:
        assert torch.equal(obs[k], again[k]), k
    assert int(obs["q_states_p1_len"].max()) >= 4 and int(obs["q_states_p2_len"].max()) == 4


def test_encode_at_one_million_boards_against_a_torch_restatement():
    from qtttgym_amd import VecEnv
    n = 1 << 20
    env = VecEnv(n, seed=6, auto_reset=False)
    depth = (torch.arange(n, device="cuda") * 2654435761 % 11).to(torch.uint8)
    a = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
    for t in range(10):
        env.sample_actions(out=a)
        a[depth <= t] = 0
        env.step_raw(a)
    vec, mask = env.encode()
    wv, wm = to_vector_torch(env.export_boards())
    assert torch.equal(mask, wm)
    assert torch.equal(vec, wv)


def test_rollout_at_one_million_boards_equals_ply_by_ply_play():
    """MCTS._simulate (mcts.py:185-198): the fused playout's (result, plies, final state) against the same policy
    played one launch per ply on a copy, where every board is looked at the moment it first terminates."""
    from qtttgym_amd import VecEnv
    n, s0 = 1 << 20, 40
    env = VecEnv(n, seed=9)
    depth = (torch.arange(n, device="cuda") * 2654435761 % 7).to(torch.uint8)                  # parents at depths 0..6
    a = torch.empty((n, 2), dtype=torch.uint8, device="cuda")
    for t in range(6):
        env.sample_actions(out=a)
        a[depth <= t] = 0
        env.step_raw(a)
    result, plies, final = env.rollout(step_idx0=s0, return_final=True)
    cp = VecEnv.from_state(env.state.clone(), n, seed=9)
    cp.step_idx = s0
    info = cp.node_info()
    done = info["terminal"].clone()                                                             # terminal parents: zero plies
    want_res = torch.where(info["winner"] < 0, torch.zeros_like(info["winner"]), info["winner"] * 2 - 1)
    want_plies = torch.zeros(n, dtype=torch.uint8, device="cuda")
    planes = lambda st: st.view(torch.int64).view(2, -1)[:, :n]
    want_final = planes(cp.state).clone()
    for p in range(9):
        cp.step_random()                                                                        # finished boards play on (or noop): ignored below
        info = cp.node_info(out=info)
        newly = info["terminal"] & ~done
        alive = ~done
        want_plies = torch.where(alive, torch.full_like(want_plies, p + 1), want_plies)
        res = torch.where(info["winner"] < 0, torch.zeros_like(info["winner"]), info["winner"] * 2 - 1)   # mcts.py:200-209
        want_res = torch.where(alive, res, want_res)
        want_final = torch.where(alive[None, :], planes(cp.state), want_final)
        done |= newly
    assert bool(done.all())
    assert torch.equal(plies, want_plies) and torch.equal(result, want_res)
    assert torch.equal(planes(final.state), want_final)
    assert int((plies == 0).sum()) > 0 and int(plies.max()) == 9 and int((result == 0).sum()) > n // 20
