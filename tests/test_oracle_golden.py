"""T1: the C oracle (oracle/qttt_oracle.c) against the golden traces recorded from the
unmodified reference (tests/golden/make_golden.py).  Bit-exact on every reference-visible
quantity, reward compared as IEEE bits so that -0.0 (env.py:49) is pinned."""
import numpy as np

import oracle


def replay(golden):
    acts, bits = golden["actions"], golden["bits"]
    E, T = bits.shape
    ob = oracle.OracleBoards(E)
    for t in range(T):
        reward, term = ob.step(acts[:, t], bits[:, t])
        yield t, ob, reward, term


def test_oracle_matches_golden_state_and_outputs(golden):
    for t, ob, reward, term in replay(golden):
        assert np.array_equal(ob.board, golden["board"][:, t]), t
        assert np.array_equal(ob.n_moves, golden["n_moves"][:, t]), t
        assert np.array_equal(ob.moves, golden["moves"][:, t]), t
        assert np.array_equal(ob.n_q, golden["n_q"][:, t]), t
        assert np.array_equal(ob.qmask, golden["qmask"][:, t]), t
        want = golden["reward"][:, t].astype(np.float32)
        assert np.array_equal(reward.view(np.uint32), want.view(np.uint32)), t
        assert np.array_equal(term, golden["terminated"][:, t]), t
        p1, p2 = ob.check_win()
        assert np.array_equal(p1, golden["p1_round"][:, t]), t
        assert np.array_equal(p2, golden["p2_round"][:, t]), t


def test_oracle_matches_golden_observation(golden):
    for t, ob, _, _ in replay(golden):
        classical, q1, l1, q2, l2, turn = ob.observe()
        assert np.array_equal(classical, golden["board"][:, t])
        assert np.array_equal(q1, golden["q_p1"][:, t])
        assert np.array_equal(l1, golden["q_p1_len"][:, t])
        assert np.array_equal(q2, golden["q_p2"][:, t])
        assert np.array_equal(l2, golden["q_p2_len"][:, t])
        assert np.array_equal(turn, golden["turn"][:, t])


def test_reward_is_negative_zero_or_minus_one(golden):
    r = golden["reward"]
    assert np.signbit(r).all()
    assert set(np.unique(np.abs(r))) <= {0.0, 1.0}
    bits = r.astype(np.float32).view(np.uint32)
    assert set(np.unique(bits)) <= {0x80000000, 0xBF800000}


def test_known_answer_traces(golden):
    """SURVEY.md Appendix A, restated independently of the .npz outputs."""
    kind = list(golden["kind"])
    k4 = kind.index("K4")
    assert golden["board"][k4, 6].tolist() == [-1, 0, 2, -1, 6, 3, 1, 4, 5]
    assert golden["terminated"][k4, 6] == 1 and golden["reward"][k4, 6] == -1.0
    assert (golden["p1_round"][k4, 6], golden["p2_round"][k4, 6]) == (6, -1)
    k6 = kind.index("K6")
    assert golden["board"][k6, 7].tolist() == [3, 5, 4, 8, 2, 7, 1, 6, 0]
    assert golden["moves"][k6, 7, 8].tolist() == [3, 3] and golden["n_moves"][k6, 7] == 9
    k5 = kind.index("K5")
    assert (golden["p1_round"][k5, 8], golden["p2_round"][k5, 8]) == (8, 7)
    k3 = kind.index("K3")
    assert golden["n_moves"][k3, 0] == 0 and golden["n_moves"][k3, 4] == 3
    assert golden["board"][k3, 5].tolist() == [0, 1, 2, 3, -1, -1, -1, -1, -1]
    k1, k2 = kind.index("K1"), kind.index("K2")
    assert golden["board"][k1, 1, :2].tolist() == [1, 0]
    assert golden["board"][k2, 1, :2].tolist() == [0, 1]


def test_oracle_state_invariants_under_random_play():
    """SURVEY.md §4 invariants, on 20 000 fresh random episodes of the oracle: rounds on the board
    are distinct, qstructs are disjoint node sets of the forest of un-collapsed moves (<= 4 of
    them, each >= 2 squares), <= 9 moves, autofill round is always 8, reward in {-0.0, -1.0}."""
    n, seed = 20000, 77
    ob = oracle.OracleBoards(n)
    rng = np.random.default_rng(seed)
    for t in range(10):
        a = ob.sample_actions(seed, t)
        reward, term = ob.step(a, rng.integers(0, 2, n).astype(np.uint8))
        assert set(np.unique(reward.view(np.uint32))) <= {0x80000000, 0xBF800000}
        board, nm, q, nq = ob.board, ob.b["n_moves"], ob.qmask, ob.b["n_q"]
        assert nm.max() <= 9 and nq.max() <= 4
        for i in range(0, n, 97):
            rounds = [int(r) for r in board[i] if r >= 0]
            assert len(rounds) == len(set(rounds))
            live = [(int(ob.b["moves"][i][j][0]), int(ob.b["moves"][i][j][1])) for j in range(int(nm[i]))
                    if j not in rounds]
            nodes = set(x for m in live for x in m)
            masks = [int(q[i][k]) for k in range(int(nq[i]))]
            union = 0
            for m in masks:
                assert bin(m).count("1") >= 2 and (union & m) == 0
                union |= m
            assert union == sum(1 << x for x in nodes)
            assert len(live) == len(nodes) - len(masks)            # a forest: edges = nodes - trees
            for j in range(int(nm[i])):
                lo, hi = int(ob.b["moves"][i][j][0]), int(ob.b["moves"][i][j][1])
                if lo == hi:
                    assert j == 8 and int(board[i][lo]) == 8       # autofill is always round 8


def test_pure_python_restatement_matches_golden(golden):
    from oracle.py_env import PyEnv
    acts, bits = golden["actions"], golden["bits"]
    for e in range(0, acts.shape[0], 3):
        env = PyEnv()
        for t in range(acts.shape[1]):
            r, term = env.step(int(acts[e, t, 0]), int(acts[e, t, 1]), int(bits[e, t]))
            assert env.b.board == golden["board"][e, t].tolist()
            assert len(env.b.moves) == int(golden["n_moves"][e, t])
            assert np.float64(r).view(np.uint64) == golden["reward"][e, t].view(np.uint64)
            assert term == bool(golden["terminated"][e, t])
    # ... and the form that returns what the reference's Env.step returns, observation included (env.py:46,68-85)
    for e in range(1, acts.shape[0], 7):
        env = PyEnv()
        for t in range(acts.shape[1]):
            obs, r, term, trunc, info = env.step_full(int(acts[e, t, 0]), int(acts[e, t, 1]), int(bits[e, t]))
            assert np.float64(r).view(np.uint64) == golden["reward"][e, t].view(np.uint64)
            assert term == bool(golden["terminated"][e, t]) and trunc is False and info == {}
            assert obs["classical"] == golden["board"][e, t].tolist() and obs["turn"] == int(golden["turn"][e, t])
            for key, g, ln in (("q_states_p1", "q_p1", "q_p1_len"), ("q_states_p2", "q_p2", "q_p2_len")):
                k = int(golden[ln][e, t])
                assert [list(p) for p in obs[key]] == golden[g][e, t, :k].tolist(), (e, t, key)
