/* TEST INFRASTRUCTURE — NOT PRODUCT CODE.  See qttt_oracle.h.
 *
 * Every function restates one reference function (Oxel40/qtttgym @ v1) and cites it.
 * Python sets over squares 0..8 become 9-bit masks; sets of moves become masks over the
 * index of the move in the list handed to eval().  Nothing here is shared with the HIP
 * kernels: the kernels use a different formulation (rooted forest + cached component
 * masks), so agreement between the two is evidence, not tautology.
 */
#include "qttt_oracle.h"
#include <string.h>

/* ------------------------------------------------------------------ board.py:2-7 */
void qo_init(qo_board *b) {
    memset(b, 0, sizeof(*b));
    for (int i = 0; i < 9; ++i) {
        b->board[i] = -1;
        b->moves[i][0] = b->moves[i][1] = -1;
    }
}

/* ------------------------------------------------------------------ qeval.py:5-51
 * entangled: k moves (lo, hi, round); the closing move is the last one.
 * out[i]: square that move i collapses onto.  `bit` replaces random.choice at qeval.py:35:
 * bit 0 -> move[0] (lo), bit 1 -> move[1] (hi). */
static void qeval_classic(const int ent[][3], int k, int bit, int out[]) {
    uint32_t rutor[9];                     /* qeval.py:12-15: per-square set of moves   */
    for (int i = 0; i < 9; ++i) rutor[i] = 0;
    for (int i = 0; i < k; ++i) out[i] = -1;             /* qeval.py:6 */
    for (int i = 0; i < k; ++i) {                         /* qeval.py:17-19 */
        rutor[ent[i][0]] |= 1u << i;
        rutor[ent[i][1]] |= 1u << i;
    }
    /* qeval.py:23-31: moves not in the cycle; a square with one move left receives it,
     * then follow that move to its other square. */
    for (int i0 = 0; i0 < 9; ++i0) {
        int i = i0;
        while (__builtin_popcount(rutor[i]) == 1) {
            int m = __builtin_ctz(rutor[i]);              /* rutor[i].pop() */
            rutor[i] = 0;
            int index = (i == ent[m][0]) ? 1 : 0;         /* qeval.py:26 */
            int next_i = ent[m][index];
            int move_res = ent[m][1 - index];
            out[m] = move_res;                            /* qeval.py:29 */
            rutor[next_i] &= ~(1u << m);                  /* qeval.py:30 */
            i = next_i;
        }
    }
    /* qeval.py:35-49: closing move takes lo/hi by the bit, the rest of the cycle is forced */
    const int last = k - 1;
    out[last] = ent[last][bit ? 1 : 0];                   /* qeval.py:35 */
    int r_start = ent[last][0];
    int r = ent[last][1];
    int fell_in_r = (r == out[last]);
    rutor[r] &= ~(1u << last);                            /* qeval.py:40 */
    while (r != r_start) {
        int m = __builtin_ctz(rutor[r]);                  /* rutor[r].pop() */
        rutor[r] &= ~(1u << m);
        int m0_is_r = (ent[m][0] == r);
        int move_res = (fell_in_r ^ m0_is_r) ? ent[m][0] : ent[m][1];   /* qeval.py:44 */
        out[m] = move_res;
        r = (ent[m][1] == r) ? ent[m][0] : ent[m][1];     /* qeval.py:47 */
        rutor[r] &= ~(1u << m);                           /* qeval.py:48 */
        fell_in_r = (r == move_res);                      /* qeval.py:49 */
    }
}

/* ------------------------------------------------------------------ board.py:27-69 */
static void update_qstructs(qo_board *b, int lo, int hi, int bit, int *consumed) {
    int m0 = -1;
    for (int i = 0; i < b->n_q; ++i)                      /* board.py:28-33 */
        if (b->q[i] >> lo & 1) { m0 = i; break; }
    int m1 = -2;
    for (int j = 0; j < b->n_q; ++j)                      /* board.py:35-40 */
        if (b->q[j] >> hi & 1) { m1 = j; break; }

    if (m0 == m1) {                                       /* board.py:42-56 */
        int ent[9][3], rounds[9], out[9], k = 0;
        for (int i = 0; i < b->n_moves; ++i) {            /* board.py:44-50 */
            int first = b->moves[i][0];
            if (b->q[m0] >> first & 1) {
                ent[k][0] = b->moves[i][0];
                ent[k][1] = b->moves[i][1];
                ent[k][2] = i;
                rounds[k] = i;
                ++k;
            }
        }
        qeval_classic(ent, k, bit, out);                  /* board.py:51 */
        *consumed = 1;
        for (int i = 0; i < k; ++i) b->board[out[i]] = (int8_t)rounds[i];   /* board.py:53-54 */
        for (int i = m1; i + 1 < b->n_q; ++i) b->q[i] = b->q[i + 1];       /* board.py:56 pop */
        b->n_q -= 1;
    } else if (m0 >= 0 && m1 >= 0) {                      /* board.py:58-61 */
        b->q[m0] = b->q[m0] | b->q[m1];
        for (int i = m1; i + 1 < b->n_q; ++i) b->q[i] = b->q[i + 1];
        b->n_q -= 1;
    } else {                                              /* board.py:62-69 */
        int i = m0 > m1 ? m0 : m1;
        if (i < 0) {
            b->q[b->n_q] = 0;
            i = b->n_q;
            b->n_q += 1;
        }
        b->q[i] |= (uint16_t)(1u << lo);
        b->q[i] |= (uint16_t)(1u << hi);
    }
}

/* ------------------------------------------------------------------ board.py:9-25
 * a, c are what Env.step hands over as action[0], action[1] (env.py:37-40).  Anything outside
 * 0..8 reaches `self.board[...]` at board.py:14 and raises IndexError there (swallowed at
 * env.py:41) unless the same-square test (board.py:10) fired first.  Negative Python indices
 * are outside the action space (env.py:19) and outside the u8 C ABI; they are rejected here. */
int qo_make_move(qo_board *b, int a, int c, int bit, int *consumed) {
    *consumed = 0;
    if (a == c) return QO_ERR_SAME_SQUARE;                /* board.py:10-12 */
    /* board.py:14: `board[a] != -1 or board[c] != -1`, left to right, short-circuit */
    if (a < 0 || a > 8) return QO_ERR_INDEX;
    if (b->board[a] != -1) return QO_ERR_CLASSICAL;
    if (c < 0 || c > 8) return QO_ERR_INDEX;
    if (b->board[c] != -1) return QO_ERR_CLASSICAL;
    int lo = a, hi = c;
    if (lo > hi) { lo = c; hi = a; }                      /* board.py:16-18 */
    b->moves[b->n_moves][0] = (int8_t)lo;                 /* board.py:19 */
    b->moves[b->n_moves][1] = (int8_t)hi;
    b->n_moves += 1;
    update_qstructs(b, lo, hi, bit, consumed);            /* board.py:20 */
    int count = 0, idx = -1;                              /* board.py:22-25 autofill */
    for (int i = 0; i < 9; ++i)
        if (b->board[i] == -1) { if (idx < 0) idx = i; ++count; }
    if (count == 1) {
        b->board[idx] = (int8_t)b->n_moves;
        b->moves[b->n_moves][0] = (int8_t)idx;
        b->moves[b->n_moves][1] = (int8_t)idx;
        b->n_moves += 1;
    }
    return QO_OK;
}

/* ------------------------------------------------------------------ board.py:71-115 */
void qo_check_win(const qo_board *b, int *p1_round, int *p2_round) {
    int mark[9];
    for (int i = 0; i < 9; ++i) {                         /* board.py:73-78 */
        int m = b->board[i];
        mark[i] = (m < 0) ? 0 : (m % 2) * 2 - 1;
    }
    int p1 = 10, p2 = 10;                                 /* board.py:81-82 */
    /* rows :85-90, cols :93-98, 2-4-6 :101-105, 0-4-8 :106-110 — in the reference's order */
    static const int lines[8][3] = {{0, 1, 2}, {3, 4, 5}, {6, 7, 8}, {0, 3, 6}, {1, 4, 7},
                                    {2, 5, 8}, {2, 4, 6}, {0, 4, 8}};
    for (int l = 0; l < 8; ++l) {
        int s = 0, mx = -1;
        for (int j = 0; j < 3; ++j) {
            s += mark[lines[l][j]];
            if (b->board[lines[l][j]] > mx) mx = b->board[lines[l][j]];
        }
        if (s == -3) { if (mx < p1) p1 = mx; }
        else if (s == 3) { if (mx < p2) p2 = mx; }
    }
    *p1_round = p1 < 10 ? p1 : -1;                        /* board.py:112-113 */
    *p2_round = p2 < 10 ? p2 : -1;
}

/* ------------------------------------------------------------------ env.py:48,51 */
int qo_terminated(const qo_board *b) {
    int p1, p2;
    qo_check_win(b, &p1, &p2);
    return (p1 > 0 || p2 > 0) || b->n_moves > 8;
}

/* ------------------------------------------------------------------ env.py:34-53 */
int qo_step(qo_board *b, int a, int c, int bit, double *reward, int *terminated, int *consumed) {
    /* env.py:35 cur_player is computed but, by precedence, never reaches the reward */
    int status = qo_make_move(b, a, c, bit, consumed);    /* env.py:36-43: errors -> noop */
    int p1, p2;
    qo_check_win(b, &p1, &p2);                            /* env.py:48 */
    int win = (p1 > 0 || p2 > 0);
    /* env.py:49: (-1 ** cur_player) * float(win) == -(1 ** cur_player) * float(win)
     *          == -1 * {0.0, 1.0} == {-0.0, -1.0} */
    *reward = -1.0 * (win ? 1.0 : 0.0);
    *terminated = win || b->n_moves > 8;                  /* env.py:51 */
    return status;
}

/* ------------------------------------------------------------------ env.py:68-85 */
void qo_observe(const qo_board *b, int8_t classical[9], uint8_t q_p1[5][2], int *q_p1_len,
                uint8_t q_p2[4][2], int *q_p2_len, int *turn) {
    int n1 = 0, n2 = 0;
    for (int i = 0; i < b->n_moves; ++i) {                /* env.py:72 */
        int in_classical = 0;                             /* env.py:73: round not on the board */
        for (int s = 0; s < 9; ++s) if (b->board[s] == i) in_classical = 1;
        if (in_classical) continue;
        if (i % 2) {                                      /* env.py:74-77 */
            q_p2[n2][0] = (uint8_t)b->moves[i][0]; q_p2[n2][1] = (uint8_t)b->moves[i][1]; ++n2;
        } else {
            q_p1[n1][0] = (uint8_t)b->moves[i][0]; q_p1[n1][1] = (uint8_t)b->moves[i][1]; ++n1;
        }
    }
    for (int s = 0; s < 9; ++s) classical[s] = b->board[s];
    *q_p1_len = n1;
    *q_p2_len = n2;
    *turn = b->n_moves % 2;                               /* env.py:83 */
}

/* ================================================================== batch forms */
void qo_reset_batch(qo_board *b, int64_t n) {
    for (int64_t i = 0; i < n; ++i) qo_init(&b[i]);       /* env.py:55-57 */
}

/* ---- synthetic-input spec (the build's own). DESIGN.md §5 states it. ---- */
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static uint32_t lowbias32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
/* launch key: depends on (seed, step_idx) only, so it is uniform over a launch */
static uint64_t launch_key(uint64_t seed, uint32_t step_idx) {
    return splitmix64(seed ^ ((uint64_t)step_idx * 0xD1B54A32D192ED03ull));
}
uint64_t qo_hash(uint64_t seed, uint64_t board_id, uint32_t step_idx) {
    uint64_t key = launch_key(seed, step_idx);
    uint32_t id = (uint32_t)board_id ^ ((uint32_t)(board_id >> 32) * 0x9E3779B9u);
    uint32_t h1 = lowbias32(id ^ (uint32_t)key);
    uint32_t h2 = lowbias32(h1 ^ (uint32_t)(key >> 32));
    return ((uint64_t)h2 << 32) | h1;
}
int qo_collapse_bit(uint64_t seed, uint64_t board_id, uint32_t step_idx) {
    return (int)((qo_hash(seed, board_id, step_idx) >> 31) & 1u);   /* top bit of h1 */
}

uint64_t qo_legal_mask(const qo_board *b) {               /* mcts.py:20-27 */
    uint64_t mask = 0;
    for (int a = 0; a < 36; ++a) {
        int lo, hi;
        qo_ind2move(a, &lo, &hi);
        if (b->board[lo] != -1 || b->board[hi] != -1) continue;
        mask |= 1ull << a;
    }
    return mask;
}

void qo_sample_action(const qo_board *b, uint64_t seed, uint64_t board_id, uint32_t step_idx,
                      uint8_t out[2]) {
    uint64_t mask = qo_legal_mask(b);
    int n_legal = __builtin_popcountll(mask);
    out[0] = out[1] = 0;
    if (n_legal == 0) return;
    uint32_t h2 = (uint32_t)(qo_hash(seed, board_id, step_idx) >> 32);
    int k = (int)(((uint64_t)h2 * (uint64_t)n_legal) >> 32);        /* 0..n_legal-1 */
    for (int a = 0; a < 36; ++a) {
        if (!(mask >> a & 1)) continue;
        if (k-- == 0) {
            int lo, hi;
            qo_ind2move(a, &lo, &hi);
            out[0] = (uint8_t)lo;
            out[1] = (uint8_t)hi;
            return;
        }
    }
}

void qo_sample_actions_batch(const qo_board *b, int64_t n, uint64_t seed, uint32_t step_idx,
                             int64_t board_offset, int auto_reset, uint8_t *actions) {
    qo_board fresh;
    qo_init(&fresh);
    for (int64_t i = 0; i < n; ++i) {
        const qo_board *src = (auto_reset && qo_terminated(&b[i])) ? &fresh : &b[i];
        qo_sample_action(src, seed, (uint64_t)(board_offset + i), step_idx, &actions[2 * i]);
    }
}

void qo_step_batch(qo_board *b, int64_t n, const uint8_t *actions, const uint8_t *bits,
                   uint64_t seed, uint32_t step_idx, int64_t board_offset, int auto_reset,
                   float *reward, uint8_t *terminated) {
    for (int64_t i = 0; i < n; ++i) {
        if (auto_reset && qo_terminated(&b[i])) qo_init(&b[i]);
        int bit = bits ? (bits[i] & 1) : qo_collapse_bit(seed, (uint64_t)(board_offset + i), step_idx);
        double r;
        int term, consumed;
        qo_step(&b[i], actions[2 * i], actions[2 * i + 1], bit, &r, &term, &consumed);
        reward[i] = (float)r;
        terminated[i] = (uint8_t)term;
    }
}

/* n_steps consecutive qo_step_batch passes over one slice of boards from a recorded action stream
 * actions[t][stride boards][2] (the slice starts at `actions`; the next step's actions are stride boards on): what
 * bench.py's cpu_baseline times — one C call per (thread, replay), so that a host with hundreds of threads is not
 * bound by the interpreter lock between steps.  reward / terminated: n-element scratch, overwritten every step. */
void qo_replay_batch(qo_board *b, int64_t n, const uint8_t *actions, int64_t stride, int32_t n_steps,
                     uint64_t seed, uint32_t step_idx0, int64_t board_offset, int auto_reset,
                     float *reward, uint8_t *terminated) {
    for (int32_t t = 0; t < n_steps; ++t)
        qo_step_batch(b, n, actions + (int64_t)t * stride * 2, NULL, seed, step_idx0 + (uint32_t)t, board_offset,
                      auto_reset, reward, terminated);
}

/* ================================================================== MCTS expand row */
void qo_ind2move(int n, int *lo, int *hi) {               /* mcts.py:339-343 */
    /* the reference inverts the triangular numbering with a float sqrt; the table it produces
     *  is lexicographic pairs (0,1),(0,2)..(7,8) — restated exactly. */
    /* n outside 0..35 is not an action (the reference's formula gives (8,9) for 36 and a math
     * domain error beyond): the loop stops at i = 8, so hi > 8 and make_move rejects it. */
    int i = 0, base = 0;
    while (i < 8 && n >= base + (8 - i)) { base += 8 - i; ++i; }
    *lo = i;
    *hi = i + 1 + (n - base);
}

void qo_update_winner(const qo_board *b, int *winner, int *terminal) {   /* mcts.py:52-65 */
    int p1, p2;
    qo_check_win(b, &p1, &p2);
    *winner = -1;                                         /* None (mcts.py:238 ctor arg) */
    *terminal = 0;
    if (p1 > 0 && p2 > 0) { *winner = p1 < p2; *terminal = 1; }
    else if (p2 < 0 && p1 > 0) { *winner = 1; *terminal = 1; }
    else if (p1 < 0 && p2 > 0) { *winner = 0; *terminal = 1; }
    *terminal = (b->n_moves == 9) || *terminal;
}

int qo_expand(const qo_board *parent, int action36, qo_board child[2], int winner[2],
              int terminal[2], uint64_t legal_mask[2]) {  /* mcts.py:233-267 */
    int lo, hi, consumed;
    if (action36 < 0 || action36 > 35) return 0;          /* not an action: no children */
    qo_ind2move(action36, &lo, &hi);
    child[0] = *parent;                                   /* mcts.py:235-241 copies */
    if (qo_make_move(&child[0], lo, hi, 0, &consumed) != QO_OK) return 0;
    qo_update_winner(&child[0], &winner[0], &terminal[0]);
    legal_mask[0] = qo_legal_mask(&child[0]);
    if (!consumed) return 1;                              /* mcts.py:245-246 board unchanged */
    /* mcts.py:252-261 resamples until the other value of the random bit shows up; the two
     * values are enumerated directly instead. */
    child[1] = *parent;
    qo_make_move(&child[1], lo, hi, 1, &consumed);
    qo_update_winner(&child[1], &winner[1], &terminal[1]);
    legal_mask[1] = qo_legal_mask(&child[1]);
    return 2;
}

void qo_to_vector(const qo_board *b, double out[180]) {   /* mcts.py:67-85 */
    for (int i = 0; i < 180; ++i) out[i] = 0.0;
    for (int i = 0; i < 9; ++i) {                         /* :68-70; board -1 indexes column 9 */
        int col = b->board[i] < 0 ? 9 : b->board[i];
        out[i * 10 + col] = 1.0;
    }
    const double third = 1.0 / 3.0;                       /* :72 1/math.sqrt(9) */
    for (int t = 0; t < b->n_moves; ++t) {                /* :73-75 every move, collapsed too */
        out[(9 + b->moves[t][0]) * 10 + t] = third;
        out[(9 + b->moves[t][1]) * 10 + t] = third;
    }
    uint16_t qsets = 0;                                   /* :77-79 */
    for (int i = 0; i < b->n_q; ++i) qsets |= b->q[i];
    for (int s = 0; s < 9; ++s)                           /* :81-84 */
        if (!(qsets >> s & 1)) out[(9 + s) * 10 + 9] = 1.0;
}

/* ------------------------------------------------------------------ mcts.py:93-94
 * GameState.__hash__ = hash(tuple(self.board) + tuple(self.moves)).  The arithmetic is CPython's
 * (third party to the reference): Objects/tupleobject.c `tuplehash` (xxHash-style, CPython >= 3.8,
 * pinned here to the 3.10.12 of this image) and Objects/longobject.c `long_hash` (small ints hash
 * to themselves, -1 hashes to -2).  Checked against the running interpreter's hash() in
 * tests/test_next_rows_cpu.py and against the hashes recorded from the reference. */
static uint64_t py_acc(uint64_t acc, uint64_t lane) {
    acc += lane * 14029467366897019727ull;
    acc = (acc << 31) | (acc >> 33);
    return acc * 11400714785074694791ull;
}
static uint64_t py_fin(uint64_t acc, uint64_t len) {
    acc += len ^ (2870177450012600261ull ^ 3527539ull);
    return acc == ~0ull ? 1546275796ull : acc;
}
int64_t qo_pyhash(const qo_board *b) {
    uint64_t acc = 2870177450012600261ull;
    for (int i = 0; i < 9; ++i)
        acc = py_acc(acc, b->board[i] == -1 ? (uint64_t)(int64_t)-2 : (uint64_t)b->board[i]);
    for (int t = 0; t < b->n_moves; ++t) {
        uint64_t in = 2870177450012600261ull;
        in = py_acc(in, (uint64_t)b->moves[t][0]);
        in = py_acc(in, (uint64_t)b->moves[t][1]);
        in = py_acc(in, (uint64_t)t);
        acc = py_acc(acc, py_fin(in, 3));
    }
    return (int64_t)py_fin(acc, (uint64_t)(9 + b->n_moves));
}
