// qttt_board_forms.h — unpacked views of the state for everything that is not the step: the generic
// `Cold` form (import / export / board_op / encode), the `Lite` form of the MCTS-side kernels, win
// check, winner, legal-action mask and the CPython tuple hash.
#ifndef QTTT_BOARD_FORMS_H
#define QTTT_BOARD_FORMS_H
#include "qttt_state.h"

namespace {

// ====================================================================== cold paths
// Friendly unpacked form for the kernels that are not on the hot path.  Everything is kept in
// packed words with shift accessors — no per-thread arrays: runtime-indexed arrays would live in
// scratch memory, and the scratch-backed version of these kernels returned an occasional wrong
// element under 512-thread workgroups on this part (round 1).
// tests/test_abi_and_host.py asserts that no kernel of this file uses scratch.
struct Cold {
    u32 n;          // n_moves, autofill move included
    u32 cl;         // classical mask, autofilled square included
    u32 done;
    u64 mvq;        // moves of rounds 0..7: byte t = lo | hi<<4
    u32 mv8;        // move of round 8
    u64 sq;         // 9 nibbles, true values (0xF = root / isolated / empty)
    u64 comps;      // 4 x 9-bit masks, list order
    __device__ u32 mv(u32 t) const { return t >= 8u ? mv8 : (u32)(mvq >> (t * 8u)) & 0xFFu; }
    __device__ void set_mv(u32 t, u32 m) {
        if (t >= 8u) mv8 = m & 0xFFu;
        else mvq = (mvq & ~(0xFFull << (t * 8u))) | ((u64)(m & 0xFFu) << (t * 8u));
    }
    __device__ u32 sqv(u32 v) const { return (u32)(sq >> (v * 4u)) & 0xFu; }
    __device__ void set_sq(u32 v, u32 x) { sq = (sq & ~(0xFull << (v * 4u))) | ((u64)(x & 0xFu) << (v * 4u)); }
    __device__ u32 comp(u32 k) const { return (u32)(comps >> (9u * k)) & 0x1FFu; }
};

// x = lo ^ hi of the move of round e (e < n real moves), see the layout notes at the top
__device__ __forceinline__ u32 cold_move_x(u32 Q0, u32 P1, u32 n_real, u32 e) {
    const u32 last_x = (P1 >> P1_LX_SHIFT) & 0xFu;
    if (e >= 8u) return last_x;
    u32 x = (rotr32(Q0, 4u * (7u - e)) >> 2) & 0xFu;
    if (e == 0u && n_real == 9u) x ^= last_x;                   // round 8's x was XORed onto round 0's nibble
    return x;
}

__device__ __forceinline__ void cold_unpack(u64 P, u64 Q, Cold &s, bool autofill = true) {
    const u32 P1 = (u32)(P >> 32), Q0 = (u32)Q;
    s.n = (P1 >> P1_N_SHIFT) & 0xFu;
    s.cl = (P1 >> P1_CL_SHIFT) & 0x1FFu;
    s.done = P1 >> 31;
    s.sq = ((P >> 2) & 0xFFFFFFFFFull) ^ 0xFFFFFFFFFull;         // stored complemented
    s.comps = (Q >> 32) | ((u64)((P1 >> P1_CHI_SHIFT) & 0xFu) << 32);
    // Board.moves from the holders: the square c with sq[c] = e is one end of the move of round e
    // (child end if un-collapsed, landing square if collapsed), the other end is c ^ x_e
    s.mvq = 0;
    s.mv8 = 0;
    const u32 n_real = s.n;
    for (u32 c = 0; c < 9; ++c) {
        const u32 e = s.sqv(c);
        if (e >= n_real) continue;                               // 0xF = root / isolated / empty
        const u32 o = c ^ cold_move_x(Q0, P1, n_real, e);
        s.set_mv(e, min(c, o) | (max(c, o) << 4));
    }
    // materialise the implicit autofill (board.py:22-25): exactly 8 classical squares
    if (autofill && __builtin_popcount(s.cl) == 8 && s.n < 9u) {
        const u32 idx = (u32)__builtin_ctz(~s.cl);
        s.set_sq(idx, s.n);                                      // board[idx] = len(self.moves)
        s.cl |= 1u << idx;
        s.set_mv(s.n, idx | (idx << 4));                         // moves.append((idx, idx, len))
        s.n += 1u;
    }
}

__device__ __forceinline__ void cold_pack(const Cold &in, u64 &P, u64 &Q) {
    Cold s = in;
    // strip an explicit autofill move (lo == hi, always the last one) back to the implicit form
    if (s.n >= 1u && s.n <= 9u) {
        const u32 last = s.mv(s.n - 1u);
        if ((last & 0xFu) == (last >> 4)) {
            const u32 idx = last & 0xFu;
            if (idx < 9u) {
                s.cl &= ~(1u << idx);
                s.set_sq(idx, 0xFu);
            }
            s.n -= 1u;
        }
    }
    u32 Q0 = 0, last_x = 0;
    for (u32 t = 0; t < s.n && t < 9u; ++t) {
        const u32 m = s.mv(t);
        last_x = ((m & 0xFu) ^ (m >> 4)) & 0xFu;
        Q0 ^= rotr32(last_x << 2, 4u * t + 4u);                  // as the step kernel appends it
    }
    const u64 sqc = (s.sq & 0xFFFFFFFFFull) ^ 0xFFFFFFFFFull;
    const u32 P1f = (s.n << P1_N_SHIFT) | (((u32)(s.comps >> 32) & 0xFu) << P1_CHI_SHIFT) |
                    (last_x << P1_LX_SHIFT) | (s.cl << P1_CL_SHIFT) | (s.done ? P1_DONE : 0u);
    P = (sqc << 2) | ((u64)P1f << 32);
    Q = (u64)Q0 | ((u64)(u32)s.comps << 32);
}

// one line of board.py:85-110: p1/p2 = min over completed lines of the max round in the line
__device__ __forceinline__ void cold_line(const Cold &s, u32 X, u32 O, u32 L, int &p1, int &p2) {
    int mx = -1;
    for (u32 v = 0; v < 9; ++v)
        if (L >> v & 1u) mx = max(mx, (int)s.sqv(v));
    const bool c1 = (X & L) == L, c2 = !c1 && (O & L) == L;     // selects, not a choice of address:
    p1 = c1 ? min(p1, mx) : p1;                                  // keeps p1/p2 in registers
    p2 = c2 ? min(p2, mx) : p2;
}

__device__ __forceinline__ void cold_check_win(const Cold &s, int &p1, int &p2) {
    // board.py:71-115, lines in the reference's order (rows, cols, 2-4-6, 0-4-8)
    u32 X = 0, O = 0;
    for (u32 v = 0; v < 9; ++v)
        if (s.cl >> v & 1u) { if (s.sqv(v) & 1u) O |= 1u << v; else X |= 1u << v; }
    p1 = 10;
    p2 = 10;
    cold_line(s, X, O, 0x007u, p1, p2);
    cold_line(s, X, O, 0x038u, p1, p2);
    cold_line(s, X, O, 0x1C0u, p1, p2);
    cold_line(s, X, O, 0x049u, p1, p2);
    cold_line(s, X, O, 0x092u, p1, p2);
    cold_line(s, X, O, 0x124u, p1, p2);
    cold_line(s, X, O, 0x054u, p1, p2);
    cold_line(s, X, O, 0x111u, p1, p2);
    if (p1 >= 10) p1 = -1;
    if (p2 >= 10) p2 = -1;
}

// ---------------------------------------------------------------------------------------------
// The MCTS-side kernels (node_info / expand / rollout / check_win) do not go through the generic
// `Cold` form: what they need is computed straight from the packed words.
struct Lite {
    u64 P;          // plane P with the implicit autofill materialised in the nibbles
    u32 cl;         // classical mask, autofilled square included
    u32 n;          // len(moves), autofill move included
    u32 n_real;     // moves played (the autofill move is not one)
};

__device__ __forceinline__ Lite lite_unpack(u64 P) {
    Lite s;
    const u32 P1 = (u32)(P >> 32);
    s.cl = (P1 >> P1_CL_SHIFT) & 0x1FFu;
    s.n_real = s.n = (P1 >> P1_N_SHIFT) & 0xFu;
    if (__builtin_popcount(s.cl) == 8 && s.n < 9u) {                 // board.py:22-25, implicit in the state
        const u32 idx = (u32)__builtin_ctz(~s.cl);
        P |= (u64)(15u - s.n) << (4u * idx + 2u);                    // its code was 0 (isolated square)
        s.cl = 0x1FFu;
        s.n += 1u;
    }
    s.P = P;
    return s;
}

// Board.check_win (board.py:71-115) without visiting lines: p1_round = min over completed X lines of
// the line's latest round = the smallest m in {4,6,8} such that the X squares of round <= m contain
// a line (three X marks need rounds 0,2,4 at least); likewise p2_round over {5,7}.  "round <= m" is
// "code >= 15-m" on the complemented nibbles, tested for eight squares at once (classical codes are
// 7..15: bit 3 set and low three bits >= T-8 <=> adding 16-T carries into bit 3).
__device__ __forceinline__ void fast_check_win(const Lite &s, const uint8_t *lds_lut, int &p1, int &p2) {
    const u32 W = (u32)(s.P >> 2);                                   // codes of squares 0..7
    const u32 c8 = (u32)(s.P >> 34) & 0xFu;
    const u32 par = W & 0x11111111u;                                 // odd code = even round = X
    const u32 even = __builtin_amdgcn_udot8(par, 0x00008421u, 0u, false) |
                     (__builtin_amdgcn_udot8(par, 0x84210000u, 0u, false) << 4) | ((c8 & 1u) << 8);
    const u32 X = s.cl & even, O = s.cl & ~even;
    u32 ge[3];
#pragma unroll
    for (u32 k = 0; k < 3; ++k) {                                    // code >= 9, 10, 11  <=>  round <= 6, 5, 4
        const u32 T = 9u + k;
        const u32 y = ((W & 0x77777777u) + 0x11111111u * (16u - T)) & W & 0x88888888u;
        ge[k] = ((__builtin_amdgcn_udot8(y, 0x00008421u, 0u, false) |
                  (__builtin_amdgcn_udot8(y, 0x84210000u, 0u, false) << 4)) >> 3) | ((c8 >= T ? 1u : 0u) << 8);
    }
    // five reads of the workgroup's LDS line table (a dword per mask), all issued up front
    const u32 *l32 = reinterpret_cast<const u32 *>(lds_lut);
    const u32 x4 = l32[X & ge[2]], x6 = l32[X & ge[0]], x8 = l32[X], o5 = l32[O & ge[1]], o7 = l32[O];
    p1 = x4 ? 4 : (x6 ? 6 : (x8 ? 8 : -1));
    p2 = o5 ? 5 : (o7 ? 7 : -1);
}

// GameState.update_winner (mcts.py:52-65): winner 1 True / 0 False / -1 None; terminal = a line or
// nine moves.  Through the workgroup's LDS line table (the step kernel's: a dword per mask, read at
// mask * 4): who holds a line takes two gathers; the rounds matter only when both players do
// (p1 in {4,6,8}, p2 in {5,7}: p1 < p2 <=> p1 == 4, or p1 == 6 and p2 == 7), a branch most waves skip.
__device__ __forceinline__ void lite_update_winner(const Lite &s, const uint8_t *lut, int &winner, int &terminal) {
    const u32 W = (u32)(s.P >> 2);
    const u32 c8 = (u32)(s.P >> 34) & 0xFu;
    const u32 par = W & 0x11111111u;
    const u32 even = __builtin_amdgcn_udot8(par, 0x00008421u, 0u, false) |
                     (__builtin_amdgcn_udot8(par, 0x84210000u, 0u, false) << 4) | ((c8 & 1u) << 8);
    const u32 X = s.cl & even, O = s.cl & ~even;
    const u32 *l32 = reinterpret_cast<const u32 *>(lut);
    const bool hx = l32[X] != 0u, ho = l32[O] != 0u;
    winner = hx ? 1 : (ho ? 0 : -1);
    if (hx && ho) {
        u32 ge[3];
#pragma unroll
        for (u32 k = 0; k < 3; ++k) {                                // code >= 9, 10, 11  <=>  round <= 6, 5, 4
            const u32 T = 9u + k;
            const u32 y = ((W & 0x77777777u) + 0x11111111u * (16u - T)) & W & 0x88888888u;
            ge[k] = ((__builtin_amdgcn_udot8(y, 0x00008421u, 0u, false) |
                      (__builtin_amdgcn_udot8(y, 0x84210000u, 0u, false) << 4)) >> 3) | ((c8 >= T ? 1u : 0u) << 8);
        }
        winner = (l32[X & ge[2]] != 0u || (l32[X & ge[0]] != 0u && l32[O & ge[1]] == 0u)) ? 1 : 0;
    }
    terminal = (s.n == 9u || hx || ho) ? 1 : 0;
}

// The same from what the step already knows: xo = who holds a line (step_line_xo), the state's done bit = a line or
// nine moves.  Only when both players hold a line do the rounds matter (mcts.py:54-56): that rare case takes the
// round-aware path above.
__device__ __forceinline__ void update_winner_from_step(u64 P, u32 xo, const uint8_t *lut, int &winner, int &terminal) {
    winner = (xo & 1u) ? 1 : ((xo & 2u) ? 0 : -1);
    terminal = (int)((u32)(P >> 63));
    if (xo == 3u) lite_update_winner(lite_unpack(P), lut, winner, terminal);
}

// GameState.actions (mcts.py:20-27) in ind2move order (mcts.py:339-343): the pairs (i, j > i) of
// row i are the empty squares above i, eight rows at offsets 0, 8, 15, 21, 26, 30, 33, 35
__device__ __forceinline__ u64 fast_legal_mask(u32 cl) {
    const u32 E = ~cl & 0x1FFu;
    u64 m = 0;
    u32 off = 0;
#pragma unroll
    for (u32 i = 0; i < 8; ++i) {
        m |= (u64)((E >> i & 1u) ? (E >> (i + 1u)) : 0u) << off;
        off += 8u - i;
    }
    return m;
}

// the same as a table over the nine classical bits, for kernels that have an LDS copy of it
__host__ __device__ constexpr u64 legal_mask_of(u32 cl) {
    const u32 E = ~cl & 0x1FFu;
    u64 m = 0;
    u32 off = 0;
    for (u32 i = 0; i < 8; ++i) {
        m |= (u64)((E >> i & 1u) ? (E >> (i + 1u)) : 0u) << off;
        off += 8u - i;
    }
    return m;
}
struct LegalLut {
    u64 m[512];
    constexpr LegalLut() : m() {
        for (u32 cl = 0; cl < 512; ++cl) m[cl] = legal_mask_of(cl);
    }
};
__device__ const LegalLut g_legal_lut = LegalLut();
template <int BLOCK>
__device__ inline void fill_legal_lut(u64 *dst) {
    for (u32 w = threadIdx.x; w < 512u; w += BLOCK) dst[w] = g_legal_lut.m[w];
}
// The same in two halves, for kernels that stream their state: the table words are REQUESTED before the state loads
// and stored to LDS behind them.  Vector loads return in order, so the wait in front of the LDS store then covers
// the table words only and the workgroup barrier is passed while the state is still in flight (a table loaded
// after the state ties the barrier — and every wave of the workgroup — to the slowest wave's state data).
template <int BLOCK>
struct LegalLutWords {
    static constexpr int N = (512 + BLOCK - 1) / BLOCK;
    u64 w[N];
    __device__ __forceinline__ void request() {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const u32 i = threadIdx.x + (u32)k * BLOCK;
            w[k] = i < 512u ? g_legal_lut.m[i] : 0ull;
        }
    }
    __device__ __forceinline__ void store(u64 *dst) const {
#pragma unroll
        for (int k = 0; k < N; ++k) {
            const u32 i = threadIdx.x + (u32)k * BLOCK;
            if (i < 512u) dst[i] = w[k];
        }
    }
};

// GameState.__hash__ (mcts.py:93-94) = hash(tuple(board) + tuple(moves)) under CPython >= 3.8
// (Objects/tupleobject.c tuplehash, xxHash-style; hash(int) = the int, hash(-1) = -2).  One
// accumulator step is acc = rotl(acc + lane * P2, 31) * P1; the products lane * P2 are tabulated
// for the ten board values and for the hash of every possible move tuple (lo, hi, round).
constexpr u64 PYH_P1 = 11400714785074694791ull, PYH_P2 = 14029467366897019727ull, PYH_P5 = 2870177450012600261ull;
__host__ __device__ constexpr u64 pyh_step(u64 acc, u64 lane_times_p2) {
    acc += lane_times_p2;
    acc = (acc << 31) | (acc >> 33);
    return acc * PYH_P1;
}
// the same on the device: the 64-bit rotate by 31 as two v_alignbit_b32 (the compiler's own lowering is a
// 64-bit shift pair + or)
__device__ __forceinline__ u64 pyh_step_dev(u64 acc, u64 lane_times_p2) {
    acc += lane_times_p2;
    const u32 lo = (u32)acc, hi = (u32)(acc >> 32);
    const u32 rlo = __builtin_amdgcn_alignbit(lo, hi, 1u);           // (lo << 31) | (hi >> 1)
    const u32 rhi = __builtin_amdgcn_alignbit(hi, lo, 1u);           // (hi << 31) | (lo >> 1)
    return ((u64)rlo | ((u64)rhi << 32)) * PYH_P1;
}
__host__ __device__ constexpr u64 pyh_fin(u64 acc, u64 len) {
    acc += len ^ (PYH_P5 ^ 3527539ull);
    return acc == ~0ull ? 1546275796ull : acc;
}
struct PyHashLut {
    u64 board[10];          // [v + 1] for Board.board value v = -1..8
    u64 move[9][9][9];      // [a][b][round] for the move on squares {a, b}, either order
    constexpr PyHashLut() : board(), move() {
        board[0] = (u64)(long long)-2 * PYH_P2;
        for (u64 v = 0; v < 9; ++v) board[v + 1] = v * PYH_P2;
        for (u64 a = 0; a < 9; ++a)
            for (u64 b = 0; b < 9; ++b)
                for (u64 t = 0; t < 9; ++t) {
                    u64 in = PYH_P5;
                    in = pyh_step(in, (a < b ? a : b) * PYH_P2);    // Board.moves holds (lo, hi, round)
                    in = pyh_step(in, (a < b ? b : a) * PYH_P2);
                    in = pyh_step(in, t * PYH_P2);
                    move[a][b][t] = pyh_fin(in, 3) * PYH_P2;
                }
    }
};
__device__ const PyHashLut g_pyhash_lut = PyHashLut();
constexpr u32 PYHASH_LUT_WORDS = 10 + 729;          // u64 entries: board[10] then move[9][9][9]

// the table is gathered 9 + n times per board with a different entry in every lane: it is served
// from an LDS copy (5.9 KB per workgroup), not from the vector cache
template <int BLOCK>
__device__ inline void fill_pyhash_lut(u64 *dst) {
    const u64 *src = reinterpret_cast<const u64 *>(&g_pyhash_lut);
    for (u32 w = threadIdx.x; w < PYHASH_LUT_WORDS; w += BLOCK) dst[w] = src[w];
}

// One board's walk over its moves in round order: round t is held by the one square whose code is
// 15 - t (zero nibble of W ^ 0x1111_1111 * code; the lowest flag of the borrow trick is always a true
// zero; no flag = square 8), and is (c, c ^ x_t).  x nibbles: round 0 in bits 0..3 of Qr, round t >= 1 at
// 32 - 4t.
struct PyHashWalk {
    u64 acc;
    u32 W, c8, Qr, kk, sh, n8, last_x, V, v8;
    bool nine_real, nine;
    __device__ __forceinline__ void init(const Lite &s, u32 P1_stored, u32 Q0) {
        acc = PYH_P5;
        W = (u32)(s.P >> 2);                                        // codes of squares 0..7
        c8 = (u32)(s.P >> 34) & 0xFu;
        // index of every board element into the table, for eight squares at once: value + 1 = 16 - code where
        // the square is classical, 0 (the entry of -1) elsewhere.  M = 0xF on the classical squares' nibbles (the
        // 8 mask bits spread to nibble LSBs, times 15); a classical code is >= 7, so the nibble-wise two's
        // complement (~code + 1) never carries out of its nibble once the other nibbles are forced to 0xF first.
        u32 m = s.cl & 0xFFu;
        m = (m | (m << 12)) & 0x000F000Fu;
        m = (m | (m << 6)) & 0x03030303u;
        m = (m | (m << 3)) & 0x11111111u;
        const u32 M = (m << 4) - m;
        V = (~(W | ~M) + 0x11111111u) & M;
        v8 = (s.cl & 0x100u) ? 16u - c8 : 0u;
        last_x = (P1_stored >> P1_LX_SHIFT) & 0xFu;
        nine_real = s.n_real == 9u;
        nine = s.n == 9u;
        Qr = rotr32(Q0, 30u) ^ (nine_real ? last_x : 0u);           // round 8's x was XORed onto round 0's
        n8 = min(s.n, 8u);
        kk = 0xFFFFFFFFu;
        sh = 0u;
    }
    __device__ __forceinline__ void board_elem(u32 v, u32 cl, const u64 *tbl) {
        (void)cl;
        acc = pyh_step_dev(acc, tbl[v < 8u ? (V >> (4u * v)) & 0xFu : v8]);   // board[value + 1], value = 15 - code
    }
    __device__ __forceinline__ void move_elem(u32 t, const u64 *tbl) {   // t < n8 (an autofill move is always round 8)
        const u32 z = W ^ kk;
        const u32 f = (z - 0x11111111u) & ~z & 0x88888888u;
        const u32 c = min(ffbl_raw(f) >> 2, 8u);                    // no flag: v_ffbl_b32 gives 0xFFFFFFFF -> square 8
        const u32 o = c ^ ((Qr >> sh) & 0xFu);                      // <= 8 for any state the kernels produce
        // entry 10 + (c * 9 + o) * 9 + t as a byte offset: two multiply-adds, no shift
        acc = pyh_step_dev(acc, *reinterpret_cast<const u64 *>(reinterpret_cast<const char *>(tbl) + (80u + 8u * t) + c * 648u + o * 72u));
        kk -= 0x11111111u;
        sh = (sh - 4u) & 31u;
    }
    __device__ __forceinline__ int64_t finish(u32 n, const u64 *tbl) {
        if (nine) {
            const u32 z = W ^ 0x77777777u;
            const u32 f = (z - 0x11111111u) & ~z & 0x88888888u;
            const u32 c = f ? (u32)__builtin_ctz(f) >> 2 : 8u;
            const u32 o = min(c ^ (nine_real ? last_x : 0u), 8u);   // autofill = (idx, idx)
            acc = pyh_step_dev(acc, tbl[10u + (c * 9u + o) * 9u + 8u]);
        }
        return (int64_t)pyh_fin(acc, 9u + n);
    }
};

// Two boards at once: the hash is a chain of dependent steps (add, rotate, 64-bit multiply, next table
// address) and every step waits on a 64-bit LDS gather, so one chain per lane leaves the SIMD waiting
// (~6.6 cycles per instruction at 8 waves per SIMD); two independent chains in one
// instruction stream fill those slots.  The rounds both boards have are walked together, the rest of
// the longer game on its own (boards of one batch are usually at similar depths).
__device__ __forceinline__ void fast_py_hash_pair(const Lite &sa, u32 P1a, u32 Q0a, const Lite &sb, u32 P1b, u32 Q0b,
                                                  const u64 *tbl, int64_t &ka, int64_t &kb) {
    PyHashWalk a, b;
    a.init(sa, P1a, Q0a);
    b.init(sb, P1b, Q0b);
#pragma unroll
    for (u32 v = 0; v < 9; ++v) { a.board_elem(v, sa.cl, tbl); b.board_elem(v, sb.cl, tbl); }
    const u32 both = min(a.n8, b.n8);
    u32 t = 0;
    for (; t < both; ++t) { a.move_elem(t, tbl); b.move_elem(t, tbl); }
    for (u32 u = t; u < a.n8; ++u) a.move_elem(u, tbl);
    for (u32 u = t; u < b.n8; ++u) b.move_elem(u, tbl);
    ka = a.finish(sa.n, tbl);
    kb = b.finish(sb.n, tbl);
}

}  // namespace

#endif  // QTTT_BOARD_FORMS_H
