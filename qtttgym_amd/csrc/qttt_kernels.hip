// qttt_kernels.hip — gfx950 (MI355X / CDNA4) kernels + the C ABI of include/qttt.h.
//
// Mapping: ONE LANE PER BOARD (64 boards per wavefront), everything in VGPRs, structure-of-
// arrays state so that every load/store of a wave is one contiguous, fully coalesced segment.
// No LDS, no MFMA: the path is HBM-bound integer/bit work (DESIGN.md §2 explains why the
// wave-per-board mapping was rejected after measurement).
//
// Formulation — deliberately NOT the reference's algorithm:
//   * the un-collapsed moves of a board form a forest on the 9 squares (a move that closes a
//     cycle collapses its whole component at once, board.py:42-56).  The forest is kept ROOTED:
//     nibble sq[v] of a non-classical square v is the round of the move joining v to its parent
//     (0xF = root / isolated).  For a classical square, sq[v] is the round that landed there
//     (= Board.board[v]).
//   * QEvalClassic.eval (qeval.py:5-51: leaf-peel + forced walk round the cycle) is equivalent
//     to: re-root the tree at the square t the closing move lands on (bit picks lo/hi), then
//     every other square of the component receives its parent edge.  So a collapse is one path
//     reversal + `classical |= component`; no per-edge work.
//   * Board.qstructs (board.py:6) is cached as 4 slots x 9-bit square masks, in the reference's
//     list order, so "same component?" is two shifts and an AND.
//
// Packed state, 20 B/board, planes A[n] u64 | B[n] u64 | C[n] u32:
//   A : moves 0..7, byte i = lo | hi<<4 (board.py:19), unused bytes 0
//   B : [0,36) sq nibbles | [36,44) move 8 | [44,48) n_moves | [48,57) classical mask |
//       57 done (terminated at the end of the last step) | [60,64) comps bits 32..35
//   C : comps bits 0..31   (comps = 4 x 9-bit masks, slot k at bit 9k)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "qttt.h"

typedef unsigned long long u64;
typedef unsigned int u32;

#define QTTT_BLOCK 256
#define QTTT_DEFAULT_BPL 2
#define QTTT_DEFAULT_PIPE 0

namespace {

constexpr u64 SQ_EMPTY = 0xFFFFFFFFFull;   // nine 0xF nibbles
constexpr u32 SLOT_LSB = 0x08040201u;      // bit 0 of each 9-bit comps slot

struct Regs {
    u64 A;      // moves 0..7
    u64 sq;     // 9 nibbles
    u64 comps;  // 4 x 9 bits
    u32 mv8;    // move 8
    u32 n;      // n_moves
    u32 cl;     // classical mask
    u32 done;   // cached terminated flag
};

struct Planes {
    u64 *A;
    u64 *B;
    u32 *C;
};

// plane stride: n rounded up to 64 boards, so every plane starts 512-byte aligned
__host__ __device__ inline int64_t plane_stride(int64_t n) { return (n + 63) & ~(int64_t)63; }

__host__ __device__ inline Planes planes(void *state, int64_t n) {
    Planes p;
    const int64_t s = plane_stride(n);
    p.A = reinterpret_cast<u64 *>(state);
    p.B = p.A + s;
    p.C = reinterpret_cast<u32 *>(p.B + s);
    return p;
}

__device__ inline void unpack(u64 A, u64 B, u32 C, Regs &r) {
    u32 hi = (u32)(B >> 32);
    r.A = A;
    r.sq = B & SQ_EMPTY;
    r.mv8 = (hi >> 4) & 0xFFu;
    r.n = (hi >> 12) & 0xFu;
    r.cl = (hi >> 16) & 0x1FFu;
    r.done = (hi >> 25) & 1u;
    r.comps = (u64)C | ((u64)(hi >> 28) << 32);
}

__device__ inline void pack(const Regs &r, u64 &A, u64 &B, u32 &C) {
    u32 chi = (u32)(r.comps >> 32);
    u32 hi = (u32)(r.sq >> 32) | (r.mv8 << 4) | (r.n << 12) | (r.cl << 16) | (r.done << 25) |
             (chi << 28);
    A = r.A;
    B = (u64)(u32)r.sq | ((u64)hi << 32);
    C = (u32)r.comps;
}

__device__ inline void regs_reset(Regs &r) {
    r.A = 0;
    r.sq = SQ_EMPTY;
    r.comps = 0;
    r.mv8 = 0;
    r.n = 0;
    r.cl = 0;
    r.done = 0;
}

__device__ inline u32 get_move(const Regs &r, u32 idx) {
    u32 m = (u32)(r.A >> ((idx & 7u) * 8u)) & 0xFFu;
    return idx >= 8u ? r.mv8 : m;
}

__device__ inline void append_move(Regs &r, u32 idx, u32 mv) {
    if (idx >= 8u) r.mv8 = mv;
    else r.A |= (u64)mv << (idx * 8u);
}

__device__ inline u32 get_sq(const Regs &r, u32 v) { return (u32)(r.sq >> (v * 4u)) & 0xFu; }

// drop the 9-bit slot that starts at bit `s` and close the gap (list.pop, board.py:56,61)
__device__ inline u64 comps_pop(u64 comps, u32 s) {
    u64 low = (1ull << s) - 1ull;
    return (comps & low) | ((comps >> 9) & ~low);
}

// 9-bit mask of the squares whose sq nibble is odd
__device__ inline u32 odd_mask(u64 sq) {
    u32 x = (u32)sq & 0x11111111u;
    // 4 nibble-LSBs -> 4 adjacent bits: (x & 0x1111) * 0x249 puts bit 4i at 9+i, no carries
    u32 lo = (__umul24(x & 0x1111u, 0x249u) >> 9) & 0xFu;
    u32 hi = (__umul24(x >> 16, 0x249u) >> 9) & 0xFu;
    return lo | (hi << 4) | (((u32)(sq >> 32) & 1u) << 8);
}

// any completed 3-in-a-row in X (bits 0..8) or O (bits 16..24) of w
__device__ inline u32 any_line(u32 w) {
    u32 rows = w & (w >> 1) & (w >> 2) & 0x00490049u;
    u32 cols = w & (w >> 3) & (w >> 6) & 0x00070007u;
    u32 diag = w & (w >> 4) & (w >> 8) & 0x00010001u;
    u32 anti = (w >> 2) & (w >> 4) & (w >> 6) & 0x00010001u;
    return rows | cols | diag | anti;
}

__device__ inline u32 lowbias32(u32 x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

__host__ __device__ inline u64 splitmix64(u64 x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__host__ __device__ inline u64 launch_key(u64 seed, u32 step_idx) {
    return splitmix64(seed ^ ((u64)step_idx * 0xD1B54A32D192ED03ull));
}

__host__ __device__ inline u32 fold_id(u64 board_id) {
    return (u32)board_id ^ ((u32)(board_id >> 32) * 0x9E3779B9u);
}

// One Env.step on the board in registers.  Returns non-zero iff a line exists afterwards.
template <bool AUTO_RESET>
__device__ inline u32 step_board(Regs &r, u32 a, u32 b, u32 bit) {
    if (AUTO_RESET) {
        if (r.done) regs_reset(r);
    }
    u32 lo = min(a, b), hi = max(a, b);
    // board.py:10-15 (+ IndexError for >8, env.py:41): reject before mutating anything
    u32 lo_c = lo & 15u, hi_c = hi & 15u;
    bool valid = (hi < 9u) && (lo != hi) && (((r.cl >> lo_c) | (r.cl >> hi_c)) & 1u) == 0u;
    if (valid) {
        const u32 n = r.n;
        append_move(r, n, lo | (hi << 4));                       // board.py:19
        u32 n1 = n + 1u;
        u32 mlo = (u32)(r.comps >> lo) & SLOT_LSB;               // slot holding lo (board.py:28-33)
        u32 mhi = (u32)(r.comps >> hi) & SLOT_LSB;               // slot holding hi (board.py:35-40)
        u32 both = mlo & mhi;
        bool cyc = both != 0u;                                   // board.py:42
        // x: the square that becomes the child end of the new edge.  On a cycle it is the
        // square the closing move lands on (qeval.py:35), which becomes the root.
        u32 x = cyc ? (bit ? hi : lo) : hi;
        {   // re-root x's tree at x: reverse parent edges along the path x -> old root
            u32 v = x, prev = 0xFu;
            for (int i = 0; i < 9; ++i) {
                u32 sh = v * 4u;
                u32 e = (u32)(r.sq >> sh) & 0xFu;
                r.sq ^= (u64)(e ^ prev) << sh;                   // sq[v] = prev
                if (e == 0xFu) break;
                u32 m = get_move(r, e);
                v = (m & 0xFu) ^ (m >> 4) ^ v;                   // other end of edge e
                prev = e;
            }
        }
        r.sq ^= (u64)(0xFu ^ n) << (x * 4u);                     // sq[x]: 0xF -> n
        if (cyc) {
            // board.py:44-56 + qeval.py:5-51: all squares of the component go classical, each
            // holding its parent edge's round; x holds the closing move's round.
            u32 s = (u32)__builtin_ctz(both);
            u32 comp = (u32)(r.comps >> s) & 0x1FFu;
            r.cl |= comp;
            r.comps = comps_pop(r.comps, s);
            if (__builtin_popcount(r.cl) == 8) {                 // board.py:22-25 autofill
                u32 idx = (u32)__builtin_ctz(~r.cl & 0x1FFu);
                r.sq ^= (u64)(0xFu ^ n1) << (idx * 4u);          // board[idx] = len(moves)
                r.cl |= 1u << idx;
                append_move(r, n1, idx | (idx << 4));
                n1 += 1u;
            }
        } else if (mlo != 0u && mhi != 0u) {                     // board.py:58-61 union, pop(m1)
            u32 s0 = (u32)__builtin_ctz(mlo), s1 = (u32)__builtin_ctz(mhi);
            u64 c1 = (r.comps >> s1) & 0x1FFull;
            r.comps |= c1 << s0;
            r.comps = comps_pop(r.comps, s1);
        } else {                                                 // board.py:62-69
            u32 m = mlo | mhi;
            u32 c = (u32)r.comps;
            u32 s_new = (c & 0x1FFu) == 0u ? 0u
                        : (c & (0x1FFu << 9)) == 0u ? 9u
                        : (c & (0x1FFu << 18)) == 0u ? 18u : 27u;
            u32 s = m ? (u32)__builtin_ctz(m) : s_new;
            r.comps |= (u64)((1u << lo) | (1u << hi)) << s;
        }
        r.n = n1;
    }
    // board.py:71-115 reduced to "does any line exist" (all env.py:49,51 need)
    u32 odd = odd_mask(r.sq);
    u32 w = (r.cl & ~odd) | ((r.cl & odd) << 16);
    u32 win = any_line(w);
    r.done = (win != 0u || r.n > 8u) ? 1u : 0u;                  // env.py:51
    return win;
}

// ------------------------------------------------------------------ kernels
// BPL boards per lane: lane j owns boards [j*BPL, (j+1)*BPL), so every plane is read and written
// with 16-byte (or 2x16-byte) vector accesses that are contiguous across the wave, and the
// fixed per-wave cost (dispatch, address setup, waits) is paid once per BPL boards.
#ifdef QTTT_DEBUG_STAMPS
__device__ u64 *g_debug_stamps = nullptr;   // diagnostic builds only (tools/stepbench --stamps)
#endif

template <typename T, int N>
struct alignas(sizeof(T) * N) Vec {
    T v[N];
};

template <int BPL, bool HAS_BITS, bool AUTO_RESET>
__global__ __launch_bounds__(QTTT_BLOCK) void step_kernel(
    u64 *__restrict__ pA, u64 *__restrict__ pB, u32 *__restrict__ pC,
    const uint16_t *__restrict__ actions, const uint8_t *__restrict__ bits, u32 key_lo,
    u64 board_offset, u32 *__restrict__ reward_bits, uint8_t *__restrict__ terminated,
    int64_t i_begin, int64_t n_groups) {
    int64_t j = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
#ifdef QTTT_DEBUG_STAMPS
    const u64 st0 = __builtin_amdgcn_s_memrealtime();
#endif
    if (j >= n_groups) return;
    const int64_t i0 = i_begin + j * BPL;
    typedef Vec<u64, BPL> V64;
    typedef Vec<u32, BPL> V32;
    typedef Vec<uint16_t, BPL> V16;
    typedef Vec<uint8_t, BPL> V8;
    V64 a = *reinterpret_cast<const V64 *>(pA + i0);
    V64 b = *reinterpret_cast<const V64 *>(pB + i0);
    V32 c = *reinterpret_cast<const V32 *>(pC + i0);
    V16 act = *reinterpret_cast<const V16 *>(actions + i0);
    V8 bt;
    if (HAS_BITS) bt = *reinterpret_cast<const V8 *>(bits + i0);
    V32 rw;
    V8 tm;
#ifdef QTTT_DEBUG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u64 st1 = __builtin_amdgcn_s_memrealtime();
#endif
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        Regs r;
        unpack(a.v[k], b.v[k], c.v[k], r);
        u32 bit;
        if (HAS_BITS) bit = bt.v[k] & 1u;
        else bit = lowbias32(fold_id(board_offset + (u64)(i0 + k)) ^ key_lo) >> 31;
        u32 av = act.v[k];
        u32 win = step_board<AUTO_RESET>(r, av & 0xFFu, av >> 8, bit);
        pack(r, a.v[k], b.v[k], c.v[k]);
        rw.v[k] = win ? 0xBF800000u : 0x80000000u;               // env.py:49: -1.0f / -0.0f
        tm.v[k] = (uint8_t)r.done;
    }
#ifdef QTTT_DEBUG_STAMPS
    const u64 st2 = __builtin_amdgcn_s_memrealtime();
#endif
    *reinterpret_cast<V64 *>(pA + i0) = a;
    *reinterpret_cast<V64 *>(pB + i0) = b;
    *reinterpret_cast<V32 *>(pC + i0) = c;
    *reinterpret_cast<V32 *>(reward_bits + i0) = rw;
    *reinterpret_cast<V8 *>(terminated + i0) = tm;
#ifdef QTTT_DEBUG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u64 st3 = __builtin_amdgcn_s_memrealtime();
    if (g_debug_stamps && (threadIdx.x & 63) == 0) {
        u64 *o = g_debug_stamps + ((int64_t)blockIdx.x * (QTTT_BLOCK / 64) + (threadIdx.x >> 6)) * 4;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
    }
#endif
}

// Persistent, software-pipelined form of step_kernel: the grid is sized to a fixed number of
// waves per SIMD and every lane walks lane-groups j, j+stride, ...; the loads of the NEXT group
// are issued before the current group is computed, so a wave always has memory traffic in
// flight while its VALU work runs, and waves drift out of phase instead of all loading, then all
// computing, then all storing (measured: DESIGN.md §6).
template <int BPL, bool HAS_BITS>
struct Tile {
    Vec<u64, BPL> a, b;
    Vec<u32, BPL> c;
    Vec<uint16_t, BPL> act;
    Vec<uint8_t, BPL> bt;
};

template <int BPL, bool HAS_BITS, bool AUTO_RESET>
__global__ __launch_bounds__(QTTT_BLOCK) void step_kernel_pipe(
    u64 *__restrict__ pA, u64 *__restrict__ pB, u32 *__restrict__ pC,
    const uint16_t *__restrict__ actions, const uint8_t *__restrict__ bits, u32 key_lo,
    u64 board_offset, u32 *__restrict__ reward_bits, uint8_t *__restrict__ terminated,
    int64_t n_groups) {
    const int64_t stride = (int64_t)gridDim.x * QTTT_BLOCK;
    int64_t j = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    if (j >= n_groups) return;
    typedef Vec<u64, BPL> V64;
    typedef Vec<u32, BPL> V32;
    typedef Vec<uint16_t, BPL> V16;
    typedef Vec<uint8_t, BPL> V8;
    Tile<BPL, HAS_BITS> cur, nxt;
    auto load = [&](Tile<BPL, HAS_BITS> &t, int64_t g) {
        const int64_t i0 = g * BPL;
        t.a = *reinterpret_cast<const V64 *>(pA + i0);
        t.b = *reinterpret_cast<const V64 *>(pB + i0);
        t.c = *reinterpret_cast<const V32 *>(pC + i0);
        t.act = *reinterpret_cast<const V16 *>(actions + i0);
        if (HAS_BITS) t.bt = *reinterpret_cast<const V8 *>(bits + i0);
    };
    auto process = [&](Tile<BPL, HAS_BITS> &t, int64_t g) {
        const int64_t i0 = g * BPL;
        V32 rw;
        V8 tm;
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            Regs r;
            unpack(t.a.v[k], t.b.v[k], t.c.v[k], r);
            u32 bit;
            if (HAS_BITS) bit = t.bt.v[k] & 1u;
            else bit = lowbias32(fold_id(board_offset + (u64)(i0 + k)) ^ key_lo) >> 31;
            u32 av = t.act.v[k];
            u32 win = step_board<AUTO_RESET>(r, av & 0xFFu, av >> 8, bit);
            pack(r, t.a.v[k], t.b.v[k], t.c.v[k]);
            rw.v[k] = win ? 0xBF800000u : 0x80000000u;
            tm.v[k] = (uint8_t)r.done;
        }
        *reinterpret_cast<V64 *>(pA + i0) = t.a;
        *reinterpret_cast<V64 *>(pB + i0) = t.b;
        *reinterpret_cast<V32 *>(pC + i0) = t.c;
        *reinterpret_cast<V32 *>(reward_bits + i0) = rw;
        *reinterpret_cast<V8 *>(terminated + i0) = tm;
    };
    // Ping-pong between two register tiles (no copies).  The first tile is peeled so that both
    // ways into the loop header carry the same outstanding-memory-op pattern (4 loads, then the
    // previous tile's 5 stores); otherwise the compiler's merged s_waitcnt makes every tile wait
    // for the previous tile's stores.  Prefetches are unconditional (clamped index) for the same
    // reason: a branch around them merges to "wait for everything".
    int64_t j1 = j + stride;
    bool more = j1 < n_groups;
    load(cur, j);
    load(nxt, more ? j1 : j);
    process(cur, j);
    if (!more) return;
    for (;;) {
        const int64_t j2 = j1 + stride;
        const bool m2 = j2 < n_groups;
        load(cur, m2 ? j2 : j1);
        process(nxt, j1);
        if (!m2) break;
        const int64_t j3 = j2 + stride;
        const bool m3 = j3 < n_groups;
        load(nxt, m3 ? j3 : j2);
        process(cur, j2);
        if (!m3) break;
        j1 = j3;
    }
}

__global__ __launch_bounds__(QTTT_BLOCK) void reset_kernel(u64 *pA, u64 *pB, u32 *pC, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    if (i >= n) return;
    pA[i] = 0;
    pB[i] = SQ_EMPTY;
    pC[i] = 0;
}

__global__ __launch_bounds__(QTTT_BLOCK) void observe_kernel(
    const u64 *pA, const u64 *pB, const u32 *pC, int8_t *classical, uint8_t *q_p1,
    uint8_t *q_p1_len, uint8_t *q_p2, uint8_t *q_p2_len, uint8_t *turn, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    if (i >= n) return;
    Regs r;
    unpack(pA[i], pB[i], pC[i], r);
    for (u32 v = 0; v < 9; ++v)                                   // env.py:71,82
        classical[i * 9 + v] = (r.cl >> v & 1u) ? (int8_t)get_sq(r, v) : (int8_t)-1;
    u32 n1 = 0, n2 = 0;
    for (u32 t = 0; t < 9; ++t) {                                 // env.py:72-77
        u32 m = get_move(r, t);
        u32 lo = m & 0xFu, hi = m >> 4;
        bool live = t < r.n && !(r.cl >> lo & 1u);                // round t not on the board
        if (live && (t & 1u)) {
            q_p2[i * 8 + n2 * 2] = (uint8_t)lo;
            q_p2[i * 8 + n2 * 2 + 1] = (uint8_t)hi;
            ++n2;
        } else if (live) {
            q_p1[i * 10 + n1 * 2] = (uint8_t)lo;
            q_p1[i * 10 + n1 * 2 + 1] = (uint8_t)hi;
            ++n1;
        }
    }
    for (u32 k = n1; k < 5; ++k) q_p1[i * 10 + k * 2] = q_p1[i * 10 + k * 2 + 1] = 255;
    for (u32 k = n2; k < 4; ++k) q_p2[i * 8 + k * 2] = q_p2[i * 8 + k * 2 + 1] = 255;
    q_p1_len[i] = (uint8_t)n1;
    q_p2_len[i] = (uint8_t)n2;
    turn[i] = (uint8_t)(r.n & 1u);                                // env.py:83
}

__device__ inline void check_win_regs(const Regs &r, int &p1, int &p2) {
    // board.py:71-115, lines in the reference's order (rows, cols, 2-4-6, 0-4-8)
    const u32 lines[8] = {0x007u, 0x038u, 0x1C0u, 0x049u, 0x092u, 0x124u, 0x054u, 0x111u};
    u32 odd = odd_mask(r.sq);
    u32 X = r.cl & ~odd, O = r.cl & odd;
    p1 = 10;
    p2 = 10;
    for (int l = 0; l < 8; ++l) {
        u32 L = lines[l];
        int mx = -1;
        for (u32 v = 0; v < 9; ++v)
            if (L >> v & 1u) mx = max(mx, (int)get_sq(r, v));
        if ((X & L) == L) p1 = min(p1, mx);
        else if ((O & L) == L) p2 = min(p2, mx);
    }
    if (p1 >= 10) p1 = -1;
    if (p2 >= 10) p2 = -1;
}

__global__ __launch_bounds__(QTTT_BLOCK) void check_win_kernel(
    const u64 *pA, const u64 *pB, const u32 *pC, int8_t *p1_round, int8_t *p2_round, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    if (i >= n) return;
    Regs r;
    unpack(pA[i], pB[i], pC[i], r);
    int p1, p2;
    check_win_regs(r, p1, p2);
    p1_round[i] = (int8_t)p1;
    p2_round[i] = (int8_t)p2;
}

__global__ __launch_bounds__(QTTT_BLOCK) void export_kernel(
    const u64 *pA, const u64 *pB, const u32 *pC, uint8_t *moves, uint8_t *n_moves,
    int8_t *board, uint16_t *qmask, uint8_t *n_q, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    if (i >= n) return;
    Regs r;
    unpack(pA[i], pB[i], pC[i], r);
    for (u32 t = 0; t < 9; ++t) {
        u32 m = get_move(r, t);
        bool used = t < r.n;
        moves[i * 18 + t * 2] = used ? (uint8_t)(m & 0xFu) : (uint8_t)255;
        moves[i * 18 + t * 2 + 1] = used ? (uint8_t)(m >> 4) : (uint8_t)255;
    }
    n_moves[i] = (uint8_t)r.n;
    for (u32 v = 0; v < 9; ++v)
        board[i * 9 + v] = (r.cl >> v & 1u) ? (int8_t)get_sq(r, v) : (int8_t)-1;
    u32 nq = 0;
    for (u32 k = 0; k < 4; ++k) {
        u32 c = (u32)(r.comps >> (9u * k)) & 0x1FFu;
        qmask[i * 4 + k] = (uint16_t)c;
        nq += c != 0u;
    }
    n_q[i] = (uint8_t)nq;
}

// Builds the packed state (incl. the rooted forest) from Board attributes assigned by a caller
// (mcts.py:11-17,241 assign .board/.moves/.qstructs directly).  Not a hot path.
__global__ __launch_bounds__(QTTT_BLOCK) void import_kernel(
    u64 *pA, u64 *pB, u32 *pC, const uint8_t *moves, const uint8_t *n_moves, const int8_t *board,
    const uint16_t *qmask, const uint8_t *n_q, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    if (i >= n) return;
    Regs r;
    regs_reset(r);
    u32 nm = min((u32)n_moves[i], 9u);
    r.n = nm;
    for (u32 t = 0; t < nm; ++t)
        append_move(r, t, (u32)(moves[i * 18 + t * 2] & 0xFu) | ((u32)(moves[i * 18 + t * 2 + 1] & 0xFu) << 4));
    for (u32 v = 0; v < 9; ++v) {
        int bv = board[i * 9 + v];
        if (bv >= 0) {
            r.cl |= 1u << v;
            r.sq ^= (u64)(0xFu ^ ((u32)bv & 0xFu)) << (v * 4u);
        }
    }
    u32 nq = min((u32)n_q[i], 4u);
    for (u32 k = 0; k < nq; ++k) r.comps |= (u64)(qmask[i * 4 + k] & 0x1FFu) << (9u * k);
    // root every tree of live edges: grow from the lowest square of each tree
    u32 rooted = 0;
    for (int pass = 0; pass < 9; ++pass) {
        bool grew = false;
        for (u32 t = 0; t < nm; ++t) {
            u32 m = get_move(r, t);
            u32 lo = m & 0xFu, hi = m >> 4;
            if (lo == hi || lo > 8u || hi > 8u || (r.cl >> lo & 1u) || (r.cl >> hi & 1u)) continue;
            bool rl = rooted >> lo & 1u, rh = rooted >> hi & 1u;
            if (rl && !rh) { r.sq ^= (u64)(get_sq(r, hi) ^ t) << (hi * 4u); rooted |= 1u << hi; grew = true; }
            else if (rh && !rl) { r.sq ^= (u64)(get_sq(r, lo) ^ t) << (lo * 4u); rooted |= 1u << lo; grew = true; }
        }
        if (!grew) {
            // start a new tree at the lowest un-rooted square that has a live edge
            u32 cand = 0;
            for (u32 t = 0; t < nm; ++t) {
                u32 m = get_move(r, t);
                u32 lo = m & 0xFu, hi = m >> 4;
                if (lo == hi || lo > 8u || hi > 8u || (r.cl >> lo & 1u) || (r.cl >> hi & 1u)) continue;
                cand |= (1u << lo) | (1u << hi);
            }
            cand &= ~rooted;
            if (cand == 0u) break;
            rooted |= cand & (0u - cand);
        }
    }
    int p1, p2;
    check_win_regs(r, p1, p2);
    r.done = (p1 > 0 || p2 > 0 || r.n > 8u) ? 1u : 0u;
    u64 A, B;
    u32 C;
    pack(r, A, B, C);
    pA[i] = A;
    pB[i] = B;
    pC[i] = C;
}

// legal pairs in ind2move order: for lo ascending, hi ascending (mcts.py:20-27, 339-343)
__global__ __launch_bounds__(QTTT_BLOCK) void sample_actions_kernel(
    const u64 *pA, const u64 *pB, const u32 *pC, u32 key_lo, u32 key_hi, u64 board_offset,
    u32 auto_reset, uint8_t *actions, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    if (i >= n) return;
    Regs r;
    unpack(pA[i], pB[i], pC[i], r);
    u32 cl = (auto_reset && r.done) ? 0u : r.cl;
    u32 empty = ~cl & 0x1FFu;
    u32 e = (u32)__builtin_popcount(empty);
    u32 n_legal = e * (e - 1u) / 2u;
    u32 lo = 0, hi = 0;
    if (n_legal != 0u) {
        u32 h1 = lowbias32(fold_id(board_offset + (u64)i) ^ key_lo);
        u32 h2 = lowbias32(h1 ^ key_hi);
        u32 k = __umulhi(h2, n_legal);
        // walk the empty squares: the j-th empty square (ascending) pairs with the e-1-j later ones
        u32 rest = empty, left = e;
        for (int it = 0; it < 9; ++it) {
            u32 v = (u32)__builtin_ctz(rest);
            rest &= rest - 1u;
            left -= 1u;
            if (k < left) {
                lo = v;
                u32 rr = rest;
                for (u32 j = 0; j < k; ++j) rr &= rr - 1u;
                hi = (u32)__builtin_ctz(rr);
                break;
            }
            k -= left;
        }
    }
    actions[i * 2] = (uint8_t)lo;
    actions[i * 2 + 1] = (uint8_t)hi;
}

// tuning knobs (bench / profiling): boards per lane (1|2|4) and, for the persistent pipelined
// form, waves per SIMD (0 = plain one-shot grid).  Initialised from QTTT_STEP_BPL /
// QTTT_STEP_PIPE, changeable at run time through qttt_set_tuning().
struct Tuning {
    int bpl;
    int pipe;
};
inline Tuning &tuning() {
    static Tuning t = [] {
        Tuning v{QTTT_DEFAULT_BPL, QTTT_DEFAULT_PIPE};
        if (const char *e = getenv("QTTT_STEP_BPL")) { int k = atoi(e); if (k == 1 || k == 2 || k == 4) v.bpl = k; }
        if (const char *e = getenv("QTTT_STEP_PIPE")) { int k = atoi(e); if (k >= 0 && k <= 8) v.pipe = k; }
        return v;
    }();
    return t;
}
inline int step_bpl_override() { return tuning().bpl; }
inline int step_pipe_override() { return tuning().pipe; }

inline int grid_for(int64_t n) { return (int)((n + QTTT_BLOCK - 1) / QTTT_BLOCK); }

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

}  // namespace

// ====================================================================== C ABI
extern "C" {

int qttt_abi_version(void) { return QTTT_ABI_VERSION; }

#ifdef QTTT_DEBUG_STAMPS
int qttt_debug_set_stamps(void *buf) {
    u64 *p = (u64 *)buf;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_debug_stamps), &p, sizeof(p));
}
#endif

int qttt_set_tuning(int boards_per_lane, int pipe_waves_per_simd) {
    if (!(boards_per_lane == 1 || boards_per_lane == 2 || boards_per_lane == 4)) return QTTT_ERR_SIZE;
    if (pipe_waves_per_simd < 0 || pipe_waves_per_simd > 8) return QTTT_ERR_SIZE;
    tuning().bpl = boards_per_lane;
    tuning().pipe = pipe_waves_per_simd;
    return 0;
}

int64_t qttt_state_bytes(int64_t n) { return n < 0 ? (int64_t)QTTT_ERR_SIZE : plane_stride(n) * 20; }

uint64_t qttt_hash(uint64_t seed, uint64_t board_id, uint32_t step_idx) {
    u64 key = launch_key(seed, step_idx);
    u32 x = fold_id(board_id) ^ (u32)key;
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    u32 h1 = x;
    x = h1 ^ (u32)(key >> 32);
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return ((u64)x << 32) | h1;
}

int qttt_reset(void *state, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state) return QTTT_ERR_NULL;
    Planes p = planes(state, n);
    hipLaunchKernelGGL(reset_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0, (hipStream_t)stream,
                       p.A, p.B, p.C, n);
    return launch_status();
}

int qttt_step(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
              uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
              uint8_t *terminated, int64_t n, void *stream) {
    if (n < 0 || board_offset < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !actions || !reward || !terminated) return QTTT_ERR_NULL;
    if ((uintptr_t)actions & 1u) return QTTT_ERR_ACTION;   // actions are read as u16 pairs
    Planes p = planes(state, n);
    u32 key_lo = (u32)launch_key(seed, step_idx);
    hipStream_t s = (hipStream_t)stream;
    const uint16_t *a16 = reinterpret_cast<const uint16_t *>(actions);
    u32 *rb = reinterpret_cast<u32 *>(reward);
    const bool ar = (flags & QTTT_FLAG_AUTO_RESET) != 0;
    // widest boards-per-lane the caller's pointers are aligned for (planes always are)
    int bpl = step_bpl_override();
    auto aligned = [&](int k) {
        return ((uintptr_t)actions % (2u * k)) == 0 && ((uintptr_t)reward % (4u * k)) == 0 &&
               ((uintptr_t)terminated % (unsigned)k) == 0 && (!bits || ((uintptr_t)bits % (unsigned)k) == 0);
    };
    while (bpl > 1 && !aligned(bpl)) bpl >>= 1;
    const int64_t n_groups = n / bpl, n_main = n_groups * bpl;
#define QTTT_LAUNCH(BPL, HB, AR, I0, NG)                                                        \
    hipLaunchKernelGGL((step_kernel<BPL, HB, AR>), dim3(grid_for(NG)), dim3(QTTT_BLOCK), 0, s,  \
                       p.A, p.B, p.C, a16, bits, key_lo, (u64)board_offset, rb, terminated,     \
                       (int64_t)(I0), (int64_t)(NG))
#define QTTT_DISPATCH(BPL, I0, NG)                                   \
    do {                                                             \
        if (bits) { if (ar) QTTT_LAUNCH(BPL, true, true, I0, NG); else QTTT_LAUNCH(BPL, true, false, I0, NG); } \
        else      { if (ar) QTTT_LAUNCH(BPL, false, true, I0, NG); else QTTT_LAUNCH(BPL, false, false, I0, NG); } \
    } while (0)
    const int pipe = step_pipe_override();      // waves per SIMD of the persistent form, 0 = off
    if (n_groups > 0 && pipe > 0) {
        int64_t blocks = (n_groups + QTTT_BLOCK - 1) / QTTT_BLOCK;
        const int64_t cap = (int64_t)256 * pipe;  // 256 CUs x (pipe waves/SIMD x 4 SIMDs / 4 waves per block)
        if (blocks > cap) blocks = cap;
#define QTTT_LAUNCH_P(BPL, HB, AR)                                                               \
    hipLaunchKernelGGL((step_kernel_pipe<BPL, HB, AR>), dim3((unsigned)blocks), dim3(QTTT_BLOCK), 0, s, \
                       p.A, p.B, p.C, a16, bits, key_lo, (u64)board_offset, rb, terminated, n_groups)
#define QTTT_DISPATCH_P(BPL)                                          \
    do {                                                              \
        if (bits) { if (ar) QTTT_LAUNCH_P(BPL, true, true); else QTTT_LAUNCH_P(BPL, true, false); } \
        else      { if (ar) QTTT_LAUNCH_P(BPL, false, true); else QTTT_LAUNCH_P(BPL, false, false); } \
    } while (0)
        if (bpl == 4) QTTT_DISPATCH_P(4);
        else if (bpl == 2) QTTT_DISPATCH_P(2);
        else QTTT_DISPATCH_P(1);
#undef QTTT_DISPATCH_P
#undef QTTT_LAUNCH_P
    } else if (n_groups > 0) {
        if (bpl == 4) QTTT_DISPATCH(4, 0, n_groups);
        else if (bpl == 2) QTTT_DISPATCH(2, 0, n_groups);
        else QTTT_DISPATCH(1, 0, n_groups);
    }
    if (n_main < n) QTTT_DISPATCH(1, n_main, n - n_main);        // ragged tail, one board per lane
#undef QTTT_DISPATCH
#undef QTTT_LAUNCH
    return launch_status();
}

int qttt_step_many(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
                   uint32_t step_idx0, int64_t board_offset, uint32_t flags, float *reward,
                   uint8_t *terminated, int64_t out_stride, int64_t n, int32_t n_steps,
                   void *stream) {
    if (n_steps < 0 || out_stride < 0) return QTTT_ERR_SIZE;
    for (int32_t t = 0; t < n_steps; ++t) {
        int rc = qttt_step(state, actions + (int64_t)t * 2 * n, bits ? bits + (int64_t)t * n : nullptr,
                           seed, step_idx0 + (uint32_t)t, board_offset, flags,
                           reward + (int64_t)t * out_stride, terminated + (int64_t)t * out_stride,
                           n, stream);
        if (rc != 0) return rc;
    }
    return 0;
}

int qttt_observe(const void *state, int8_t *classical, uint8_t *q_p1, uint8_t *q_p1_len,
                 uint8_t *q_p2, uint8_t *q_p2_len, uint8_t *turn, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !classical || !q_p1 || !q_p1_len || !q_p2 || !q_p2_len || !turn) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    hipLaunchKernelGGL(observe_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0, (hipStream_t)stream,
                       p.A, p.B, p.C, classical, q_p1, q_p1_len, q_p2, q_p2_len, turn, n);
    return launch_status();
}

int qttt_check_win(const void *state, int8_t *p1_round, int8_t *p2_round, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !p1_round || !p2_round) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    hipLaunchKernelGGL(check_win_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0, (hipStream_t)stream,
                       p.A, p.B, p.C, p1_round, p2_round, n);
    return launch_status();
}

int qttt_export(const void *state, uint8_t *moves, uint8_t *n_moves, int8_t *board,
                uint16_t *qmask, uint8_t *n_q, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !moves || !n_moves || !board || !qmask || !n_q) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    hipLaunchKernelGGL(export_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0, (hipStream_t)stream,
                       p.A, p.B, p.C, moves, n_moves, board, qmask, n_q, n);
    return launch_status();
}

int qttt_import(void *state, const uint8_t *moves, const uint8_t *n_moves, const int8_t *board,
                const uint16_t *qmask, const uint8_t *n_q, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !moves || !n_moves || !board || !qmask || !n_q) return QTTT_ERR_NULL;
    Planes p = planes(state, n);
    hipLaunchKernelGGL(import_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0, (hipStream_t)stream,
                       p.A, p.B, p.C, moves, n_moves, board, qmask, n_q, n);
    return launch_status();
}

int qttt_sample_actions(const void *state, uint64_t seed, uint32_t step_idx, int64_t board_offset,
                        uint32_t flags, uint8_t *actions, int64_t n, void *stream) {
    if (n < 0 || board_offset < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !actions) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    u64 key = launch_key(seed, step_idx);
    hipLaunchKernelGGL(sample_actions_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0,
                       (hipStream_t)stream, p.A, p.B, p.C, (u32)key, (u32)(key >> 32),
                       (u64)board_offset, (u32)((flags & QTTT_FLAG_AUTO_RESET) != 0), actions, n);
    return launch_status();
}

}  // extern "C"
