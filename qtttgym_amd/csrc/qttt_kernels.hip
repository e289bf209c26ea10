// qttt_kernels.hip — gfx950 (MI355X / CDNA4) kernels + the C ABI of include/qttt.h.
//
// Mapping: ONE LANE PER BOARD (64 boards per wavefront, BPL consecutive boards per lane),
// everything in VGPRs, structure-of-arrays state so that every load/store of a wave is one
// contiguous 16-byte-per-lane segment.  No MFMA; LDS holds one lookup table (and, in the kernels
// that return the observation, the output tiles).  DESIGN.md §2 explains why the wave-per-board
// mapping was rejected after measurement and why the kernel is written for minimum VALU
// *instruction count* (measured issue cost ~4 cycles per wave-instruction for this instruction mix,
// tools/valu_rates.cpp).
//
// Formulation — deliberately NOT the reference's algorithm:
//   * the un-collapsed moves of a board form a forest on the 9 squares (a move that closes a
//     cycle collapses its whole component at once, board.py:42-56).  The forest is kept ROOTED:
//     nibble sq[v] of a non-classical square v is the round of the move joining v to its parent
//     (root / isolated = none).  For a classical square, sq[v] is the round that landed there
//     (= Board.board[v]).
//   * QEvalClassic.eval (qeval.py:5-51: leaf-peel + forced walk round the cycle) is equivalent
//     to: re-root the tree at the square t the closing move lands on (bit picks lo/hi), then
//     every other square of the component receives its parent edge.  So a collapse is one path
//     reversal + `classical |= component`; no per-edge work.
//   * Every move ever played is therefore HELD by exactly one square c (sq[c] = its round): the
//     child end of an un-collapsed move, the landing square of a collapsed one.  The move itself
//     is then (c, c ^ x) with x = lo ^ hi, so the state stores only the 4-bit x of each move — the
//     re-rooting walk needs nothing else ("other end of edge e" = v ^ x_e) and the cold kernels
//     rebuild Board.moves from the holders.
//   * Board.qstructs (board.py:6) is cached as 4 slots x 9-bit square masks, in the reference's
//     list order, so "same component?" is two shifts and an AND.
//
// Packed state, 16 B/board = 39 algorithmic bytes per step, planes
// P[s] u64 | Q[s] u64 (s = n rounded up to 64).  The all-zero state is the empty board.
//   P bits [2,38)  nine nibbles, square v at bits [4v+2, 4v+6), COMPLEMENT-coded: 0 = root /
//                  isolated / empty, round e is stored as 15-e.  The 2-bit offset makes
//                  `(P >> 4v) & 0x3C` the code times four, the unit every shift amount below wants.
//   P1 = P >> 32:  [0,6) nibbles | [6,8) 0 | [8,12) n = moves PLAYED | [12,16) comps bits 32..35 |
//                  [16,20) x of the last move | [20,22) 0 | [22,31) classical mask | 31 done
//   Q0:            x = lo^hi of the moves of rounds 0..7: round e in the nibble at bit
//                  (4(7-e)+2) mod 32, so that rotating Q0 right by four times the CODE of e
//                  (4(15-e) = 4(7-e) mod 32) lands 4x on bits 2..5.  The move of round 8 can only
//                  be the last one of a game: its x is the `last x` field of P1 (it is also XORed
//                  onto round 0's nibble, where it is harmless: the game is over; the cold
//                  kernels undo it).
//   Q1:            comps bits 0..31 (comps = 4 x 9-bit masks, slot k at bit 9k, list order, compact)
//   The autofill of board.py:22-25 is IMPLICIT: a board with exactly 8 classical squares stands
//   for the reference state in which the 9th square holds round 8 and moves ends with (idx,idx,8)
//   (the autofill round is always 8); the cold kernels materialise it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "qttt.h"

typedef unsigned long long u64;
typedef unsigned int u32;

#ifndef QTTT_BLOCK
#define QTTT_BLOCK 512
#endif
#define QTTT_DEFAULT_BPL 2
#define QTTT_STATE_BYTES 16

namespace {

constexpr u32 SLOT_LSB = 0x08040201u;      // bit 0 of each 9-bit comps slot

// P1 = high word of plane P
constexpr u32 P1_N_SHIFT = 8, P1_CHI_SHIFT = 12, P1_LX_SHIFT = 16, P1_CL_SHIFT = 22;
constexpr u32 P1_DONE = 0x80000000u;

struct Planes {
    u64 *P;
    u64 *Q;
};

// plane stride: n rounded up to 64 boards, so every plane starts 512-byte aligned
__host__ __device__ inline int64_t plane_stride(int64_t n) { return (n + 63) & ~(int64_t)63; }

__host__ __device__ inline Planes planes(void *state, int64_t n) {
    Planes p;
    p.P = reinterpret_cast<u64 *>(state);
    p.Q = p.P + plane_stride(n);
    return p;
}

template <typename T, int N>
struct alignas(sizeof(T) * N) Vec {
    T v[N];
};

// same-size raw integer type for a Vec, so cache-policy builtins (which want scalars / ext vectors)
// can be applied to it
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x8 __attribute__((ext_vector_type(8)));
template <int BYTES> struct RawOf;
template <> struct RawOf<1> { typedef uint8_t type; };
template <> struct RawOf<2> { typedef uint16_t type; };
template <> struct RawOf<4> { typedef u32 type; };
template <> struct RawOf<8> { typedef u32x2 type; };
template <> struct RawOf<16> { typedef u32x4 type; };
template <> struct RawOf<32> { typedef u32x8 type; };

// Every access of the step kernel is streaming within a launch (each byte is touched once) and L2
// contents do not survive to the next launch, so all of them carry the non-temporal hint
// (measured, stores only: nt 7.6 / sc1 7.9 / plain 8.2 us per 1 M-board launch)
template <typename V>
__device__ __forceinline__ V load_stream(const V *p) {
    typedef typename RawOf<sizeof(V)>::type R;
    R r = __builtin_nontemporal_load(reinterpret_cast<const R *>(p));
    V v;
    __builtin_memcpy(&v, &r, sizeof(V));
    return v;
}
template <typename V>
__device__ __forceinline__ void store_stream(V *p, const V &v) {
    typedef typename RawOf<sizeof(V)>::type R;
    R r;
    __builtin_memcpy(&r, &v, sizeof(V));
    __builtin_nontemporal_store(r, reinterpret_cast<R *>(p));
}

__device__ __forceinline__ u32 rotr32(u32 x, u32 s) { return __builtin_amdgcn_alignbit(x, x, s); }
// v_ffbl_b32 as the hardware defines it: index of the lowest set bit, 0xFFFFFFFF for 0
__device__ __forceinline__ u32 ffbl_raw(u32 x) {
    u32 r;
    asm("v_ffbl_b32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

#ifdef QTTT_DEBUG_STAMPS
__device__ u64 *g_debug_stamps = nullptr;   // diagnostic builds only (tools/stepbench stamps)
#endif

// ------------------------------------------------------------------ 3-in-a-row lookup table
// line_lut[m] = 0x7F iff the 9-bit square mask m contains one of the 8 lines of board.py:85-110.
// The LDS copy keeps one entry per DWORD (LINE_LUT_BYTES = 2 KB), because every mask of the hot
// path lives "times four" (the nibbles sit at bit 4v+2): the byte offset into the table is the
// mask itself, no shift.
__host__ __device__ constexpr bool mask_has_line(u32 m) {
    return (m & 0x007u) == 0x007u || (m & 0x038u) == 0x038u || (m & 0x1C0u) == 0x1C0u ||
           (m & 0x049u) == 0x049u || (m & 0x092u) == 0x092u || (m & 0x124u) == 0x124u ||
           (m & 0x054u) == 0x054u || (m & 0x111u) == 0x111u;
}

struct LineLut {
    uint8_t b[512];
    constexpr LineLut() : b() {
        for (u32 m = 0; m < 512; ++m) b[m] = mask_has_line(m) ? 0x7F : 0;   // 0x7F << 23 = 1.0f
    }
};
__constant__ LineLut g_line_lut = LineLut();
constexpr u32 LINE_LUT_BYTES = 2048;

// The LDS copy is COMPUTED (thread w makes entry w, a dozen instructions once per launch), not
// loaded: a global load in front of the workgroup barrier would tie the barrier — and with it every
// wave of the workgroup — to the slowest wave's state loads (its `s_waitcnt vmcnt(0)` covers them
// too).  Computed, the barrier is passed while the loads are still in flight and every wave then
// waits for its own data only: 7.2 – 7.4 against 7.45 – 7.6 us per 1 M boards, 3.8 against 4.05 us
// at 262 144 (tools/stepbench, interleaved).
__device__ __forceinline__ u32 line_lut_entry(u32 m) {
    const u32 rows = m & (m >> 1) & (m >> 2) & 0x049u;                 // 0-1-2, 3-4-5, 6-7-8
    const u32 cols = m & (m >> 3) & (m >> 6) & 0x007u;                 // 0-3-6, 1-4-7, 2-5-8
    const bool diag = (m & 0x111u) == 0x111u || (m & 0x054u) == 0x054u;
    return ((rows | cols) != 0u || diag) ? 0x7Fu : 0u;
}
template <int BLOCK>
__device__ inline void fill_line_lut_nosync(uint8_t *lut) {
    for (u32 w = threadIdx.x; w < 512u; w += BLOCK) reinterpret_cast<u32 *>(lut)[w] = line_lut_entry(w);
}
template <int BLOCK>
__device__ inline void fill_line_lut(uint8_t *lut) {
    fill_line_lut_nosync<BLOCK>(lut);
    __syncthreads();
}

// ------------------------------------------------------------------ counter hash (the build's
// synthetic-input spec)
__host__ __device__ inline u32 lowbias32(u32 x) {
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
    u32 f = (u32)board_id;
    u32 h = (u32)(board_id >> 32);
    if (h) f ^= h * 0x9E3779B9u;           // never taken below 2^32 boards: no multiply on the hot path
    return f;
}
// top bit of lowbias32(x): the final xor-shift cannot change bit 31, so it is skipped
__device__ inline u32 collapse_bit_of(u32 x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    return x >> 31;
}

// ---- uniform-legal policy tables (GameState.actions rule, mcts.py:20-27, in ind2move order) ----
// rank_pair[e][k]: the k-th pair (i < j) of e items in lexicographic order, as i | j<<4.
// nth5[m][r] / nth4[m][r]: index of the r-th set bit of a 5-bit / 4-bit mask: the r-th set bit of the
// 9-bit empty-square mask is looked up in its low five bits or, past their population, in its
// high four.  584 bytes in all, so that filling it per workgroup costs next to nothing.
struct PolicyLut {
    uint8_t rank_pair[10 * 36];
    uint8_t nth5[32 * 5];
    uint8_t nth4[16 * 4];
    constexpr PolicyLut() : rank_pair(), nth5(), nth4() {
        for (int e = 0; e < 10; ++e) {
            int k = 0;
            for (int i = 0; i < e; ++i)
                for (int j = i + 1; j < e; ++j) rank_pair[e * 36 + k++] = (uint8_t)(i | (j << 4));
            for (; k < 36; ++k) rank_pair[e * 36 + k] = 0;
        }
        for (int m = 0; m < 32; ++m) {
            int r = 0;
            for (int v = 0; v < 5; ++v)
                if (m >> v & 1) nth5[m * 5 + r++] = (uint8_t)v;
            for (; r < 5; ++r) nth5[m * 5 + r] = 0;
        }
        for (int m = 0; m < 16; ++m) {
            int r = 0;
            for (int v = 0; v < 4; ++v)
                if (m >> v & 1) nth4[m * 4 + r++] = (uint8_t)(5 + v);
            for (; r < 4; ++r) nth4[m * 4 + r] = 0;
        }
    }
};
__constant__ PolicyLut g_policy_lut = PolicyLut();
constexpr u32 POLICY_LUT_WORDS = (10 * 36 + 32 * 5 + 16 * 4) / 4;
constexpr u32 POLICY_NTH5 = 360, POLICY_NTH4 = 360 + 160;

template <int BLOCK>
__device__ inline void fill_policy_lut(uint8_t *dst) {
    const u32 *src = reinterpret_cast<const u32 *>(&g_policy_lut);
    for (u32 w = threadIdx.x; w < POLICY_LUT_WORDS; w += BLOCK) reinterpret_cast<u32 *>(dst)[w] = src[w];
}

// the r-th (0-based) set bit of the 9-bit mask `m`
__device__ __forceinline__ u32 policy_nth(const uint8_t *plut, u32 m, u32 c5, u32 r) {
    return r < c5 ? (u32)plut[POLICY_NTH5 + (m & 31u) * 5u + r] : (u32)plut[POLICY_NTH4 + (m >> 5) * 4u + (r - c5)];
}

// the policy's action for a board whose empty-square mask is `empty`, from hash word h2: lo | hi<<8
__device__ __forceinline__ u32 policy_action(const uint8_t *plut, u32 empty, u32 h2) {
    const u32 e = (u32)__builtin_popcount(empty);
    const u32 k = __umulhi(h2, (e * (e - 1u)) >> 1);
    const u32 ij = plut[e * 36u + k];
    const u32 c5 = (u32)__builtin_popcount(empty & 31u);
    return policy_nth(plut, empty, c5, ij & 0xFu) | (policy_nth(plut, empty, c5, ij >> 4) << 8);
}

// ====================================================================== the hot path
// One Env.step (env.py:34-53) on the board held in (P0,P1,Q0,Q1).  `lut` is the LDS copy of
// g_line_lut (one entry per dword).  Returns 0x7F iff a completed line exists afterwards (else 0);
// P1's done bit is updated.
template <bool AUTO_RESET>
__device__ __forceinline__ u32 step_core(u32 &P0, u32 &P1, u32 &Q0, u32 &Q1, u32 act, u32 bit,
                                         const uint8_t *lut) {
    if (AUTO_RESET) {                                   // finished boards restart: empty = all zero
        const u32 keep = ~(u32)((int)P1 >> 31);         // 0 iff done
        P0 &= keep;
        P1 &= keep;
        Q0 &= keep;
        Q1 &= keep;
    }
    const u32 a = act & 0xFFu, b = act >> 8;            // action[0], action[1] (env.py:37-38)
    const u32 lo = min(a, b), hi = max(a, b);           // board.py:16-18
    // the two squares as a mask at the classical mask's place in P1 (only looked at when hi < 9)
    const u32 pmS = ((1u << P1_CL_SHIFT) << (lo & 31u)) | ((1u << P1_CL_SHIFT) << (hi & 31u));
    // board.py:10-15 (+ IndexError for >8, swallowed at env.py:41): reject before mutating
    if (hi < 9u && lo != hi && (P1 & pmS) == 0u) {
        const u32 pm = pmS >> P1_CL_SHIFT;
        const u32 n4 = (P1 >> (P1_N_SHIFT - 2u)) & 0x3Cu;            // 4 * moves played (bits 6,7 of P1 are 0)
        u64 comps = (u64)Q1 | ((u64)((P1 >> P1_CHI_SHIFT) & 0xFu) << 32);
        const u32 mlo = (u32)(comps >> lo) & SLOT_LSB;   // slot holding lo (board.py:28-33)
        const u32 mhi = (u32)(comps >> hi) & SLOT_LSB;   // slot holding hi (board.py:35-40)
        const bool has_lo = mlo != 0u;
        const bool cyc = (mlo & mhi) != 0u;              // board.py:42: same component -> cycle
        // x: the square that becomes the child end of the new edge (its tree is re-rooted at it).
        // On a cycle it is the square the closing move lands on (qeval.py:35: bit 0 -> lo, 1 -> hi);
        // otherwise either end will do, and an isolated square is the cheap one: hi, unless only lo
        // is isolated (no walk at all instead of a walk up hi's tree)
        const u32 x4 = (((cyc && bit == 0u) || (!has_lo && mhi != 0u)) ? lo : hi) * 4u;
        u64 P = (u64)P0 | ((u64)P1 << 32);
        {   // re-root x's tree at x: reverse the parent edges along the path x -> old root.
            // All quantities are "times four": v4 = 4v is the shift that brings square v's nibble
            // to bits 2..5, ec4 = 4 * code of the edge found there, and rotating Q0 right by ec4
            // brings 4 * (lo^hi) of that edge to bits 2..5: the other end of the edge is one
            // rotate and one xor-and away.
            // x itself receives this move as its parent edge (code of round n), every later node
            // on the path receives the edge its child used to have.
            u32 v4 = x4, prev4 = n4 ^ 0x3Cu;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const u32 t = (u32)(P >> v4);
                const u32 ec4 = t & 0x3Cu;
                P ^= (u64)((t ^ prev4) & 0x3Cu) << v4;   // sq[v] = prev
                if (ec4 == 0u) break;                    // v was the root
                v4 ^= rotr32(Q0, ec4) & 0x3Cu;
                prev4 = ec4;
            }
        }
        P0 = (u32)P;
        P1 = (u32)(P >> 32);
        // board.py:19: append.  Only x = lo^hi is kept (see the header): round n <= 7 goes to its
        // nibble of Q0 (x*4 rotated right by 4n+4, i.e. x<<16 rotated by 4n+18), every move to the
        // `last x` field; n += 1.
        const u32 x16 = (a ^ b) << 16;
        Q0 ^= rotr32(x16, (P1 >> (P1_N_SHIFT - 2u)) + 18u);            // a rotate only looks at the low five bits
        P1 = ((P1 & ~(0xFu << P1_LX_SHIFT)) | x16) + (1u << P1_N_SHIFT);
        // ---- board.py:42-69 on the cached qstructs, all cases in one straight line ----
        // ffbl_raw(0) = -1, a 64-bit shift by -1 (= 63) gives 0: c1 = component of hi, 0 if none
        const u32 c1 = (u32)(comps >> (ffbl_raw(mhi) & 63u)) & 0x1FFu;
        // the slot the move goes to (board.py:58-69): lo's, else hi's, else the first empty one.
        // Slots are compact, so the first empty slot's bit lies above every occupied slot's and a
        // single "lowest set bit" picks the right one: nz = non-empty flags of slots 0..2 (bits
        // 8,17,26), t = the LSBs of slots 0..count, t & ~(t >> 9) = the LSB of slot `count`.
        const u32 tsel = has_lo ? mlo : mhi;
        const u32 c32 = (u32)comps;
        const u32 nz = (((c32 & 0x03FDFEFFu) + 0x03FDFEFFu) | c32) & 0x04020100u;
        const u32 t = (nz << 1) | 1u;
        const u32 sT = ffbl_raw(tsel | (t & ~(t >> 9)));
        // the move's squares join the slot; so does hi's component (a no-op unless this is a
        // union, board.py:58-61: on a cycle or when only hi is in a slot it is that slot already)
        comps |= (u64)(pm | c1) << sT;
        // pop hi's slot on a cycle (board.py:56) or a union (board.py:61) <=> both are in a slot
        const u32 mpop = has_lo ? mhi : 0u;
        const u32 low = mpop - 1u;                                      // all ones = keep everything
        const u32 chi2 = (u32)(comps >> 32);
        Q1 = ((u32)comps & low) | (__builtin_amdgcn_alignbit(chi2, (u32)comps, 9u) & ~low);
        // after a pop at most three slots are left: bits 27..35 are empty
        P1 = (P1 & ~(0xFu << P1_CHI_SHIFT)) | ((mpop ? 0u : chi2) << P1_CHI_SHIFT);
        // board.py:44-56 + qeval.py:5-51: on a cycle every square of the component goes classical
        // and already holds its parent edge's round; x holds the closing move's round
        P1 |= (cyc ? c1 : 0u) << P1_CL_SHIFT;
    }
    // board.py:71-115 reduced to "does any line exist" (all that env.py:49,51 need): parity of the
    // round on each classical square -> X / O masks -> table lookup.  Codes are complemented, so
    // a set low bit means an EVEN round (X).  All masks here are "times four" (bit v+2 = square v).
    // Eight classical squares = the autofill of board.py:22-25 is due: the ninth square counts as
    // X (round 8) — with 8 or 9 classical squares X is simply "everything that is not O".
    const u32 par4 = P0 & 0x44444444u;
    const u32 even4 = __builtin_amdgcn_udot8(par4, 0x00008421u, 0u, false) |
                      (__builtin_amdgcn_udot8(par4, 0x84210000u, 0u, false) << 4) | ((P1 << 8) & 0x400u);
    const u32 cl4 = (P1 >> (P1_CL_SHIFT - 2u)) & 0x7FCu;                // bits 20,21 of P1 are 0
    const u32 pc = (u32)__builtin_popcount(cl4);
    const u32 O4 = cl4 & ~even4;
    const u32 X4 = pc >= 8u ? (O4 ^ 0x7FCu) : (cl4 & even4);
    const u32 win = (u32)lut[X4] | (u32)lut[O4];
    // env.py:51: a line, or len(moves) > 8  <=>  at least 8 classical squares.  win is 0 or 0x7F,
    // pc <= 9: bit 3 of (win | pc & 8) is the answer
    P1 = (P1 & ~P1_DONE) | (((win | (pc & 8u)) << 28) & P1_DONE);
    return win;
}

// classical mask the policy sees (a finished board counts as empty under auto-reset)
template <bool AUTO_RESET>
__device__ __forceinline__ u32 policy_empty_mask(u32 P1) {
    const u32 cl = (AUTO_RESET && (P1 >> 31)) ? 0u : (P1 >> P1_CL_SHIFT) & 0x1FFu;
    return ~cl & 0x1FFu;
}

// ====================================================================== observation tiles
// Env._observation (env.py:68-85) is written through LDS tiles laid out exactly like the outputs
// (row-major per board), so that the workgroup can stream every tile out with coalesced dword
// stores (a lane-per-board store would be ~30 single-byte stores per lane, 8..10 bytes apart).
struct ObsOut {                 // global outputs, indexed by the board's local index i
    int8_t *classical;          // [n,9]
    uint8_t *q_p1, *q_p1_len;   // [n,5,2], [n]
    uint8_t *q_p2, *q_p2_len;   // [n,4,2], [n]
    uint8_t *turn;              // [n]
};
struct ObsTiles {               // LDS rows of the workgroup's boards (already phase-shifted)
    uint8_t *cl, *p1, *p2, *l1, *l2, *tn;
};

// The tile of an output whose first byte lands at global address g starts at LDS offset (g & 3) of
// a 16-byte aligned buffer: global and LDS addresses then share their alignment phase and the bulk
// of the copy is aligned dwords on both sides, whatever board the workgroup starts at.
__device__ __forceinline__ u32 obs_phase(const void *g) { return (u32)(uintptr_t)g & 3u; }

template <int BLOCK>
__device__ inline void tile_copy_out(uint8_t *gdst, const uint8_t *tile16, u32 nbytes) {
    const u32 phase = obs_phase(gdst);
    const uint8_t *src = tile16 + phase;
    const u32 head = min((4u - phase) & 3u, nbytes);
    if (threadIdx.x < head) gdst[threadIdx.x] = src[threadIdx.x];
    const u32 body = (nbytes - head) >> 2;
    u32 *gd = reinterpret_cast<u32 *>(gdst + head);
    const u32 *sd = reinterpret_cast<const u32 *>(src + head);
    for (u32 k = threadIdx.x; k < body; k += BLOCK) __builtin_nontemporal_store(sd[k], &gd[k]);
    const u32 k = head + (body << 2) + threadIdx.x;
    if (k < nbytes) gdst[k] = src[k];
}

// LDS bytes of the six tiles for `boards` boards (each tile padded for its phase, 16-byte aligned)
__host__ __device__ constexpr u32 obs_tile_bytes(u32 boards, u32 row) { return (boards * row + 4u + 15u) & ~15u; }
__host__ __device__ constexpr u32 obs_lds_bytes(u32 boards) {
    return obs_tile_bytes(boards, 9) + obs_tile_bytes(boards, 10) + obs_tile_bytes(boards, 8) +
           3u * obs_tile_bytes(boards, 1);
}

// Wave-private copy-out: the 64 * BPL boards of one wave occupy one contiguous, dword-aligned span
// of every tile (as long as the tile itself starts on a dword, phase 0), so the wave that wrote the
// rows can stream them out itself right away — LDS operations of one wave execute in order, no
// workgroup barrier is needed, and its stores overlap the other waves' compute.
__device__ inline void wave_copy_out(uint8_t *gdst, const uint8_t *tile16, u32 begin, u32 end) {
    const u32 lane = threadIdx.x & 63u;
    u32 *gd = reinterpret_cast<u32 *>(gdst);
    const u32 *sd = reinterpret_cast<const u32 *>(tile16);
    const u32 d0 = begin >> 2, d1 = end >> 2;                       // begin is a multiple of 4
    for (u32 k = d0 + lane; k < d1; k += 64u) __builtin_nontemporal_store(sd[k], &gd[k]);
    const u32 k = (d1 << 2) + lane;                                  // the last board of the batch may end mid-dword
    if (k < end) gdst[k] = tile16[k];
}

// true iff every tile of this workgroup starts on a dword in global memory (block-uniform)
__device__ __forceinline__ bool obs_all_phase0(const ObsOut &o, int64_t first) {
    return ((obs_phase(reinterpret_cast<const uint8_t *>(o.classical) + first * 9) | obs_phase(o.q_p1 + first * 10) |
             obs_phase(o.q_p2 + first * 8) | obs_phase(o.q_p1_len + first) | obs_phase(o.q_p2_len + first) |
             obs_phase(o.turn + first)) == 0u);
}

template <u32 BOARDS>
__device__ inline void obs_wave_copy_out(uint8_t *lds, const ObsOut &o, int64_t first, u32 b0, u32 b1) {
    wave_copy_out(reinterpret_cast<uint8_t *>(o.classical) + first * 9, lds, b0 * 9u, b1 * 9u);
    lds += obs_tile_bytes(BOARDS, 9);
    wave_copy_out(o.q_p1 + first * 10, lds, b0 * 10u, b1 * 10u);
    lds += obs_tile_bytes(BOARDS, 10);
    wave_copy_out(o.q_p2 + first * 8, lds, b0 * 8u, b1 * 8u);
    lds += obs_tile_bytes(BOARDS, 8);
    wave_copy_out(o.q_p1_len + first, lds, b0, b1);
    lds += obs_tile_bytes(BOARDS, 1);
    wave_copy_out(o.q_p2_len + first, lds, b0, b1);
    lds += obs_tile_bytes(BOARDS, 1);
    wave_copy_out(o.turn + first, lds, b0, b1);
}

template <u32 BOARDS>
__device__ __forceinline__ ObsTiles obs_tiles(uint8_t *lds, const ObsOut &o, int64_t first) {
    ObsTiles t;
    t.cl = lds + obs_phase(reinterpret_cast<const uint8_t *>(o.classical) + first * 9);
    lds += obs_tile_bytes(BOARDS, 9);
    t.p1 = lds + obs_phase(o.q_p1 + first * 10);
    lds += obs_tile_bytes(BOARDS, 10);
    t.p2 = lds + obs_phase(o.q_p2 + first * 8);
    lds += obs_tile_bytes(BOARDS, 8);
    t.l1 = lds + obs_phase(o.q_p1_len + first);
    lds += obs_tile_bytes(BOARDS, 1);
    t.l2 = lds + obs_phase(o.q_p2_len + first);
    lds += obs_tile_bytes(BOARDS, 1);
    t.tn = lds + obs_phase(o.turn + first);
    return t;
}

template <int BLOCK, u32 BOARDS>
__device__ inline void obs_copy_out(uint8_t *lds, const ObsOut &o, int64_t first, u32 valid) {
    tile_copy_out<BLOCK>(reinterpret_cast<uint8_t *>(o.classical) + first * 9, lds, valid * 9u);
    lds += obs_tile_bytes(BOARDS, 9);
    tile_copy_out<BLOCK>(o.q_p1 + first * 10, lds, valid * 10u);
    lds += obs_tile_bytes(BOARDS, 10);
    tile_copy_out<BLOCK>(o.q_p2 + first * 8, lds, valid * 8u);
    lds += obs_tile_bytes(BOARDS, 8);
    tile_copy_out<BLOCK>(o.q_p1_len + first, lds, valid);
    lds += obs_tile_bytes(BOARDS, 1);
    tile_copy_out<BLOCK>(o.q_p2_len + first, lds, valid);
    lds += obs_tile_bytes(BOARDS, 1);
    tile_copy_out<BLOCK>(o.turn + first, lds, valid);
}

// Compaction table of the observation's move lists.  A list has four candidate entries in fixed
// places (byte j of a register = the move of one round, byte 3 the EARLIEST round); a 4-bit
// liveness mask m selects the v_perm_b32 selectors that gather the live ones in move order into
// two dwords of (lo, hi) byte pairs — sources: lo bytes = selector 0..3, hi bytes = 4..7 — and pad
// the rest with 0xFF (selector 0x0D).
struct ObsLut {
    u32 sel[16][2];
    constexpr ObsLut() : sel() {
        for (u32 m = 0; m < 16; ++m) {
            u32 pos[4] = {0x0D0Du, 0x0D0Du, 0x0D0Du, 0x0D0Du};
            u32 p = 0;
            for (int j = 3; j >= 0; --j)
                if (m >> j & 1u) pos[p++] = (u32)j | ((4u + (u32)j) << 8);
            sel[m][0] = pos[0] | (pos[1] << 16);
            sel[m][1] = pos[2] | (pos[3] << 16);
        }
    }
};
__constant__ ObsLut g_obs_lut = ObsLut();
constexpr u32 OBS_LUT_BYTES = 128;

// One move list of the observation.  h: byte j = holder square + 1 of the candidate move j (0 =
// not live), x: byte j = lo^hi of that move.  Returns the (lo,hi) pairs of the live moves in move
// order as w0 | w1 (two pairs each, 0xFF-padded) and their number.
__device__ __forceinline__ u32 obs_list(u32 h, u32 x, const u32 *olut, u32 &w0, u32 &w1) {
    const u32 live01 = ((h + 0x0F0F0F0Fu) >> 4) & 0x01010101u;        // 1 where h != 0 (h <= 9)
    const u32 idx = __builtin_amdgcn_udot4(live01, 0x08040201u, 0u, false);
    const u32 c = h - live01;                                          // the holder square
    const u32 o = c ^ x;                                               // the other end of its move
    // bytewise min / max of c, o (both < 16): bit 4 of (c | 0x10) - o survives iff c >= o
    const u32 ge = (((c | 0x10101010u) - o) >> 4) & 0x01010101u;
    const u32 gm = (ge << 8) - ge;
    const u32 lo = (o & gm) | (c & ~gm);
    const u32 hi = c ^ o ^ lo;
    const u32 s0 = olut[idx * 2u], s1 = olut[idx * 2u + 1u];
    w0 = __builtin_amdgcn_perm(hi, lo, s0);
    w1 = __builtin_amdgcn_perm(hi, lo, s1);
    return (u32)__builtin_popcount(idx);
}

// The observation of one board, from its packed words, into row b of the tiles.
//   classical (env.py:71,82): Board.board, -1 for an empty square: nibbles -> bytes (two v_perm),
//     15 - code where classical, 0xFF elsewhere;
//   q_states_p1 / p2 (env.py:72-77): (lo,hi) of the un-collapsed moves of even / odd round in move
//     order, 255-padded.  An un-collapsed move is the parent edge of exactly one non-classical
//     square c (its holder) and is (c, c ^ x).  H inverts the holders: nibble code-8 = holder + 1
//     (one 64-bit shift per square: a square that holds no live edge has code 0 and lands in the
//     low word, which is ignored).  Nibble j of H and nibble j of the x word Q0 >>> 2 belong to
//     the same round 7-j, odd nibbles = even rounds = player 1, so both lists are built bytewise
//     for four moves at a time and compacted with one table lookup (obs_list);
//   turn (env.py:83): len(moves) % 2, the implicit autofill move included.
__device__ __forceinline__ void obs_board(u32 P0, u32 P1, u32 Q0, const ObsTiles &T, u32 b, const u32 *olut) {
    u64 P = (u64)P0 | ((u64)P1 << 32);
    u32 cl = (P1 >> P1_CL_SHIFT) & 0x1FFu;
    const u32 n = (P1 >> P1_N_SHIFT) & 0xFu;
    const bool fill = __builtin_popcount(cl) == 8;      // the autofill of board.py:22-25 is implicit
    if (fill) {
        // the last empty square is isolated (code 0); it holds round n (always 8: SURVEY.md §8a)
        const u32 idx = (u32)__builtin_ctz(~cl);
        P |= (u64)(15u - n) << (4u * idx + 2u);
        cl = 0x1FFu;
    }
    const u32 W = (u32)(P >> 2);
    const u32 ev = W & 0x0F0F0F0Fu, od = (W >> 4) & 0x0F0F0F0Fu;
    const u32 c03 = __builtin_amdgcn_perm(od, ev, 0x05010400u);          // codes of squares 0..3
    const u32 c47 = __builtin_amdgcn_perm(od, ev, 0x07030602u);          // codes of squares 4..7
    const u32 c8 = (u32)(P >> 34) & 0xFu;
    const u32 t03 = __umul24(cl & 0xFu, 0x204081u) & 0x01010101u;        // bit v -> byte v
    const u32 t47 = __umul24((cl >> 4) & 0xFu, 0x204081u) & 0x01010101u;
    const u32 m03 = (t03 << 8) - t03, m47 = (t47 << 8) - t47;            // 0xFF where classical
    const u32 o03 = (c03 ^ 0x0F0F0F0Fu) | ~m03;
    const u32 o47 = (c47 ^ 0x0F0F0F0Fu) | ~m47;
    const u32 o8 = (cl & 0x100u) ? (c8 ^ 0xFu) : 0xFFu;
    uint8_t *rc = T.cl + b * 9u;
    rc[0] = (uint8_t)o03;
    rc[1] = (uint8_t)(o03 >> 8);
    rc[2] = (uint8_t)(o03 >> 16);
    rc[3] = (uint8_t)(o03 >> 24);
    rc[4] = (uint8_t)o47;
    rc[5] = (uint8_t)(o47 >> 8);
    rc[6] = (uint8_t)(o47 >> 16);
    rc[7] = (uint8_t)(o47 >> 24);
    rc[8] = (uint8_t)o8;
    // ---- holders by code: four times the code of a square that holds a live edge (0 otherwise)
    const u32 S03 = (c03 & ~m03) << 2, S47 = (c47 & ~m47) << 2, S8 = (cl & 0x100u) ? 0u : c8 << 2;
    u32 H = 0;
#define QTTT_HOLD(v, S, k) H |= (u32)(((u64)((v) + 1u) << (((S) >> (8 * (k))) & 0xFFu)) >> 32)
    QTTT_HOLD(0, S03, 0);
    QTTT_HOLD(1, S03, 1);
    QTTT_HOLD(2, S03, 2);
    QTTT_HOLD(3, S03, 3);
    QTTT_HOLD(4, S47, 0);
    QTTT_HOLD(5, S47, 1);
    QTTT_HOLD(6, S47, 2);
    QTTT_HOLD(7, S47, 3);
    QTTT_HOLD(8, S8, 0);
#undef QTTT_HOLD
    const u32 X = rotr32(Q0, 2);
    u32 a0, a1, b0, b1;
    const u32 n1 = obs_list((H >> 4) & 0x0F0F0F0Fu, (X >> 4) & 0x0F0F0F0Fu, olut, a0, a1);   // even rounds
    const u32 n2 = obs_list(H & 0x0F0F0F0Fu, X & 0x0F0F0F0Fu, olut, b0, b1);                 // odd rounds
    uint16_t *r1 = reinterpret_cast<uint16_t *>(T.p1 + b * 10u);
    r1[0] = (uint16_t)a0;
    r1[1] = (uint16_t)(a0 >> 16);
    r1[2] = (uint16_t)a1;
    r1[3] = (uint16_t)(a1 >> 16);
    r1[4] = (uint16_t)0xFFFFu;                            // round 8 can never be un-collapsed
    *reinterpret_cast<u64 *>(T.p2 + b * 8u) = (u64)b0 | ((u64)b1 << 32);
    T.l1[b] = (uint8_t)n1;
    T.l2[b] = (uint8_t)n2;
    T.tn[b] = (uint8_t)((n + (fill ? 1u : 0u)) & 1u);                   // env.py:83
}

// ====================================================================== the step kernels
// BPL boards per lane: lane j owns boards [j*BPL, (j+1)*BPL), so every plane is read and written
// with 16-byte vector accesses that are contiguous across the wave.  Addresses are a block-uniform
// 64-bit base (scalar unit) plus a 32-bit lane offset.
// SAMPLE: the action is not read but drawn in the kernel from the uniform-legal policy (and written to
// `actions` when that is not null) — qttt_sample_actions + qttt_step in one launch.
// OBS: Env.step returns the observation too (env.py:46,53): it is written from the registers the
// step already holds, through the LDS tiles above — qttt_step + qttt_observe in one launch.
// BLOCK: workgroup size, chosen by the host per launch: 1024 for batches that fill the chip with
// 1024-thread workgroups (7.2-7.5 us instead of 7.6-7.7 per 1 M boards), QTTT_BLOCK = 512 below
// (262 144 boards: 3.8 us with 512 against 4.9 with 1024, which would leave half the CUs idle).
template <int BLOCK, int BPL, bool HAS_BITS, bool AUTO_RESET, bool SAMPLE = false, bool OBS = false>
__global__ __launch_bounds__(BLOCK) void step_kernel(
    u64 *__restrict__ pP, u64 *__restrict__ pQ, uint16_t *__restrict__ actions,
    const uint8_t *__restrict__ bits, u32 key_fold, u32 key_hi, u32 id_base,
    u32 *__restrict__ reward_bits, uint8_t *__restrict__ terminated, ObsOut obs, int64_t i_begin,
    u32 last_groups) {
    constexpr u32 TILE_BOARDS = BLOCK * BPL;
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t plut[SAMPLE ? POLICY_LUT_WORDS * 4 : 4];
    __shared__ __attribute__((aligned(16))) uint8_t otile[OBS ? obs_lds_bytes(TILE_BOARDS) : 16];
    __shared__ __attribute__((aligned(16))) u32 olut[OBS ? OBS_LUT_BYTES / 4 : 4];
#ifdef QTTT_DEBUG_STAMPS
    const u64 st0 = __builtin_amdgcn_s_memrealtime();
#endif
    typedef Vec<u64, BPL> V64;
    typedef Vec<u32, BPL> V32;
    typedef Vec<uint16_t, BPL> V16;
    typedef Vec<uint8_t, BPL> V8;
    const int64_t jb = (int64_t)blockIdx.x * BLOCK;                // first lane-group of the block
    const int64_t ib = i_begin + jb * BPL;                              // first board of the block
    // lane-groups of this block: every block is full except possibly the last one of the grid
    const u32 ng = blockIdx.x + 1u == gridDim.x ? last_groups : (u32)BLOCK;
    const bool active = threadIdx.x < ng;
    const u32 g = active ? threadIdx.x : 0u;                            // idle lanes re-read group 0
    // The small tables that are LOADED (policy, observation) are requested first and stored after
    // the streaming loads have been issued: vector loads return in order, so the wait in front of
    // the table's LDS store then covers the table word only, not this wave's state.  The line table
    // is computed.  Either way the workgroup barrier is passed while the state is still in flight.
    static_assert(!SAMPLE || BLOCK >= (int)POLICY_LUT_WORDS, "one policy-table word per thread");
    u32 plw = 0, olw = 0;
    if (SAMPLE && threadIdx.x < POLICY_LUT_WORDS) plw = reinterpret_cast<const u32 *>(&g_policy_lut)[threadIdx.x];
    if (OBS && threadIdx.x < OBS_LUT_BYTES / 4) olw = (&g_obs_lut.sel[0][0])[threadIdx.x];
    V64 p = load_stream(&reinterpret_cast<const V64 *>(pP + ib)[g]);
    V64 q = load_stream(&reinterpret_cast<const V64 *>(pQ + ib)[g]);
    V16 act;
    V8 bt;
    if (!SAMPLE) act = load_stream(&reinterpret_cast<const V16 *>(actions + ib)[g]);
    if (HAS_BITS) bt = load_stream(&reinterpret_cast<const V8 *>(bits + ib)[g]);
    fill_line_lut_nosync<BLOCK>(lut);
    if (SAMPLE && threadIdx.x < POLICY_LUT_WORDS) reinterpret_cast<u32 *>(plut)[threadIdx.x] = plw;
    if (OBS && threadIdx.x < OBS_LUT_BYTES / 4) olut[threadIdx.x] = olw;
    ObsTiles T;
    if (OBS) T = obs_tiles<TILE_BOARDS>(otile, obs, ib);
    __syncthreads();
#ifdef QTTT_DEBUG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u64 st1 = __builtin_amdgcn_s_memrealtime();
#endif
    if (active) {
        V32 rw;
        V8 tm;
        const u32 id0 = id_base + ((u32)jb + g) * BPL;                  // low 32 bits of the global board id
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            u32 P0 = (u32)p.v[k], P1 = (u32)(p.v[k] >> 32);
            u32 Q0 = (u32)q.v[k], Q1 = (u32)(q.v[k] >> 32);
            u32 bit, av;
            if (SAMPLE) {
                // the policy sees the board the step will act on: a finished board counts as empty
                const u32 h1 = lowbias32((id0 + (u32)k) ^ key_fold);
                const u32 h2 = lowbias32(h1 ^ key_hi);
                const u32 empty = policy_empty_mask<AUTO_RESET>(P1);
                av = (empty & (empty - 1u)) ? policy_action(plut, empty, h2) : 0u;
                act.v[k] = (uint16_t)av;
                bit = h1 >> 31;
            } else {
                av = act.v[k];
                if (HAS_BITS) bit = bt.v[k] & 1u;
                else bit = collapse_bit_of((id0 + (u32)k) ^ key_fold);
            }
            const u32 win = step_core<AUTO_RESET>(P0, P1, Q0, Q1, av, bit, lut);
            p.v[k] = (u64)P0 | ((u64)P1 << 32);
            q.v[k] = (u64)Q0 | ((u64)Q1 << 32);
            rw.v[k] = 0x80000000u | (win << 23);                         // env.py:49: -1.0f / -0.0f
            tm.v[k] = (uint8_t)(P1 >> 31);
            if (OBS) obs_board(P0, P1, Q0, T, g * BPL + (u32)k, olut);
        }
        store_stream(&reinterpret_cast<V64 *>(pP + ib)[g], p);
        store_stream(&reinterpret_cast<V64 *>(pQ + ib)[g], q);
        if (SAMPLE && actions) store_stream(&reinterpret_cast<V16 *>(actions + ib)[g], act);
        store_stream(&reinterpret_cast<V32 *>(reward_bits + ib)[g], rw);
        store_stream(&reinterpret_cast<V8 *>(terminated + ib)[g], tm);
    }
    if (OBS) {
        // a wave's boards [64w * BPL, 64(w+1) * BPL) start on a multiple of 4 bytes in every tile
        if ((64u * BPL) % 4u == 0u && obs_all_phase0(obs, ib)) {
            const u32 w0 = (threadIdx.x & ~63u) * BPL;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // rows written by other lanes of this wave
            __builtin_amdgcn_wave_barrier();
            if (w0 < ng * BPL) obs_wave_copy_out<TILE_BOARDS>(otile, obs, ib, w0, min(w0 + 64u * BPL, ng * BPL));
        } else {
            __syncthreads();
            obs_copy_out<BLOCK, TILE_BOARDS>(otile, obs, ib, ng * BPL);
        }
    }
#ifdef QTTT_DEBUG_STAMPS
    const u64 st2 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u64 st3 = __builtin_amdgcn_s_memrealtime();
    if (g_debug_stamps && (threadIdx.x & 63) == 0) {
        u64 *o = g_debug_stamps + ((int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 4;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = st3;
    }
#endif
}

// Mapping study: ONE WAVEFRONT PER BOARD, the mapping BASELINE.json's north_star
// sketches.  A board's step is a chain of dependent operations on a 9-node graph (validity ->
// component lookup -> path walk -> collapse -> line test), so whatever the 64 lanes of a wave do
// with __shfl/__ballot, the wave cannot retire a board faster than one lane can run that chain.
// This kernel is that lower bound made concrete: lane 0 of every wave runs the same step_core, the
// other 63 lanes are idle, state is staged through LDS by the workgroup.  Same results as
// step_kernel (tested); measured beside it in tools/stepbench.
template <bool HAS_BITS, bool AUTO_RESET>
__global__ __launch_bounds__(256) void step_wave_per_board_kernel(
    u64 *__restrict__ pP, u64 *__restrict__ pQ, const uint16_t *__restrict__ actions,
    const uint8_t *__restrict__ bits, u32 key_fold, u32 id_base, u32 *__restrict__ reward_bits,
    uint8_t *__restrict__ terminated, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ u64 sP[4], sQ[4];
    __shared__ u32 sAct[4], sBit[4];
    fill_line_lut_nosync<256>(lut);
    const int64_t i0 = (int64_t)blockIdx.x * 4;                   // 4 waves = 4 boards per workgroup
    if (threadIdx.x < 4 && i0 + threadIdx.x < n) {                 // cooperative tile load into LDS
        const int64_t i = i0 + threadIdx.x;
        sP[threadIdx.x] = pP[i];
        sQ[threadIdx.x] = pQ[i];
        sAct[threadIdx.x] = actions[i];
        sBit[threadIdx.x] = HAS_BITS ? bits[i] & 1u : collapse_bit_of((id_base + (u32)i) ^ key_fold);
    }
    __syncthreads();
    const u32 w = threadIdx.x >> 6;
    const int64_t i = i0 + w;
    if (i >= n || (threadIdx.x & 63u) != 0u) return;              // lane 0 of each wave owns the board
    u32 P0 = (u32)sP[w], P1 = (u32)(sP[w] >> 32), Q0 = (u32)sQ[w], Q1 = (u32)(sQ[w] >> 32);
    const u32 win = step_core<AUTO_RESET>(P0, P1, Q0, Q1, sAct[w], sBit[w], lut);
    pP[i] = (u64)P0 | ((u64)P1 << 32);
    pQ[i] = (u64)Q0 | ((u64)Q1 << 32);
    reward_bits[i] = 0x80000000u | (win << 23);
    terminated[i] = (uint8_t)(P1 >> 31);
}

// T consecutive steps in ONE launch (qttt_step_many with QTTT_FLAG_FUSED): the boards stay in
// registers, only the per-step streams move (2 B action in, 5 B reward/terminated out per step), so
// the loop is VALU-bound and pays one launch instead of T.  Same results as T launches of
// step_kernel; meant for replay / evaluation where the actions are known up front (a policy that
// looks at the state between steps needs the one-launch-per-step form).
template <bool HAS_BITS, bool AUTO_RESET>
__global__ __launch_bounds__(QTTT_BLOCK) void step_fused_kernel(
    u64 *__restrict__ pP, u64 *__restrict__ pQ, const uint16_t *__restrict__ actions,
    const uint8_t *__restrict__ bits, u64 seed, u32 step_idx0, u32 id_hi_fold, u32 id_base,
    u32 *__restrict__ reward_bits, uint8_t *__restrict__ terminated, int64_t out_stride, int64_t n,
    int32_t n_steps) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    fill_line_lut<QTTT_BLOCK>(lut);
    const int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    if (i >= n) return;
    const u64 P = pP[i], Q = pQ[i];
    u32 P0 = (u32)P, P1 = (u32)(P >> 32), Q0 = (u32)Q, Q1 = (u32)(Q >> 32);
    const u32 id = id_base + (u32)i;
    u32 win = 0;
    for (int32_t t = 0; t < n_steps; ++t) {
        const u32 act = load_stream(&actions[(int64_t)t * n + i]);
        u32 bit;
        if (HAS_BITS) bit = load_stream(&bits[(int64_t)t * n + i]) & 1u;
        else bit = collapse_bit_of(id ^ ((u32)launch_key(seed, step_idx0 + (u32)t) ^ id_hi_fold));
        win = step_core<AUTO_RESET>(P0, P1, Q0, Q1, act, bit, lut);
        if (out_stride != 0 || t == n_steps - 1) {
            const u32 rwv = 0x80000000u | (win << 23);
            const uint8_t tmv = (uint8_t)(P1 >> 31);
            store_stream(&reward_bits[(int64_t)t * out_stride + i], rwv);
            store_stream(&terminated[(int64_t)t * out_stride + i], tmv);
        }
    }
    pP[i] = (u64)P0 | ((u64)P1 << 32);
    pQ[i] = (u64)Q0 | ((u64)Q1 << 32);
}

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
__device__ __forceinline__ void fast_check_win(const Lite &s, int &p1, int &p2) {
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
    const uint8_t *lut = g_line_lut.b;
    p1 = lut[X & ge[2]] ? 4 : (lut[X & ge[0]] ? 6 : (lut[X] ? 8 : -1));
    p2 = lut[O & ge[1]] ? 5 : (lut[O] ? 7 : -1);
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

__device__ __forceinline__ int64_t fast_py_hash(const Lite &s, u32 P1_stored, u32 Q0, const u64 *tbl) {
    u64 acc = PYH_P5;
    const u32 W = (u32)(s.P >> 2);                                  // codes of squares 0..7
    const u32 c8 = (u32)(s.P >> 34) & 0xFu;
#pragma unroll
    for (u32 v = 0; v < 9; ++v) {
        const u32 c = v < 8u ? (W >> (4u * v)) & 0xFu : c8;
        acc = pyh_step(acc, tbl[(s.cl >> v & 1u) ? 16u - c : 0u]);  // board[value + 1], value = 15 - c
    }
    // moves in round order: round t is held by the one square whose code is 15 - t (zero nibble of
    // W ^ 0x1111_1111 * code; the lowest flag of the borrow trick is always a true zero; no flag =
    // square 8), and is (c, c ^ x_t).  x nibbles: round 0 in bits 0..3 of Qr, round t >= 1 at 32 - 4t.
    const u32 last_x = (P1_stored >> P1_LX_SHIFT) & 0xFu;
    const u32 Qr = rotr32(Q0, 30u) ^ (s.n_real == 9u ? last_x : 0u);  // round 8's x was XORed onto round 0's
    const u32 n8 = min(s.n, 8u);
    u32 kk = 0xFFFFFFFFu, sh = 0u;
    for (u32 t = 0; t < n8; ++t) {                                  // (an autofill move is always round 8)
        const u32 z = W ^ kk;
        const u32 f = (z - 0x11111111u) & ~z & 0x88888888u;
        const u32 c = f ? (u32)__builtin_ctz(f) >> 2 : 8u;
        const u32 o = min(c ^ ((Qr >> sh) & 0xFu), 8u);             // (only a corrupted import could exceed 8)
        acc = pyh_step(acc, tbl[10u + (c * 9u + o) * 9u + t]);
        kk -= 0x11111111u;
        sh = (sh - 4u) & 31u;
    }
    if (s.n == 9u) {
        const u32 z = W ^ 0x77777777u;
        const u32 f = (z - 0x11111111u) & ~z & 0x88888888u;
        const u32 c = f ? (u32)__builtin_ctz(f) >> 2 : 8u;
        const u32 o = min(c ^ (s.n_real == 9u ? last_x : 0u), 8u);  // autofill = (idx, idx)
        acc = pyh_step(acc, tbl[10u + (c * 9u + o) * 9u + 8u]);
    }
    return (int64_t)pyh_fin(acc, 9u + s.n);
}

#define QTTT_COLD_BLOCK 256

// Env._observation (env.py:68-85) of stored boards: two boards per lane (one 16-byte load per
// plane, as the step kernel) through the same LDS tiles and the same obs_board() as the fused step
// kernel.  A workgroup owns 2 * QTTT_BLOCK consecutive boards; the last board of an odd batch is
// read with scalar loads.
__global__ __launch_bounds__(QTTT_BLOCK) void observe_kernel(const u64 *pP, const u64 *pQ, ObsOut obs, int64_t n) {
    constexpr u32 TILE_BOARDS = QTTT_BLOCK * 2;
    __shared__ __attribute__((aligned(16))) uint8_t otile[obs_lds_bytes(TILE_BOARDS)];
    __shared__ __attribute__((aligned(16))) u32 olut[OBS_LUT_BYTES / 4];
    const u32 olw = threadIdx.x < OBS_LUT_BYTES / 4 ? (&g_obs_lut.sel[0][0])[threadIdx.x] : 0u;   // see step_kernel
    const int64_t base = (int64_t)blockIdx.x * TILE_BOARDS;
    const u32 valid = (u32)min((int64_t)TILE_BOARDS, n - base);
    const ObsTiles T = obs_tiles<TILE_BOARDS>(otile, obs, base);
    const u32 b0 = threadIdx.x * 2u;
    typedef Vec<u64, 2> V64;
    V64 p, q;
    if (b0 + 1u < valid) {
        p = load_stream(&reinterpret_cast<const V64 *>(pP + base)[threadIdx.x]);
        q = load_stream(&reinterpret_cast<const V64 *>(pQ + base)[threadIdx.x]);
    } else if (b0 < valid) {
        p.v[0] = pP[base + b0];
        q.v[0] = pQ[base + b0];
    }
    if (threadIdx.x < OBS_LUT_BYTES / 4) olut[threadIdx.x] = olw;
    __syncthreads();
    if (b0 < valid) obs_board((u32)p.v[0], (u32)(p.v[0] >> 32), (u32)q.v[0], T, b0, olut);
    if (b0 + 1u < valid) obs_board((u32)p.v[1], (u32)(p.v[1] >> 32), (u32)q.v[1], T, b0 + 1u, olut);
    if (obs_all_phase0(obs, base)) {                      // every wave streams out the rows it wrote (see step_kernel)
        const u32 w0 = (threadIdx.x & ~63u) * 2u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (w0 < valid) obs_wave_copy_out<TILE_BOARDS>(otile, obs, base, w0, min(w0 + 128u, valid));
    } else {
        __syncthreads();
        obs_copy_out<QTTT_BLOCK, TILE_BOARDS>(otile, obs, base, valid);
    }
}

__global__ __launch_bounds__(QTTT_COLD_BLOCK) void check_win_kernel(
    const u64 *pP, const u64 *pQ, int8_t *p1_round, int8_t *p2_round, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * QTTT_COLD_BLOCK + threadIdx.x;
    if (i >= n) return;
    (void)pQ;
    const Lite s = lite_unpack(load_stream(&pP[i]));
    int p1, p2;
    fast_check_win(s, p1, p2);
    p1_round[i] = (int8_t)p1;
    p2_round[i] = (int8_t)p2;
}

// Board.moves / .board / .qstructs (board.py:4-6) as arrays, straight from the packed words: the
// move of round t is (c, c ^ x_t) for its holder c (found through the same inverse map as in
// fast_py_hash), the board is the nibbles of the classical squares, the qstructs are the cached
// slots.
__global__ __launch_bounds__(QTTT_COLD_BLOCK) void export_kernel(
    const u64 *pP, const u64 *pQ, uint8_t *moves, uint8_t *n_moves,
    int8_t *board, uint16_t *qmask, uint8_t *n_q, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * QTTT_COLD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const u64 P = load_stream(&pP[i]), Q = load_stream(&pQ[i]);
    const u32 P1 = (u32)(P >> 32), Q0 = (u32)Q;
    const Lite s = lite_unpack(P);
    u64 H = 0;                                                // nibble (code - 7) = holder square + 1
#pragma unroll
    for (u32 v = 0; v < 9; ++v) {
        const u32 c = (u32)(s.P >> (4u * v + 2u)) & 0xFu;
        H |= (u64)(v + 1u) << ((4u * c + 36u) & 63u);
        board[i * 9 + v] = (s.cl >> v & 1u) ? (int8_t)(15u - c) : (int8_t)-1;
    }
#pragma unroll
    for (u32 t = 0; t < 9; ++t) {
        const u32 h = (u32)(H >> (4u * (8u - t))) & 0xFu;
        const u32 c = h ? h - 1u : 0u;
        const u32 x = (t >= s.n_real) ? 0u : cold_move_x(Q0, P1, s.n_real, t);     // autofill = (idx, idx)
        const u32 o = c ^ x;
        const bool used = t < s.n;
        moves[i * 18 + t * 2] = used ? (uint8_t)min(c, o) : (uint8_t)255;
        moves[i * 18 + t * 2 + 1] = used ? (uint8_t)max(c, o) : (uint8_t)255;
    }
    n_moves[i] = (uint8_t)s.n;
    const u64 comps = (Q >> 32) | ((u64)((P1 >> P1_CHI_SHIFT) & 0xFu) << 32);
    u32 nq = 0;
#pragma unroll
    for (u32 k = 0; k < 4; ++k) {
        const u32 m = (u32)(comps >> (9u * k)) & 0x1FFu;
        qmask[i * 4 + k] = (uint16_t)m;
        nq += m != 0u;
    }
    n_q[i] = (uint8_t)nq;
}

// Builds the unpacked board (incl. the rooted forest) from Board attributes assigned by a caller
// (mcts.py:11-17,241 assign .board/.moves/.qstructs directly).  Not a hot path.
__device__ __forceinline__ void cold_from_attrs(const uint8_t *moves, u32 n_moves, const int8_t *board,
                                                const uint16_t *qmask, u32 n_q, Cold &s) {
    s.n = min(n_moves, 9u);
    s.cl = 0;
    s.mvq = 0;
    s.mv8 = 0;
    for (u32 t = 0; t < s.n; ++t)
        s.set_mv(t, (u32)(moves[t * 2] & 0xFu) | ((u32)(moves[t * 2 + 1] & 0xFu) << 4));
    s.sq = 0xFFFFFFFFFull;
    for (u32 v = 0; v < 9; ++v) {
        const int bv = board[v];
        if (bv >= 0) {
            s.cl |= 1u << v;
            s.set_sq(v, (u32)bv & 0xFu);
        }
    }
    const u32 nq = min(n_q, 4u);
    s.comps = 0;
    for (u32 k = 0; k < nq; ++k) s.comps |= (u64)(qmask[k] & 0x1FFu) << (9u * k);
    // root every tree of live edges: grow from the lowest square of each tree
    u32 rooted = 0;
    for (int pass = 0; pass < 9; ++pass) {
        bool grew = false;
        u32 cand = 0;
        for (u32 t = 0; t < s.n; ++t) {
            const u32 m = s.mv(t);
            const u32 lo = m & 0xFu, hi = m >> 4;
            if (lo == hi || lo > 8u || hi > 8u || (s.cl >> lo & 1u) || (s.cl >> hi & 1u)) continue;
            const bool rl = rooted >> lo & 1u, rh = rooted >> hi & 1u;
            if (rl && !rh) { s.set_sq(hi, t); rooted |= 1u << hi; grew = true; }
            else if (rh && !rl) { s.set_sq(lo, t); rooted |= 1u << lo; grew = true; }
            cand |= (1u << lo) | (1u << hi);
        }
        if (!grew) {
            cand &= ~rooted;                 // start a new tree at the lowest un-rooted square
            if (cand == 0u) break;
            rooted |= cand & (0u - cand);
        }
    }
    int p1, p2;
    cold_check_win(s, p1, p2);
    s.done = (p1 > 0 || p2 > 0 || s.n > 8u) ? 1u : 0u;
}

__global__ __launch_bounds__(QTTT_COLD_BLOCK) void import_kernel(
    u64 *pP, u64 *pQ, const uint8_t *moves, const uint8_t *n_moves, const int8_t *board,
    const uint16_t *qmask, const uint8_t *n_q, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * QTTT_COLD_BLOCK + threadIdx.x;
    if (i >= n) return;
    Cold s;
    cold_from_attrs(moves + i * 18, n_moves[i], board + i * 9, qmask + i * 4, n_q[i], s);
    u64 P, Q;
    cold_pack(s, P, Q);
    pP[i] = P;
    pQ[i] = Q;
}

// Board.make_move / update_qstructs / check_win (board.py:9-115) on caller-assigned attributes, one
// 64-byte record in, one out (include/qttt.h: qttt_board_op): import -> the SAME step_core the
// batch kernels run -> export + check_win, in one launch, so that the single-board façade costs one
// round trip.  The records may live in pinned host memory (the kernel reads and writes them
// directly).
__global__ __launch_bounds__(QTTT_COLD_BLOCK) void board_op_kernel(const uint8_t *in, uint8_t *out, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    fill_line_lut<QTTT_COLD_BLOCK>(lut);
    int64_t i = (int64_t)blockIdx.x * QTTT_COLD_BLOCK + threadIdx.x;
    if (i >= n) return;
    const uint8_t *r = in + i * QTTT_BOARD_RECORD_BYTES;
    uint8_t *o = out + i * QTTT_BOARD_RECORD_BYTES;
    uint8_t mv[18];
    int8_t bd[9];
    uint16_t qm[4];
    for (int k = 0; k < 18; ++k) mv[k] = r[k];
    for (int k = 0; k < 9; ++k) bd[k] = (int8_t)r[19 + k];
    for (int k = 0; k < 4; ++k) qm[k] = (uint16_t)(r[30 + 2 * k] | (r[31 + 2 * k] << 8));
    const u32 op = r[29];
    Cold s;
    cold_from_attrs(mv, r[18], bd, qm, r[28], s);
    u64 P, Q;
    cold_pack(s, P, Q);
    u32 P0 = (u32)P, P1 = (u32)(P >> 32), Q0 = (u32)Q, Q1 = (u32)(Q >> 32);
    const u32 n_before = (P1 >> P1_N_SHIFT) & 0xFu;
    u32 win = 0;
    if (op != QTTT_OP_CHECK_WIN) win = step_core<false>(P0, P1, Q0, Q1, (u32)r[38] | ((u32)r[39] << 8), r[40] & 1u, lut);
    const u32 rejected = (op != QTTT_OP_CHECK_WIN && ((P1 >> P1_N_SHIFT) & 0xFu) == n_before) ? 1u : 0u;
    // update_qstructs alone (board.py:27-69) never autofills: that is make_move's job (board.py:22-25)
    cold_unpack((u64)P0 | ((u64)P1 << 32), (u64)Q0 | ((u64)Q1 << 32), s, op != QTTT_OP_UPDATE_QSTRUCTS);
    for (u32 t = 0; t < 9; ++t) {
        const bool used = t < s.n;
        const u32 m = s.mv(t);
        o[t * 2] = used ? (uint8_t)(m & 0xFu) : (uint8_t)255;
        o[t * 2 + 1] = used ? (uint8_t)(m >> 4) : (uint8_t)255;
    }
    o[18] = (uint8_t)s.n;
    for (u32 v = 0; v < 9; ++v) o[19 + v] = (s.cl >> v & 1u) ? (uint8_t)s.sqv(v) : (uint8_t)0xFF;
    u32 nq = 0;
    for (u32 k = 0; k < 4; ++k) {
        o[30 + 2 * k] = (uint8_t)s.comp(k);
        o[31 + 2 * k] = (uint8_t)(s.comp(k) >> 8);
        nq += s.comp(k) != 0u;
    }
    o[28] = (uint8_t)nq;
    o[29] = (uint8_t)op;
    o[38] = r[38];
    o[39] = r[39];
    o[40] = r[40];
    o[41] = (uint8_t)rejected;
    int p1, p2;
    cold_check_win(s, p1, p2);
    const u32 any = (p1 > 0 || p2 > 0) ? 1u : 0u;
    const u32 rb = 0x80000000u | (any ? 0x3F800000u : 0u);             // env.py:49: -1.0f / -0.0f
    o[44] = (uint8_t)rb;
    o[45] = (uint8_t)(rb >> 8);
    o[46] = (uint8_t)(rb >> 16);
    o[47] = (uint8_t)(rb >> 24);
    o[48] = (uint8_t)((any || s.n > 8u) ? 1u : 0u);                    // env.py:51
    o[49] = (uint8_t)(int8_t)p1;
    o[50] = (uint8_t)(int8_t)p2;
    (void)win;
}

// legal pairs in ind2move order: for lo ascending, hi ascending (mcts.py:20-27, 339-343)
__global__ __launch_bounds__(QTTT_BLOCK) void sample_actions_kernel(
    const u64 *pP, u32 key_lo, u32 key_hi, u64 board_offset, u32 auto_reset, uint16_t *actions,
    int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    const u64 Pw = i < n ? load_stream(&pP[i]) : 0ull;      // requested before the table: the latencies overlap
    fill_policy_lut<QTTT_BLOCK>(plut);
    __syncthreads();
    if (i >= n) return;
    const u32 P1 = (u32)(Pw >> 32);
    const u32 cl = (auto_reset && (P1 >> 31)) ? 0u : (P1 >> P1_CL_SHIFT) & 0x1FFu;
    const u32 empty = ~cl & 0x1FFu;
    const u32 h1 = lowbias32(fold_id(board_offset + (u64)i) ^ key_lo);
    const u32 h2 = lowbias32(h1 ^ key_hi);
    // fewer than two empty squares: rank_pair gives (0,0) and nth_bit[..][0] twice -> a == b, a noop;
    // the spec says (0,0)
    const u32 act = (empty & (empty - 1u)) ? policy_action(plut, empty, h2) : 0u;
    actions[i] = (uint16_t)act;
}

// ====================================================================== §8(f) rows
// ind2move (mcts.py:339-343): lexicographic pairs (0,1),(0,2)..(7,8) as lo | hi<<4
struct PairLut {
    uint8_t b[36];
    constexpr PairLut() : b() {
        int a = 0;
        for (int i = 0; i < 9; ++i)
            for (int j = i + 1; j < 9; ++j) b[a++] = (uint8_t)(i | (j << 4));
    }
};
__constant__ PairLut g_pair_lut = PairLut();

__global__ __launch_bounds__(QTTT_BLOCK) void node_info_kernel(
    const u64 *pP, const u64 *pQ, int8_t *winner, uint8_t *terminal, u64 *legal,
    int64_t *key, int64_t n) {
    __shared__ u64 htbl[PYHASH_LUT_WORDS];
    __shared__ u64 ltbl[512];
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    const u64 P = i < n ? load_stream(&pP[i]) : 0ull, Q = i < n ? load_stream(&pQ[i]) : 0ull;   // before the table fill
    fill_pyhash_lut<QTTT_BLOCK>(htbl);
    fill_legal_lut<QTTT_BLOCK>(ltbl);
    fill_line_lut<QTTT_BLOCK>(lut);                       // ends with the workgroup barrier
    if (i >= n) return;
    const Lite s = lite_unpack(P);
    int w, t;
    lite_update_winner(s, lut, w, t);
    winner[i] = (int8_t)w;
    terminal[i] = (uint8_t)t;
    legal[i] = ltbl[s.cl];
    key[i] = fast_py_hash(s, (u32)(P >> 32), (u32)Q, htbl);
}

// MCTS._step (mcts.py:233-267): both values of the collapse bit computed directly instead of
// re-sampling make_move until the other branch appears.
__global__ __launch_bounds__(QTTT_BLOCK) void expand_kernel(
    const u64 *pP, const u64 *pQ, const uint8_t *action36,
    u64 *c0P, u64 *c0Q, u64 *c1P, u64 *c1Q, uint8_t *n_children,
    int8_t *winner, uint8_t *terminal, u64 *legal, int64_t *key, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ u64 htbl[PYHASH_LUT_WORDS];
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    const u64 P = i < n ? pP[i] : 0ull, Q = i < n ? pQ[i] : 0ull;  // requested before the table fills
    const u32 a = i < n ? (u32)action36[i] : 0u;
    fill_pyhash_lut<QTTT_BLOCK>(htbl);
    fill_line_lut<QTTT_BLOCK>(lut);                       // ends with the workgroup barrier
    if (i >= n) return;
    const u32 pr = a < 36u ? (u32)g_pair_lut.b[a] : 0u;            // (0,0) = a noop for bad indices
    const u32 act = (pr & 0xFu) | ((pr >> 4) << 8);
    u64 kidP[2], kidQ[2];
    for (u32 bit = 0; bit < 2; ++bit) {
        u32 P0 = (u32)P, P1 = (u32)(P >> 32), Q0 = (u32)Q, Q1 = (u32)(Q >> 32);
        step_core<false>(P0, P1, Q0, Q1, act, bit, lut);
        kidP[bit] = (u64)P0 | ((u64)P1 << 32);
        kidQ[bit] = (u64)Q0 | ((u64)Q1 << 32);
    }
    const u32 n_before = ((u32)(P >> 32) >> P1_N_SHIFT) & 0xFu;
    const u32 n_after = ((u32)(kidP[0] >> 32) >> P1_N_SHIFT) & 0xFu;
    const u32 cl_before = ((u32)(P >> 32) >> P1_CL_SHIFT) & 0x1FFu;
    const u32 cl_after = ((u32)(kidP[0] >> 32) >> P1_CL_SHIFT) & 0x1FFu;
    const u32 kids = n_after == n_before ? 0u : (cl_after != cl_before ? 2u : 1u);   // mcts.py:245
    n_children[i] = (uint8_t)kids;
    c0P[i] = kidP[0]; c0Q[i] = kidQ[0];
    c1P[i] = kidP[1]; c1Q[i] = kidQ[1];
    for (u32 c = 0; c < 2; ++c) {
        int w = -1, t = 0;
        u64 lm = 0;
        int64_t k = 0;
        if (c < kids) {
            const Lite s = lite_unpack(kidP[c]);
            lite_update_winner(s, lut, w, t);
            lm = fast_legal_mask(s.cl);
            k = fast_py_hash(s, (u32)(kidP[c] >> 32), (u32)kidQ[c], htbl);
        }
        winner[i * 2 + c] = (int8_t)w;
        terminal[i * 2 + c] = (uint8_t)t;
        legal[i * 2 + c] = lm;
        key[i * 2 + c] = k;
    }
}

// MCTS._simulate (mcts.py:185-198) under the uniform priors of mcts.py:287-292: play uniform-legal
// random moves to the end with the board in registers.  Ply p uses the counter hash of
// (seed, board id, step_idx0 + p) exactly like qttt_sample_actions + qttt_step would.
__global__ __launch_bounds__(QTTT_BLOCK) void rollout_kernel(
    const u64 *pP, const u64 *pQ, u64 seed, u32 step_idx0, u64 board_offset,
    int8_t *result, uint8_t *plies, u64 *fP, u64 *fQ, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    const u64 P = i < n ? pP[i] : 0ull, Q = i < n ? pQ[i] : 0ull;  // requested before the table fills
    fill_policy_lut<QTTT_BLOCK>(plut);
    fill_line_lut<QTTT_BLOCK>(lut);                       // ends with the workgroup barrier
    if (i >= n) return;
    u32 P0 = (u32)P, P1 = (u32)(P >> 32), Q0 = (u32)Q, Q1 = (u32)(Q >> 32);
    const u32 id = fold_id(board_offset + (u64)i);
    u32 played = 0;
    for (u32 p = 0; p < 9u; ++p) {
        const u32 empty = ~(P1 >> P1_CL_SHIFT) & 0x1FFu;
        if ((P1 >> 31) || (empty & (empty - 1u)) == 0u) break;   // terminal (mcts.py:188) / nothing legal
        const u64 key = launch_key(seed, step_idx0 + p);
        const u32 h1 = lowbias32(id ^ (u32)key);
        const u32 h2 = lowbias32(h1 ^ (u32)(key >> 32));
        step_core<false>(P0, P1, Q0, Q1, policy_action(plut, empty, h2), h1 >> 31, lut);
        played += 1u;
    }
    const u64 oP = (u64)P0 | ((u64)P1 << 32), oQ = (u64)Q0 | ((u64)Q1 << 32);
    int w, t;
    lite_update_winner(lite_unpack(oP), lut, w, t);
    result[i] = (int8_t)(w < 0 ? 0 : (w ? 1 : -1));       // MCTS._reward, mcts.py:200-209
    plies[i] = (uint8_t)played;
    if (fP) { fP[i] = oP; fQ[i] = oQ; }
}

// GameState.to_vector (mcts.py:67-85) as f32[18][10] and action_mask (mcts.py:87-91).
// A 256-thread workgroup owns 64 boards: thread (board b, part p) builds the rows of squares
// p, p+4, p+8 in an LDS tile, then all four waves stream the tile out as fully coalesced 16-byte
// stores (a lane-per-board store would scatter 16-byte pieces 720 bytes apart; one wave per tile
// would leave the CU at 3 waves because of the 46 KB tile).
#define QTTT_ENC_BOARDS 64
#define QTTT_ENC_BLOCK 256
__global__ __launch_bounds__(QTTT_ENC_BLOCK) void encode_kernel(
    const u64 *pP, const u64 *pQ, float *vec, uint8_t *mask, int64_t n) {
    __shared__ __attribute__((aligned(16))) float tile[QTTT_ENC_BOARDS * 180];
    __shared__ __attribute__((aligned(16))) uint8_t mtile[QTTT_ENC_BOARDS * 36];
    const int64_t base = (int64_t)blockIdx.x * QTTT_ENC_BOARDS;
    const u32 b = threadIdx.x & 63u, part = threadIdx.x >> 6;
    const int64_t i = base + b;
    const u32 valid = (u32)min((int64_t)QTTT_ENC_BOARDS, n - base);
    if (b < valid) {
        Cold s;
        cold_unpack(pP[i], pQ[i], s);
        float *o = tile + b * 180;
        const u32 qsets = s.comp(0) | s.comp(1) | s.comp(2) | s.comp(3);
        for (u32 v = part; v < 9; v += 4) {
            const u32 col = (s.cl >> v & 1u) ? s.sqv(v) : 9u;        // board -1 indexes column 9
            u32 touched = 0;                                       // rounds whose move touches v
            for (u32 t = 0; t < s.n; ++t)
                if ((s.mv(t) & 0xFu) == v || (s.mv(t) >> 4) == v) touched |= 1u << t;
            for (u32 c = 0; c < 10; ++c) {
                o[v * 10 + c] = c == col ? 1.0f : 0.0f;
                float q = (touched >> c & 1u) ? (1.0f / 3.0f) : 0.0f;   // 1/math.sqrt(9)
                if (c == 9u && !(qsets >> v & 1u)) q = 1.0f;        // square in no qstruct
                o[90 + v * 10 + c] = q;
            }
        }
        if (mask && part == 3u) {                                  // the lightest part also does the mask
            const u64 lm = fast_legal_mask(s.cl);
            for (int a = 0; a < 36; ++a) mtile[b * 36 + a] = (uint8_t)(lm >> a & 1ull);
        }
    }
    __syncthreads();
    {
        const u32 n4 = valid * 45u;                                // float4 pieces in this tile
        const u32x4 *src = reinterpret_cast<const u32x4 *>(tile);
        u32x4 *dst = reinterpret_cast<u32x4 *>(vec + base * 180);
        for (u32 k = threadIdx.x; k < n4; k += QTTT_ENC_BLOCK) __builtin_nontemporal_store(src[k], &dst[k]);
    }
    if (mask) {
        const u32 n4 = valid * 9u;                                 // 4-byte pieces (36 = 9 x 4)
        const u32 *src = reinterpret_cast<const u32 *>(mtile);
        u32 *dst = reinterpret_cast<u32 *>(mask + base * 36);
        for (u32 k = threadIdx.x; k < n4; k += QTTT_ENC_BLOCK) dst[k] = src[k];
    }
}

// tuning knob (bench / profiling): boards per lane of the step kernel (1|2|4).  Initialised from
// QTTT_STEP_BPL, changeable at run time through qttt_set_tuning().
inline int &tuning_bpl() {
    static int v = [] {
        int k = QTTT_DEFAULT_BPL;
        if (const char *e = getenv("QTTT_STEP_BPL")) { int q = atoi(e); if (q == 1 || q == 2 || q == 4) k = q; }
        return k;
    }();
    return v;
}

inline int grid_for(int64_t n) { return (int)((n + QTTT_BLOCK - 1) / QTTT_BLOCK); }
inline int cold_grid_for(int64_t n) { return (int)((n + QTTT_COLD_BLOCK - 1) / QTTT_COLD_BLOCK); }
inline int blocks_for(int64_t n_groups, int block) { return (int)((n_groups + block - 1) / block); }
// lane-groups from which qttt_step uses 1024-thread workgroups: 512 such workgroups = 16 waves on every SIMD pair
#define QTTT_BIG_BLOCK_MIN_GROUPS (512 * 1024)

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

int qttt_set_tuning(int boards_per_lane, int reserved) {
    (void)reserved;
    if (!(boards_per_lane == 1 || boards_per_lane == 2 || boards_per_lane == 4)) return QTTT_ERR_SIZE;
    tuning_bpl() = boards_per_lane;
    return 0;
}

int64_t qttt_state_bytes(int64_t n) { return n < 0 ? (int64_t)QTTT_ERR_SIZE : plane_stride(n) * QTTT_STATE_BYTES; }

uint64_t qttt_hash(uint64_t seed, uint64_t board_id, uint32_t step_idx) {
    const u64 key = launch_key(seed, step_idx);
    const u32 h1 = lowbias32(fold_id(board_id) ^ (u32)key);
    const u32 h2 = lowbias32(h1 ^ (u32)(key >> 32));
    return ((u64)h2 << 32) | h1;
}

int qttt_reset(void *state, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state) return QTTT_ERR_NULL;
    // the empty board is the all-zero state
    hipError_t e = hipMemsetAsync(state, 0, (size_t)(plane_stride(n) * QTTT_STATE_BYTES), (hipStream_t)stream);
    return e == hipSuccess ? 0 : (int)e;
}

static int launch_step(void *state, uint8_t *actions, const uint8_t *bits, uint64_t seed,
                       uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
                       uint8_t *terminated, int64_t n, void *stream, bool sample, const ObsOut *obs) {
    if (n < 0 || board_offset < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !reward || !terminated || (!sample && !actions)) return QTTT_ERR_NULL;
    if ((uintptr_t)actions & 1u) return QTTT_ERR_ACTION;   // actions are accessed as u16 pairs
    Planes p = planes(state, n);
    const u64 key = launch_key(seed, step_idx);
    const u32 key_lo = (u32)key, key_hi = (u32)(key >> 32);
    hipStream_t s = (hipStream_t)stream;
    uint16_t *a16 = reinterpret_cast<uint16_t *>(actions);
    u32 *rb = reinterpret_cast<u32 *>(reward);
    const bool ar = (flags & QTTT_FLAG_AUTO_RESET) != 0;
    // widest boards-per-lane the caller's pointers are aligned for (the planes always are)
    int bpl_max = obs ? (tuning_bpl() > 2 ? 2 : tuning_bpl()) : tuning_bpl();   // the tiles are sized for <= 2
    auto aligned = [&](int k) {
        return ((uintptr_t)actions % (2u * k)) == 0 && ((uintptr_t)reward % (4u * k)) == 0 &&
               ((uintptr_t)terminated % (unsigned)k) == 0 && (!bits || ((uintptr_t)bits % (unsigned)k) == 0);
    };
    while (bpl_max > 1 && !aligned(bpl_max)) bpl_max >>= 1;
    const ObsOut oo = obs ? *obs : ObsOut{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
#define QTTT_LAUNCH_B(BLK, BPL, HB, AR, SM, OB, I0, NG, KF, IDB)                                        \
    hipLaunchKernelGGL((step_kernel<BLK, BPL, HB, AR, SM, OB>), dim3(blocks_for(NG, BLK)), dim3(BLK), 0, s, \
                       p.P, p.Q, a16, bits, (u32)(KF), key_hi, (u32)(IDB), rb, terminated, oo,         \
                       (int64_t)(I0), (u32)((NG) - (int64_t)(blocks_for(NG, BLK) - 1) * (BLK)))
    // workgroup size: 1024 threads once that still gives every CU two workgroups' worth of waves
#define QTTT_LAUNCH(BPL, HB, AR, SM, OB, I0, NG, KF, IDB)                                              \
    do {                                                                                              \
        if ((BPL) == 2 && (NG) >= (int64_t)QTTT_BIG_BLOCK_MIN_GROUPS) QTTT_LAUNCH_B(1024, 2, HB, AR, SM, OB, I0, NG, KF, IDB); \
        else QTTT_LAUNCH_B(QTTT_BLOCK, BPL, HB, AR, SM, OB, I0, NG, KF, IDB);                          \
    } while (0)
#define QTTT_DISPATCH(BPL, I0, NG, KF, IDB)                                                           \
    do {                                                                                              \
        if (obs) {                                                                                    \
            if (bits) { if (ar) QTTT_LAUNCH(BPL, true, true, false, true, I0, NG, KF, IDB); else QTTT_LAUNCH(BPL, true, false, false, true, I0, NG, KF, IDB); } \
            else { if (ar) QTTT_LAUNCH(BPL, false, true, false, true, I0, NG, KF, IDB); else QTTT_LAUNCH(BPL, false, false, false, true, I0, NG, KF, IDB); } \
        }                                                                                             \
        else if (sample) { if (ar) QTTT_LAUNCH(BPL, false, true, true, false, I0, NG, KF, IDB); else QTTT_LAUNCH(BPL, false, false, true, false, I0, NG, KF, IDB); } \
        else if (bits) { if (ar) QTTT_LAUNCH(BPL, true, true, false, false, I0, NG, KF, IDB); else QTTT_LAUNCH(BPL, true, false, false, false, I0, NG, KF, IDB); } \
        else { if (ar) QTTT_LAUNCH(BPL, false, true, false, false, I0, NG, KF, IDB); else QTTT_LAUNCH(BPL, false, false, false, false, I0, NG, KF, IDB); } \
    } while (0)
    // The hash folds the global board id as lo32 ^ hi32*C (fold_id).  hi32 is uniform over a
    // range of boards unless the range crosses a multiple of 2^32; the batch is cut there (at most
    // once), so the kernel only ever adds a lane index to a 32-bit base.
    int64_t seg_begin = 0;
    while (seg_begin < n) {
        const u64 first = (u64)board_offset + (u64)seg_begin;
        const u64 to_boundary = (((first >> 32) + 1u) << 32) - first;
        const int64_t seg_n = (int64_t)((u64)(n - seg_begin) < to_boundary ? (u64)(n - seg_begin) : to_boundary);
        const u32 key_fold = key_lo ^ ((u32)(first >> 32) * 0x9E3779B9u);
        const u32 id_base = (u32)first;
        int bpl = bpl_max;
        while (bpl > 1 && (seg_begin % bpl) != 0) bpl >>= 1;     // vector accesses need an aligned start
        const int64_t n_groups = seg_n / bpl, n_main = n_groups * bpl;
        if (n_groups > 0) {
            if (bpl == 4) QTTT_DISPATCH(4, seg_begin, n_groups, key_fold, id_base);
            else if (bpl == 2) QTTT_DISPATCH(2, seg_begin, n_groups, key_fold, id_base);
            else QTTT_DISPATCH(1, seg_begin, n_groups, key_fold, id_base);
        }
        if (n_main < seg_n)                                      // ragged tail, one board per lane
            QTTT_DISPATCH(1, seg_begin + n_main, seg_n - n_main, key_fold, id_base + (u32)n_main);
        seg_begin += seg_n;
    }
#undef QTTT_DISPATCH
#undef QTTT_LAUNCH
#undef QTTT_LAUNCH_B
    return launch_status();
}

int qttt_step(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
              uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
              uint8_t *terminated, int64_t n, void *stream) {
    return launch_step(state, const_cast<uint8_t *>(actions), bits, seed, step_idx, board_offset, flags,
                       reward, terminated, n, stream, false, nullptr);
}

int qttt_step_observe(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
                      uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
                      uint8_t *terminated, int8_t *classical, uint8_t *q_p1, uint8_t *q_p1_len,
                      uint8_t *q_p2, uint8_t *q_p2_len, uint8_t *turn, int64_t n, void *stream) {
    if (n > 0 && (!classical || !q_p1 || !q_p1_len || !q_p2 || !q_p2_len || !turn)) return QTTT_ERR_NULL;
    if (((uintptr_t)q_p1 & 1u) || ((uintptr_t)q_p2 & 7u)) return QTTT_ERR_ACTION;   // 2- / 8-byte LDS row stores
    const ObsOut o = {classical, q_p1, q_p1_len, q_p2, q_p2_len, turn};
    return launch_step(state, const_cast<uint8_t *>(actions), bits, seed, step_idx, board_offset, flags,
                       reward, terminated, n, stream, false, &o);
}

int qttt_step_wave_per_board(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
                             uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
                             uint8_t *terminated, int64_t n, void *stream) {
    if (n < 0 || board_offset < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !actions || !reward || !terminated) return QTTT_ERR_NULL;
    if ((uintptr_t)actions & 1u) return QTTT_ERR_ACTION;
    const u64 first = (u64)board_offset;
    if ((first >> 32) != ((first + (u64)n - 1u) >> 32)) return QTTT_ERR_SIZE;   // study kernel: one id range
    Planes p = planes(state, n);
    const u32 key_fold = (u32)launch_key(seed, step_idx) ^ ((u32)(first >> 32) * 0x9E3779B9u);
    const bool ar = (flags & QTTT_FLAG_AUTO_RESET) != 0;
    dim3 g((unsigned)((n + 3) / 4)), b(256);
    hipStream_t s = (hipStream_t)stream;
    const uint16_t *a16 = reinterpret_cast<const uint16_t *>(actions);
    u32 *rb = reinterpret_cast<u32 *>(reward);
#define QTTT_WPB(HB, AR) hipLaunchKernelGGL((step_wave_per_board_kernel<HB, AR>), g, b, 0, s, p.P, p.Q, \
                                            a16, bits, key_fold, (u32)first, rb, terminated, n)
    if (bits) { if (ar) QTTT_WPB(true, true); else QTTT_WPB(true, false); }
    else      { if (ar) QTTT_WPB(false, true); else QTTT_WPB(false, false); }
#undef QTTT_WPB
    return launch_status();
}

int qttt_step_random(void *state, uint64_t seed, uint32_t step_idx, int64_t board_offset,
                     uint32_t flags, uint8_t *actions_out, float *reward, uint8_t *terminated,
                     int64_t n, void *stream) {
    return launch_step(state, actions_out, nullptr, seed, step_idx, board_offset, flags, reward,
                       terminated, n, stream, true, nullptr);
}

int qttt_step_many(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
                   uint32_t step_idx0, int64_t board_offset, uint32_t flags, float *reward,
                   uint8_t *terminated, int64_t out_stride, int64_t n, int32_t n_steps,
                   void *stream) {
    if (n_steps < 0 || out_stride < 0) return QTTT_ERR_SIZE;
    const u64 first = (u64)(board_offset < 0 ? 0 : board_offset);
    const bool one_hi = n > 0 && (first >> 32) == ((first + (u64)n - 1u) >> 32);
    if ((flags & QTTT_FLAG_FUSED) && n > 0 && n_steps > 0 && one_hi) {
        if (board_offset < 0) return QTTT_ERR_SIZE;
        if (!state || !actions || !reward || !terminated) return QTTT_ERR_NULL;
        if ((uintptr_t)actions & 1u) return QTTT_ERR_ACTION;
        Planes p = planes(state, n);
        const bool ar = (flags & QTTT_FLAG_AUTO_RESET) != 0;
        const u32 hi_fold = (u32)(first >> 32) * 0x9E3779B9u;
        dim3 g(grid_for(n)), b(QTTT_BLOCK);
        hipStream_t s = (hipStream_t)stream;
        const uint16_t *a16 = reinterpret_cast<const uint16_t *>(actions);
        u32 *rb = reinterpret_cast<u32 *>(reward);
#define QTTT_FUSED(HB, AR)                                                                        \
    hipLaunchKernelGGL((step_fused_kernel<HB, AR>), g, b, 0, s, p.P, p.Q, a16, bits, (u64)seed, \
                       step_idx0, hi_fold, (u32)first, rb, terminated, out_stride, n, n_steps)
        if (bits) { if (ar) QTTT_FUSED(true, true); else QTTT_FUSED(true, false); }
        else      { if (ar) QTTT_FUSED(false, true); else QTTT_FUSED(false, false); }
#undef QTTT_FUSED
        return launch_status();
    }
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
    if (((uintptr_t)q_p1 & 1u) || ((uintptr_t)q_p2 & 7u)) return QTTT_ERR_ACTION;   // 2- / 8-byte LDS row stores
    Planes p = planes(const_cast<void *>(state), n);
    const ObsOut o = {classical, q_p1, q_p1_len, q_p2, q_p2_len, turn};
    hipLaunchKernelGGL(observe_kernel, dim3((unsigned)((n + 2 * QTTT_BLOCK - 1) / (2 * QTTT_BLOCK))), dim3(QTTT_BLOCK), 0,
                       (hipStream_t)stream, p.P, p.Q, o, n);
    return launch_status();
}

int qttt_check_win(const void *state, int8_t *p1_round, int8_t *p2_round, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !p1_round || !p2_round) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    hipLaunchKernelGGL(check_win_kernel, dim3(cold_grid_for(n)), dim3(QTTT_COLD_BLOCK), 0, (hipStream_t)stream,
                       p.P, p.Q, p1_round, p2_round, n);
    return launch_status();
}

int qttt_export(const void *state, uint8_t *moves, uint8_t *n_moves, int8_t *board,
                uint16_t *qmask, uint8_t *n_q, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !moves || !n_moves || !board || !qmask || !n_q) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    hipLaunchKernelGGL(export_kernel, dim3(cold_grid_for(n)), dim3(QTTT_COLD_BLOCK), 0, (hipStream_t)stream,
                       p.P, p.Q, moves, n_moves, board, qmask, n_q, n);
    return launch_status();
}

int qttt_import(void *state, const uint8_t *moves, const uint8_t *n_moves, const int8_t *board,
                const uint16_t *qmask, const uint8_t *n_q, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !moves || !n_moves || !board || !qmask || !n_q) return QTTT_ERR_NULL;
    Planes p = planes(state, n);
    hipLaunchKernelGGL(import_kernel, dim3(cold_grid_for(n)), dim3(QTTT_COLD_BLOCK), 0, (hipStream_t)stream,
                       p.P, p.Q, moves, n_moves, board, qmask, n_q, n);
    return launch_status();
}

int qttt_board_op(const void *records_in, void *records_out, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!records_in || !records_out) return QTTT_ERR_NULL;
    hipLaunchKernelGGL(board_op_kernel, dim3(cold_grid_for(n)), dim3(QTTT_COLD_BLOCK), 0, (hipStream_t)stream,
                       (const uint8_t *)records_in, (uint8_t *)records_out, n);
    return launch_status();
}

int qttt_sample_actions(const void *state, uint64_t seed, uint32_t step_idx, int64_t board_offset,
                        uint32_t flags, uint8_t *actions, int64_t n, void *stream) {
    if (n < 0 || board_offset < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !actions) return QTTT_ERR_NULL;
    if ((uintptr_t)actions & 1u) return QTTT_ERR_ACTION;   // written as u16 pairs
    Planes p = planes(const_cast<void *>(state), n);
    const u64 key = launch_key(seed, step_idx);
    hipLaunchKernelGGL(sample_actions_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0,
                       (hipStream_t)stream, p.P, (u32)key, (u32)(key >> 32), (u64)board_offset,
                       (u32)((flags & QTTT_FLAG_AUTO_RESET) != 0), reinterpret_cast<uint16_t *>(actions), n);
    return launch_status();
}

int qttt_node_info(const void *state, int8_t *winner, uint8_t *terminal, uint64_t *legal,
                   int64_t *key, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !winner || !terminal || !legal || !key) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    hipLaunchKernelGGL(node_info_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0, (hipStream_t)stream,
                       p.P, p.Q, winner, terminal, (u64 *)legal, key, n);
    return launch_status();
}

int qttt_expand(const void *state, const uint8_t *action36, void *child0, void *child1,
                uint8_t *n_children, int8_t *winner, uint8_t *terminal, uint64_t *legal,
                int64_t *key, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !action36 || !child0 || !child1 || !n_children || !winner || !terminal || !legal || !key)
        return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n), c0 = planes(child0, n), c1 = planes(child1, n);
    hipLaunchKernelGGL(expand_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0, (hipStream_t)stream,
                       p.P, p.Q, action36, c0.P, c0.Q, c1.P, c1.Q, n_children, winner,
                       terminal, (u64 *)legal, key, n);
    return launch_status();
}

int qttt_rollout(const void *state, uint64_t seed, uint32_t step_idx0, int64_t board_offset,
                 int8_t *result, uint8_t *plies, void *final_state, int64_t n, void *stream) {
    if (n < 0 || board_offset < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !result || !plies) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    Planes f = {nullptr, nullptr};
    if (final_state) f = planes(final_state, n);
    hipLaunchKernelGGL(rollout_kernel, dim3(grid_for(n)), dim3(QTTT_BLOCK), 0, (hipStream_t)stream,
                       p.P, p.Q, (u64)seed, step_idx0, (u64)board_offset, result, plies, f.P, f.Q, n);
    return launch_status();
}

int qttt_encode(const void *state, float *vec, uint8_t *mask, int64_t n, void *stream) {
    if (n < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !vec) return QTTT_ERR_NULL;
    Planes p = planes(const_cast<void *>(state), n);
    if (((uintptr_t)vec & 15u) || ((uintptr_t)mask & 3u)) return QTTT_ERR_ACTION;   // vector stores
    hipLaunchKernelGGL(encode_kernel, dim3((unsigned)((n + QTTT_ENC_BOARDS - 1) / QTTT_ENC_BOARDS)),
                       dim3(QTTT_ENC_BLOCK), 0, (hipStream_t)stream, p.P, p.Q, vec, mask, n);
    return launch_status();
}

}  // extern "C"
