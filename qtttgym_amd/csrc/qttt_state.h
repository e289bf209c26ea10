// qttt_state.h — the packed 16-byte board, its loads and stores, and the small tables every kernel
// shares (3-in-a-row table, counter hash, uniform-legal policy table).  Layout: see qttt_kernels.hip.
#ifndef QTTT_STATE_H
#define QTTT_STATE_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "qttt.h"

//@isa lane
typedef unsigned long long u64;
typedef unsigned int u32;

#ifndef QTTT_BLOCK
#define QTTT_BLOCK 512
#endif
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
#ifdef QTTT_PLAIN_LOADS                       // (A/B builds only: the loads without the non-temporal hint)
    R r = *reinterpret_cast<const R *>(p);
#else
    R r = __builtin_nontemporal_load(reinterpret_cast<const R *>(p));
#endif
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

// The same store through a block-uniform base in scalar registers + a 32-bit lane offset (the addressing the loads get
// by themselves): written out, because for these stores the compiler builds per-lane 64-bit addresses instead
// (two v_lshl_add_u64 per lane in the step kernel).  8-, 16- and 32-byte vectors.
template <typename V>
__device__ __forceinline__ void store_stream_sbase(void *block_base, u32 byte_offset, const V &v) {
    static_assert(sizeof(V) == 8 || sizeof(V) == 16 || sizeof(V) == 32, "plane vectors only");
#ifdef QTTT_NO_SBASE_STORES                    // (A/B builds only: the compiler's own addressing)
    store_stream(reinterpret_cast<V *>(static_cast<uint8_t *>(block_base) + byte_offset), v);
    return;
#endif
    if constexpr (sizeof(V) == 8) {
        u32x2 r;
        __builtin_memcpy(&r, &v, 8);
        asm volatile("global_store_dwordx2 %0, %1, %2 nt" : : "v"(byte_offset), "v"(r), "s"(block_base) : "memory");
    } else if constexpr (sizeof(V) == 16) {
        u32x4 r;
        __builtin_memcpy(&r, &v, 16);
        // (s_nop: a store of more than 64 bits followed by a VALU write of its data registers needs one wait state;
        // the compiler inserts it for its own stores, it cannot see into this one)
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : : "v"(byte_offset), "v"(r), "s"(block_base) : "memory");
    } else {
        u32x4 r[2];
        __builtin_memcpy(r, &v, 32);
        asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 0" : : "v"(byte_offset), "v"(r[0]), "s"(block_base) : "memory");
        asm volatile("global_store_dwordx4 %0, %1, %2 offset:16 nt\n\ts_nop 0" : : "v"(byte_offset), "v"(r[1]), "s"(block_base) : "memory");
    }
}

// ... and for the 1-, 2- and 4-byte per-ply outputs of the fused kernels (terminated, action, reward): the value travels
// in the low bits of one register.
template <int BYTES>
__device__ __forceinline__ void store_stream_sbase_word(void *block_base, u32 byte_offset, u32 value) {
    static_assert(BYTES == 1 || BYTES == 2 || BYTES == 4, "one register");
#ifdef QTTT_NO_SBASE_STORES
    if constexpr (BYTES == 1) store_stream(static_cast<uint8_t *>(block_base) + byte_offset, (uint8_t)value);
    else if constexpr (BYTES == 2) store_stream(reinterpret_cast<uint16_t *>(static_cast<uint8_t *>(block_base) + byte_offset), (uint16_t)value);
    else store_stream(reinterpret_cast<u32 *>(static_cast<uint8_t *>(block_base) + byte_offset), value);
    return;
#endif
    if constexpr (BYTES == 1) asm volatile("global_store_byte %0, %1, %2 nt" : : "v"(byte_offset), "v"(value), "s"(block_base) : "memory");
    else if constexpr (BYTES == 2) asm volatile("global_store_short %0, %1, %2 nt" : : "v"(byte_offset), "v"(value), "s"(block_base) : "memory");
    else asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(byte_offset), "v"(value), "s"(block_base) : "memory");
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

//@isa table
// ------------------------------------------------------------------ 3-in-a-row lookup table
// Line table: entry m = 0x7F iff the 9-bit square mask m contains one of the 8 lines of
// board.py:85-110 (0x7F << 23 = 1.0f), one entry per DWORD (LINE_LUT_BYTES = 2 KB) in LDS, because
// every mask of the hot path lives "times four" (the nibbles sit at bit 4v+2): the byte offset into
// the table is the mask itself, no shift.
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
// The same 512 answers as one bit each (bit m of the 512-bit word: mask m contains a line).  A wave covers 64
// consecutive masks = one 64-bit piece, fetched with a SCALAR load (its own counter: nothing of the wave's vector loads is
// waited for) — a shift, an AND and a multiply per table entry instead of the twenty instructions of line_lut_entry().
struct LineBits {
    u64 w[8];
    constexpr LineBits() : w() {
        for (u32 m = 0; m < 512u; ++m) {
            const u32 rows = m & (m >> 1) & (m >> 2) & 0x049u, cols = m & (m >> 3) & (m >> 6) & 0x007u;
            const bool diag = (m & 0x111u) == 0x111u || (m & 0x054u) == 0x054u;
            if ((rows | cols) != 0u || diag) w[m >> 6] |= 1ull << (m & 63u);
        }
    }
};
__constant__ LineBits g_line_bits = LineBits();
// SCALAR: 1 = always the scalar piece (kernels that are not launch-latency-bound at any size, e.g. import), -1 = by BLOCK
template <int BLOCK, int SCALAR = -1>
__device__ inline void fill_line_lut_nosync(uint8_t *lut) {
    static_assert(BLOCK % 64 == 0, "a wave covers one 64-bit piece of the table");
    // From 512 threads up (one entry per thread at most) the scalar piece; in 256-thread workgroups — the launch shape of
    // the latency-bound batches, where the scalar load's own latency sits in front of the barrier — the entry is computed
    // (tools/stepbench, 262 144 boards: 3.52 against 3.74 us best, 3.83 / 3.89 median; 1 M boards, policy in the step
    // kernel: 7.31 with the scalar piece against 7.50 computed).
#ifdef QTTT_LUT_COMPUTED                       // (A/B builds only: always computed, as up to round 4)
    constexpr bool SCALAR_PIECE = false;
#else
    constexpr bool SCALAR_PIECE = SCALAR == 1 || BLOCK >= 512;
#endif
    for (u32 w = threadIdx.x; w < 512u; w += BLOCK) {
        if constexpr (SCALAR_PIECE) {
            const u64 piece = g_line_bits.w[__builtin_amdgcn_readfirstlane(w >> 6)];  // wave-uniform: s_load_dwordx2
            reinterpret_cast<u32 *>(lut)[w] = ((u32)(piece >> (w & 63u)) & 1u) * 0x7Fu;
        } else {
            reinterpret_cast<u32 *>(lut)[w] = line_lut_entry(w);
        }
    }
}
template <int BLOCK>
__device__ inline void fill_line_lut(uint8_t *lut) {
    fill_line_lut_nosync<BLOCK>(lut);
    __syncthreads();
}

//@isa hash
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

//@isa other
// ------------------------------------------------------------------ native 64-bit position key
// GameState.__hash__ / __eq__ (mcts.py:93-97) are only ever dict keys (mcts.py:160-164,210-221): what a search
// needs is "equal keys <=> equal (board, moves)".  The packed state already is a canonical form of
// (board, moves) — reached by stepping or written by qttt_import, every field below is a function of the move
// sequence and of where the collapsed moves landed — so the key is a mix of the state's own
// words: three 64-bit multiplies instead of the 9 + n dependent multiply steps of CPython's tuple hash.
// Left out: the cached qstructs (plane Q's high word and P1 bits 12..15: list ORDER is not part of a
// position's identity and an importing caller may hand them over in another order) and the done bit.
constexpr u64 KEY_P_MASK = ~((0xFull << (32u + P1_CHI_SHIFT)) | ((u64)P1_DONE << 32));
__host__ __device__ inline u64 state_key(u64 P, u32 Q0) {
    u64 h = (P & KEY_P_MASK) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29;
    h += (u64)Q0 * 0xD6E8FEB86659FD93ull;
    h *= 0xBF58476D1CE4E5B9ull;
    return h ^ (h >> 32);
}

//@isa policy
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

// nth9[m * 9 + r] = the r-th empty square of the 9-bit mask m: kernels that run the policy for many plies per
// board (rollout, fused random stepping) use this full table (4.5 KB, computed here: thread m writes row m)
// instead of the two-level lookup of policy_nth(); behind it, nth9[NTH9_PAIRS + m] = the number of unordered pairs of
// the empty squares of m, e (e - 1) / 2 (one LDS read at a constant offset from the mask instead of a second
// popcount, a multiply and a shift per ply).
constexpr u32 NTH9_PAIRS = 512u * 9u, NTH9_BYTES = NTH9_PAIRS + 512u;
template <int BLOCK>
__device__ inline void fill_nth9(uint8_t *nth9) {
    for (u32 m = threadIdx.x; m < 512u; m += BLOCK) {
        u32 r = 0;
#pragma unroll
        for (u32 v = 0; v < 9; ++v) {
            nth9[m * 9u + r] = (uint8_t)v;                 // kept only if bit v is set (r advances), else overwritten
            r += m >> v & 1u;                              // r <= v inside the loop: the store stays in row m
        }
        nth9[NTH9_PAIRS + m] = (uint8_t)((r * (r - 1u)) >> 1);
    }
}
// The one-launch-per-step policy kernel keeps the same information as ONE word per mask: nibble r of row32[m] = the
// r-th empty square of m, r = 0..7 (rank 8 exists only for the empty board, where the square is 8), and the pair table
// pre-scaled to bit offsets: pair16[e][k] = 4 * rank_lo | (4 * rank_hi) << 5.  The row (indexed by the mask) and the
// pair (indexed by the hash) are then two INDEPENDENT LDS reads followed by two bit-field extracts, instead of the pair
// read followed by two dependent byte reads; and the tables are 2 KB + 720 B, loaded (requested in front of the state
// loads, stored to LDS behind them) rather than computed.
struct PolicyRows {
    u32 row32[512];
    uint16_t pair16[10 * 36];
    constexpr PolicyRows() : row32(), pair16() {
        for (u32 m = 0; m < 512u; ++m) {
            u32 r = 0, w = 0;
            for (u32 v = 0; v < 9; ++v)
                if (m >> v & 1u) { if (r < 8u) w |= v << (4u * r); ++r; }
            row32[m] = w;
        }
        for (int e = 0; e < 10; ++e) {
            int k = 0;
            for (int i = 0; i < e; ++i)
                for (int j = i + 1; j < e; ++j) pair16[e * 36 + k++] = (uint16_t)((4 * i) | ((4 * j) << 5));
            for (; k < 36; ++k) pair16[e * 36 + k] = 0;
        }
    }
};
__device__ const PolicyRows g_policy_rows = PolicyRows();
constexpr u32 POLICY_ROWS_WORDS = (512 * 4 + 10 * 36 * 2) / 4;      // 692
// the policy's action for the empty-square mask `empty` (>= 2 squares) from hash word h2: lo | hi << 8
__device__ __forceinline__ u32 policy_action_rows(const u32 *rows, u32 empty, u32 h2) {
    const uint16_t *pair16 = reinterpret_cast<const uint16_t *>(rows + 512);
    const u32 e = (u32)__builtin_popcount(empty);
    const u32 row = rows[empty];
    const u32 ent = pair16[e * 36u + __umulhi(h2, (e * (e - 1u)) >> 1)];
    const u32 lo = __builtin_amdgcn_ubfe(row, ent & 31u, 4u);
    const u32 hs = ent >> 5;
    const u32 hi = hs == 32u ? 8u : __builtin_amdgcn_ubfe(row, hs, 4u);
    return lo | (hi << 8);
}

// the policy's action for the empty-square mask `empty` (>= 2 squares) from hash word h2: lo | hi << 8
__device__ __forceinline__ u32 policy_action_nth9(const uint8_t *plut, const uint8_t *nth9, u32 empty, u32 h2) {
    const u32 e = (u32)__builtin_popcount(empty);
    const u32 ij = plut[e * 36u + __umulhi(h2, (u32)nth9[NTH9_PAIRS + empty])];
    return (u32)nth9[empty * 9u + (ij & 0xFu)] | ((u32)nth9[empty * 9u + (ij >> 4)] << 8);
}

}  // namespace

#endif  // QTTT_STATE_H
