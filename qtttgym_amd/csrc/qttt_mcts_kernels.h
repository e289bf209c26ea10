// qttt_mcts_kernels.h — SURVEY §8(f) rows: node info, batched expand, fused rollout, tensor encoding.
#ifndef QTTT_MCTS_KERNELS_H
#define QTTT_MCTS_KERNELS_H
#include "qttt_step_core.h"
#include "qttt_board_forms.h"
#include "qttt_aux_kernels.h"

namespace {

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
    // nth9[m * 9 + r] = the r-th empty square of the 9-bit mask m: nine plies of policy per board make a
    // full table (4.5 KB, computed here: thread m writes row m) cheaper than the two-level lookup of
    // policy_nth()
    __shared__ uint8_t nth9[512 * 9];
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    const u64 P = i < n ? pP[i] : 0ull, Q = i < n ? pQ[i] : 0ull;  // requested before the table fills
    fill_policy_lut<QTTT_BLOCK>(plut);
    for (u32 m = threadIdx.x; m < 512u; m += QTTT_BLOCK) {
        u32 r = 0;
#pragma unroll
        for (u32 v = 0; v < 9; ++v) {
            nth9[m * 9u + r] = (uint8_t)v;                 // kept only if bit v is set (r advances), else overwritten
            r += m >> v & 1u;                              // r <= v inside the loop: the store stays in row m
        }
    }
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
        // policy_action() with the full table: the k-th legal pair (i < j) -> squares (a < b)
        const u32 e = (u32)__builtin_popcount(empty);
        const u32 ij = plut[e * 36u + __umulhi(h2, (e * (e - 1u)) >> 1)];
        const u32 act = (u32)nth9[empty * 9u + (ij & 0xFu)] | ((u32)nth9[empty * 9u + (ij >> 4)] << 8);
        step_core<false, true>(P0, P1, Q0, Q1, act, h1 >> 31, lut);   // legal and sorted
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

}  // namespace

#endif  // QTTT_MCTS_KERNELS_H
