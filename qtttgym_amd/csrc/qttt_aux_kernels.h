// qttt_aux_kernels.h — kernels beside the step: observation of stored boards, check_win, export /
// import of Board attributes, the single-record Board façade op, action sampling.
#ifndef QTTT_AUX_KERNELS_H
#define QTTT_AUX_KERNELS_H
#include "qttt_step_core.h"
#include "qttt_observation.h"
#include "qttt_board_forms.h"

namespace {

#define QTTT_COLD_BLOCK 256

// Env._observation (env.py:68-85) of stored boards: two boards per lane (one 16-byte load per
// plane, as the step kernel) through the same LDS tiles and the same obs_board() as the fused step
// kernel.  A workgroup owns 2 * BLOCK consecutive boards; the last board of an odd batch is
// read with scalar loads.
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void observe_kernel(const u64 *pP, const u64 *pQ, ObsOut obs, int64_t n) {
    constexpr u32 TILE_BOARDS = BLOCK * 2;
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
        obs_copy_out<BLOCK, TILE_BOARDS>(otile, obs, base, valid);
    }
}

// Board.check_win (board.py:71-115) of stored boards: two boards per lane (one 16-byte load of plane
// P, one 2-byte store per output), the line table in LDS.
__global__ __launch_bounds__(QTTT_COLD_BLOCK) void check_win_kernel(
    const u64 *pP, const u64 *pQ, int8_t *p1_round, int8_t *p2_round, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    (void)pQ;
    const int64_t j = (int64_t)blockIdx.x * QTTT_COLD_BLOCK + threadIdx.x;   // boards 2j, 2j+1
    const int64_t i0 = 2 * j;
    typedef Vec<u64, 2> V64;
    V64 p;
    p.v[0] = p.v[1] = 0ull;
    if (i0 + 1 < n) p = load_stream(&reinterpret_cast<const V64 *>(pP)[j]);    // requested before the table fill
    else if (i0 < n) p.v[0] = pP[i0];
    fill_line_lut<QTTT_COLD_BLOCK>(lut);                  // ends with the workgroup barrier
    if (i0 >= n) return;
    int a1, a2, b1, b2;
    fast_check_win(lite_unpack(p.v[0]), lut, a1, a2);
    fast_check_win(lite_unpack(p.v[1]), lut, b1, b2);
    if (i0 + 1 < n && ((reinterpret_cast<uintptr_t>(p1_round) | reinterpret_cast<uintptr_t>(p2_round)) & 1u) == 0u) {
        reinterpret_cast<uint16_t *>(p1_round)[j] = (uint16_t)((u32)(a1 & 0xFF) | ((u32)(b1 & 0xFF) << 8));
        reinterpret_cast<uint16_t *>(p2_round)[j] = (uint16_t)((u32)(a2 & 0xFF) | ((u32)(b2 & 0xFF) << 8));
    } else {
        p1_round[i0] = (int8_t)a1;
        p2_round[i0] = (int8_t)a2;
        if (i0 + 1 < n) { p1_round[i0 + 1] = (int8_t)b1; p2_round[i0 + 1] = (int8_t)b2; }
    }
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

// legal pairs in ind2move order: for lo ascending, hi ascending (mcts.py:20-27, 339-343).  Two boards
// per lane: one 16-byte load of plane P, one 4-byte store of the two actions.
__device__ __forceinline__ u32 sampled_action(const uint8_t *plut, u64 Pw, u64 board_id, u32 key_lo, u32 key_hi,
                                              u32 auto_reset) {
    const u32 P1 = (u32)(Pw >> 32);
    const u32 cl = (auto_reset && (P1 >> 31)) ? 0u : (P1 >> P1_CL_SHIFT) & 0x1FFu;
    const u32 empty = ~cl & 0x1FFu;
    const u32 h1 = lowbias32(fold_id(board_id) ^ key_lo);
    const u32 h2 = lowbias32(h1 ^ key_hi);
    // fewer than two empty squares: rank_pair gives (0,0) and nth_bit[..][0] twice -> a == b, a noop;
    // the spec says (0,0)
    return (empty & (empty - 1u)) ? policy_action(plut, empty, h2) : 0u;
}

__global__ __launch_bounds__(QTTT_BLOCK) void sample_actions_kernel(
    const u64 *pP, u32 key_lo, u32 key_hi, u64 board_offset, u32 auto_reset, uint16_t *actions,
    int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    const int64_t j = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;   // boards 2j, 2j+1
    const int64_t i0 = 2 * j;
    typedef Vec<u64, 2> V64;
    V64 p;
    p.v[0] = p.v[1] = 0ull;
    if (i0 + 1 < n) p = load_stream(&reinterpret_cast<const V64 *>(pP)[j]);   // requested before the table: the latencies overlap
    else if (i0 < n) p.v[0] = pP[i0];
    fill_policy_lut<QTTT_BLOCK>(plut);
    __syncthreads();
    if (i0 >= n) return;
    const u32 a0 = sampled_action(plut, p.v[0], board_offset + (u64)i0, key_lo, key_hi, auto_reset);
    if (i0 + 1 < n) {
        const u32 a1 = sampled_action(plut, p.v[1], board_offset + (u64)i0 + 1u, key_lo, key_hi, auto_reset);
        if ((reinterpret_cast<uintptr_t>(actions) & 3u) == 0u) reinterpret_cast<u32 *>(actions)[j] = a0 | (a1 << 16);
        else { actions[i0] = (uint16_t)a0; actions[i0 + 1] = (uint16_t)a1; }
    } else {
        actions[i0] = (uint16_t)a0;
    }
}

}  // namespace

#endif  // QTTT_AUX_KERNELS_H
