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
// WAVE_LUT: every wave keeps its own copy of the 128-byte selector table (lanes 0..31 copy one dword each), so
// the kernel has no workgroup barrier at all: a wave depends on nothing but its own loads.
template <int BLOCK, bool WAVE_LUT = true>
__global__ __launch_bounds__(BLOCK) void observe_kernel(const u64 *pP, const u64 *pQ, ObsOut obs, int64_t n) {
    constexpr u32 TILE_BOARDS = BLOCK * 2;
    __shared__ __attribute__((aligned(16))) uint8_t otile[obs_lds_bytes(TILE_BOARDS)];
    __shared__ __attribute__((aligned(16))) u32 olut_all[(WAVE_LUT ? BLOCK / 64 : 1) * (OBS_LUT_BYTES / 4)];
    const u32 lane = threadIdx.x & 63u;
    u32 *olut = olut_all + (WAVE_LUT ? (threadIdx.x >> 6) * (OBS_LUT_BYTES / 4) : 0u);
    const u32 lw = WAVE_LUT ? lane : threadIdx.x;
    const u32 olw = lw < OBS_LUT_BYTES / 4 ? (&g_obs_lut.sel[0][0])[lw] : 0u;   // requested first (see step_kernel)
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
    if (lw < OBS_LUT_BYTES / 4) olut[lw] = olw;
    if (WAVE_LUT) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
    if (b0 < valid) obs_board((u32)p.v[0], (u32)(p.v[0] >> 32), (u32)q.v[0], T, b0, olut);
    if (b0 + 1u < valid) obs_board((u32)p.v[1], (u32)(p.v[1] >> 32), (u32)q.v[1], T, b0 + 1u, olut);
    const u32 ph = obs_all_phases(obs, base, 15u);
    if ((ph & 3u) == 0u) {                                // every wave streams out the rows it wrote (see step_kernel)
        const u32 w0 = (threadIdx.x & ~63u) * 2u;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (w0 < valid) {
            if (ph == 0u) obs_wave_copy_out<TILE_BOARDS, true>(otile, obs, base, w0, min(w0 + 128u, valid));
            else obs_wave_copy_out<TILE_BOARDS, false>(otile, obs, base, w0, min(w0 + 128u, valid));
        }
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

// Board.moves / .board / .qstructs (board.py:4-6) as arrays, straight from the packed words, through
// LDS tiles laid out like the outputs (as the observation): a lane builds its board's rows as packed
// words (7 LDS stores), the wave then streams its 64 rows of every tile out with 16-byte stores.
// Every output is nullable (VecEnv.turn() asks for n_moves alone: 8 B read, 1 B written per board).
//   moves: the move of round t is (c, c ^ x_t) for its holder c — the one square whose nibble holds
//     round t (child end of a live move, landing square of a collapsed one, the autofilled square for
//     the autofill move, whose x is 0).  H inverts the holders: nibble 8 - t = holder + 1.  Rounds
//     0..7 are then handled bytewise, four at a time (even / odd nibbles of H and of the x word), and
//     put in round order with v_perm_b32; unused rounds read 255, 255.
struct ExpOut {
    uint8_t *moves;             // [n,9,2]
    uint8_t *n_moves;           // [n]
    int8_t *board;              // [n,9]
    uint16_t *qmask;            // [n,4]
    uint8_t *n_q;               // [n]
};
constexpr u32 exp_lds_bytes(u32 boards) {
    return obs_tile_bytes(boards, 18) + obs_tile_bytes(boards, 9) + obs_tile_bytes(boards, 8) +
           2u * obs_tile_bytes(boards, 1);
}

// (lo, hi) byte pairs of four rounds: h = holder + 1 per byte (0 = round not played), x = lo ^ hi per byte.
// Returns lo bytes, hi bytes (255 where not played).
__device__ __forceinline__ void exp_pairs(u32 h, u32 x, u32 &lo, u32 &hi) {
    const u32 live01 = ((h + 0x0F0F0F0Fu) >> 4) & 0x01010101u;        // 1 where h != 0 (h <= 9)
    const u32 c = h - live01;                                          // the holder square
    const u32 o = c ^ x;                                               // the other end of its move
    const u32 ge = (((c | 0x10101010u) - o) >> 4) & 0x01010101u;       // bytewise c >= o (both < 16)
    const u32 gm = (ge << 8) - ge;
    const u32 dead01 = live01 ^ 0x01010101u;
    const u32 dead = (dead01 << 8) - dead01;                           // 0xFF where not played
    lo = ((o & gm) | (c & ~gm)) | dead;
    hi = ((c & gm) | (o & ~gm)) | dead;
}

// the rows of board b of the tile, from its packed words
struct ExpTiles {
    uint8_t *mv, *bd, *qm, *nm, *nq;                                  // already phase-shifted; null = not asked for
};
__device__ __forceinline__ void export_board(u64 Pw, u64 Qw, const ExpTiles &T, u32 b) {
    const u32 P1 = (u32)(Pw >> 32), Q0 = (u32)Qw;
    const Lite s = lite_unpack(Pw);                                   // the implicit autofill materialised
    const u32 W = (u32)(s.P >> 2);                                    // codes of squares 0..7
    const u32 c8 = (u32)(s.P >> 34) & 0xFu;
    if (T.nm) T.nm[b] = (uint8_t)s.n;
    if (T.bd) {
        // Board.board: 15 - code where classical, -1 elsewhere (nibbles -> bytes with two v_perm)
        const u32 ev = W & 0x0F0F0F0Fu, od = (W >> 4) & 0x0F0F0F0Fu;
        const u32 c03 = __builtin_amdgcn_perm(od, ev, 0x05010400u), c47 = __builtin_amdgcn_perm(od, ev, 0x07030602u);
        const u32 t03 = __umul24(s.cl & 0xFu, 0x204081u) & 0x01010101u;
        const u32 t47 = __umul24((s.cl >> 4) & 0xFu, 0x204081u) & 0x01010101u;
        const u32 m03 = (t03 << 8) - t03, m47 = (t47 << 8) - t47;
        const u64 o07 = (u64)((c03 ^ 0x0F0F0F0Fu) | ~m03) | ((u64)((c47 ^ 0x0F0F0F0Fu) | ~m47) << 32);
        uint8_t *r = T.bd + b * 9u;
        __builtin_memcpy(r, &o07, 8);
        r[8] = (uint8_t)((s.cl & 0x100u) ? (c8 ^ 0xFu) : 0xFFu);
    }
    if (T.mv) {
        u64 H = 0;                                                    // nibble (code - 7) = holder square + 1
#pragma unroll
        for (u32 v = 0; v < 8; ++v) H |= (u64)(v + 1u) << ((((W >> (4u * v)) & 0xFu) * 4u + 36u) & 63u);
        H |= 9ull << ((c8 * 4u + 36u) & 63u);
        const u32 last_x = (P1 >> P1_LX_SHIFT) & 0xFu;
        const bool nine = s.n_real == 9u;
        const u32 Hs = (u32)(H >> 4);                                 // nibble 7 - t = holder + 1 of round t <= 7
        const u32 X = rotr32(Q0, 2u) ^ (nine ? last_x << 28 : 0u);     // nibble 7 - t = x of round t (round 8's x undone)
        u32 lo_e, hi_e, lo_o, hi_o;                                   // bytes 0..3 = rounds 7,5,3,1 / 6,4,2,0
        exp_pairs(Hs & 0x0F0F0F0Fu, X & 0x0F0F0F0Fu, lo_e, hi_e);
        exp_pairs((Hs >> 4) & 0x0F0F0F0Fu, (X >> 4) & 0x0F0F0F0Fu, lo_o, hi_o);
        const u32 A = __builtin_amdgcn_perm(hi_o, lo_o, 0x06020703u);       // rounds 0, 2 as (lo, hi) pairs
        const u32 B = __builtin_amdgcn_perm(hi_e, lo_e, 0x06020703u);       // rounds 1, 3
        const u32 C = __builtin_amdgcn_perm(hi_o, lo_o, 0x04000501u);       // rounds 4, 6
        const u32 D = __builtin_amdgcn_perm(hi_e, lo_e, 0x04000501u);       // rounds 5, 7
        const u64 m03 = (u64)__builtin_amdgcn_perm(B, A, 0x05040100u) | ((u64)__builtin_amdgcn_perm(B, A, 0x07060302u) << 32);
        const u64 m47 = (u64)__builtin_amdgcn_perm(D, C, 0x05040100u) | ((u64)__builtin_amdgcn_perm(D, C, 0x07060302u) << 32);
        const u32 h8 = (u32)H & 0xFu;                                 // round 8: the last real move or the autofill (x = 0)
        const u32 c = h8 - 1u, o = c ^ (nine ? last_x : 0u);
        const uint16_t m8 = h8 ? (uint16_t)(min(c, o) | (max(c, o) << 8)) : (uint16_t)0xFFFFu;
        uint8_t *r = T.mv + b * 18u;
        __builtin_memcpy(r, &m03, 8);
        __builtin_memcpy(r + 8, &m47, 8);
        __builtin_memcpy(r + 16, &m8, 2);
    }
    if (T.qm || T.nq) {
        const u64 comps = (Qw >> 32) | ((u64)((P1 >> P1_CHI_SHIFT) & 0xFu) << 32);
        const u32 c32 = (u32)comps;
        if (T.qm) {
            const u64 q = (u64)((c32 & 0x1FFu) | ((c32 << 7) & 0x01FF0000u)) |
                          ((u64)(((c32 >> 18) & 0x1FFu) | ((u32)(comps >> 11) & 0x01FF0000u)) << 32);
            __builtin_memcpy(T.qm + b * 8u, &q, 8);
        }
        if (T.nq) {                                                   // slots are compact: count the non-empty ones
            const u32 nz = ((((c32 & 0x03FDFEFFu) + 0x03FDFEFFu) | c32) & 0x04020100u) | ((comps >> 27) ? 1u : 0u);
            T.nq[b] = (uint8_t)__builtin_popcount(nz);
        }
    }
}

// BPL consecutive boards per lane (16-byte plane loads with two), a workgroup owns BLOCK * BPL boards.
template <int BLOCK, int BPL>
__global__ __launch_bounds__(BLOCK) void export_kernel(const u64 *pP, const u64 *pQ, ExpOut out, int64_t n) {
    constexpr u32 TILE = BLOCK * BPL;
    __shared__ __attribute__((aligned(16))) uint8_t tile[exp_lds_bytes(TILE)];
    const int64_t base = (int64_t)blockIdx.x * TILE;
    const u32 valid = (u32)min((int64_t)TILE, n - base);
    // global rows of this workgroup and the tiles that mirror them (same alignment phase, see obs_tiles)
    uint8_t *g_mv = out.moves + base * 18, *g_bd = reinterpret_cast<uint8_t *>(out.board) + base * 9;
    uint8_t *g_qm = reinterpret_cast<uint8_t *>(out.qmask) + base * 8, *g_nm = out.n_moves + base, *g_nq = out.n_q + base;
    uint8_t *l_mv = tile, *l_bd = l_mv + obs_tile_bytes(TILE, 18), *l_qm = l_bd + obs_tile_bytes(TILE, 9);
    uint8_t *l_nm = l_qm + obs_tile_bytes(TILE, 8), *l_nq = l_nm + obs_tile_bytes(TILE, 1);
    ExpTiles T;
    T.mv = out.moves ? l_mv + obs_phase(g_mv) : nullptr;
    T.bd = out.board ? l_bd + obs_phase(g_bd) : nullptr;
    T.qm = out.qmask ? l_qm + obs_phase(g_qm) : nullptr;
    T.nm = out.n_moves ? l_nm + obs_phase(g_nm) : nullptr;
    T.nq = out.n_q ? l_nq + obs_phase(g_nq) : nullptr;
    const bool want_q = out.moves || out.qmask || out.n_q;            // plane Q: the x nibbles and the comps
    const u32 b0 = threadIdx.x * BPL;
    typedef Vec<u64, BPL> V64;
    V64 p, q;
#pragma unroll
    for (int k = 0; k < BPL; ++k) p.v[k] = q.v[k] = 0ull;
    if (b0 + BPL <= valid) {
        p = load_stream(&reinterpret_cast<const V64 *>(pP + base)[threadIdx.x]);
        if (want_q) q = load_stream(&reinterpret_cast<const V64 *>(pQ + base)[threadIdx.x]);
    } else {
#pragma unroll
        for (int k = 0; k < BPL; ++k)
            if (b0 + k < valid) { p.v[k] = pP[base + b0 + k]; if (want_q) q.v[k] = pQ[base + b0 + k]; }
    }
#pragma unroll
    for (int k = 0; k < BPL; ++k)
        if (b0 + k < valid) export_board(p.v[k], q.v[k], T, b0 + k);
    const uintptr_t all = (out.moves ? (uintptr_t)g_mv : 0) | (out.board ? (uintptr_t)g_bd : 0) |
                          (out.qmask ? (uintptr_t)g_qm : 0) | (out.n_moves ? (uintptr_t)g_nm : 0) | (out.n_q ? (uintptr_t)g_nq : 0);
    if ((all & 3u) == 0u) {                                           // every wave streams out the rows it wrote
        const u32 w0 = (threadIdx.x & ~63u) * BPL, w1 = min(w0 + 64u * BPL, valid);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (w0 < valid) {
#define QTTT_XCO(PTR, G, L, ROW) do { if (PTR) { if ((all & 15u) == 0u) wave_copy_out16(G, L, w0 * (ROW), w1 * (ROW)); else wave_copy_out(G, L, w0 * (ROW), w1 * (ROW)); } } while (0)
            QTTT_XCO(out.moves, g_mv, l_mv, 18u);
            QTTT_XCO(out.board, g_bd, l_bd, 9u);
            QTTT_XCO(out.qmask, g_qm, l_qm, 8u);
            QTTT_XCO(out.n_moves, g_nm, l_nm, 1u);
            QTTT_XCO(out.n_q, g_nq, l_nq, 1u);
#undef QTTT_XCO
        }
    } else {
        __syncthreads();
        if (out.moves) tile_copy_out<BLOCK>(g_mv, l_mv, valid * 18u);
        if (out.board) tile_copy_out<BLOCK>(g_bd, l_bd, valid * 9u);
        if (out.qmask) tile_copy_out<BLOCK>(g_qm, l_qm, valid * 8u);
        if (out.n_moves) tile_copy_out<BLOCK>(g_nm, l_nm, valid);
        if (out.n_q) tile_copy_out<BLOCK>(g_nq, l_nq, valid);
    }
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
    // root every tree of live edges: grow from the lowest square of each tree.  The given qstructs name the trees
    // (their lowest squares start as roots, all trees grow at once: passes = the deepest tree's depth); whatever
    // they do not cover — attributes a caller assigned inconsistently — is picked up tree by tree below
    u32 rooted = 0;
    for (u32 k = 0; k < nq; ++k) {
        const u32 m = (u32)(s.comps >> (9u * k)) & 0x1FFu & ~s.cl;
        rooted |= m & (0u - m);
    }
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

// ---------------------------------------------------------------------------------------------------------
// qttt_import for batches: the inverse of export_kernel.  The five input arrays come in through LDS tiles laid out
// like the inputs (16-byte global loads by the wave that owns the rows), a lane reads its board's rows with wide
// LDS loads and builds the packed words directly:
//   * classical mask and round codes from the nine board bytes, nibble-parallel;
//   * the x = lo ^ hi nibbles of Q0 from the (lo, hi) byte pairs (padding 255 ^ 255 and an autofill move (idx, idx)
//     give 0 by themselves);
//   * the explicit autofill move (board.py:22-25) stripped back to the implicit form;
//   * the rooted forest: the un-collapsed moves are inserted in round order exactly as the step inserts a move —
//     re-root the tree of one end (path reversal, the step's own walk, the step's own choice of the end), hang it
//     under the other end: import(export(state)) is the state again, bit for bit;
//   * qstructs from the caller's masks as they are; done = a line or nine moves (env.py:51).
// Valid (reachable) attribute sets give a state whose export is the input again; anything else gives some state
// without faults or unbounded loops (the walk is bounded by nine nodes).  The single-record Board façade keeps the
// generic cold_from_attrs path.
//@isa imp_copy
template <int BLOCK>
__device__ inline void tile_copy_in(uint8_t *tile16, const uint8_t *gsrc, u32 nbytes) {
    uint8_t *dst = tile16 + obs_phase(gsrc);
    for (u32 k = threadIdx.x; k < nbytes; k += BLOCK) dst[k] = gsrc[k];
}
__device__ inline void wave_copy_in16(uint8_t *tile16, const uint8_t *gsrc, u32 begin, u32 end) {
    const u32 lane = threadIdx.x & 63u;
    const u32x4 *gq = reinterpret_cast<const u32x4 *>(gsrc);
    u32x4 *sq = reinterpret_cast<u32x4 *>(tile16);
    const u32 q0 = begin >> 4, q1 = end >> 4;                       // begin is a multiple of 16
    for (u32 k = q0 + lane; k < q1; k += 64u) sq[k] = __builtin_nontemporal_load(&gq[k]);
    const u32 kb = (q1 << 4) + lane;                                 // < 16 bytes left
    if (kb < end) tile16[kb] = gsrc[kb];
}

//@isa imp_head
// low nibbles of the four bytes of x -> 16 bits (byte k -> nibble k)
__device__ __forceinline__ u32 nibbles_of_bytes(u32 x) {
    const u32 y = (x | (x >> 4)) & 0x00FF00FFu;
    return (y | (y >> 8)) & 0xFFFFu;
}
// (lo, hi) byte pairs of four rounds e0..e0+3 -> x nibbles in the order Q0 wants: x_e0 in the highest nibble
__device__ __forceinline__ u32 x_nibbles_of_pairs(u64 m) {
    const u64 xr = m ^ (m >> 8);                                     // bytes 0, 2, 4, 6 = lo ^ hi
    const u32 a = (u32)xr & 0x000F000Fu, b = (u32)(xr >> 32) & 0x000F000Fu;
    return (((a << 12) | (a >> 8)) & 0xFF00u) | (((b << 4) | (b >> 16)) & 0x00FFu);
}

__device__ __forceinline__ void import_board(u64 m03, u64 m47, u32 m8, u32 nmv, u64 b07, u32 b8, u64 qm, u32 nq,
                                             const uint8_t *lut, u64 &Pout, u64 &Qout) {
    const u32 n = min(nmv, 9u);
    // ---- Board.board: classical mask + codes (15 - round), squares 0..7 nibble-parallel
    const u32 blo = (u32)b07, bhi = (u32)(b07 >> 32);
    const u32 clo = (~blo >> 7) & 0x01010101u, chi = (~bhi >> 7) & 0x01010101u;         // 1 per classical byte
    u32 cl = ((clo * 0x01020408u) >> 24) | (((chi * 0x01020408u) >> 24) << 4) | ((b8 & 0x80u) ? 0u : 0x100u);
    u32 W = nibbles_of_bytes(~blo & 0x0F0F0F0Fu & ((clo << 8) - clo)) |
            (nibbles_of_bytes(~bhi & 0x0F0F0F0Fu & ((chi << 8) - chi)) << 16);
    u32 c8 = (b8 & 0x80u) ? 0u : (~b8 & 0xFu);
    // ---- Board.moves: x nibbles; the last move (an explicit autofill move is stripped: board.py:22-25 is implicit here)
    // the autofill move is (idx, idx, 8), always of round 8 (eight squares must have collapsed first: SURVEY.md §8a), so
    // it can only be the ninth entry
    const u32 l_lo = m8 & 0xFFu, l_hi = (m8 >> 8) & 0xFFu;
    const bool is_auto = n == 9u && l_lo == l_hi && l_lo < 9u;
    const u32 n_real = is_auto ? 8u : n;
    if (is_auto) {
        cl &= ~(1u << l_lo);
        if (l_lo < 8u) W &= ~(0xFu << (4u * l_lo)); else c8 = 0u;
    }
    const u32 Xn = (x_nibbles_of_pairs(m03) << 16) | x_nibbles_of_pairs(m47);   // nibble 7 - e = x of round e
    const u32 x8 = n_real == 9u ? ((m8 ^ (m8 >> 8)) & 0xFu) : 0u;
    u32 Q0 = __builtin_amdgcn_alignbit(Xn, Xn, 30u) ^ rotr32(x8, 2u);             // rotl 2; round 8's x onto round 0's nibble
    const u32 last_x = n_real == 0u ? 0u : (n_real == 9u ? x8 : (Xn >> (4u * (8u - n_real))) & 0xFu);
    // ---- Board.qstructs as given: 4 x 9 bits, list order
    const u32 nqc = min(nq, 4u);
    const u64 comps_all = (u64)((u32)qm & 0x1FFu) | ((u64)((u32)(qm >> 16) & 0x1FFu) << 9) |
                          ((u64)((u32)(qm >> 32) & 0x1FFu) << 18) | ((u64)((u32)(qm >> 48) & 0x1FFu) << 27);
    const u64 comps = nqc >= 4u ? comps_all : comps_all & ((1ull << (9u * nqc)) - 1ull);
//@isa imp_moves
    // ---- the rooted forest: insert the un-collapsed moves in round order (the step's own path reversal)
    // The child end is chosen as the step chose it when the move was played (step_child_end4: lo when lo was free of
    // un-collapsed moves, else hi), so that an imported position is bit for bit the state stepping reaches —
    // which is what lets state_key() stand for (board, moves).  "In a component then" = touched by an earlier move
    // that is still un-collapsed now: a component collapses as a whole, so a square whose old component is gone is
    // classical, and so would be every move on it.  (A move of round 8 always closes a cycle: never live.)
    // WHICH moves are un-collapsed is read off the board once, for all rounds: a move has collapsed iff its round
    // stands on a square (env.py:72-74 uses the same test), so live = rounds < n_real that are on no square — nine
    // one-bit shifts instead of two classical-square tests and three range tests per move.  (Attributes no game
    // reaches give SOME state: a garbage square only ever becomes a shift count, which the hardware reduces, and the
    // walk is bounded by nine nodes.)
    u32 on_board = (1u << (b8 & 31u));                                   // (an empty square, -1, sets bit 31: never looked at)
#pragma unroll
    for (u32 v = 0; v < 8u; ++v) on_board |= 1u << ((u32)(b07 >> (8u * v)) & 31u);
    if (is_auto) on_board &= ~(1u << (n_real & 31u));                   // the stripped autofill round is not a move's
    const u32 live = ~on_board & ((1u << n_real) - 1u) & 0xFFu;
    u64 P = ((u64)W << 2) | ((u64)c8 << 34);
    u32 touched = 0;
#pragma unroll
    for (u32 t = 0; t < 8u; ++t) {
        if (live & (1u << t)) {
            const u32 pr = (u32)((t < 4u ? m03 : m47) >> (16u * (t & 3u)));
            const u32 lo = pr & 0xFu, hi = (pr >> 8) & 0xFu;                 // (squares are 0..8; four bits keep every shift below defined)
            const u32 x = ((touched >> lo) & 1u) == 0u ? lo : hi;            // step_child_end4 without a cycle
            P = step_reroot(P, Q0, x * 4u, t * 4u);                          // x becomes the child end of move t
            touched |= (1u << lo) | (1u << hi);
        }
    }
//@isa imp_tail
    u32 P0 = (u32)P, P1 = (u32)(P >> 32) & 0x3Fu;
    P1 |= (n_real << P1_N_SHIFT) | (((u32)(comps >> 32) & 0xFu) << P1_CHI_SHIFT) | (last_x << P1_LX_SHIFT) | (cl << P1_CL_SHIFT);
    step_line(P0, P1, lut);                                              // done = a completed line or nine moves (env.py:51)
    Pout = (u64)P0 | ((u64)P1 << 32);
    Qout = (u64)Q0 | ((u64)(u32)comps << 32);
}

//@isa imp_copy
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void import_kernel(u64 *pP, u64 *pQ, ExpOut in, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t tile[exp_lds_bytes(BLOCK)];
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    const int64_t base = (int64_t)blockIdx.x * BLOCK;
    const u32 valid = (u32)min((int64_t)BLOCK, n - base);
    const uint8_t *g_mv = in.moves + base * 18, *g_bd = reinterpret_cast<const uint8_t *>(in.board) + base * 9;
    const uint8_t *g_qm = reinterpret_cast<const uint8_t *>(in.qmask) + base * 8, *g_nm = in.n_moves + base, *g_nq = in.n_q + base;
    uint8_t *l_mv = tile, *l_bd = l_mv + obs_tile_bytes(BLOCK, 18), *l_qm = l_bd + obs_tile_bytes(BLOCK, 9);
    uint8_t *l_nm = l_qm + obs_tile_bytes(BLOCK, 8), *l_nq = l_nm + obs_tile_bytes(BLOCK, 1);
    const uintptr_t all = (uintptr_t)g_mv | (uintptr_t)g_bd | (uintptr_t)g_qm | (uintptr_t)g_nm | (uintptr_t)g_nq;
    fill_line_lut_nosync<BLOCK, 1>(lut);
    if ((all & 15u) == 0u) {                                             // every wave fetches the rows it will read
        const u32 w0 = threadIdx.x & ~63u, w1 = min(w0 + 64u, valid);
        if (w1 == w0 + 64u) {
            // a full wave: 64 rows = 72 + 36 + 32 + 4 + 4 sixteen-byte pieces.  ALL six loads are requested before the
            // first one is waited for (the generic loop below waits for each piece before it asks for the next: six
            // memory round trips in a row per wave — 20.9 us per 1 M boards against 9 for the export, round 4)
            const u32 lane = threadIdx.x & 63u;
            const u32x4 *q_mv = reinterpret_cast<const u32x4 *>(g_mv + w0 * 18u), *q_bd = reinterpret_cast<const u32x4 *>(g_bd + w0 * 9u);
            const u32x4 *q_qm = reinterpret_cast<const u32x4 *>(g_qm + w0 * 8u), *q_nm = reinterpret_cast<const u32x4 *>(g_nm + w0);
            const u32x4 *q_nq = reinterpret_cast<const u32x4 *>(g_nq + w0);
            u32x4 a0, a1, b0, c0, d0, e0;
            a0 = __builtin_nontemporal_load(&q_mv[lane]);
            if (lane < 8u) a1 = __builtin_nontemporal_load(&q_mv[64u + lane]);
            if (lane < 36u) b0 = __builtin_nontemporal_load(&q_bd[lane]);
            if (lane < 32u) c0 = __builtin_nontemporal_load(&q_qm[lane]);
            if (lane < 4u) { d0 = __builtin_nontemporal_load(&q_nm[lane]); e0 = __builtin_nontemporal_load(&q_nq[lane]); }
            reinterpret_cast<u32x4 *>(l_mv + w0 * 18u)[lane] = a0;
            if (lane < 8u) reinterpret_cast<u32x4 *>(l_mv + w0 * 18u)[64u + lane] = a1;
            if (lane < 36u) reinterpret_cast<u32x4 *>(l_bd + w0 * 9u)[lane] = b0;
            if (lane < 32u) reinterpret_cast<u32x4 *>(l_qm + w0 * 8u)[lane] = c0;
            if (lane < 4u) { reinterpret_cast<u32x4 *>(l_nm + w0)[lane] = d0; reinterpret_cast<u32x4 *>(l_nq + w0)[lane] = e0; }
        } else if (w0 < valid) {
            wave_copy_in16(l_mv, g_mv, w0 * 18u, w1 * 18u);
            wave_copy_in16(l_bd, g_bd, w0 * 9u, w1 * 9u);
            wave_copy_in16(l_qm, g_qm, w0 * 8u, w1 * 8u);
            wave_copy_in16(l_nm, g_nm, w0, w1);
            wave_copy_in16(l_nq, g_nq, w0, w1);
        }
    } else {
        tile_copy_in<BLOCK>(l_mv, g_mv, valid * 18u);
        tile_copy_in<BLOCK>(l_bd, g_bd, valid * 9u);
        tile_copy_in<BLOCK>(l_qm, g_qm, valid * 8u);
        tile_copy_in<BLOCK>(l_nm, g_nm, valid);
        tile_copy_in<BLOCK>(l_nq, g_nq, valid);
    }
    __syncthreads();                                                     // the line table (and, misaligned, the tiles)
    const u32 b = threadIdx.x;
    if (b >= valid) return;
    const uint8_t *r_mv = l_mv + obs_phase(g_mv) + b * 18u, *r_bd = l_bd + obs_phase(g_bd) + b * 9u;
    u64 m03, m47, b07, qm;
    uint16_t m8;
    __builtin_memcpy(&m03, r_mv, 8);
    __builtin_memcpy(&m47, r_mv + 8, 8);
    __builtin_memcpy(&m8, r_mv + 16, 2);
    __builtin_memcpy(&b07, r_bd, 8);
    __builtin_memcpy(&qm, l_qm + obs_phase(g_qm) + b * 8u, 8);
    u64 P, Q;
    import_board(m03, m47, m8, (l_nm + obs_phase(g_nm))[b], b07, r_bd[8], qm, (l_nq + obs_phase(g_nq))[b], lut, P, Q);
    store_stream(&pP[base + b], P);
    store_stream(&pQ[base + b], Q);
}

// Board.make_move / update_qstructs / check_win (board.py:9-115) on caller-assigned attributes, one
//@isa other
// 64-byte record in, one out (include/qttt.h: qttt_board_op): import -> the SAME step_core the
// batch kernels run -> export + check_win, in one launch, so that the single-board façade costs one
// round trip.  The records may live in pinned host memory (the kernel reads and writes them
// directly).
// stamp != 0: the record's last byte receives it after everything else of the record is visible system-wide, so a host
// that owns the (pinned) records can poll for completion instead of synchronising the stream (qttt_board_op_host).
// one record: r -> o (any address space: global, pinned host memory, or the mailbox kernel's LDS copies)
__device__ __forceinline__ void board_op_record(const uint8_t *r, uint8_t *o, const uint8_t *lut) {
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

__global__ __launch_bounds__(QTTT_COLD_BLOCK) void board_op_kernel(const uint8_t *in, uint8_t *out, int64_t n, u32 stamp) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    fill_line_lut<QTTT_COLD_BLOCK>(lut);
    int64_t i = (int64_t)blockIdx.x * QTTT_COLD_BLOCK + threadIdx.x;
    if (i >= n) return;
    uint8_t *o = out + i * QTTT_BOARD_RECORD_BYTES;
    board_op_record(in + i * QTTT_BOARD_RECORD_BYTES, o, lut);
    if (stamp) {
        __threadfence_system();
        *reinterpret_cast<volatile uint8_t *>(o + QTTT_BOARD_RECORD_BYTES - 1) = (uint8_t)stamp;
    }
}

// The BOUNDED MAILBOX behind qttt_board_op_host for single records (qttt_kernels.hip: board_mailbox): ONE wave stays
// resident on a private stream and serves one request slot in pinned host memory, so that a call costs a
// doorbell write and a poll instead of a kernel launch (tools/sync_latency: 4.4 - 4.8 us against 7.9 - 8.6 for the
// same echo through a launch).  It is never left resident: it returns BY ITSELF when no request has come for
// `idle_ticks` of the constant 100 MHz s_memrealtime counter, `resident_ticks` after it started whatever the traffic
// (checked between requests: a caller that keeps it busy cannot keep it resident — a device-wide synchronise in another
// thread waits for at most that long), when the host asks it to (MBOX_LEAVE in the request numbers:
// qttt_board_mailbox_retire), after `max_rings` requests, or after `max_polls` polls of one wait — every loop below has
// those exits, all wave-uniform.
//   slot_in  : FOUR 16-byte pieces, piece k = record bytes [12k, 12k + 12) + the 4-byte request number (the 41 input
//              bytes of a record fit the 48 data bytes).  The host writes the data of every piece, then the four numbers;
//              lane k reads piece k with ONE 16-byte load, and the request is taken only when ALL FOUR pieces carry the
//              wanted number.  Nothing here assumes that the four loads are one snapshot of the 64-byte line: a piece
//              whose number is new holds new data as soon as a single aligned 16-byte read is served from one moment
//              of the line (the host's stores to the data precede the store of the number in program order: x86 TSO),
//              and a piece whose number is still old keeps the wave polling.
//   slot_out : the answer in the record's own layout, bytes 60..63 = the request number, written after the rest is
//              visible system-wide (__threadfence_system between the two: device -> host posted writes)
//   exited   : receives `generation` when the kernel leaves (the host then knows it must launch again)
typedef u32 mbox_u32x4 __attribute__((ext_vector_type(4)));
constexpr u32 MBOX_LEAVE = 0xFFFFFFFFu;                     // never a request number (mbox_next skips it and 0)
__host__ __device__ __forceinline__ u32 mbox_next(u32 ring) {
    ++ring;
    return (ring == 0u || ring == MBOX_LEAVE) ? 1u : ring;
}
__global__ __launch_bounds__(64) void board_mailbox_kernel(const mbox_u32x4 *slot_in, mbox_u32x4 *slot_out, u32 *exited,
                                                           u32 generation, u32 first_ring, u32 max_rings, u64 idle_ticks,
                                                           u32 max_polls, u64 resident_ticks) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t rec_in[QTTT_BOARD_RECORD_BYTES], rec_out[QTTT_BOARD_RECORD_BYTES];
    fill_line_lut<64>(lut);
    const u32 lane = threadIdx.x;
    const u64 t_start = __builtin_amdgcn_s_memrealtime();
    if (lane < 16) reinterpret_cast<u32 *>(rec_in)[lane] = 0u;                  // bytes 48..63 are never sent
    u32 want = first_ring;
    for (u32 served = 0; served < max_rings; ++served) {
        const u64 t0 = __builtin_amdgcn_s_memrealtime();
        if (t0 - t_start > resident_ticks) break;                               // between two requests: none is taken
        bool rung = false;
        mbox_u32x4 v = {0u, 0u, 0u, 0u};
        for (u32 polls = 0; !rung; ++polls) {
            // system-coherent 16-byte loads (sc0 sc1: past the GPU's caches), piece k for lane k
            const mbox_u32x4 *src = slot_in + (lane & 3u);
            asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(src) : "memory");
            const u32 n0 = __builtin_amdgcn_readlane(v.w, 0), n1 = __builtin_amdgcn_readlane(v.w, 1);
            const u32 n2 = __builtin_amdgcn_readlane(v.w, 2), n3 = __builtin_amdgcn_readlane(v.w, 3);
            rung = n0 == want && n1 == want && n2 == want && n3 == want;        // wave-uniform
            const u64 now = __builtin_amdgcn_s_memrealtime();
            if (!rung && (n3 == MBOX_LEAVE || polls >= max_polls || now - t0 > idle_ticks || now - t_start > resident_ticks)) {
                if (lane == 0) __hip_atomic_store(exited, generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
        }
        if (lane < 4) {
            u32 *dst = reinterpret_cast<u32 *>(rec_in) + 3u * lane;
            dst[0] = v.x; dst[1] = v.y; dst[2] = v.z;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane == 0) {
            board_op_record(rec_in, rec_out, lut);
            rec_out[60] = rec_out[61] = rec_out[62] = rec_out[63] = 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (lane < 4) slot_out[lane] = reinterpret_cast<const mbox_u32x4 *>(rec_out)[lane];   // the record, number still 0
        __threadfence_system();
        if (lane == 0) __hip_atomic_store(reinterpret_cast<u32 *>(slot_out) + 15, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        want = mbox_next(want);
    }
    if (lane == 0) __hip_atomic_store(exited, generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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

// step_ctr (nullable): the step index is key_hi + *step_ctr and the launch key is made here (qttt_env.step_counter,
// see step_kernel's DEVSTEP); then key_lo is unused and `seed` is the environment's seed
__global__ __launch_bounds__(QTTT_BLOCK) void sample_actions_kernel(
    const u64 *pP, u32 key_lo, u32 key_hi, u64 board_offset, u32 auto_reset, uint16_t *actions,
    int64_t n, const u32 *step_ctr, u64 seed) {
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    if (step_ctr) {
        const u64 key = launch_key(seed, key_hi + *step_ctr);
        key_lo = (u32)key;
        key_hi = (u32)(key >> 32);
    }
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

// the device-side step counter of qttt_env.step_counter, advanced on the stream (one lane)
__global__ void counter_add_kernel(u32 *counter, u32 by) { *counter += by; }

}  // namespace

#endif  // QTTT_AUX_KERNELS_H
