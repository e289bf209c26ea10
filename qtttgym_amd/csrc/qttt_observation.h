// qttt_observation.h — Env._observation (env.py:68-85) from the packed words, through LDS tiles that are
// streamed out with coalesced stores.  Used by the fused step kernel and by observe_kernel.
#ifndef QTTT_OBSERVATION_H
#define QTTT_OBSERVATION_H
#include "qttt_state.h"

namespace {

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

// The same with 16-byte pieces (one global_store_dwordx4 per lane = 1 KB per wave instruction) when the
// wave's span starts on a 16-byte boundary on both sides; the tail (< 16 bytes x 64) goes as dwords + bytes.
__device__ inline void wave_copy_out16(uint8_t *gdst, const uint8_t *tile16, u32 begin, u32 end) {
    const u32 lane = threadIdx.x & 63u;
    u32x4 *gq = reinterpret_cast<u32x4 *>(gdst);
    const u32x4 *sq = reinterpret_cast<const u32x4 *>(tile16);
    const u32 q0 = begin >> 4, q1 = end >> 4;                       // begin is a multiple of 16
    for (u32 k = q0 + lane; k < q1; k += 64u) __builtin_nontemporal_store(sq[k], &gq[k]);
    const u32 t0 = q1 << 4;                                          // < 16 bytes left: at most 3 dwords + 3 bytes
    const u32 kd = (t0 >> 2) + lane;
    if (kd < (end >> 2)) reinterpret_cast<u32 *>(gdst)[kd] = reinterpret_cast<const u32 *>(tile16)[kd];
    const u32 kb = (end & ~3u) + lane;
    if (kb < end) gdst[kb] = tile16[kb];
}

// true iff every tile of this workgroup starts on a dword in global memory (block-uniform)
__device__ __forceinline__ u32 obs_all_phases(const ObsOut &o, int64_t first, u32 mask) {
    return (u32)((uintptr_t)(reinterpret_cast<const uint8_t *>(o.classical) + first * 9) | (uintptr_t)(o.q_p1 + first * 10) |
                 (uintptr_t)(o.q_p2 + first * 8) | (uintptr_t)(o.q_p1_len + first) | (uintptr_t)(o.q_p2_len + first) |
                 (uintptr_t)(o.turn + first)) & mask;
}
__device__ __forceinline__ bool obs_all_phase0(const ObsOut &o, int64_t first) { return obs_all_phases(o, first, 3u) == 0u; }

// VEC16: every tile starts on a 16-byte boundary in global memory and b0 is a multiple of 64 boards
template <u32 BOARDS, bool VEC16 = false>
__device__ inline void obs_wave_copy_out(uint8_t *lds, const ObsOut &o, int64_t first, u32 b0, u32 b1) {
#define QTTT_WCO(G, ROW) do { if (VEC16) wave_copy_out16(G, lds, b0 * (ROW), b1 * (ROW)); else wave_copy_out(G, lds, b0 * (ROW), b1 * (ROW)); } while (0)
    QTTT_WCO(reinterpret_cast<uint8_t *>(o.classical) + first * 9, 9u);
    lds += obs_tile_bytes(BOARDS, 9);
    QTTT_WCO(o.q_p1 + first * 10, 10u);
    lds += obs_tile_bytes(BOARDS, 10);
    QTTT_WCO(o.q_p2 + first * 8, 8u);
    lds += obs_tile_bytes(BOARDS, 8);
    QTTT_WCO(o.q_p1_len + first, 1u);
    lds += obs_tile_bytes(BOARDS, 1);
    QTTT_WCO(o.q_p2_len + first, 1u);
    lds += obs_tile_bytes(BOARDS, 1);
    QTTT_WCO(o.turn + first, 1u);
#undef QTTT_WCO
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
    // a 9-byte row: one 8-byte and one 1-byte LDS store (gfx950 does unaligned DS accesses)
    uint8_t *rc = T.cl + b * 9u;
    const u64 o07 = (u64)o03 | ((u64)o47 << 32);
    __builtin_memcpy(rc, &o07, 8);
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
    uint8_t *r1 = T.p1 + b * 10u;                         // a 10-byte row: one 8-byte and one 2-byte LDS store
    const u64 a01 = (u64)a0 | ((u64)a1 << 32);
    const uint16_t pad = (uint16_t)0xFFFFu;               // round 8 can never be un-collapsed
    __builtin_memcpy(r1, &a01, 8);
    __builtin_memcpy(r1 + 8, &pad, 2);
    *reinterpret_cast<u64 *>(T.p2 + b * 8u) = (u64)b0 | ((u64)b1 << 32);
    T.l1[b] = (uint8_t)n1;
    T.l2[b] = (uint8_t)n2;
    T.tn[b] = (uint8_t)((n + (fill ? 1u : 0u)) & 1u);                   // env.py:83
}

}  // namespace

#endif  // QTTT_OBSERVATION_H
