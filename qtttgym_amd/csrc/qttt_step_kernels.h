// qttt_step_kernels.h — the step kernels: lane per board, one launch per step (the metric's form), and the
// T-steps-in-registers forms (recorded actions / in-kernel policy).  The wave-per-board mapping study
// lives in tools/qttt_study.hip, outside the product library.
#ifndef QTTT_STEP_KERNELS_H
#define QTTT_STEP_KERNELS_H
#include "qttt_step_core.h"
#include "qttt_observation.h"

namespace {

// ====================================================================== the step kernels
// BPL boards per lane: lane j owns boards [j*BPL, (j+1)*BPL), so every plane is read and written
// with 16-byte vector accesses that are contiguous across the wave.  Addresses are a block-uniform
// 64-bit base (scalar unit) plus a 32-bit lane offset.
// SAMPLE: the action is not read but drawn in the kernel from the uniform-legal policy (and written to
// `actions` when that is not null) — qttt_sample_actions + qttt_step in one launch.
// OBS: Env.step returns the observation too (env.py:46,53): it is written from the registers the
// step already holds, through the LDS tiles above — qttt_step + qttt_observe in one launch.
// BLOCK, BPL: workgroup size and boards per lane, chosen by the host per launch from the batch size
// (auto_tuning() in qttt_kernels.hip holds the measured table: one board per lane in 256-thread
// workgroups below ~450 K boards, 1024-thread workgroups where they fill the chip exactly once,
// two boards per lane in 256-thread workgroups above 1 536 K boards).
// DEVSTEP: the step index lives on the DEVICE (qttt_env.step_counter, for launches captured in a hipGraph: the
// host cannot bake a step index into a node that is replayed).  key_hi then carries the node's offset, key_fold
// the fold of the ids' high word, and the launch key is made in the kernel (scalar unit, ~30 SALU instructions).
// A separate instantiation, so that the kernels of the ordinary path carry neither the two extra arguments nor
// the branch (measured: 2-3 % on a 1 M-board launch when they did).
//@isa lane
template <bool DEVSTEP> struct StepKeySource {};
template <> struct StepKeySource<true> {
    const u32 *ctr;
    u64 seed;
};

template <int BLOCK, int BPL, bool HAS_BITS, bool AUTO_RESET, bool SAMPLE = false, bool OBS = false, bool DEVSTEP = false>
__global__ __launch_bounds__(BLOCK) void step_kernel(
    u64 *__restrict__ pP, u64 *__restrict__ pQ, uint16_t *__restrict__ actions,
    const uint8_t *__restrict__ bits, u32 key_fold, u32 key_hi, u32 id_base,
    u32 *__restrict__ reward_bits, uint8_t *__restrict__ terminated, ObsOut obs, int64_t i_begin,
    u32 last_groups, StepKeySource<DEVSTEP> sk) {
    constexpr u32 TILE_BOARDS = BLOCK * BPL;
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    // the policy's tables: from 512 threads up qttt_state.h's PolicyRows (one word per empty-square mask + the
    // pre-scaled pair table, 692 words), in 256-thread workgroups — where that would be three table words per thread —
    // the two-level PolicyLut (146 words).  tools/stepbench STEPBENCH_RANDOM=1, interleaved, us per launch, round 3's
    // kernel / trusted step + two-level table / trusted step + rows: 1 M boards (2 x 1024) 8.06 / 7.79 / 7.66,
    // 768 K (2 x 512) 6.69 / 6.43 / 6.45, 2 M (2 x 256) 15.03 / 14.80 / 15.16, 256 K (1 x 256) 4.13 / 3.75 / 3.88.
    constexpr bool ROWS = SAMPLE && BLOCK >= 512;
    constexpr u32 PT_WORDS = ROWS ? POLICY_ROWS_WORDS : POLICY_LUT_WORDS;
    constexpr int PRW = SAMPLE ? (int)((PT_WORDS + BLOCK - 1) / BLOCK) : 1;
    __shared__ __attribute__((aligned(16))) u32 prows[SAMPLE ? PT_WORDS : 1];
    __shared__ __attribute__((aligned(16))) uint8_t otile[OBS ? obs_lds_bytes(TILE_BOARDS) : 16];
    __shared__ __attribute__((aligned(16))) u32 olut[OBS ? OBS_LUT_BYTES / 4 : 4];
#ifdef QTTT_DEBUG_STAMPS
    const u64 st0 = __builtin_amdgcn_s_memrealtime();
#endif
    typedef Vec<u64, BPL> V64;
    typedef Vec<u32, BPL> V32;
    typedef Vec<uint16_t, BPL> V16;
    typedef Vec<uint8_t, BPL> V8;
    if constexpr (DEVSTEP) {
        const u64 key = launch_key(sk.seed, key_hi + *sk.ctr);
        key_fold ^= (u32)key;
        key_hi = (u32)(key >> 32);
    }
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
    u32 olw = 0;
    if (OBS && threadIdx.x < OBS_LUT_BYTES / 4) olw = (&g_obs_lut.sel[0][0])[threadIdx.x];
    u32 prw[PRW];
    if (SAMPLE) {
#pragma unroll
        for (int k = 0; k < PRW; ++k) {
            const u32 w = threadIdx.x + (u32)k * BLOCK;
            const u32 *src = ROWS ? reinterpret_cast<const u32 *>(&g_policy_rows) : reinterpret_cast<const u32 *>(&g_policy_lut);
            prw[k] = w < PT_WORDS ? src[w] : 0u;
        }
    }
    V64 p = load_stream(&reinterpret_cast<const V64 *>(pP + ib)[g]);
    V64 q = load_stream(&reinterpret_cast<const V64 *>(pQ + ib)[g]);
    V16 act;
    V8 bt;
    if (!SAMPLE) act = load_stream(&reinterpret_cast<const V16 *>(actions + ib)[g]);
    if (HAS_BITS) bt = load_stream(&reinterpret_cast<const V8 *>(bits + ib)[g]);
    fill_line_lut_nosync<BLOCK>(lut);
    if (OBS && threadIdx.x < OBS_LUT_BYTES / 4) olut[threadIdx.x] = olw;
    if (SAMPLE) {
#pragma unroll
        for (int k = 0; k < PRW; ++k) {
            const u32 w = threadIdx.x + (u32)k * BLOCK;
            if (w < PT_WORDS) prows[w] = prw[k];
        }
    }
    ObsTiles T;
    if (OBS) T = obs_tiles<TILE_BOARDS>(otile, obs, ib);
//@isa hash
    // The counter hash (collapse bit; with SAMPLE the policy's word too) depends on the board id and the launch key
    // only: it is computed HERE, while the state loads are in flight and the SIMD has nothing else to issue, and
    // pinned in registers (the empty asm) so that the compiler does not sink it back behind the loads' wait — 8
    // (SAMPLE: 16) VALU instructions per board off the critical path between "data landed" and "stores issued".
    const u32 id0 = id_base + ((u32)jb + g) * BPL;                      // low 32 bits of the global board id
    u32 hbit[BPL], hpol[BPL];
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        hbit[k] = hpol[k] = 0u;
        if (SAMPLE) {
            const u32 h1 = lowbias32((id0 + (u32)k) ^ key_fold);
            hpol[k] = lowbias32(h1 ^ key_hi);
            hbit[k] = h1 >> 31;
#ifndef QTTT_NO_HASH_HOIST                  // (A/B builds only: without the pin the compiler sinks the hash behind the loads' wait)
            asm volatile("" : "+v"(hbit[k]), "+v"(hpol[k]));
#endif
        } else if (!HAS_BITS) {
            hbit[k] = collapse_bit_of((id0 + (u32)k) ^ key_fold);
#ifndef QTTT_NO_HASH_HOIST
            asm volatile("" : "+v"(hbit[k]));
#endif
        }
    }
//@isa pack
    __syncthreads();
#ifdef QTTT_DEBUG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const u64 st1 = __builtin_amdgcn_s_memrealtime();
#endif
    if (active) {
        V32 rw;
        V8 tm;
#pragma unroll
        for (int k = 0; k < BPL; ++k) {
            u32 P0 = (u32)p.v[k], P1 = (u32)(p.v[k] >> 32);
            u32 Q0 = (u32)q.v[k], Q1 = (u32)(q.v[k] >> 32);
            u32 bit, av, win;
            if (SAMPLE) {
                const u32 h2 = hpol[k];
                bit = hbit[k];
                if (AUTO_RESET) {
                    // a finished board restarts first (empty = all zero); a board that is not done has >= 2 empty
                    // squares (8 classical squares set the done bit), so the policy always finds a legal pair,
                    // lo < hi by construction: the step runs TRUSTED (no validation, no sorting), as in
                    // step_random_fused_kernel
                    step_auto_reset(P0, P1, Q0, Q1);
                    const u32 empty = ~(P1 >> P1_CL_SHIFT) & 0x1FFu;
                    av = ROWS ? policy_action_rows(prows, empty, h2) : policy_action(reinterpret_cast<const uint8_t *>(prows), empty, h2);
                    win = step_core<false, true>(P0, P1, Q0, Q1, av, bit, lut);
                } else {
                    // post-terminal legal moves are accepted; no legal pair -> (0,0), a noop
                    const u32 empty = ~(P1 >> P1_CL_SHIFT) & 0x1FFu;
                    av = !(empty & (empty - 1u)) ? 0u : ROWS ? policy_action_rows(prows, empty, h2)
                                                             : policy_action(reinterpret_cast<const uint8_t *>(prows), empty, h2);
                    win = step_core<false>(P0, P1, Q0, Q1, av, bit, lut);
                }
                act.v[k] = (uint16_t)av;
            } else {
                av = act.v[k];
                if (HAS_BITS) bit = bt.v[k] & 1u;
                else bit = hbit[k];
                win = step_core<AUTO_RESET>(P0, P1, Q0, Q1, av, bit, lut);
            }
            p.v[k] = (u64)P0 | ((u64)P1 << 32);
            q.v[k] = (u64)Q0 | ((u64)Q1 << 32);
            rw.v[k] = 0x80000000u | (win << 23);                         // env.py:49: -1.0f / -0.0f
            tm.v[k] = (uint8_t)(P1 >> 31);
            if (OBS) obs_board(P0, P1, Q0, T, g * BPL + (u32)k, olut);
        }
        store_stream_sbase(pP + ib, g * (u32)sizeof(V64), p);          // block-uniform base + 32-bit lane offset, like the loads
        store_stream_sbase(pQ + ib, g * (u32)sizeof(V64), q);
        if (SAMPLE && actions) store_stream(&reinterpret_cast<V16 *>(actions + ib)[g], act);
        store_stream(&reinterpret_cast<V32 *>(reward_bits + ib)[g], rw);
        store_stream(&reinterpret_cast<V8 *>(terminated + ib)[g], tm);
    }
    if (OBS) {
        // a wave's boards [64w * BPL, 64(w+1) * BPL) start on a multiple of 4 bytes in every tile
        const u32 ph = obs_all_phases(obs, ib, 15u);                 // block-uniform
        if ((ph & 3u) == 0u) {
            const u32 w0 = (threadIdx.x & ~63u) * BPL;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // rows written by other lanes of this wave
            __builtin_amdgcn_wave_barrier();
            if (w0 < ng * BPL) {
                if (ph == 0u) obs_wave_copy_out<TILE_BOARDS, true>(otile, obs, ib, w0, min(w0 + 64u * BPL, ng * BPL));
                else obs_wave_copy_out<TILE_BOARDS, false>(otile, obs, ib, w0, min(w0 + 64u * BPL, ng * BPL));
            }
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

//@isa lane
// T consecutive steps in ONE launch (qttt_step_many with QTTT_FLAG_FUSED): the boards stay in
// registers, only the per-step streams move (2 B action in, 5 B reward/terminated out per step), so
// the loop is VALU-bound and pays one launch instead of T.  Same results as T launches of
// step_kernel; meant for replay / evaluation where the actions are known up front (a policy that
// looks at the state between steps needs the one-launch-per-step form).
// The launch keys of the plies come from the host as a kernel argument (FusedKeys, at most FUSED_MAX_PLIES per launch; the
// library splits longer runs): splitmix64 per ply on the scalar unit was 25 of the loop's 86 scalar instructions — measured:
// 3.57 -> 3.44 us per ply at 1 M boards, 1.19 -> 1.10 at 262 144 (profiles/r05/fused_kernarg_keys_check.txt).
constexpr int FUSED_MAX_PLIES = 64;
struct FusedKeys { u64 k[FUSED_MAX_PLIES]; };
template <bool HAS_BITS, bool AUTO_RESET>
__global__ __launch_bounds__(QTTT_BLOCK) void step_fused_kernel(
    u64 *__restrict__ pP, u64 *__restrict__ pQ, const uint16_t *__restrict__ actions,
    const uint8_t *__restrict__ bits, FusedKeys keys, u32 id_hi_fold, u32 id_base,
    u32 *__restrict__ reward_bits, uint8_t *__restrict__ terminated, int64_t out_stride, int64_t n,
    int32_t n_steps) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    const int64_t ib = (int64_t)blockIdx.x * QTTT_BLOCK;            // first board of the workgroup (block-uniform)
    const int64_t i = ib + threadIdx.x;
    const int64_t il = i < n ? i : 0;                               // idle lanes re-read board 0, store nothing
    const u64 P = load_stream(&pP[il]), Q = load_stream(&pQ[il]);   // requested before the table fills
    fill_line_lut<QTTT_BLOCK>(lut);
    if (i >= n) return;
    u32 P0 = (u32)P, P1 = (u32)(P >> 32), Q0 = (u32)Q, Q1 = (u32)(Q >> 32);
    const u32 id = (id_base + (u32)i) ^ id_hi_fold;
    // per-ply streams: a block-uniform 64-bit base (scalar unit, advanced by the stride every ply) plus the lane's offset
    const uint16_t *a_blk = actions + ib;
    const uint8_t *b_blk = HAS_BITS ? bits + ib : nullptr;
    u32 *r_blk = reward_bits + ib;
    uint8_t *t_blk = terminated + ib;
    const u32 lane = threadIdx.x;
    u32 win = 0;
    u64 key = keys.k[0];
    u32 act = load_stream(&a_blk[lane]), bit_in = HAS_BITS ? (u32)load_stream(&b_blk[lane]) : 0u;
    for (int32_t t = 0; t < n_steps; ++t) {
        const u64 key_next = keys.k[(t + 1) & (FUSED_MAX_PLIES - 1)];   // one scalar load, requested a ply ahead
        // the next ply's action (and bit) are requested before this ply's step: their latency hides behind it
        const bool more = t + 1 < n_steps;
        a_blk += more ? n : 0;
        if (HAS_BITS) b_blk += more ? n : 0;
        const u32 act_next = load_stream(&a_blk[lane]), bit_next = HAS_BITS ? (u32)load_stream(&b_blk[lane]) : 0u;
        const u32 bit = HAS_BITS ? (bit_in & 1u) : collapse_bit_of(id ^ (u32)key);
        win = step_core<AUTO_RESET>(P0, P1, Q0, Q1, act, bit, lut);
        if (out_stride != 0 || t == n_steps - 1) {
            // (the compiler's own stores here: it counts them when it waits for the action requested a ply ahead)
            store_stream(&r_blk[lane], 0x80000000u | (win << 23));
            store_stream(&t_blk[lane], (uint8_t)(P1 >> 31));
        }
        r_blk += out_stride;
        t_blk += out_stride;
        key = key_next;
        act = act_next;
        bit_in = bit_next;
    }
    store_stream(&pP[i], (u64)P0 | ((u64)P1 << 32));
    store_stream(&pQ[i], (u64)Q0 | ((u64)Q1 << 32));
}

// The same with the uniform-legal policy IN the kernel (qttt_step_random_many): T consecutive
// qttt_step_random steps — policy -> collapse bit -> step, ply t keyed by the counter hash of
// (seed, global board id, step_idx0 + t) exactly as the launch-by-launch form — with the boards in
// registers.  This is MCTS._simulate's loop (mcts.py:185-198: policy -> step until terminal) turned
// into the env's throughput mode: with AUTO_RESET a finished board restarts on its next ply, so every
// ply of every lane is a live transition.  Per ply it moves only what the caller keeps (nothing, or
// action 2 B + reward 4 B + terminated 1 B per board), so it is VALU-bound and pays one launch per T
// steps: this is what takes batches of <= 512 K boards (BASELINE configs 2-4: one partial occupancy
// round per launch) out of the launch-bound regime.
//   nth9 / policy_action_nth9: qttt_state.h (the full 4.5 KB "r-th empty square" table, computed by the workgroup).
// With AUTO_RESET the policy always has a legal pair (a board that is not done has >= 2 empty squares:
// 8 classical squares set the done bit), so the step runs TRUSTED (no validation, no sorting).
// KEEP: every ply's action, reward and terminated are kept (out_stride != 0, no null output) — the loop then carries no
// test of what to store.
template <int BLOCK, bool AUTO_RESET, bool RETURNS = false, bool KEEP = false>
__global__ __launch_bounds__(BLOCK) void step_random_fused_kernel(
    u64 *__restrict__ pP, u64 *__restrict__ pQ, FusedKeys keys, u64 board_offset,
    uint16_t *__restrict__ actions_out, u32 *__restrict__ reward_bits, uint8_t *__restrict__ terminated,
    int64_t out_stride, int64_t n, int32_t n_steps, float *__restrict__ returns) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    __shared__ uint8_t nth9[NTH9_BYTES];
    const int64_t ib = (int64_t)blockIdx.x * BLOCK;                 // first board of the workgroup (block-uniform)
    const int64_t i = ib + threadIdx.x;
    const bool active = i < n;
    const int64_t il = active ? i : 0;                              // idle lanes re-read board 0, store nothing
    const u64 P = load_stream(&pP[il]), Q = load_stream(&pQ[il]);   // requested before the table fills
    fill_policy_lut<BLOCK>(plut);
    fill_nth9<BLOCK>(nth9);
    fill_line_lut<BLOCK>(lut);                                      // ends with the workgroup barrier
    if (!active) return;
    u32 P0 = (u32)P, P1 = (u32)(P >> 32), Q0 = (u32)Q, Q1 = (u32)(Q >> 32);
    const u32 id = fold_id(board_offset + (u64)i);
    // per-ply outputs: a block-uniform 64-bit base (scalar unit, advanced by the stride every ply) plus the
    // lane's 32-bit offset — no 64-bit vector address arithmetic in the loop
    uint16_t *a_blk = actions_out ? actions_out + ib : nullptr;
    u32 *r_blk = reward_bits ? reward_bits + ib : nullptr;
    uint8_t *t_blk = terminated ? terminated + ib : nullptr;
    const int64_t a_step = actions_out ? out_stride : 0, r_step = reward_bits ? out_stride : 0;   // a null output stays null
    const u32 lane = threadIdx.x;
    u32 lines = 0;                                                  // plies whose reward was -1.0 (env.py:49): -(the return)
    u64 key = keys.k[0];
    // The state is consumed HERE, in front of the loop: the per-ply stores below are written out (store_stream_sbase_word),
    // so the compiler sees no memory operation in the loop and would leave its wait for the state loads inside it — where
    // a vmcnt wait also waits for the previous ply's stores (+9 % per ply with one wave per SIMD).
    asm volatile("" : "+v"(P0), "+v"(P1), "+v"(Q0), "+v"(Q1));
    for (int32_t t = 0; t < n_steps; ++t) {
        const u64 key_next = keys.k[(t + 1) & (FUSED_MAX_PLIES - 1)];   // one scalar load, requested a ply ahead
        const u32 h1 = lowbias32(id ^ (u32)key);
        const u32 h2 = lowbias32(h1 ^ (u32)(key >> 32));
        u32 act, win;
        if (AUTO_RESET) {
            step_auto_reset(P0, P1, Q0, Q1);                        // a finished board restarts: empty = all zero
            const u32 empty = ~(P1 >> P1_CL_SHIFT) & 0x1FFu;        // >= 2 squares: the board is not done
            act = policy_action_nth9(plut, nth9, empty, h2);
            win = step_core<false, true>(P0, P1, Q0, Q1, act, h1 >> 31, lut);
        } else {
            // post-terminal legal moves are accepted; no legal pair -> (0,0), a noop
            const u32 empty = ~(P1 >> P1_CL_SHIFT) & 0x1FFu;
            act = (empty & (empty - 1u)) ? policy_action_nth9(plut, nth9, empty, h2) : 0u;
            win = step_core<false, false>(P0, P1, Q0, Q1, act, h1 >> 31, lut);
        }
        if (KEEP || out_stride != 0 || t == n_steps - 1) {
            if (KEEP || a_blk) store_stream_sbase_word<2>(a_blk, lane * 2u, act);
            if (KEEP || r_blk) {
                store_stream_sbase_word<4>(r_blk, lane * 4u, 0x80000000u | (win << 23));   // env.py:49: -1.0f / -0.0f
                store_stream_sbase_word<1>(t_blk, lane, P1 >> 31);                          // env.py:51
            }
        }
        if (RETURNS) lines += win & 1u;                             // (its own instantiation: the ordinary ply carries nothing extra)
        a_blk += a_step;
        r_blk += r_step;
        t_blk += r_step;
        key = key_next;
    }
    if (RETURNS) returns[i] -= (float)lines;                        // the sum of the rewards of these plies, accumulated
    store_stream(&pP[i], (u64)P0 | ((u64)P1 << 32));
    store_stream(&pQ[i], (u64)Q0 | ((u64)Q1 << 32));
}

}  // namespace

#endif  // QTTT_STEP_KERNELS_H
