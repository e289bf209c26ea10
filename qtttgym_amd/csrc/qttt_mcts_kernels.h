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

// Two boards per lane (one 16-byte load per plane, 16-byte stores of the two legal masks and the two
// keys): the CPython tuple hash is a dependent chain per board, two chains in one lane interleave
// (fast_py_hash_pair).  The last board of an odd batch is handled alone.
// PYKEY: the CPython-exact key (GameState.__hash__, for host-side dicts built by reference code) is its own
// instantiation: without it the kernel carries neither the 5.9 KB table nor the chain.  `skey` is the native
// 64-bit position key (state_key(): a mix of the packed words, for device-side tables).
template <int BLOCK, bool PYKEY>
__global__ __launch_bounds__(BLOCK) void node_info_kernel(
    const u64 *pP, const u64 *pQ, int8_t *winner, uint8_t *terminal, u64 *legal,
    int64_t *key, u64 *skey, int64_t n) {
    __shared__ u64 htbl[PYKEY ? PYHASH_LUT_WORDS : 1];
    __shared__ u64 ltbl[512];
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;   // boards 2j, 2j + 1
    const int64_t i0 = 2 * j;
    typedef Vec<u64, 2> V64;
    V64 p, q;
    p.v[0] = p.v[1] = q.v[0] = q.v[1] = 0ull;
    const bool want_q = PYKEY || skey != nullptr;                       // plane Q: only the keys read it (the x nibbles)
    LegalLutWords<BLOCK> lw;
    if (legal) lw.request();                                            // in front of the state loads, stored behind them
    if (i0 + 1 < n) {
        p = load_stream(&reinterpret_cast<const V64 *>(pP)[j]);
        if (want_q) q = load_stream(&reinterpret_cast<const V64 *>(pQ)[j]);
    } else if (i0 < n) {
        p.v[0] = pP[i0];
        if (want_q) q.v[0] = pQ[i0];
    }
    if (PYKEY) fill_pyhash_lut<BLOCK>(htbl);
    if (legal) lw.store(ltbl);
    if (winner || terminal || legal || PYKEY) fill_line_lut<BLOCK>(lut);   // computed; ends with the workgroup barrier
    if (i0 >= n) return;
    const bool two = i0 + 1 < n;
    typedef Vec<u64, 2> V64x;
    if (skey) {                                            // needs no table: out first
        const u64 ka = state_key(p.v[0], (u32)q.v[0]), kb = state_key(p.v[1], (u32)q.v[1]);
        if (two && (reinterpret_cast<uintptr_t>(skey) & 15u) == 0u) {
            V64x k2;
            k2.v[0] = ka; k2.v[1] = kb;
            store_stream(&reinterpret_cast<V64x *>(skey)[j], k2);
        } else {
            skey[i0] = ka;
            if (two) skey[i0 + 1] = kb;
        }
    }
    if (!(winner || terminal || legal || PYKEY)) return;
    const Lite sa = lite_unpack(p.v[0]), sb = lite_unpack(p.v[1]);
    if (winner || terminal) {
        int wa, ta, wb, tb;
        lite_update_winner(sa, lut, wa, ta);
        lite_update_winner(sb, lut, wb, tb);
        if (two && ((reinterpret_cast<uintptr_t>(winner) | reinterpret_cast<uintptr_t>(terminal)) & 1u) == 0u) {
            if (winner) reinterpret_cast<uint16_t *>(winner)[j] = (uint16_t)((u32)(wa & 0xFF) | ((u32)(wb & 0xFF) << 8));
            if (terminal) reinterpret_cast<uint16_t *>(terminal)[j] = (uint16_t)((u32)ta | ((u32)tb << 8));
        } else {
            if (winner) { winner[i0] = (int8_t)wa; if (two) winner[i0 + 1] = (int8_t)wb; }
            if (terminal) { terminal[i0] = (uint8_t)ta; if (two) terminal[i0 + 1] = (uint8_t)tb; }
        }
    }
    if (legal) {
        const u64 la = ltbl[sa.cl], lb = ltbl[sb.cl];
        if (two && (reinterpret_cast<uintptr_t>(legal) & 15u) == 0u) {
            V64x l2;
            l2.v[0] = la; l2.v[1] = lb;
            store_stream(&reinterpret_cast<V64x *>(legal)[j], l2);
        } else {
            legal[i0] = la;
            if (two) legal[i0 + 1] = lb;
        }
    }
    if (PYKEY) {                                           // the expensive part: skipped when the caller keeps no host table
        int64_t ka, kb;
        fast_py_hash_pair(sa, (u32)(p.v[0] >> 32), (u32)q.v[0], sb, (u32)(p.v[1] >> 32), (u32)q.v[1], htbl, ka, kb);
        if (two && (reinterpret_cast<uintptr_t>(key) & 15u) == 0u) {
            V64x k2;
            k2.v[0] = (u64)ka; k2.v[1] = (u64)kb;
            store_stream(&reinterpret_cast<V64x *>(key)[j], k2);
        } else {
            key[i0] = ka;
            if (two) key[i0 + 1] = kb;
        }
    }
}

// MCTS._step (mcts.py:233-267): both values of the collapse bit computed directly instead of
// re-sampling make_move until the other branch appears.  The two children's bookkeeping (winner, legal
// mask, keys) is computed as a pair — with PYKEY two independent hash chains in one lane, as in node_info — and
// masked by n_children afterwards (child 1 is a valid state even when there is no collapse: it equals
// child 0).  Every per-child output is nullable; PYKEY (the CPython-exact key) is its own instantiation.
struct ExpandOut {
    uint8_t *n_children;        // [n]
    int8_t *winner;             // [n,2]
    uint8_t *terminal;          // [n,2]
    u64 *legal;                 // [n,2]
    int64_t *key;               // [n,2] CPython tuple hash (PYKEY)
    u64 *skey;                  // [n,2] native position key
};
// the bookkeeping of the two children of one pair, from their packed words (shared by expand_kernel and
// expand_rollout_kernel's writer lanes)
template <bool PYKEY>
__device__ __forceinline__ void expand_bookkeeping(u32 kids, const u64 kidP[2], const u64 kidQ[2], u32 xo0, u32 xo1,
                                                   const ExpandOut &o, int64_t i, const uint8_t *lut, const u64 *htbl, const u64 *ltbl) {
    typedef Vec<u64, 2> V64;
    const bool h0 = kids >= 1u, h1 = kids >= 2u;
    if (o.n_children) o.n_children[i] = (uint8_t)kids;
    if (o.skey) {
        V64 k2;
        k2.v[0] = h0 ? state_key(kidP[0], (u32)kidQ[0]) : 0ull;
        k2.v[1] = h1 ? state_key(kidP[1], (u32)kidQ[1]) : 0ull;
        store_stream(&reinterpret_cast<V64 *>(o.skey)[i], k2);
    }
    if (o.winner || o.terminal) {                    // from the line test the step just made (update_winner, mcts.py:52-65)
        int w0, t0, w1, t1;
        update_winner_from_step(kidP[0], xo0, lut, w0, t0);
        update_winner_from_step(kidP[1], xo1, lut, w1, t1);
        const u32 wv = (u32)((h0 ? w0 : -1) & 0xFF) | ((u32)((h1 ? w1 : -1) & 0xFF) << 8);
        const u32 tv = (h0 ? (u32)t0 : 0u) | ((h1 ? (u32)t1 : 0u) << 8);
        if (o.winner) reinterpret_cast<uint16_t *>(o.winner)[i] = (uint16_t)wv;      // [n,2] rows: 2-byte / 16-byte aligned by the host check
        if (o.terminal) reinterpret_cast<uint16_t *>(o.terminal)[i] = (uint16_t)tv;
    }
    if (o.legal) {                                   // GameState.actions (mcts.py:20-27): a function of the classical squares,
        V64 l2;                                      // the implicit autofill (eight classical squares) counted as the ninth
        const u32 c0 = (u32)(kidP[0] >> (32u + P1_CL_SHIFT)) & 0x1FFu, c1 = (u32)(kidP[1] >> (32u + P1_CL_SHIFT)) & 0x1FFu;
        l2.v[0] = h0 ? ltbl[__builtin_popcount(c0) == 8 ? 0x1FFu : c0] : 0ull;
        l2.v[1] = h1 ? ltbl[__builtin_popcount(c1) == 8 ? 0x1FFu : c1] : 0ull;
        store_stream(&reinterpret_cast<V64 *>(o.legal)[i], l2);
    }
    if (PYKEY) {
        const Lite s0 = lite_unpack(kidP[0]), s1 = lite_unpack(kidP[1]);
        int64_t k0, k1;
        fast_py_hash_pair(s0, (u32)(kidP[0] >> 32), (u32)kidQ[0], s1, (u32)(kidP[1] >> 32), (u32)kidQ[1], htbl, k0, k1);
        V64 k2;
        k2.v[0] = h0 ? (u64)k0 : 0ull;     k2.v[1] = h1 ? (u64)k1 : 0ull;
        store_stream(&reinterpret_cast<V64 *>(o.key)[i], k2);
    }
}

// One (state, action) pair per lane.  Measured and not adopted (tools/rowbench, 1 M pairs, us per launch): two pairs per
// lane with 16-byte plane accesses — 31 - 33 against 19 (the two children of two pairs do not fit the registers of a
// full-occupancy wave); 1024-thread workgroups 20.3 against 19.1 with 256.
template <int BLOCK, bool PYKEY>
__global__ __launch_bounds__(BLOCK) void expand_kernel(
    const u64 *pP, const u64 *pQ, const uint8_t *action36,
    u64 *c0P, u64 *c0Q, u64 *c1P, u64 *c1Q, ExpandOut out, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ u64 htbl[PYKEY ? PYHASH_LUT_WORDS : 1];
    __shared__ u64 ltbl[512];
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    LegalLutWords<BLOCK> lw;
    if (out.legal) lw.request();                     // in front of the state loads, stored behind them
    const u64 P = i < n ? load_stream(&pP[i]) : 0ull, Q = i < n ? load_stream(&pQ[i]) : 0ull;
    const u32 a = i < n ? (u32)action36[i] : 0u;
    if (PYKEY) fill_pyhash_lut<BLOCK>(htbl);
    if (out.legal) lw.store(ltbl);
    fill_line_lut<BLOCK>(lut);                       // computed; ends with the workgroup barrier
    if (i >= n) return;
    const u32 pr = a < 36u ? (u32)g_pair_lut.b[a] : 0u;            // (0,0) = a noop for bad indices
    const u32 act = (pr & 0xFu) | ((pr >> 4) << 8);
    // both children in one pass: validity, components, append, qstructs and classical update are shared, only the
    // path reversal and the line test run per child (step_core_both)
    u32 Q0 = (u32)Q, Q1 = (u32)(Q >> 32), P0a, P1a, P0b, P1b, xo0, xo1;
    const u32 kids = step_core_both((u32)P, (u32)(P >> 32), Q0, Q1, act, lut, P0a, P1a, P0b, P1b, xo0, xo1);   // mcts.py:245
    u64 kidP[2], kidQ[2];
    kidP[0] = (u64)P0a | ((u64)P1a << 32);
    kidP[1] = (u64)P0b | ((u64)P1b << 32);
    kidQ[0] = kidQ[1] = (u64)Q0 | ((u64)Q1 << 32);
    store_stream(&c0P[i], kidP[0]); store_stream(&c0Q[i], kidQ[0]);
    store_stream(&c1P[i], kidP[1]); store_stream(&c1Q[i], kidQ[1]);
    expand_bookkeeping<PYKEY>(kids, kidP, kidQ, xo0, xo1, out, i, lut, htbl, ltbl);
}

// MCTS._simulate (mcts.py:185-198) under the uniform priors of mcts.py:287-292: play uniform-legal
// random moves to the end with the board in registers.  Ply p uses the counter hash of
// (seed, board id, step_idx0 + p) exactly like qttt_sample_actions + qttt_step would.
// The launch keys of a playout's plies come from a table in LDS (splitmix64 of (seed, step index): 25 scalar instructions
// per ply when the step index is wave-uniform, and ~30 VECTOR instructions per ply when it differs per lane — the
// simulations of rollout_many / expand_rollout use step_idx0 + slot * QTTT_SIM_STRIDE + ply).  A table row = the nine keys
// of one slot; PLAYOUT_KEY_SLOTS rows fit one key per thread of a 256-thread workgroup.  More slots than that: the keys are
// computed in the loop (TABLE = false).
constexpr u32 PLAYOUT_PLIES = 9u, PLAYOUT_KEY_SLOTS = 28u;
template <int BLOCK>
__device__ __forceinline__ void fill_playout_keys_nosync(u64 *keytab, u64 seed, u32 step_idx0, u32 n_slots) {
    for (u32 k = threadIdx.x; k < n_slots * PLAYOUT_PLIES; k += BLOCK) {
        const u32 slot = k / PLAYOUT_PLIES, ply = k - slot * PLAYOUT_PLIES;
        keytab[k] = launch_key(seed, step_idx0 + slot * QTTT_SIM_STRIDE + ply);
    }
}
// one playout of the board in (P0, P1, Q0, Q1) to the end; returns the number of plies played.  TABLE: `keys` = the nine
// keys of this lane's slot (LDS); else they are launch_key(seed, step_idx0 + ply).
template <bool TABLE>
__device__ __forceinline__ u32 playout(u32 &P0, u32 &P1, u32 &Q0, u32 &Q1, u32 id, u64 seed, u32 step_idx0, const u64 *keys,
                                       const uint8_t *lut, const uint8_t *plut, const uint8_t *nth9) {
    u32 played = 0;
    u64 key_tab = TABLE ? keys[0] : 0ull;
    for (u32 p = 0; p < PLAYOUT_PLIES; ++p) {
        const u32 empty = ~(P1 >> P1_CL_SHIFT) & 0x1FFu;
        if ((P1 >> 31) || (empty & (empty - 1u)) == 0u) break;   // terminal (mcts.py:188) / nothing legal
        const u64 key = TABLE ? key_tab : launch_key(seed, step_idx0 + p);
        if (TABLE) key_tab = keys[p + 1u < PLAYOUT_PLIES ? p + 1u : p];      // the next ply's, requested a ply ahead
        const u32 h1 = lowbias32(id ^ (u32)key);
        const u32 h2 = lowbias32(h1 ^ (u32)(key >> 32));
        const u32 act = policy_action_nth9(plut, nth9, empty, h2);   // the k-th legal pair, squares a < b
        step_core<false, true>(P0, P1, Q0, Q1, act, h1 >> 31, lut);   // legal and sorted
        played += 1u;
    }
    return played;
}

__global__ __launch_bounds__(QTTT_BLOCK) void rollout_kernel(
    const u64 *pP, const u64 *pQ, u64 seed, u32 step_idx0, u64 board_offset,
    int8_t *result, uint8_t *plies, u64 *fP, u64 *fQ, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    __shared__ uint8_t nth9[NTH9_BYTES];
    __shared__ u64 keytab[PLAYOUT_PLIES];
    int64_t i = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    const u64 P = i < n ? pP[i] : 0ull, Q = i < n ? pQ[i] : 0ull;  // requested before the table fills
    fill_policy_lut<QTTT_BLOCK>(plut);
    fill_nth9<QTTT_BLOCK>(nth9);
    fill_playout_keys_nosync<QTTT_BLOCK>(keytab, seed, step_idx0, 1u);
    fill_line_lut<QTTT_BLOCK>(lut);                       // ends with the workgroup barrier
    if (i >= n) return;
    u32 P0 = (u32)P, P1 = (u32)(P >> 32), Q0 = (u32)Q, Q1 = (u32)(Q >> 32);
    const u32 played = playout<true>(P0, P1, Q0, Q1, fold_id(board_offset + (u64)i), seed, step_idx0, keytab, lut, plut, nth9);
    const u64 oP = (u64)P0 | ((u64)P1 << 32), oQ = (u64)Q0 | ((u64)Q1 << 32);
    int w, t;
    lite_update_winner(lite_unpack(oP), lut, w, t);
    result[i] = (int8_t)(w < 0 ? 0 : (w ? 1 : -1));       // MCTS._reward, mcts.py:200-209
    plies[i] = (uint8_t)played;
    if (fP) { fP[i] = oP; fQ[i] = oQ; }
}

// MCTS._rollout's `for _ in range(self.num_simulations): r = self._simulate(leaf)` (mcts.py:170-176) in ONE launch:
// lane j plays simulation j % n_sims of board j / n_sims, with the step indices of qttt_rollout(step_idx0 +
// sim * QTTT_SIM_STRIDE) — so a batch too small to fill the chip (65 536 leaves) still does, n_sims times over.
// Outputs [n, n_sims], contiguous in lane order.
__global__ __launch_bounds__(QTTT_BLOCK) void rollout_many_kernel(
    const u64 *pP, const u64 *pQ, u64 seed, u32 step_idx0, u64 board_offset, u32 n_sims,
    int8_t *result, uint8_t *plies, int64_t n_lanes) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    __shared__ uint8_t nth9[NTH9_BYTES];
    const int64_t j = (int64_t)blockIdx.x * QTTT_BLOCK + threadIdx.x;
    const int64_t jl = j < n_lanes ? j : 0;
    const int64_t i = jl / n_sims;                        // board
    const u32 sim = (u32)(jl - i * n_sims);
    const u64 P = pP[i], Q = pQ[i];                       // n_sims neighbouring lanes read the same 16 bytes
    __shared__ u64 keytab[PLAYOUT_KEY_SLOTS * PLAYOUT_PLIES];
    const bool table = n_sims <= PLAYOUT_KEY_SLOTS;       // wave-uniform
    fill_policy_lut<QTTT_BLOCK>(plut);
    fill_nth9<QTTT_BLOCK>(nth9);
    if (table) fill_playout_keys_nosync<QTTT_BLOCK>(keytab, seed, step_idx0, n_sims);
    fill_line_lut<QTTT_BLOCK>(lut);                       // ends with the workgroup barrier
    if (j >= n_lanes) return;
    u32 P0 = (u32)P, P1 = (u32)(P >> 32), Q0 = (u32)Q, Q1 = (u32)(Q >> 32);
    const u32 id = fold_id(board_offset + (u64)i);
    const u32 played = table ? playout<true>(P0, P1, Q0, Q1, id, seed, 0u, keytab + sim * PLAYOUT_PLIES, lut, plut, nth9)
                             : playout<false>(P0, P1, Q0, Q1, id, seed, step_idx0 + sim * QTTT_SIM_STRIDE, nullptr, lut, plut, nth9);
    int w, t;
    lite_update_winner(lite_unpack((u64)P0 | ((u64)P1 << 32)), lut, w, t);
    result[j] = (int8_t)(w < 0 ? 0 : (w ? 1 : -1));
    if (plies) plies[j] = (uint8_t)played;
}

// One MCTS._rollout below the selected node (mcts.py:166-176 with _select's expansion, :210-221,233-267) in ONE launch:
// expand (state, action) into its one or two children, the children's bookkeeping, and n_sims playouts FROM EACH
// CHILD — qttt_expand followed by qttt_rollout_many on child 0 (step indices from step_idx0) and on child 1 (from
// step_idx0 + n_sims * QTTT_SIM_STRIDE), bit for bit.  One lane per (pair, simulation, child): lanes 2k, 2k + 1 are the
// two children of one (pair, simulation), so a batch of 65 536 pairs with one playout per child is two waves per
// SIMD instead of one with twice the chain; a workgroup owns BLOCK / (2 n_sims) whole pairs, so the per-child sums
// are LDS adds.  Every lane redoes the (cheap) expansion of its pair; the lane of (simulation 0, child 0) writes the
// pair's children and bookkeeping.
//   value_sum i32[n,2]: sum over the child's simulations of `r if leaf.turn else -r` (mcts.py:174; leaf.turn is
//   True after an even number of real moves: reset's len(moves) % 2 == 0 flipped once per _step, mcts.py:140,243);
//   0 for a child that does not exist.  result i8[n,2,n_sims] (nullable): every simulation's MCTS._reward.
template <int BLOCK, bool PYKEY>
__global__ __launch_bounds__(BLOCK) void expand_rollout_kernel(
    const u64 *pP, const u64 *pQ, const uint8_t *action36,
    u64 *c0P, u64 *c0Q, u64 *c1P, u64 *c1Q, ExpandOut out,
    u64 seed, u32 step_idx0, u64 board_offset, u32 n_sims, u32 pairs_per_block,
    int32_t *value_sum, int8_t *result, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    __shared__ uint8_t nth9[NTH9_BYTES];
    __shared__ u64 htbl[PYKEY ? PYHASH_LUT_WORDS : 1];
    __shared__ u64 ltbl[512];
    __shared__ int acc[BLOCK];                                  // [pair of the workgroup][child]
    __shared__ u64 keytab[PLAYOUT_KEY_SLOTS * PLAYOUT_PLIES];   // slot = child * n_sims + sim
    const bool table = 2u * n_sims <= PLAYOUT_KEY_SLOTS;        // wave-uniform
    const u32 per_pair = 2u * n_sims;
    const u32 pl = threadIdx.x / per_pair;                      // pair of the workgroup
    const u32 rem = threadIdx.x - pl * per_pair;
    const u32 sim = rem >> 1, child = rem & 1u;
    const int64_t i = (int64_t)blockIdx.x * pairs_per_block + pl;
    const bool valid = pl < pairs_per_block && i < n;
    const int64_t il = valid ? i : 0;
    const u64 P = pP[il], Q = pQ[il];                           // 2 n_sims neighbouring lanes read the same 16 bytes
    const u32 a = (u32)action36[il];
    acc[threadIdx.x] = 0;
    fill_policy_lut<BLOCK>(plut);
    fill_nth9<BLOCK>(nth9);
    if (PYKEY) fill_pyhash_lut<BLOCK>(htbl);
    if (out.legal) fill_legal_lut<BLOCK>(ltbl);
    if (table) fill_playout_keys_nosync<BLOCK>(keytab, seed, step_idx0, 2u * n_sims);
    fill_line_lut<BLOCK>(lut);                                  // ends with the workgroup barrier
    int r = 0;
    u32 kids = 0;
    if (valid) {
        const u32 pr = a < 36u ? (u32)g_pair_lut.b[a] : 0u;
        const u32 act = (pr & 0xFu) | ((pr >> 4) << 8);
        u32 Q0 = (u32)Q, Q1 = (u32)(Q >> 32), P0a, P1a, P0b, P1b, xo0, xo1;
        kids = step_core_both((u32)P, (u32)(P >> 32), Q0, Q1, act, lut, P0a, P1a, P0b, P1b, xo0, xo1);
        if (rem == 0u) {                                        // this pair's writer
            u64 kidP[2], kidQ[2];
            kidP[0] = (u64)P0a | ((u64)P1a << 32);
            kidP[1] = (u64)P0b | ((u64)P1b << 32);
            kidQ[0] = kidQ[1] = (u64)Q0 | ((u64)Q1 << 32);
            if (c0P) { c0P[i] = kidP[0]; c0Q[i] = kidQ[0]; }
            if (c1P) { c1P[i] = kidP[1]; c1Q[i] = kidQ[1]; }
            expand_bookkeeping<PYKEY>(kids, kidP, kidQ, xo0, xo1, out, i, lut, htbl, ltbl);
        }
        if (child < kids) {
            u32 P0 = child ? P0b : P0a, P1 = child ? P1b : P1a;
            const u32 slot = child * n_sims + sim, id = fold_id(board_offset + (u64)i);
            if (table) playout<true>(P0, P1, Q0, Q1, id, seed, 0u, keytab + slot * PLAYOUT_PLIES, lut, plut, nth9);
            else playout<false>(P0, P1, Q0, Q1, id, seed, step_idx0 + slot * QTTT_SIM_STRIDE, nullptr, lut, plut, nth9);
            int w, t;
            lite_update_winner(lite_unpack((u64)P0 | ((u64)P1 << 32)), lut, w, t);
            r = w < 0 ? 0 : (w ? 1 : -1);                       // MCTS._reward, mcts.py:200-209
            // leaf.turn (mcts.py:174): the child has one real move more than the parent
            const u32 child_real = ((child ? P1b : P1a) >> P1_N_SHIFT) & 0xFu;
            if (r) atomicAdd(&acc[pl * 2u + child], (child_real & 1u) ? -r : r);
        }
        if (result) result[(i * 2 + child) * (int64_t)n_sims + sim] = (int8_t)r;
    }
    __syncthreads();
    if (threadIdx.x < 2u * pairs_per_block) {
        const int64_t o = (int64_t)blockIdx.x * pairs_per_block * 2 + threadIdx.x;
        if (o < 2 * n) value_sum[o] = acc[threadIdx.x];
    }
}

// The same operator for batches where the playouts are the work (many simulations per child, or many pairs): a lane per
// (pair, simulation, child) leaves the lanes of children that do not exist idle (a collapse happens on ~22 % of the
// moves: 39 % of the child slots are empty) and redoes the expansion in every lane.  Here a workgroup owns P pairs:
//   1. lane t < P expands pair t once, writes the children and their bookkeeping, and leaves the two child states in LDS;
//   2. a workgroup scan over the pairs' child counts numbers the children that exist ("units");
//   3. the playouts of all units — unit u, simulation s is job u * n_sims + s — are dealt to the lanes in job order, so
//      every wave but the last of a workgroup runs full, and neighbouring lanes play the same child (similar lengths);
//   4. per-child sums through LDS adds, as above.
// Same results as expand_rollout_kernel, bit for bit.
constexpr int XR_MAX_PAIRS = 256;
template <int BLOCK, bool PYKEY>
__global__ __launch_bounds__(BLOCK) void expand_rollout_jobs_kernel(
    const u64 *pP, const u64 *pQ, const uint8_t *action36,
    u64 *c0P, u64 *c0Q, u64 *c1P, u64 *c1Q, ExpandOut out,
    u64 seed, u32 step_idx0, u64 board_offset, u32 n_sims, u32 pairs_per_block,
    int32_t *value_sum, int8_t *result, int64_t n) {
    static_assert(BLOCK >= XR_MAX_PAIRS, "one pair per lane in the expansion phase");
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    __shared__ uint8_t nth9[NTH9_BYTES];
    __shared__ u64 htbl[PYKEY ? PYHASH_LUT_WORDS : 1];
    __shared__ u64 ltbl[512];
    __shared__ u64 kidPs[XR_MAX_PAIRS * 2];                     // [pair][child] plane-P word; plane Q is the same for both
    __shared__ u64 kidQs[XR_MAX_PAIRS];
    __shared__ int acc[XR_MAX_PAIRS * 2];
    __shared__ uint16_t unit_tbl[XR_MAX_PAIRS * 2];             // unit -> pair << 1 | child
    __shared__ u32 wave_tot[BLOCK / 64];
    __shared__ u64 keytab[PLAYOUT_KEY_SLOTS * PLAYOUT_PLIES];   // slot = child * n_sims + sim
    const bool table = 2u * n_sims <= PLAYOUT_KEY_SLOTS;        // wave-uniform
    const u32 t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * pairs_per_block;
    const int64_t i = base + t;
    const bool valid = t < pairs_per_block && i < n;
    LegalLutWords<BLOCK> lw;
    if (out.legal) lw.request();                                // in front of the state loads, stored behind them
    const int64_t il = valid ? i : 0;
    const u64 P = pP[il], Q = pQ[il];
    const u32 a = (u32)action36[il];
    for (u32 k = t; k < 2u * XR_MAX_PAIRS; k += BLOCK) acc[k] = 0;
    fill_policy_lut<BLOCK>(plut);
    fill_nth9<BLOCK>(nth9);
    if (PYKEY) fill_pyhash_lut<BLOCK>(htbl);
    if (out.legal) lw.store(ltbl);
    if (table) fill_playout_keys_nosync<BLOCK>(keytab, seed, step_idx0, 2u * n_sims);
    fill_line_lut<BLOCK>(lut);                                  // ends with the workgroup barrier
    // ---- 1. the expansions
    u32 kids = 0;
    if (valid) {
        const u32 pr = a < 36u ? (u32)g_pair_lut.b[a] : 0u;
        const u32 act = (pr & 0xFu) | ((pr >> 4) << 8);
        u32 Q0 = (u32)Q, Q1 = (u32)(Q >> 32), P0a, P1a, P0b, P1b, xo0, xo1;
        kids = step_core_both((u32)P, (u32)(P >> 32), Q0, Q1, act, lut, P0a, P1a, P0b, P1b, xo0, xo1);
        u64 kidP[2], kidQ[2];
        kidP[0] = (u64)P0a | ((u64)P1a << 32);
        kidP[1] = (u64)P0b | ((u64)P1b << 32);
        kidQ[0] = kidQ[1] = (u64)Q0 | ((u64)Q1 << 32);
        kidPs[2u * t] = kidP[0]; kidPs[2u * t + 1u] = kidP[1]; kidQs[t] = kidQ[0];
        if (c0P) { store_stream(&c0P[i], kidP[0]); store_stream(&c0Q[i], kidQ[0]); }
        if (c1P) { store_stream(&c1P[i], kidP[1]); store_stream(&c1Q[i], kidQ[1]); }
        expand_bookkeeping<PYKEY>(kids, kidP, kidQ, xo0, xo1, out, i, lut, htbl, ltbl);
        if (result)                                             // children that do not exist: every simulation reads 0
            for (u32 c = kids; c < 2u; ++c)
                for (u32 s = 0; s < n_sims; ++s) result[(i * 2 + c) * (int64_t)n_sims + s] = 0;
    }
    // ---- 2. number the children that exist: exclusive scan of `kids` over the workgroup (wave scan + wave totals)
    u32 incl = kids;
#pragma unroll
    for (u32 d = 1; d < 64u; d <<= 1) {
        const u32 up = (u32)__shfl_up((int)incl, d, 64);
        if ((t & 63u) >= d) incl += up;
    }
    if ((t & 63u) == 63u) wave_tot[t >> 6] = incl;
    __syncthreads();
    u32 before = 0, units = 0;
#pragma unroll
    for (u32 w = 0; w < (u32)(BLOCK / 64); ++w) {
        const u32 x = wave_tot[w];
        before += w < (t >> 6) ? x : 0u;
        units += x;
    }
    const u32 first = before + incl - kids;
    for (u32 c = 0; c < kids; ++c) unit_tbl[first + c] = (uint16_t)((t << 1) | c);
    __syncthreads();
    // ---- 3. the playouts, dealt in job order
    const u32 jobs = units * n_sims;
    for (u32 j = t; j < jobs; j += BLOCK) {
        const u32 u = j / n_sims, sim = j - u * n_sims;
        const u32 e = unit_tbl[u], pl = e >> 1, child = e & 1u;
        const u64 cP = kidPs[e], cQ = kidQs[pl];
        u32 P0 = (u32)cP, P1 = (u32)(cP >> 32), Q0 = (u32)cQ, Q1 = (u32)(cQ >> 32);
        const u32 child_real = (P1 >> P1_N_SHIFT) & 0xFu;
        const int64_t ip = base + pl;
        const u32 slot = child * n_sims + sim, id = fold_id(board_offset + (u64)ip);
        if (table) playout<true>(P0, P1, Q0, Q1, id, seed, 0u, keytab + slot * PLAYOUT_PLIES, lut, plut, nth9);
        else playout<false>(P0, P1, Q0, Q1, id, seed, step_idx0 + slot * QTTT_SIM_STRIDE, nullptr, lut, plut, nth9);
        int w, tm;
        lite_update_winner(lite_unpack((u64)P0 | ((u64)P1 << 32)), lut, w, tm);
        const int r = w < 0 ? 0 : (w ? 1 : -1);                 // MCTS._reward, mcts.py:200-209
        if (r) atomicAdd(&acc[e], (child_real & 1u) ? -r : r);  // leaf.turn (mcts.py:174)
        if (result) result[(ip * 2 + child) * (int64_t)n_sims + sim] = (int8_t)r;
    }
    __syncthreads();
    for (u32 k = t; k < 2u * pairs_per_block; k += BLOCK) {     // (with one pair per lane there are two sums per lane)
        const int64_t o = base * 2 + k;
        if (o < 2 * n) value_sum[o] = acc[k];
    }
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
