// tools/xr_sims_per_lane.cpp — lab harness (not part of the product, not the judged bench): BASELINE config 5's unit with
// ten playouts per child (qttt_expand_rollout, mcts.py:166-176), A/B over HOW MANY SIMULATIONS ONE LANE PLAYS.
//
// The question: every wave runs to its longest playout; does giving a lane k playouts make the lanes'
// totals more alike (sums of k lengths have 1/sqrt(k) the relative spread) and the launch faster?  Three formulations,
// all bit-identical in value_sum to the product's kernels (checked here before timing):
//   A  ksims<KS>   the product's lane-per-(pair, simulation, child) kernel with KS simulations per lane, one after the
//                  other (KS = 1 is the product's expand_rollout_kernel).  A lane's playouts run in lock-step with its
//                  wave: each of the KS rounds still lasts as long as the wave's longest playout of that round.
//   B  jobs<P>     the product's job-list kernel (the children that exist dealt to the lanes): P pairs per workgroup is
//                  ~0.055 * P jobs per lane (P = 64, the product's choice at 65 536 pairs: 3.5).  Rounds, as in A.
//   C  refill<P>   the job list with ONE ply loop per lane: a lane whose playout has ended takes its next job at once, so
//                  a wave lasts as long as its busiest lane's TOTAL — the only formulation in which the 1/sqrt(k) argument
//                  applies.  The price: the end-of-playout work (winner, LDS add, next job's state) sits inside the ply
//                  loop, and a wave in which any lane ends a playout executes it.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=4 -Iinclude -Iqtttgym_amd/csrc \
//         tools/xr_sims_per_lane.cpp -o tools/xr_sims_per_lane
//   tools/xr_sims_per_lane [pairs = 65536] [plies before = 4] [K = 50] [reps = 9] [sims = 10]
#include "qttt_step_kernels.h"
#include "qttt_aux_kernels.h"
#include "qttt_mcts_kernels.h"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>
#include <functional>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

// ---- A: KS simulations per lane, lane = (pair, simulation group, child) ------------------------------------------
template <int BLOCK, int KS>
__global__ __launch_bounds__(BLOCK) void xr_ksims_kernel(const u64 *pP, const u64 *pQ, const uint8_t *action36, u64 seed, u32 step_idx0,
                                                        u64 board_offset, u32 n_sims, u32 pairs_per_block, int32_t *value_sum, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    __shared__ uint8_t nth9[NTH9_BYTES];
    __shared__ int acc[BLOCK];
    __shared__ u64 keytab[PLAYOUT_KEY_SLOTS * PLAYOUT_PLIES];
    const u32 groups = (n_sims + KS - 1) / KS, per_pair = 2u * groups;
    const u32 pl = threadIdx.x / per_pair, rem = threadIdx.x - pl * per_pair, grp = rem >> 1, child = rem & 1u;
    const int64_t i = (int64_t)blockIdx.x * pairs_per_block + pl;
    const bool valid = pl < pairs_per_block && i < n;
    const int64_t il = valid ? i : 0;
    const u64 P = pP[il], Q = pQ[il];
    const u32 a = (u32)action36[il];
    acc[threadIdx.x] = 0;
    fill_policy_lut<BLOCK>(plut);
    fill_nth9<BLOCK>(nth9);
    fill_playout_keys_nosync<BLOCK>(keytab, seed, step_idx0, 2u * n_sims);
    fill_line_lut<BLOCK>(lut);
    if (valid) {
        const u32 pr = a < 36u ? (u32)g_pair_lut.b[a] : 0u;
        const u32 act = (pr & 0xFu) | ((pr >> 4) << 8);
        u32 Q0 = (u32)Q, Q1 = (u32)(Q >> 32), P0a, P1a, P0b, P1b, xo0, xo1;
        const u32 kids = step_core_both((u32)P, (u32)(P >> 32), Q0, Q1, act, lut, P0a, P1a, P0b, P1b, xo0, xo1);
        if (child < kids) {
            const u32 cP0 = child ? P0b : P0a, cP1 = child ? P1b : P1a, id = fold_id(board_offset + (u64)i);
            const u32 child_real = (cP1 >> P1_N_SHIFT) & 0xFu;
            int sum = 0;
            for (u32 s = grp * KS; s < grp * KS + KS && s < n_sims; ++s) {
                u32 P0 = cP0, P1 = cP1, q0 = Q0, q1 = Q1;
                playout<true>(P0, P1, q0, q1, id, seed, 0u, keytab + (child * n_sims + s) * PLAYOUT_PLIES, lut, plut, nth9);
                int w, t;
                lite_update_winner(lite_unpack((u64)P0 | ((u64)P1 << 32)), lut, w, t);
                const int r = w < 0 ? 0 : (w ? 1 : -1);
                sum += (child_real & 1u) ? -r : r;
            }
            if (sum) atomicAdd(&acc[pl * 2u + child], sum);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2u * pairs_per_block) {
        const int64_t o = (int64_t)blockIdx.x * pairs_per_block * 2 + threadIdx.x;
        if (o < 2 * n) value_sum[o] = acc[threadIdx.x];
    }
}

// ---- B / C: the job list; REFILL = one ply loop per lane ------------------------------------------------------------
template <int BLOCK, bool REFILL>
__global__ __launch_bounds__(BLOCK) void xr_jobs_kernel(const u64 *pP, const u64 *pQ, const uint8_t *action36, u64 seed, u32 step_idx0,
                                                       u64 board_offset, u32 n_sims, u32 pairs_per_block, int32_t *value_sum, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    __shared__ uint8_t nth9[NTH9_BYTES];
    __shared__ u64 kidPs[XR_MAX_PAIRS * 2];
    __shared__ u64 kidQs[XR_MAX_PAIRS];
    __shared__ int acc[XR_MAX_PAIRS * 2];
    __shared__ uint16_t unit_tbl[XR_MAX_PAIRS * 2];
    __shared__ u32 wave_tot[BLOCK / 64];
    __shared__ u64 keytab[PLAYOUT_KEY_SLOTS * PLAYOUT_PLIES];
    const u32 t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * pairs_per_block;
    const int64_t i = base + t;
    const bool valid = t < pairs_per_block && i < n;
    const int64_t il = valid ? i : 0;
    const u64 P = pP[il], Q = pQ[il];
    const u32 a = (u32)action36[il];
    for (u32 k = t; k < 2u * XR_MAX_PAIRS; k += BLOCK) acc[k] = 0;
    fill_policy_lut<BLOCK>(plut);
    fill_nth9<BLOCK>(nth9);
    fill_playout_keys_nosync<BLOCK>(keytab, seed, step_idx0, 2u * n_sims);
    fill_line_lut<BLOCK>(lut);
    u32 kids = 0;
    if (valid) {
        const u32 pr = a < 36u ? (u32)g_pair_lut.b[a] : 0u;
        const u32 act = (pr & 0xFu) | ((pr >> 4) << 8);
        u32 Q0 = (u32)Q, Q1 = (u32)(Q >> 32), P0a, P1a, P0b, P1b, xo0, xo1;
        kids = step_core_both((u32)P, (u32)(P >> 32), Q0, Q1, act, lut, P0a, P1a, P0b, P1b, xo0, xo1);
        kidPs[2u * t] = (u64)P0a | ((u64)P1a << 32); kidPs[2u * t + 1u] = (u64)P0b | ((u64)P1b << 32);
        kidQs[t] = (u64)Q0 | ((u64)Q1 << 32);
    }
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
    const u32 jobs = units * n_sims;
    if (!REFILL) {
        for (u32 j = t; j < jobs; j += BLOCK) {
            const u32 u = j / n_sims, sim = j - u * n_sims;
            const u32 e = unit_tbl[u], pl = e >> 1, child = e & 1u;
            const u64 cP = kidPs[e], cQ = kidQs[pl];
            u32 P0 = (u32)cP, P1 = (u32)(cP >> 32), Q0 = (u32)cQ, Q1 = (u32)(cQ >> 32);
            const u32 child_real = (P1 >> P1_N_SHIFT) & 0xFu;
            playout<true>(P0, P1, Q0, Q1, fold_id(board_offset + (u64)(base + pl)), seed, 0u,
                          keytab + (child * n_sims + sim) * PLAYOUT_PLIES, lut, plut, nth9);
            int w, tm;
            lite_update_winner(lite_unpack((u64)P0 | ((u64)P1 << 32)), lut, w, tm);
            const int r = w < 0 ? 0 : (w ? 1 : -1);
            if (r) atomicAdd(&acc[e], (child_real & 1u) ? -r : r);
        }
    } else {
        // one ply loop: `ply` counts the plies of the CURRENT job; a lane that ends a playout finishes it and takes its
        // next job in the same trip.  Exit: every lane's job index runs past `jobs` (each trip either plays a ply of a
        // playout of at most nine plies or advances j by BLOCK: at most ceil(jobs / BLOCK) * 10 trips).
        u32 j = t, ply = 0, P0 = 0, P1 = 0, Q0 = 0, Q1 = 0, e = 0, id = 0, child_real = 0;
        const u64 *keys = keytab;
        bool active = j < jobs;
        auto take = [&]() {
            const u32 u = j / n_sims, sim = j - u * n_sims;
            e = unit_tbl[u];
            const u32 pl = e >> 1, child = e & 1u;
            const u64 cP = kidPs[e], cQ = kidQs[pl];
            P0 = (u32)cP; P1 = (u32)(cP >> 32); Q0 = (u32)cQ; Q1 = (u32)(cQ >> 32);
            child_real = (P1 >> P1_N_SHIFT) & 0xFu;
            id = fold_id(board_offset + (u64)(base + pl));
            keys = keytab + (child * n_sims + sim) * PLAYOUT_PLIES;
            ply = 0;
        };
        if (active) take();
        while (__builtin_amdgcn_ballot_w64(active) != 0ull) {
            if (active) {
                u32 empty = ~(P1 >> P1_CL_SHIFT) & 0x1FFu;
                bool over = (P1 >> 31) || (empty & (empty - 1u)) == 0u || ply >= PLAYOUT_PLIES;
                if (!over) {
                    const u64 key = keys[ply];
                    const u32 h1 = lowbias32(id ^ (u32)key);
                    const u32 h2 = lowbias32(h1 ^ (u32)(key >> 32));
                    const u32 act = policy_action_nth9(plut, nth9, empty, h2);
                    step_core<false, true>(P0, P1, Q0, Q1, act, h1 >> 31, lut);
                    ++ply;
                    empty = ~(P1 >> P1_CL_SHIFT) & 0x1FFu;
                    over = (P1 >> 31) || (empty & (empty - 1u)) == 0u || ply >= PLAYOUT_PLIES;
                }
                if (over) {
                    int w, tm;
                    lite_update_winner(lite_unpack((u64)P0 | ((u64)P1 << 32)), lut, w, tm);
                    const int r = w < 0 ? 0 : (w ? 1 : -1);
                    if (r) atomicAdd(&acc[e], (child_real & 1u) ? -r : r);
                    j += BLOCK;
                    active = j < jobs;
                    if (active) take();
                }
            }
        }
    }
    __syncthreads();
    for (u32 k = t; k < 2u * pairs_per_block; k += BLOCK) {
        const int64_t o = base * 2 + k;
        if (o < 2 * n) value_sum[o] = acc[k];
    }
}

struct Variant {
    std::string name;
    std::function<void(hipStream_t, int32_t *)> launch;
    std::vector<float> us;
};

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 65536;
    const int plies = argc > 2 ? atoi(argv[2]) : 4;
    const int K = argc > 3 ? atoi(argv[3]) : 50;
    const int reps = argc > 4 ? atoi(argv[4]) : 9;
    const u32 S = argc > 5 ? (u32)atoi(argv[5]) : 10u;
    if (2u * S > PLAYOUT_KEY_SLOTS || S < 1u) { fprintf(stderr, "1 <= sims <= %u (the key table)\n", PLAYOUT_KEY_SLOTS / 2u); return 2; }
    const int64_t s64 = plane_stride(n);
    u64 *state; CK(hipMalloc(&state, s64 * 16)); CK(hipMemset(state, 0, s64 * 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    Planes p = planes(state, n);
    FusedKeys fk; for (int t = 0; t < FUSED_MAX_PLIES; ++t) fk.k[t] = launch_key(7, (u32)t);
    if (plies > 0)
        hipLaunchKernelGGL((step_random_fused_kernel<256, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.P, p.Q, fk, (u64)0,
                           (uint16_t *)nullptr, (u32 *)nullptr, (uint8_t *)nullptr, (int64_t)0, n, plies, (float *)nullptr);
    uint8_t *act36; CK(hipMalloc(&act36, n));
    { std::vector<uint8_t> ha(n); u32 x = 12345u; for (auto &v : ha) { x = x * 1664525u + 1013904223u; v = (uint8_t)((x >> 16) % 36u); }
      CK(hipMemcpy(act36, ha.data(), n, hipMemcpyHostToDevice)); }
    int32_t *vs_ref, *vs_out; CK(hipMalloc(&vs_ref, 8 * n)); CK(hipMalloc(&vs_out, 8 * n));
    const ExpandOut none = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    std::vector<Variant> vs;
    // the product's own two kernels, children and bookkeeping not written (the playouts are what is compared)
    vs.push_back({"product expand_rollout_kernel (1 sim per lane)", [=](hipStream_t st, int32_t *o) {
        const u32 ppb = 256u / (2u * S);
        hipLaunchKernelGGL((expand_rollout_kernel<256, false>), dim3((unsigned)((n + ppb - 1) / ppb)), dim3(256), 0, st, p.P, p.Q, act36,
                           (u64 *)nullptr, (u64 *)nullptr, (u64 *)nullptr, (u64 *)nullptr, none, (u64)5, 0u, (u64)0, S, ppb, o, (int8_t *)nullptr, n); }, {}});
    vs.push_back({"product expand_rollout_jobs_kernel P=64", [=](hipStream_t st, int32_t *o) {
        hipLaunchKernelGGL((expand_rollout_jobs_kernel<256, false>), dim3((unsigned)((n + 63) / 64)), dim3(256), 0, st, p.P, p.Q, act36,
                           (u64 *)nullptr, (u64 *)nullptr, (u64 *)nullptr, (u64 *)nullptr, none, (u64)5, 0u, (u64)0, S, 64u, o, (int8_t *)nullptr, n); }, {}});
#define KSV(KS)                                                                                                          \
    vs.push_back({"A ksims: " #KS " sims per lane", [=](hipStream_t st, int32_t *o) {                                     \
        const u32 ppb = 256u / (2u * ((S + KS - 1) / KS));                                                               \
        hipLaunchKernelGGL((xr_ksims_kernel<256, KS>), dim3((unsigned)((n + ppb - 1) / ppb)), dim3(256), 0, st, p.P, p.Q, act36, (u64)5, 0u, \
                           (u64)0, S, ppb, o, n); }, {}});
    KSV(1) KSV(2) KSV(5) KSV(10)
#define JV(PPB, RF, TAG)                                                                                                 \
    vs.push_back({TAG " P=" #PPB, [=](hipStream_t st, int32_t *o) {                                                      \
        hipLaunchKernelGGL((xr_jobs_kernel<256, RF>), dim3((unsigned)((n + PPB - 1) / PPB)), dim3(256), 0, st, p.P, p.Q, act36, (u64)5, 0u, \
                           (u64)0, S, (u32)PPB, o, n); }, {}});
    JV(32, false, "B jobs (rounds)") JV(64, false, "B jobs (rounds)") JV(128, false, "B jobs (rounds)") JV(256, false, "B jobs (rounds)")
    JV(32, true, "C refill") JV(64, true, "C refill") JV(128, true, "C refill") JV(256, true, "C refill")
    // ---- every variant gives the product's value_sum, element for element
    vs[0].launch(s, vs_ref);
    CK(hipStreamSynchronize(s));
    std::vector<int32_t> ref(2 * n), got(2 * n);
    CK(hipMemcpy(ref.data(), vs_ref, 8 * n, hipMemcpyDeviceToHost));
    long long nz = 0; for (auto v : ref) nz += v != 0;
    for (auto &v : vs) {
        CK(hipMemsetAsync(vs_out, 0xFF, 8 * n, s));
        v.launch(s, vs_out);
        CK(hipStreamSynchronize(s));
        CK(hipGetLastError());
        CK(hipMemcpy(got.data(), vs_out, 8 * n, hipMemcpyDeviceToHost));
        if (got != ref) { fprintf(stderr, "MISMATCH: %s\n", v.name.c_str()); return 1; }
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < reps; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            Variant &v = vs[(i + r) % vs.size()];
            for (int k = 0; k < 3; ++k) v.launch(s, vs_out);
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < K; ++k) v.launch(s, vs_out);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            v.us.push_back(ms * 1e3f / K);
        }
    CK(hipGetLastError());
    printf("xr_sims_per_lane: %lld pairs after %d random plies, %u playouts per child, K=%d, %d reps, alternating; every variant's "
           "value_sum == the product's (%lld non-zero sums)\n  us per launch: min / median\n", (long long)n, plies, S, K, reps, nz);
    for (auto &v : vs) {
        std::sort(v.us.begin(), v.us.end());
        printf("  %-50s %8.2f %8.2f\n", v.name.c_str(), v.us.front(), v.us[v.us.size() / 2]);
    }
    return 0;
}
