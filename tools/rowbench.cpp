// tools/rowbench.cpp — lab harness for the cold / §8(f) row kernels (not part of the product, not the judged
// bench).  Includes the product's kernel headers directly and times launch-shape variants of one kernel
// INTERLEAVED in one process, beside a traffic-floor kernel that moves the same bytes in the same pattern
// with no board logic.  States come from the product's own fused random-policy kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Iqtttgym_amd/csrc tools/rowbench.cpp -o tools/rowbench
//   tools/rowbench N PLIES K REPS
#include "qttt_step_kernels.h"
#include "qttt_aux_kernels.h"
#include "qttt_mcts_kernels.h"
#include <cstdio>
#include <vector>
#include <algorithm>
#include <string>
#include <functional>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

// the export kernel's memory pattern alone: 16 B read per board, the five tiles streamed out by the wave
// that owns them, nothing computed (the tile holds whatever LDS held)
template <int BLOCK, int BPL>
__global__ __launch_bounds__(BLOCK) void export_floor(const u64 *pP, const u64 *pQ, ExpOut out, int64_t n) {
    constexpr u32 TILE = BLOCK * BPL;
    __shared__ __attribute__((aligned(16))) uint8_t tile[exp_lds_bytes(TILE)];
    const int64_t base = (int64_t)blockIdx.x * TILE;
    const u32 valid = (u32)min((int64_t)TILE, n - base);
    typedef Vec<u64, BPL> V64;
    const u32 b0 = threadIdx.x * BPL;
    V64 p, q;
    if (b0 + BPL <= valid) {
        p = load_stream(&reinterpret_cast<const V64 *>(pP + base)[threadIdx.x]);
        q = load_stream(&reinterpret_cast<const V64 *>(pQ + base)[threadIdx.x]);
        u64 x = 0;
#pragma unroll
        for (int k = 0; k < BPL; ++k) x ^= p.v[k] + q.v[k];
        reinterpret_cast<u64 *>(tile)[threadIdx.x] = x;                 // keeps the loads alive
    }
    uint8_t *l_mv = tile, *l_bd = l_mv + obs_tile_bytes(TILE, 18), *l_qm = l_bd + obs_tile_bytes(TILE, 9);
    uint8_t *l_nm = l_qm + obs_tile_bytes(TILE, 8), *l_nq = l_nm + obs_tile_bytes(TILE, 1);
    const u32 w0 = (threadIdx.x & ~63u) * BPL, w1 = min(w0 + 64u * BPL, valid);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    if (w0 < valid) {
        wave_copy_out16(out.moves + base * 18, l_mv, w0 * 18u, w1 * 18u);
        wave_copy_out16(reinterpret_cast<uint8_t *>(out.board) + base * 9, l_bd, w0 * 9u, w1 * 9u);
        wave_copy_out16(reinterpret_cast<uint8_t *>(out.qmask) + base * 8, l_qm, w0 * 8u, w1 * 8u);
        wave_copy_out16(out.n_moves + base, l_nm, w0, w1);
        wave_copy_out16(out.n_q + base, l_nq, w0, w1);
    }
}

// EXPERIMENT (measured, not adopted): the fused random stepper with TWO boards per lane (two independent chains in one
// instruction stream), auto-reset, every ply's outputs kept; even n only.  us per 64-ply launch, one / two boards per
// lane: 65 536 boards 41.4 / 61.5, 262 144: 80.0 / 86.0, 524 288: 134 / 140, 1 M: 243 / 254 — waves, not chains per
// wave, are what the SIMDs want
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void step_random_fused2_kernel(
    u64 *__restrict__ pP, u64 *__restrict__ pQ, u64 seed, u32 step_idx0, u64 board_offset,
    uint16_t *__restrict__ actions_out, u32 *__restrict__ reward_bits, uint8_t *__restrict__ terminated,
    int64_t out_stride, int64_t n, int32_t n_steps) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ __attribute__((aligned(16))) uint8_t plut[POLICY_LUT_WORDS * 4];
    __shared__ uint8_t nth9[NTH9_BYTES];
    typedef Vec<u64, 2> V64;
    const int64_t jb = (int64_t)blockIdx.x * BLOCK;                 // first lane-group of the workgroup
    const int64_t j = jb + threadIdx.x;
    const bool active = 2 * j + 1 < n;
    const int64_t jl = active ? j : 0;
    V64 p = load_stream(&reinterpret_cast<const V64 *>(pP)[jl]);
    V64 q = load_stream(&reinterpret_cast<const V64 *>(pQ)[jl]);
    fill_policy_lut<BLOCK>(plut);
    fill_nth9<BLOCK>(nth9);
    fill_line_lut<BLOCK>(lut);
    if (!active) return;
    u32 P0[2], P1[2], Q0[2], Q1[2], id[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        P0[k] = (u32)p.v[k]; P1[k] = (u32)(p.v[k] >> 32); Q0[k] = (u32)q.v[k]; Q1[k] = (u32)(q.v[k] >> 32);
        id[k] = fold_id(board_offset + (u64)(2 * j + k));
    }
    u32 *a_blk = reinterpret_cast<u32 *>(actions_out + 2 * jb);
    u64 *r_blk = reinterpret_cast<u64 *>(reward_bits + 2 * jb);
    uint16_t *t_blk = reinterpret_cast<uint16_t *>(terminated + 2 * jb);
    const u32 lane = threadIdx.x;
    for (int32_t t = 0; t < n_steps; ++t) {
        const u64 key = launch_key(seed, step_idx0 + (u32)t);
        u32 act[2], win[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const u32 h1 = lowbias32(id[k] ^ (u32)key);
            const u32 h2 = lowbias32(h1 ^ (u32)(key >> 32));
            const u32 keep = ~(u32)((int)P1[k] >> 31);
            P0[k] &= keep; P1[k] &= keep; Q0[k] &= keep; Q1[k] &= keep;
            const u32 empty = ~(P1[k] >> P1_CL_SHIFT) & 0x1FFu;
            act[k] = policy_action_nth9(plut, nth9, empty, h2);
            win[k] = step_core<false, true>(P0[k], P1[k], Q0[k], Q1[k], act[k], h1 >> 31, lut);
        }
        store_stream(&a_blk[lane], act[0] | (act[1] << 16));
        const u64 rw = (u64)(0x80000000u | (win[0] << 23)) | ((u64)(0x80000000u | (win[1] << 23)) << 32);
        store_stream(&r_blk[lane], rw);
        store_stream(&t_blk[lane], (uint16_t)((P1[0] >> 31) | ((P1[1] >> 31) << 8)));
        a_blk += out_stride / 2 * 1;                                  // strides in boards: two boards per element
        r_blk += out_stride / 2;
        t_blk += out_stride / 2;
    }
    V64 po, qo;
#pragma unroll
    for (int k = 0; k < 2; ++k) { po.v[k] = (u64)P0[k] | ((u64)P1[k] << 32); qo.v[k] = (u64)Q0[k] | ((u64)Q1[k] << 32); }
    store_stream(&reinterpret_cast<V64 *>(pP)[j], po);
    store_stream(&reinterpret_cast<V64 *>(pQ)[j], qo);
}

// Experiment: node_info (winner, terminal, legal mask, native key) with no LDS table and no workgroup barrier — the line
// test and the legal mask done arithmetically (more VALU, but the kernel is traffic-bound and a wave then depends on
// nothing but its own loads).  Same outputs as node_info_kernel<BLOCK, false>.
__device__ __forceinline__ bool has_line_arith(u32 m) {
    const u32 rows = m & (m >> 1) & (m >> 2) & 0x049u, cols = m & (m >> 3) & (m >> 6) & 0x007u;
    return ((rows | cols) != 0u) | ((m & 0x111u) == 0x111u) | ((m & 0x054u) == 0x054u);
}
__device__ __forceinline__ void winner_arith(const Lite &s, int &winner, int &terminal) {
    const u32 W = (u32)(s.P >> 2);
    const u32 c8 = (u32)(s.P >> 34) & 0xFu;
    const u32 par = W & 0x11111111u;
    const u32 even = __builtin_amdgcn_udot8(par, 0x00008421u, 0u, false) |
                     (__builtin_amdgcn_udot8(par, 0x84210000u, 0u, false) << 4) | ((c8 & 1u) << 8);
    const u32 X = s.cl & even, O = s.cl & ~even;
    const bool hx = has_line_arith(X), ho = has_line_arith(O);
    winner = hx ? 1 : (ho ? 0 : -1);
    if (hx && ho) {
        u32 ge[3];
#pragma unroll
        for (u32 k = 0; k < 3; ++k) {
            const u32 T = 9u + k;
            const u32 y = ((W & 0x77777777u) + 0x11111111u * (16u - T)) & W & 0x88888888u;
            ge[k] = ((__builtin_amdgcn_udot8(y, 0x00008421u, 0u, false) |
                      (__builtin_amdgcn_udot8(y, 0x84210000u, 0u, false) << 4)) >> 3) | ((c8 >= T ? 1u : 0u) << 8);
        }
        winner = (has_line_arith(X & ge[2]) || (has_line_arith(X & ge[0]) && !has_line_arith(O & ge[1]))) ? 1 : 0;
    }
    terminal = (s.n == 9u || hx || ho) ? 1 : 0;
}
template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void node_info_nolds_kernel(const u64 *pP, const u64 *pQ, int8_t *winner, uint8_t *terminal,
                                                                u64 *legal, u64 *skey, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    const int64_t i0 = 2 * j;
    if (i0 + 1 >= n) return;                                            // (experiment: even batches only)
    typedef Vec<u64, 2> V64;
    const V64 p = load_stream(&reinterpret_cast<const V64 *>(pP)[j]), q = load_stream(&reinterpret_cast<const V64 *>(pQ)[j]);
    V64 k2, l2;
    k2.v[0] = state_key(p.v[0], (u32)q.v[0]); k2.v[1] = state_key(p.v[1], (u32)q.v[1]);
    store_stream(&reinterpret_cast<V64 *>(skey)[j], k2);
    const Lite sa = lite_unpack(p.v[0]), sb = lite_unpack(p.v[1]);
    int wa, ta, wb, tb;
    winner_arith(sa, wa, ta);
    winner_arith(sb, wb, tb);
    reinterpret_cast<uint16_t *>(winner)[j] = (uint16_t)((u32)(wa & 0xFF) | ((u32)(wb & 0xFF) << 8));
    reinterpret_cast<uint16_t *>(terminal)[j] = (uint16_t)((u32)ta | ((u32)tb << 8));
    l2.v[0] = fast_legal_mask(sa.cl); l2.v[1] = fast_legal_mask(sb.cl);
    store_stream(&reinterpret_cast<V64 *>(legal)[j], l2);
}

struct Variant {
    std::string name;
    std::function<void(hipStream_t)> launch;
    std::vector<float> us;
};

int main(int argc, char **argv) {
    const int64_t n = argc > 1 ? atoll(argv[1]) : 1 << 20;
    const int plies = argc > 2 ? atoi(argv[2]) : 5;
    const int K = argc > 3 ? atoi(argv[3]) : 50;
    const int reps = argc > 4 ? atoi(argv[4]) : 9;
    const int64_t s64 = plane_stride(n);
    u64 *state; CK(hipMalloc(&state, s64 * 16));
    CK(hipMemset(state, 0, s64 * 16));
    hipStream_t s; CK(hipStreamCreate(&s));
    Planes p = planes(state, n);
    auto fused_keys = [](u64 seed) { FusedKeys k; for (int t = 0; t < FUSED_MAX_PLIES; ++t) k.k[t] = launch_key(seed, (u32)t); return k; };
    if (plies > 0)
        hipLaunchKernelGGL((step_random_fused_kernel<256, false>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p.P, p.Q,
                           fused_keys(7), (u64)0, (uint16_t *)nullptr, (u32 *)nullptr, (uint8_t *)nullptr, (int64_t)0, n,
                           plies < FUSED_MAX_PLIES ? plies : FUSED_MAX_PLIES, (float *)nullptr);
    CK(hipStreamSynchronize(s));
    ExpOut eo;
    CK(hipMalloc(&eo.moves, n * 18)); CK(hipMalloc(&eo.n_moves, n)); CK(hipMalloc(&eo.board, n * 9));
    CK(hipMalloc(&eo.qmask, n * 8)); CK(hipMalloc(&eo.n_q, n));
    int8_t *winner; uint8_t *terminal; u64 *legal; int64_t *key;
    CK(hipMalloc(&winner, n)); CK(hipMalloc(&terminal, n)); CK(hipMalloc(&legal, n * 8)); CK(hipMalloc(&key, n * 8));
    std::vector<Variant> vs;
#define EXPV(BLK, BPL)                                                                                                   \
    vs.push_back({"export<" #BLK "," #BPL ">", [=](hipStream_t st) {                                                     \
        hipLaunchKernelGGL((export_kernel<BLK, BPL>), dim3((unsigned)((n + BLK * BPL - 1) / (BLK * BPL))), dim3(BLK), 0, st, p.P, p.Q, eo, n); }, {}}); \
    vs.push_back({"export_floor<" #BLK "," #BPL ">", [=](hipStream_t st) {                                               \
        hipLaunchKernelGGL((export_floor<BLK, BPL>), dim3((unsigned)((n + BLK * BPL - 1) / (BLK * BPL))), dim3(BLK), 0, st, p.P, p.Q, eo, n); }, {}});
    EXPV(256, 1) EXPV(512, 1) EXPV(1024, 1) EXPV(256, 2) EXPV(512, 2) EXPV(1024, 2)
    u64 *skey; CK(hipMalloc(&skey, n * 8));
#define NIV(BLK)                                                                                                         \
    vs.push_back({"node_info<" #BLK "> CPython key + native key", [=](hipStream_t st) {                                  \
        hipLaunchKernelGGL((node_info_kernel<BLK, true>), dim3((unsigned)(((n + 1) / 2 + BLK - 1) / BLK)), dim3(BLK), 0, st,   \
                           p.P, p.Q, winner, terminal, legal, key, skey, n); }, {}});                                   \
    vs.push_back({"node_info<" #BLK "> native key", [=](hipStream_t st) {                                                \
        hipLaunchKernelGGL((node_info_kernel<BLK, false>), dim3((unsigned)(((n + 1) / 2 + BLK - 1) / BLK)), dim3(BLK), 0, st,  \
                           p.P, p.Q, winner, terminal, legal, (int64_t *)nullptr, skey, n); }, {}});                    \
    vs.push_back({"node_info<" #BLK "> no key", [=](hipStream_t st) {                                                    \
        hipLaunchKernelGGL((node_info_kernel<BLK, false>), dim3((unsigned)(((n + 1) / 2 + BLK - 1) / BLK)), dim3(BLK), 0, st,  \
                           p.P, p.Q, winner, terminal, legal, (int64_t *)nullptr, (u64 *)nullptr, n); }, {}});          \
    vs.push_back({"node_info<" #BLK "> native key alone", [=](hipStream_t st) {                                          \
        hipLaunchKernelGGL((node_info_kernel<BLK, false>), dim3((unsigned)(((n + 1) / 2 + BLK - 1) / BLK)), dim3(BLK), 0, st,  \
                           p.P, p.Q, (int8_t *)nullptr, (uint8_t *)nullptr, (u64 *)nullptr, (int64_t *)nullptr, skey, n); }, {}});
    NIV(256) NIV(512) NIV(1024)
#define NIA(BLK)                                                                                                         \
    vs.push_back({"node_info_nolds<" #BLK "> native key, arithmetic line test + legal mask (experiment)", [=](hipStream_t st) { \
        hipLaunchKernelGGL((node_info_nolds_kernel<BLK>), dim3((unsigned)(((n + 1) / 2 + BLK - 1) / BLK)), dim3(BLK), 0, st,  \
                           p.P, p.Q, winner, terminal, legal, skey, n); }, {}});
    NIA(256) NIA(512) NIA(1024)
    uint8_t *act36, *nch; u64 *kid0, *kid1; int8_t *w2; uint8_t *t2; u64 *l2; int64_t *k2; u64 *sk2; int32_t *vsum;
    CK(hipMalloc(&act36, n));
    { std::vector<uint8_t> ha(n); u32 x = 12345u; for (auto &v : ha) { x = x * 1664525u + 1013904223u; v = (uint8_t)((x >> 16) % 36u); }
      CK(hipMemcpy(act36, ha.data(), n, hipMemcpyHostToDevice)); } CK(hipMalloc(&nch, n)); CK(hipMalloc(&kid0, s64 * 16)); CK(hipMalloc(&kid1, s64 * 16));
    CK(hipMalloc(&w2, 2 * n)); CK(hipMalloc(&t2, 2 * n)); CK(hipMalloc(&l2, 16 * n)); CK(hipMalloc(&k2, 16 * n));
    CK(hipMalloc(&sk2, 16 * n)); CK(hipMalloc(&vsum, 8 * n));
    Planes c0 = planes(kid0, n), c1 = planes(kid1, n);
    const ExpandOut xo_py = {nch, w2, t2, l2, k2, sk2}, xo = {nch, w2, t2, l2, nullptr, sk2}, xo_lean = {nch, w2, t2, nullptr, nullptr, sk2};
#define EXV(BLK)                                                                                                         \
    vs.push_back({"expand<" #BLK "> CPython + native keys (uniform random actions)", [=](hipStream_t st) {                \
        hipLaunchKernelGGL((expand_kernel<BLK, true>), dim3((unsigned)((n + BLK - 1) / BLK)), dim3(BLK), 0, st, p.P, p.Q, act36, \
                           c0.P, c0.Q, c1.P, c1.Q, xo_py, n); }, {}});                                                   \
    vs.push_back({"expand<" #BLK "> native keys", [=](hipStream_t st) {                                                   \
        hipLaunchKernelGGL((expand_kernel<BLK, false>), dim3((unsigned)((n + BLK - 1) / BLK)), dim3(BLK), 0, st, p.P, p.Q, act36, \
                           c0.P, c0.Q, c1.P, c1.Q, xo, n); }, {}});                                                      \
    vs.push_back({"expand<" #BLK "> native keys, no legal masks", [=](hipStream_t st) {                                   \
        hipLaunchKernelGGL((expand_kernel<BLK, false>), dim3((unsigned)((n + BLK - 1) / BLK)), dim3(BLK), 0, st, p.P, p.Q, act36, \
                           c0.P, c0.Q, c1.P, c1.Q, xo_lean, n); }, {}});
    EXV(256) EXV(512) EXV(1024)
#define XRV(SIMS)                                                                                                        \
    vs.push_back({"expand_rollout<256> " #SIMS " playouts per child, native keys", [=](hipStream_t st) {                  \
        const u32 ppb = 256u / (2u * SIMS);                                                                              \
        hipLaunchKernelGGL((expand_rollout_kernel<256, false>), dim3((unsigned)((n + ppb - 1) / ppb)), dim3(256), 0, st, p.P, p.Q, act36, \
                           c0.P, c0.Q, c1.P, c1.Q, xo, (u64)5, 0u, (u64)0, (u32)SIMS, ppb, vsum, (int8_t *)nullptr, n); }, {}});
#define XJV(SIMS, PPB)                                                                                                   \
    vs.push_back({"expand_rollout_jobs<256> " #SIMS " playouts per child, " #PPB " pairs per workgroup", [=](hipStream_t st) { \
        hipLaunchKernelGGL((expand_rollout_jobs_kernel<256, false>), dim3((unsigned)((n + PPB - 1) / PPB)), dim3(256), 0, st, p.P, p.Q, act36, \
                           c0.P, c0.Q, c1.P, c1.Q, xo, (u64)5, 0u, (u64)0, (u32)SIMS, (u32)PPB, vsum, (int8_t *)nullptr, n); }, {}});
    XRV(1) XRV(10)
    XJV(1, 64) XJV(1, 128) XJV(1, 193) XJV(1, 256) XJV(10, 32) XJV(10, 48) XJV(10, 58) XJV(10, 64) XJV(10, 128) XJV(10, 251) XJV(10, 256)
    // the playouts alone, for comparison: rollout_many on the parents (what round 3's unit did) and on child 0
    {
        int8_t *rres; CK(hipMalloc(&rres, n * 10));
        vs.push_back({"rollout_many<512> 10 playouts per PARENT", [=](hipStream_t st) {
            hipLaunchKernelGGL(rollout_many_kernel, dim3((unsigned)((n * 10 + 511) / 512)), dim3(512), 0, st, p.P, p.Q, (u64)5, 0u, (u64)0, 10u,
                               rres, (uint8_t *)nullptr, n * 10); }, {}});
        vs.push_back({"rollout_many<512> 10 playouts per child 0 (of the last expand)", [=](hipStream_t st) {
            hipLaunchKernelGGL(rollout_many_kernel, dim3((unsigned)((n * 10 + 511) / 512)), dim3(512), 0, st, c0.P, c0.Q, (u64)5, 0u, (u64)0, 10u,
                               rres, (uint8_t *)nullptr, n * 10); }, {}});
    }
    ObsOut oo;
    CK(hipMalloc(&oo.classical, n * 9)); CK(hipMalloc(&oo.q_p1, n * 10)); CK(hipMalloc(&oo.q_p1_len, n));
    CK(hipMalloc(&oo.q_p2, n * 8)); CK(hipMalloc(&oo.q_p2_len, n)); CK(hipMalloc(&oo.turn, n));
#define OBSV(BLK, WL)                                                                                                    \
    vs.push_back({"observe<" #BLK "," #WL ">", [=](hipStream_t st) {                                                     \
        hipLaunchKernelGGL((observe_kernel<BLK, WL>), dim3((unsigned)((n + 2 * BLK - 1) / (2 * BLK))), dim3(BLK), 0, st, p.P, p.Q, oo, n); }, {}});
    OBSV(256, false) OBSV(256, true) OBSV(512, false) OBSV(512, true) OBSV(1024, true)
    ExpOut only_n = {nullptr, eo.n_moves, nullptr, nullptr, nullptr};
    vs.push_back({"export<256,1> n_moves only (turn)", [=](hipStream_t st) {
        hipLaunchKernelGGL((export_kernel<256, 1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p.P, p.Q, only_n, n); }, {}});
    // fused random stepping, one against two boards per lane: T = 64 plies per launch, every ply's outputs kept
    {
        u64 *st2; CK(hipMalloc(&st2, s64 * 16)); CK(hipMemset(st2, 0, s64 * 16));
        Planes q2 = planes(st2, n);
        uint16_t *fa; u32 *fr; uint8_t *ft;
        CK(hipMalloc(&fa, (size_t)64 * n * 2)); CK(hipMalloc(&fr, (size_t)64 * n * 4)); CK(hipMalloc(&ft, (size_t)64 * n));
        const FusedKeys k3 = fused_keys(3);
        vs.push_back({"random_fused 1 board/lane <256>, 64 plies (us per LAUNCH)", [=](hipStream_t st) {
            hipLaunchKernelGGL((step_random_fused_kernel<256, true>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st,
                               q2.P, q2.Q, k3, (u64)0, fa, fr, ft, (int64_t)n, n, 64, (float *)nullptr); }, {}});
#define RF2(BLK)                                                                                                        \
        vs.push_back({"random_fused 2 boards/lane <" #BLK ">, 64 plies (us per LAUNCH)", [=](hipStream_t st) {          \
            hipLaunchKernelGGL((step_random_fused2_kernel<BLK>), dim3((unsigned)((n / 2 + BLK - 1) / BLK)), dim3(BLK), 0, st, \
                               q2.P, q2.Q, (u64)3, 0u, (u64)0, fa, fr, ft, (int64_t)n, n, 64); }, {}});
        RF2(128) RF2(256) RF2(512)
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < reps; ++r)
        for (size_t i = 0; i < vs.size(); ++i) {
            Variant &v = vs[(i + r) % vs.size()];
            const int Kv = v.name.rfind("random_fused", 0) == 0 ? std::max(1, K / 10) : K;   // a 64-ply launch is long
            for (int k = 0; k < 3; ++k) v.launch(s);
            CK(hipEventRecord(e0, s));
            for (int k = 0; k < Kv; ++k) v.launch(s);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            v.us.push_back(ms * 1e3f / Kv);
        }
    CK(hipGetLastError());
    printf("rowbench: %lld boards after %d random plies, K=%d, %d reps (us per launch: min / median)\n", (long long)n, plies, K, reps);
    for (auto &v : vs) {
        std::sort(v.us.begin(), v.us.end());
        printf("  %-40s %8.2f %8.2f\n", v.name.c_str(), v.us.front(), v.us[v.us.size() / 2]);
    }
    return 0;
}
