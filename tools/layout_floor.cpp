// tools/layout_floor.cpp — experiment, not part of the product: the traffic floor (the step's loads and
// stores, no game logic) under different layouts of the 16-byte state, to see whether fewer / wider
// memory streams than the two u64 planes would move the 39 B per board any faster.
//   A  SoA planes P[n], Q[n]; a lane owns boards 2j, 2j+1 (one 16-byte access per plane)   = the product
//   B  AoS 16 B per board;   a lane owns boards 2j, 2j+1 (32 contiguous bytes, two 16-byte accesses)
//   C  AoS 16 B per board;   a lane owns boards j and j+64 of its wave's 128 (every access 1 KB per wave)
//   hipcc --offload-arch=gfx950 -O3 tools/layout_floor.cpp -o tools/layout_floor ; tools/layout_floor N K REPS
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned long long u64;
typedef unsigned int u32;
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef u32 u32x2 __attribute__((ext_vector_type(2)));

#define BLK 1024
__device__ __forceinline__ u32x4 ld4(const void *p) { return __builtin_nontemporal_load((const u32x4 *)p); }
__device__ __forceinline__ void st4(void *p, u32x4 v) { __builtin_nontemporal_store(v, (u32x4 *)p); }

__global__ __launch_bounds__(BLK) void floor_A(u64 *pP, u64 *pQ, const uint16_t *act, u32 *rew, uint8_t *term, int64_t ng) {
    int64_t j = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (j >= ng) return;
    u32x4 p = ld4(pP + 2 * j), q = ld4(pQ + 2 * j);
    u32 a = __builtin_nontemporal_load((const u32 *)act + j);
    p.x ^= a; q.y += 1; p.z ^= a >> 16; q.w += 1;
    st4(pP + 2 * j, p); st4(pQ + 2 * j, q);
    u32x2 r = {p.x, p.z};
    __builtin_nontemporal_store(r, (u32x2 *)rew + j);
    __builtin_nontemporal_store((uint16_t)(q.y | q.w << 8), (uint16_t *)term + j);
}
__global__ __launch_bounds__(BLK) void floor_B(u64 *st, const uint16_t *act, u32 *rew, uint8_t *term, int64_t ng) {
    int64_t j = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (j >= ng) return;
    u32x4 b0 = ld4(st + 4 * j), b1 = ld4(st + 4 * j + 2);
    u32 a = __builtin_nontemporal_load((const u32 *)act + j);
    b0.x ^= a; b0.w += 1; b1.x ^= a >> 16; b1.w += 1;
    st4(st + 4 * j, b0); st4(st + 4 * j + 2, b1);
    u32x2 r = {b0.x, b1.x};
    __builtin_nontemporal_store(r, (u32x2 *)rew + j);
    __builtin_nontemporal_store((uint16_t)(b0.w | b1.w << 8), (uint16_t *)term + j);
}
__global__ __launch_bounds__(BLK) void floor_C(u64 *st, const uint16_t *act, u32 *rew, uint8_t *term, int64_t ng) {
    int64_t j = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (j >= ng) return;
    const int64_t i0 = (j & ~63ll) * 2 + (j & 63), i1 = i0 + 64;
    u32x4 b0 = ld4(st + 2 * i0), b1 = ld4(st + 2 * i1);
    u32 a0 = __builtin_nontemporal_load(act + i0), a1 = __builtin_nontemporal_load(act + i1);
    b0.x ^= a0; b0.w += 1; b1.x ^= a1; b1.w += 1;
    st4(st + 2 * i0, b0); st4(st + 2 * i1, b1);
    __builtin_nontemporal_store(b0.x, rew + i0); __builtin_nontemporal_store(b1.x, rew + i1);
    __builtin_nontemporal_store((uint8_t)b0.w, term + i0); __builtin_nontemporal_store((uint8_t)b1.w, term + i1);
}

int main(int argc, char **argv) {
    int64_t n = argc > 1 ? atoll(argv[1]) : 1048576;
    int K = argc > 2 ? atoi(argv[2]) : 100, reps = argc > 3 ? atoi(argv[3]) : 15;
    u64 *st; uint16_t *act; u32 *rew; uint8_t *term;
    CK(hipMalloc(&st, n * 16)); CK(hipMemset(st, 0, n * 16));
    CK(hipMalloc(&act, (size_t)K * n * 2)); CK(hipMemset(act, 1, (size_t)K * n * 2));
    CK(hipMalloc(&rew, n * 4)); CK(hipMalloc(&term, n));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int64_t ng = n / 2;
    dim3 g((unsigned)((ng + BLK - 1) / BLK)), b(BLK);
    std::vector<float> us[3];
    for (int r = 0; r < reps; ++r)
        for (int vi = 0; vi < 3; ++vi) {
            const int v = (vi + r) % 3;                              // rotated order
            for (int pass = 0; pass < 2; ++pass) {                   // pass 0 = warm-up in front of the timed launches
                if (pass) CK(hipEventRecord(e0, s));
                for (int t = 0; t < (pass ? K : 10); ++t) {
                    const uint16_t *a = act + (size_t)t * n;
                    if (v == 0) hipLaunchKernelGGL(floor_A, g, b, 0, s, st, st + n, a, rew, term, ng);
                    if (v == 1) hipLaunchKernelGGL(floor_B, g, b, 0, s, st, a, rew, term, ng);
                    if (v == 2) hipLaunchKernelGGL(floor_C, g, b, 0, s, st, a, rew, term, ng);
                }
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            us[v].push_back(ms * 1e3f / K);
        }
    const char *names[3] = {"A SoA planes, lane = boards 2j,2j+1", "B AoS, lane = boards 2j,2j+1 (32 B contiguous)", "C AoS, lane = boards j,j+64 of the wave"};
    for (int v = 0; v < 3; ++v) {
        std::sort(us[v].begin(), us[v].end());
        printf("{\"layout\": \"%s\", \"boards\": %lld, \"us_min\": %.3f, \"us_median\": %.3f, \"GBps_median\": %.0f}\n", names[v],
               (long long)n, us[v].front(), us[v][us[v].size() / 2], 39.0 * n / us[v][us[v].size() / 2] * 1e-3);
    }
    return 0;
}
