// tools/stepbench.cpp — C++ A/B timing harness for kernel work (not part of the product, not the
// judged bench).  Loads one or more builds of libqttt_hip.so with dlopen, drives each through the
// C ABI exactly like bench.py (record the uniform-legal action stream, reset, replay under
// hipEvents), and times the variants INTERLEAVED in one process on one device, so that clock and
// device differences cancel.  Also times a "traffic floor" kernel that moves the same bytes per
// board with no game logic.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude tools/stepbench.cpp -ldl -o tools/stepbench
//   tools/stepbench N K REPS  lib.so[:boards_per_lane[:workgroup_size]] ...   (0 = the library's own choice)
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <algorithm>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

typedef unsigned long long u64;
typedef unsigned int u32;

template <typename T, int N> struct alignas(sizeof(T) * N) Vec { T v[N]; };

typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
template <int B> struct RawOf;
template <> struct RawOf<1> { typedef uint8_t type; };
template <> struct RawOf<2> { typedef uint16_t type; };
template <> struct RawOf<4> { typedef u32 type; };
template <> struct RawOf<8> { typedef u32x2 type; };
template <> struct RawOf<16> { typedef u32x4 type; };
typedef u32 u32x8 __attribute__((ext_vector_type(8)));
template <> struct RawOf<32> { typedef u32x8 type; };
template <typename V> __device__ __forceinline__ V ld_nt(const V *p) {
    typedef typename RawOf<sizeof(V)>::type R;
    R r = __builtin_nontemporal_load(reinterpret_cast<const R *>(p));
    V v; __builtin_memcpy(&v, &r, sizeof(V)); return v;
}
template <typename V> __device__ __forceinline__ void st_nt(V *p, const V &v) {
    typedef typename RawOf<sizeof(V)>::type R;
    R r; __builtin_memcpy(&r, &v, sizeof(V));
    __builtin_nontemporal_store(r, reinterpret_cast<R *>(p));
}

// same loads and stores as the step kernel (non-temporal, like the product), trivial arithmetic
template <int BPL>
__global__ __launch_bounds__(256) void floor_kernel(u64 *pA, u64 *pB, u32 *pC, const uint16_t *actions,
                                                    u32 *reward, uint8_t *term, int64_t n_groups) {
    int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_groups) return;
    int64_t i0 = j * BPL;
    Vec<u64, BPL> a = ld_nt(reinterpret_cast<const Vec<u64, BPL> *>(pA + i0));
    Vec<u64, BPL> b = ld_nt(reinterpret_cast<const Vec<u64, BPL> *>(pB + i0));
    Vec<u32, BPL> c = ld_nt(reinterpret_cast<const Vec<u32, BPL> *>(pC + i0));
    Vec<uint16_t, BPL> act = ld_nt(reinterpret_cast<const Vec<uint16_t, BPL> *>(actions + i0));
    Vec<u32, BPL> rw; Vec<uint8_t, BPL> tm;
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        a.v[k] ^= act.v[k]; b.v[k] += 1; c.v[k] ^= 1u;
        rw.v[k] = (u32)a.v[k]; tm.v[k] = (uint8_t)b.v[k];
    }
    st_nt(reinterpret_cast<Vec<u64, BPL> *>(pA + i0), a);
    st_nt(reinterpret_cast<Vec<u64, BPL> *>(pB + i0), b);
    st_nt(reinterpret_cast<Vec<u32, BPL> *>(pC + i0), c);
    st_nt(reinterpret_cast<Vec<u32, BPL> *>(reward + i0), rw);
    st_nt(reinterpret_cast<Vec<uint8_t, BPL> *>(term + i0), tm);
}

// the same with a 16-byte state (planes A and B only): what a 39 B/step layout would reach
// PLAIN_LD: the loads without the non-temporal hint (the stores keep it)
template <int BPL, int BLK = 256, bool PLAIN_LD = false>
__global__ __launch_bounds__(BLK) void floor16_kernel(u64 *pA, u64 *pB, const uint16_t *actions,
                                                      u32 *reward, uint8_t *term, int64_t n_groups) {
    int64_t j = (int64_t)blockIdx.x * BLK + threadIdx.x;
    if (j >= n_groups) return;
    int64_t i0 = j * BPL;
    Vec<u64, BPL> a, b;
    Vec<uint16_t, BPL> act;
    if (PLAIN_LD) {
        a = *reinterpret_cast<const Vec<u64, BPL> *>(pA + i0);
        b = *reinterpret_cast<const Vec<u64, BPL> *>(pB + i0);
        act = *reinterpret_cast<const Vec<uint16_t, BPL> *>(actions + i0);
    } else {
        a = ld_nt(reinterpret_cast<const Vec<u64, BPL> *>(pA + i0));
        b = ld_nt(reinterpret_cast<const Vec<u64, BPL> *>(pB + i0));
        act = ld_nt(reinterpret_cast<const Vec<uint16_t, BPL> *>(actions + i0));
    }
    Vec<u32, BPL> rw; Vec<uint8_t, BPL> tm;
#pragma unroll
    for (int k = 0; k < BPL; ++k) {
        a.v[k] ^= act.v[k]; b.v[k] += 1;
        rw.v[k] = (u32)a.v[k]; tm.v[k] = (uint8_t)b.v[k];
    }
    st_nt(reinterpret_cast<Vec<u64, BPL> *>(pA + i0), a);
    st_nt(reinterpret_cast<Vec<u64, BPL> *>(pB + i0), b);
    st_nt(reinterpret_cast<Vec<u32, BPL> *>(reward + i0), rw);
    st_nt(reinterpret_cast<Vec<uint8_t, BPL> *>(term + i0), tm);
}

struct Lib {
    std::string spec, path;
    int bpl, pipe;
    void *h;
    int64_t (*state_bytes)(int64_t);
    int (*reset)(void *, int64_t, void *);
    int (*step)(void *, const uint8_t *, const uint8_t *, uint64_t, uint32_t, int64_t, uint32_t, float *, uint8_t *, int64_t, void *);
    int (*step_many)(void *, const uint8_t *, const uint8_t *, uint64_t, uint32_t, int64_t, uint32_t, float *, uint8_t *, int64_t, int64_t, int32_t, void *);
    int (*sample)(const void *, uint64_t, uint32_t, int64_t, uint32_t, uint8_t *, int64_t, void *);
    int (*set_tuning)(int, int);
    int (*step_wpb)(void *, const uint8_t *, const uint8_t *, uint64_t, uint32_t, int64_t, uint32_t, float *, uint8_t *, int64_t, void *);
    int (*step_obs)(void *, const uint8_t *, const uint8_t *, uint64_t, uint32_t, int64_t, uint32_t, float *, uint8_t *, int8_t *,
                    uint8_t *, uint8_t *, uint8_t *, uint8_t *, uint8_t *, int64_t, void *);
    int (*step_random)(void *, uint64_t, uint32_t, int64_t, uint32_t, uint8_t *, float *, uint8_t *, int64_t, void *);
    std::vector<float> us;
};

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: stepbench N K REPS lib.so:bpl:pipe ...\n"); return 2; }
    int64_t n = atoll(argv[1]);
    int K = atoi(argv[2]), reps = atoi(argv[3]);
    int W = 20, T = K + W;
    const uint64_t seed = 1;
    std::vector<Lib> libs;
    for (int i = 4; i < argc; ++i) {
        Lib L; L.spec = argv[i];
        char path[512]; int bpl = 0, pipe = 0;      // lib.so:boards_per_lane:workgroup_size, 0 = the library's choice
        if (sscanf(argv[i], "%511[^:]:%d:%d", path, &bpl, &pipe) < 1) return 2;
        L.path = path; L.bpl = bpl; L.pipe = pipe;
        L.h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!L.h) { fprintf(stderr, "dlopen %s: %s\n", path, dlerror()); return 1; }
#define SYM(f, name) *(void **)(&L.f) = dlsym(L.h, name); if (!L.f) { fprintf(stderr, "missing %s\n", name); return 1; }
        SYM(state_bytes, "qttt_state_bytes") SYM(reset, "qttt_reset") SYM(step, "qttt_step")
        SYM(step_many, "qttt_step_many") SYM(sample, "qttt_sample_actions") SYM(set_tuning, "qttt_set_tuning")
        L.step_wpb = nullptr;
        if (libs.empty()) {                                                  // the mapping study lives in its own library
            if (void *hs = dlopen("tools/libqttt_study.so", RTLD_NOW | RTLD_LOCAL))
                *(void **)(&L.step_wpb) = dlsym(hs, "qttt_step_wave_per_board");
        }
        *(void **)(&L.step_obs) = dlsym(L.h, "qttt_step_observe");          // optional (STEPBENCH_OBS=1 times it)
        *(void **)(&L.step_random) = dlsym(L.h, "qttt_step_random");        // optional (STEPBENCH_RANDOM=1 times it)
        libs.push_back(L);
    }
    void *state; uint8_t *actions, *term; float *reward;
    CK(hipMalloc(&state, 20 * ((n + 63) & ~63ll)));               // room for either layout (16 or 20 B/board)
    CK(hipMalloc(&actions, (size_t)T * n * 2));
    CK(hipMalloc(&reward, n * 4));
    CK(hipMalloc(&term, n));
    hipStream_t s; CK(hipStreamCreate(&s));
    // STEPBENCH_OBS=1: time qttt_step_observe (step + observation, one kernel) instead of qttt_step
    const bool obs_mode = getenv("STEPBENCH_OBS") != nullptr;
    // STEPBENCH_RANDOM=1: time qttt_step_random (policy + step in one kernel, the actions played written out)
    const bool random_mode = getenv("STEPBENCH_RANDOM") != nullptr;
    int8_t *o_cl; uint8_t *o_p1, *o_l1, *o_p2, *o_l2, *o_tn;
    CK(hipMalloc(&o_cl, n * 9)); CK(hipMalloc(&o_p1, n * 10)); CK(hipMalloc(&o_l1, n)); CK(hipMalloc(&o_p2, n * 8));
    CK(hipMalloc(&o_l2, n)); CK(hipMalloc(&o_tn, n));
    Lib &L0 = libs[0];
    L0.set_tuning(L0.bpl, L0.pipe);
    L0.reset(state, n, s);
    for (int t = 0; t < T; ++t) {
        L0.sample(state, seed, t, 0, 1, actions + (size_t)t * 2 * n, n, s);
        L0.step(state, actions + (size_t)t * 2 * n, nullptr, seed, t, 0, 1, reward, term, n, s);
    }
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    u64 *pA = (u64 *)state; int64_t stride = (n + 63) & ~63ll; u64 *pB = pA + stride; u32 *pC = (u32 *)(pB + stride);
    std::vector<float> fl[3], fl16, fl16big, fl16plain;
    for (int r = 0; r < reps; ++r) {
        for (size_t li = 0; li < libs.size(); ++li) {
            Lib &L = libs[(li + (size_t)r) % libs.size()];      // rotate the order: the slot after the floor kernels is slower
            L.set_tuning(L.bpl, L.pipe);
            L.reset(state, n, s);
            L.step_many(state, actions, nullptr, seed, 0, 0, 1, reward, term, 0, n, W, s);
            CK(hipEventRecord(e0, s));
            int rc = 0;
            if (obs_mode && L.step_obs) {
                for (int t = 0; t < K && !rc; ++t)
                    rc = L.step_obs(state, actions + (size_t)(W + t) * 2 * n, nullptr, seed, W + t, 0, 1, reward, term, o_cl, o_p1,
                                    o_l1, o_p2, o_l2, o_tn, n, s);
            } else if (random_mode && L.step_random) {
                for (int t = 0; t < K && !rc; ++t)
                    rc = L.step_random(state, seed, W + t, 0, 1, actions + (size_t)(W + t) * 2 * n, reward, term, n, s);
            } else {
                rc = L.step_many(state, actions + (size_t)W * 2 * n, nullptr, seed, W, 0, 1, reward, term, 0, n, K, s);
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            if (rc) { fprintf(stderr, "step rc=%d\n", rc); return 1; }
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            L.us.push_back(ms * 1e3f / K);
        }
        int vi = 0;
        for (int bplv : {1, 2, 4}) {
            CK(hipEventRecord(e0, s));
            for (int t = 0; t < K; ++t) {
                const uint16_t *a16 = (const uint16_t *)(actions + (size_t)(W + t) * 2 * n);
                int64_t ng = n / bplv; dim3 g((unsigned)((ng + 255) / 256)), b(256);
                if (bplv == 1) hipLaunchKernelGGL(floor_kernel<1>, g, b, 0, s, pA, pB, pC, a16, (u32 *)reward, term, ng);
                if (bplv == 2) hipLaunchKernelGGL(floor_kernel<2>, g, b, 0, s, pA, pB, pC, a16, (u32 *)reward, term, ng);
                if (bplv == 4) hipLaunchKernelGGL(floor_kernel<4>, g, b, 0, s, pA, pB, pC, a16, (u32 *)reward, term, ng);
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            fl[vi++].push_back(ms * 1e3f / K);
        }
        {
            CK(hipEventRecord(e0, s));
            for (int t = 0; t < K; ++t) {
                const uint16_t *a16 = (const uint16_t *)(actions + (size_t)(W + t) * 2 * n);
                int64_t ng = n / 2; dim3 g((unsigned)((ng + 255) / 256)), b(256);
                hipLaunchKernelGGL(floor16_kernel<2>, g, b, 0, s, pA, pB, a16, (u32 *)reward, term, ng);
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            fl16.push_back(ms * 1e3f / K);
        }
        {   // the same with 1024-thread workgroups (what the step kernel uses for chip-filling batches)
            CK(hipEventRecord(e0, s));
            for (int t = 0; t < K; ++t) {
                const uint16_t *a16 = (const uint16_t *)(actions + (size_t)(W + t) * 2 * n);
                int64_t ng = n / 2; dim3 g((unsigned)((ng + 1023) / 1024)), b(1024);
                hipLaunchKernelGGL((floor16_kernel<2, 1024>), g, b, 0, s, pA, pB, a16, (u32 *)reward, term, ng);
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            fl16big.push_back(ms * 1e3f / K);
        }
        {   // and with plain (temporal) loads
            CK(hipEventRecord(e0, s));
            for (int t = 0; t < K; ++t) {
                const uint16_t *a16 = (const uint16_t *)(actions + (size_t)(W + t) * 2 * n);
                int64_t ng = n / 2; dim3 g((unsigned)((ng + 1023) / 1024)), b(1024);
                hipLaunchKernelGGL((floor16_kernel<2, 1024, true>), g, b, 0, s, pA, pB, a16, (u32 *)reward, term, ng);
            }
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            fl16plain.push_back(ms * 1e3f / K);
        }
    }
    // optional: per-wave timeline of one launch from a -DQTTT_DEBUG_STAMPS build (last lib)
    {
        Lib &L = libs.back();
        int (*set_stamps)(void *) = nullptr;
        *(void **)(&set_stamps) = dlsym(L.h, "qttt_debug_set_stamps");
        if (set_stamps) {
            int eff_bpl = L.bpl, eff_blk = 0;
            int (*shape)(int64_t, uint32_t, int, int *, int *) = nullptr;   // ABI v3 signature
            *(void **)(&shape) = dlsym(L.h, "qttt_step_launch_shape");
            L.set_tuning(L.bpl, L.pipe);
            if (shape) shape(n, 0u, 0, &eff_bpl, &eff_blk);
            if (!eff_bpl) eff_bpl = 2;
            int64_t n_waves = (n / eff_bpl + 63) / 64;
            u64 *dbuf; CK(hipMalloc(&dbuf, n_waves * 32)); CK(hipMemset(dbuf, 0, n_waves * 32));
            L.reset(state, n, s);
            L.step_many(state, actions, nullptr, seed, 0, 0, 1, reward, term, 0, n, W + 5, s);
            CK(hipStreamSynchronize(s));
            set_stamps(dbuf);
            L.step(state, actions + (size_t)(W + 5) * 2 * n, nullptr, seed, W + 5, 0, 1, reward, term, n, s);
            CK(hipStreamSynchronize(s));
            set_stamps(nullptr);
            std::vector<u64> h(n_waves * 4);
            CK(hipMemcpy(h.data(), dbuf, n_waves * 32, hipMemcpyDeviceToHost));
            u64 tmin = ~0ull, tmax = 0;
            for (int64_t w = 0; w < n_waves; ++w) { tmin = std::min(tmin, h[w * 4]); tmax = std::max(tmax, h[w * 4 + 3]); }
            printf("stamps: %lld waves, kernel span %.2f us (100 MHz realtime ticks)\n", (long long)n_waves, (tmax - tmin) * 0.01);
            const char *names[4] = {"start", "loads_done", "compute_done", "stores_done"};
            for (int k = 0; k < 4; ++k) {
                std::vector<double> v(n_waves);
                for (int64_t w = 0; w < n_waves; ++w) v[w] = (h[w * 4 + k] - tmin) * 0.01;
                std::sort(v.begin(), v.end());
                printf("  %-13s us since first start: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f\n", names[k], v[0],
                       v[n_waves / 10], v[n_waves / 2], v[n_waves * 9 / 10], v[n_waves - 1]);
            }
            const char *dn[3] = {"load_wait", "compute", "store_wait"};
            for (int k = 0; k < 3; ++k) {
                std::vector<double> v(n_waves);
                for (int64_t w = 0; w < n_waves; ++w) v[w] = (h[w * 4 + k + 1] - h[w * 4 + k]) * 0.01;
                std::sort(v.begin(), v.end());
                printf("  %-13s duration us: min %.2f p10 %.2f p50 %.2f p90 %.2f max %.2f\n", dn[k], v[0], v[n_waves / 10],
                       v[n_waves / 2], v[n_waves * 9 / 10], v[n_waves - 1]);
            }
        }
    }
    // mapping study: one wavefront per board (a few launches are enough: it is ~50x slower)
    if (libs[0].step_wpb) {
        Lib &L = libs[0];
        L.reset(state, n, s);
        const int KW = 8;
        L.step_wpb(state, actions, nullptr, seed, 0, 0, 1, reward, term, n, s);
        CK(hipEventRecord(e0, s));
        for (int t = 1; t <= KW; ++t)
            L.step_wpb(state, actions + (size_t)t * 2 * n, nullptr, seed, t, 0, 1, reward, term, n, s);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("wave-per-board study kernel (lane 0 of each wave) %14s us/launch %8.2f  %6.2f Gsteps/s\n", "",
               ms * 1e3 / KW, n / (ms * 1e3 / KW) * 1e-3);
    }
    double bytes = 47.0 * n;
    for (auto &L : libs) {
        const int64_t sb_per_board = L.state_bytes(64) / 64;      // 16 or 20, by build
        const double lib_bytes = (2.0 * sb_per_board + 7.0) * n;  // algorithmic bytes of the library's layout
        if (getenv("STEPBENCH_ALL")) {                              // every rep in time order (is the spread a drift or spikes?)
            printf("reps  %-44s", L.spec.c_str());
            for (float v : L.us) printf(" %.2f", v);
            printf("\n");
        }
        std::sort(L.us.begin(), L.us.end());
        printf("step  %-44s us/launch min %6.2f med %6.2f  %6.1f Gsteps/s %5.0f GB/s (%d B state)\n", L.spec.c_str(), L.us.front(),
               L.us[L.us.size() / 2], n / L.us.front() * 1e-3, lib_bytes / L.us.front() * 1e-3, (int)sb_per_board);
    }
    int vi = 0;
    for (int bplv : {1, 2, 4}) {
        std::sort(fl[vi].begin(), fl[vi].end());
        printf("floor bpl=%d %38s us/launch min %6.2f med %6.2f  %19s %5.0f GB/s\n", bplv, "", fl[vi].front(),
               fl[vi][fl[vi].size() / 2], "", bytes / fl[vi].front() * 1e-3);
        ++vi;
    }
    std::sort(fl16.begin(), fl16.end());
    printf("floor 16-byte state, bpl=2 %23s us/launch min %6.2f med %6.2f  %19s %5.0f GB/s (39 B/step)\n", "",
           fl16.front(), fl16[fl16.size() / 2], "", 39.0 * n / fl16.front() * 1e-3);
    std::sort(fl16big.begin(), fl16big.end());
    printf("floor 16-byte state, bpl=2, 1024-thread WG %7s us/launch min %6.2f med %6.2f  %19s %5.0f GB/s (39 B/step)\n", "",
           fl16big.front(), fl16big[fl16big.size() / 2], "", 39.0 * n / fl16big.front() * 1e-3);
    std::sort(fl16plain.begin(), fl16plain.end());
    printf("floor 16-byte state, bpl=2, 1024-thread WG, plain loads us/launch min %6.2f med %6.2f  %19s %5.0f GB/s (39 B/step)\n",
           fl16plain.front(), fl16plain[fl16plain.size() / 2], "", 39.0 * n / fl16plain.front() * 1e-3);
    return 0;
}
