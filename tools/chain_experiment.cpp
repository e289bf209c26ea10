// chain_experiment.cpp — can the ~1.5 us kernel boundary of a launch-bound step (262 144 boards: 3.5 us per launch for 1.6 us
// of traffic) be overlapped by issuing consecutive DEPENDENT launches on alternating streams, with the dependency carried
// per workgroup through a flag in device memory instead of by the queue's barrier?  (hipExtAnyOrderLaunch is not supported
// on gfx9; two hardware queues are.)  Workgroup w of step t+1 needs only what workgroup w of step t stored, so it spins on
// flag[w] == t+1 (bounded by s_memrealtime: a time-out sets an error word and the wave leaves).  Only legal while both
// launches fit on the chip at once (2 x grid <= resident workgroups): a spinning workgroup must never hold the slot its
// producer needs.
//   hipcc -O3 --offload-arch=gfx950 tools/chain_experiment.cpp -o tools/chain_experiment
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef uint64_t u64;
typedef uint32_t u32;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// the step's traffic: 16 B of state read and written, 2 B of action read, 4 + 1 B written; a little dependent arithmetic so
// that a stale read changes the result
template <int BLK, bool CHAIN>
__global__ __launch_bounds__(BLK) void chain_kernel(u64 *pP, u64 *pQ, const uint16_t *actions, u32 *reward, uint8_t *term,
                                                     u32 *flags, u32 step, u32 *err, int64_t n, u64 timeout_ticks) {
    __shared__ u32 lut[64];
    const int64_t i = (int64_t)blockIdx.x * BLK + threadIdx.x;
    const u32 a = i < n ? (u32)__builtin_nontemporal_load(&actions[i]) : 0u;      // independent of the previous step
    if (threadIdx.x < 64) lut[threadIdx.x] = threadIdx.x * 0x9E3779B9u;
    if (CHAIN && step > 0u) {
        if (threadIdx.x == 0) {
            const u64 t0 = __builtin_amdgcn_s_memrealtime();
            while (__hip_atomic_load(&flags[blockIdx.x], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != step) {
                if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) { atomicAdd(err, 1u); break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
    }
    __syncthreads();
    if (CHAIN && step > 0u) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (i < n) {
        u64 P = __builtin_nontemporal_load(&pP[i]), Q = __builtin_nontemporal_load(&pQ[i]);
        P = P * 6364136223846793005ull + a + lut[(u32)Q & 63u];
        Q ^= (P >> 17) | (P << 47);
        __builtin_nontemporal_store(P, &pP[i]);
        __builtin_nontemporal_store(Q, &pQ[i]);
        __builtin_nontemporal_store((u32)(P >> 32), &reward[i]);
        __builtin_nontemporal_store((uint8_t)(Q & 1u), &term[i]);
    }
    if (CHAIN) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(&flags[blockIdx.x], step + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
}

int main(int argc, char **argv) {
    const int K = argc > 1 ? atoi(argv[1]) : 200;
    const int REPS = 9;
    const int64_t sizes[] = {4096, 65536, 131072, 262144, 524288};
    for (int64_t n : sizes) {
        constexpr int BLK = 256;
        const int grid = (int)((n + BLK - 1) / BLK);
        u64 *pP, *pQ; uint16_t *act; u32 *rew, *flags, *err; uint8_t *term;
        CK(hipMalloc(&pP, n * 8)); CK(hipMalloc(&pQ, n * 8)); CK(hipMalloc(&act, (size_t)K * n * 2));
        CK(hipMalloc(&rew, n * 4)); CK(hipMalloc(&term, n)); CK(hipMalloc(&flags, grid * 4)); CK(hipMalloc(&err, 4));
        std::vector<uint16_t> ha((size_t)K * n);
        for (size_t j = 0; j < ha.size(); ++j) ha[j] = (uint16_t)(j * 2654435761u >> 13);
        CK(hipMemcpy(act, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
        hipStream_t s[4];
        for (auto &x : s) CK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
        std::vector<u64> ref(n), got(n);
        auto reset = [&]() { CK(hipMemset(pP, 0, n * 8)); CK(hipMemset(pQ, 0, n * 8)); CK(hipMemset(flags, 0, grid * 4)); CK(hipMemset(err, 0, 4)); CK(hipDeviceSynchronize()); };
        auto run = [&](int nstreams, bool chain) {
            std::vector<double> us;
            for (int r = 0; r < REPS; ++r) {
                reset();
                const auto t0 = std::chrono::steady_clock::now();
                for (int t = 0; t < K; ++t) {
                    hipStream_t q = s[t % nstreams];
                    if (chain) hipLaunchKernelGGL((chain_kernel<BLK, true>), dim3(grid), dim3(BLK), 0, q, pP, pQ, act + (size_t)t * n, rew, term, flags, (u32)t, err, n, (u64)2000000);
                    else hipLaunchKernelGGL((chain_kernel<BLK, false>), dim3(grid), dim3(BLK), 0, q, pP, pQ, act + (size_t)t * n, rew, term, flags, (u32)t, err, n, (u64)0);
                }
                CK(hipDeviceSynchronize());
                us.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / K);
            }
            std::sort(us.begin(), us.end());
            return std::make_pair(us[us.size() / 2], us.front());
        };
        auto base = run(1, false);
        CK(hipMemcpy(ref.data(), pP, n * 8, hipMemcpyDeviceToHost));
        printf("boards %7lld  grid %5d  one stream, queue barrier   : med %6.2f best %6.2f us per launch (K=%d, host wall)\n", (long long)n, grid, base.first, base.second, K);
        auto c1 = run(1, true);
        printf("boards %7lld              one stream + flags (cost)   : med %6.2f best %6.2f\n", (long long)n, c1.first, c1.second);
        for (int ns : {2, 3, 4}) {
            if ((int64_t)grid * ns > 2048) { printf("boards %7lld              %d streams: skipped (%d x %d workgroups do not fit on the chip at once)\n", (long long)n, ns, ns, grid); continue; }
            auto c = run(ns, true);
            u32 herr = 0;
            CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(got.data(), pP, n * 8, hipMemcpyDeviceToHost));
            printf("boards %7lld              %d streams, per-WG flags     : med %6.2f best %6.2f  time-outs %u  result %s\n", (long long)n, ns, c.first, c.second, herr,
                   got == ref ? "== the single-stream run" : "DIFFERS");
        }
        fflush(stdout);
        for (auto &x : s) CK(hipStreamDestroy(x));
        CK(hipFree(pP)); CK(hipFree(pQ)); CK(hipFree(act)); CK(hipFree(rew)); CK(hipFree(term)); CK(hipFree(flags)); CK(hipFree(err));
    }
    return 0;
}
