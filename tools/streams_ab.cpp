// tools/streams_ab.cpp — experiment, not part of the product: does splitting ONE batch of n boards
// into S independent shards on S HIP streams of the same GPU hide the kernel boundary (the ~1.1 us
// between two dependent launches on one stream) behind the other shards' kernels?
// Every shard is its own state buffer with its own board_offset (the multi-GPU shard layout of
// DESIGN.md §8, here on one device), so the results are the same as the single launch's.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude tools/streams_ab.cpp -ldl -o tools/streams_ab
//   tools/streams_ab N K REPS lib.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

// holds the streams back until all launches are queued: spins on the 100 MHz wall clock for `ticks`,
// with an iteration bound so that it always ends
__global__ void gate_kernel(unsigned long long ticks) {
    unsigned long long t0 = wall_clock64();
    for (int i = 0; i < 4000000; ++i) {
        if (wall_clock64() - t0 > ticks) break;
        __builtin_amdgcn_s_sleep(32);
    }
}

typedef int (*step_fn)(void *, const uint8_t *, const uint8_t *, uint64_t, uint32_t, int64_t, uint32_t, float *, uint8_t *, int64_t, void *);
typedef int (*sample_fn)(const void *, uint64_t, uint32_t, int64_t, uint32_t, uint8_t *, int64_t, void *);

int main(int argc, char **argv) {
    if (argc < 5) { fprintf(stderr, "usage: streams_ab N K REPS lib.so\n"); return 2; }
    int64_t n = atoll(argv[1]);
    int K = atoi(argv[2]), reps = atoi(argv[3]);
    const int W = 10, T = K + W;
    const uint64_t seed = 1;
    void *h = dlopen(argv[4], RTLD_NOW | RTLD_LOCAL);
    if (!h) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 1; }
    auto state_bytes = (int64_t (*)(int64_t))dlsym(h, "qttt_state_bytes");
    auto reset = (int (*)(void *, int64_t, void *))dlsym(h, "qttt_reset");
    auto step = (step_fn)dlsym(h, "qttt_step");
    auto sample = (sample_fn)dlsym(h, "qttt_sample_actions");
    if (!state_bytes || !reset || !step || !sample) return 1;

    const int MAXS = 8;
    hipStream_t st[MAXS];
    hipEvent_t done[MAXS], e0, e1, go;
    for (int s = 0; s < MAXS; ++s) { CK(hipStreamCreateWithFlags(&st[s], hipStreamNonBlocking)); CK(hipEventCreateWithFlags(&done[s], hipEventDisableTiming)); }
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreateWithFlags(&go, hipEventDisableTiming));

    for (int S : {1, 2, 4, 8}) {
        if (n % (S * 64)) continue;
        int64_t m = n / S;
        std::vector<void *> state(S); std::vector<uint8_t *> act(S), term(S); std::vector<float *> rew(S);
        for (int s = 0; s < S; ++s) {
            CK(hipMalloc(&state[s], state_bytes(m))); CK(hipMalloc(&act[s], (size_t)T * m * 2));
            CK(hipMalloc(&rew[s], m * 4)); CK(hipMalloc(&term[s], m));
            reset(state[s], m, st[s]);
            for (int t = 0; t < T; ++t) {                                  // record the uniform-legal action stream
                sample(state[s], seed, t, s * m, 1, act[s] + (size_t)t * 2 * m, m, st[s]);
                step(state[s], act[s] + (size_t)t * 2 * m, nullptr, seed, t, s * m, 1, rew[s], term[s], m, st[s]);
            }
            CK(hipStreamSynchronize(st[s]));
        }
        std::vector<float> us;
        for (int r = 0; r < reps; ++r) {
            for (int s = 0; s < S; ++s) reset(state[s], m, st[s]);
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(gate_kernel, dim3(1), dim3(64), 0, st[0], 400000ull);     // 4 ms
            CK(hipEventRecord(go, st[0]));
            for (int s = 1; s < S; ++s) CK(hipStreamWaitEvent(st[s], go, 0));
            for (int t = 0; t < W; ++t)
                for (int s = 0; s < S; ++s)
                    step(state[s], act[s] + (size_t)t * 2 * m, nullptr, seed, t, s * m, 1, rew[s], term[s], m, st[s]);
            // all shards meet, the clock starts on stream 0, all shards wait for it
            for (int s = 1; s < S; ++s) { CK(hipEventRecord(done[s], st[s])); CK(hipStreamWaitEvent(st[0], done[s], 0)); }
            CK(hipEventRecord(e0, st[0]));
            for (int s = 1; s < S; ++s) CK(hipStreamWaitEvent(st[s], e0, 0));
            for (int t = W; t < T; ++t)
                for (int s = 0; s < S; ++s)
                    step(state[s], act[s] + (size_t)t * 2 * m, nullptr, seed, t, s * m, 1, rew[s], term[s], m, st[s]);
            for (int s = 1; s < S; ++s) { CK(hipEventRecord(done[s], st[s])); CK(hipStreamWaitEvent(st[0], done[s], 0)); }
            CK(hipEventRecord(e1, st[0]));
            CK(hipDeviceSynchronize());
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            us.push_back(ms * 1e3f / K);
        }
        std::sort(us.begin(), us.end());
        printf("{\"boards\": %lld, \"shards_on_streams\": %d, \"K\": %d, \"reps\": %d, \"us_per_step_min\": %.3f, \"us_per_step_median\": %.3f, "
               "\"us_per_1M_boards_median\": %.3f}\n", (long long)n, S, K, reps, us.front(), us[us.size() / 2],
               us[us.size() / 2] * 1048576.0 / n);
        fflush(stdout);
        for (int s = 0; s < S; ++s) { CK(hipFree(state[s])); CK(hipFree(act[s])); CK(hipFree(rew[s])); CK(hipFree(term[s])); }
    }
    return 0;
}
