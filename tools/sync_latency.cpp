// tools/sync_latency.cpp — what a single-record Board façade call costs below Python (diagnostic, not product):
// one 64-byte record in pinned host memory through qttt_board_op, completion detected (a) by hipStreamSynchronize
// (what qttt_board_op_sync does) and (b) by spinning on a byte of the out record the kernel writes last.
//   hipcc --offload-arch=gfx950 -O3 -Iinclude tools/sync_latency.cpp -ldl -o tools/sync_latency && tools/sync_latency qtttgym_amd/libqttt_hip.so
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
    void *h = dlopen(argc > 1 ? argv[1] : "qtttgym_amd/libqttt_hip.so", RTLD_NOW);
    if (!h) { fprintf(stderr, "%s\n", dlerror()); return 1; }
    int (*op)(const void *, void *, int64_t, void *) = (int (*)(const void *, void *, int64_t, void *))dlsym(h, "qttt_board_op");
    int (*op_sync)(const void *, void *, int64_t, void *) = (int (*)(const void *, void *, int64_t, void *))dlsym(h, "qttt_board_op_sync");
    int (*op_host)(const void *, void *, int64_t, void *) = (int (*)(const void *, void *, int64_t, void *))dlsym(h, "qttt_board_op_host");
    uint8_t *in, *out;
    CK(hipHostMalloc(&in, 64 * 64, hipHostMallocDefault)); CK(hipHostMalloc(&out, 64 * 64, hipHostMallocDefault));
    memset(in, 0, 64 * 64); memset(in, 0xFF, 18); in[19 + 0] = 0xFF; for (int v = 0; v < 9; ++v) in[19 + v] = 0xFF;
    in[38] = 0; in[39] = 1;                                       // the move (0, 1) on an empty board
    hipStream_t s; CK(hipStreamCreate(&s));
    const int N = 3000;
    for (int r = 0; r < 2; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) if (op_sync(in, out, 1, s)) return 2;
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("launch + hipStreamSynchronize          : %.2f us per call (n_moves out %d)\n", us, out[18]);
    }
    for (int r = 0; r < 2; ++r) {
        long spins = 0;
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            volatile uint8_t *flag = out + 18;                     // n_moves of the out record: 1 after the move
            *flag = 0xEE;
            if (op(in, out, 1, s)) return 2;
            while (*flag == 0xEE) ++spins;
        }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        CK(hipStreamSynchronize(s));
        printf("launch + spin on the out record's byte : %.2f us per call (%.0f spins per call, n_moves out %d)\n", us, (double)spins / N, out[18]);
    }
    for (int r = 0; r < 2 && op_host; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) if (op_host(in, out, 1, s)) return 2;
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("qttt_board_op_host (launch + stamp poll): %.2f us per call (n_moves out %d, stamp %d)\n", us, out[18], out[63]);
    }
    return 0;
}
