// tools/launch_gap.cpp — back-to-back launch cost of an (almost) empty kernel on one stream:
// the platform's per-launch floor that every step launch pays (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void tiny(unsigned *p) { if (threadIdx.x == 0 && p[0] == 12345u) p[1] = 1; }
int main() {
    unsigned *d; hipMalloc(&d, 64); hipMemset(d, 0, 64);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {1, 256, 4096}) {
        for (int r = 0; r < 3; ++r) {
            hipEventRecord(e0, s);
            for (int i = 0; i < 1000; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d);
            hipEventRecord(e1, s); hipStreamSynchronize(s);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (r == 2) printf("grid %5d x 256: %.2f us per back-to-back launch\n", grid, ms);
        }
    }
    return 0;
}
