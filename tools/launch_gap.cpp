// tools/launch_gap.cpp — back-to-back launch cost of an (almost) empty kernel on one stream, eager
// and replayed from a HIP graph: the platform's per-launch floor that every step launch pays
// (diagnostic).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); return 1; } } while (0)
__global__ void tiny(unsigned *p) { if (threadIdx.x == 0 && p[0] == 12345u) p[1] = 1; }
int main() {
    unsigned *d; CK(hipMalloc(&d, 64)); CK(hipMemset(d, 0, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 1000;
    for (int grid : {1, 4096}) {
        float ms = 0;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d);
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("eager  grid %5d x 256: %.2f us per back-to-back launch\n", grid, ms * 1e3 / N);
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(tiny, dim3(grid), dim3(256), 0, s, d);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            CK(hipEventElapsedTime(&ms, e0, e1));
        }
        printf("graph  grid %5d x 256: %.2f us per launch inside a %d-node graph\n", grid, ms * 1e3 / N, N);
        CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
    return 0;
}
