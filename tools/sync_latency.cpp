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

// ---- floors below qttt_board_op_host (round 5): what the round trip costs without the board's arithmetic
// (a) an empty kernel that only stamps the out record; (b) one that also reads the 64-byte in record from pinned host
// memory with four 16-byte loads and echoes it; (c) a BOUNDED mailbox: one wave stays resident, polls a doorbell word in
// pinned host memory, answers each ring with the echo of (b), and returns BY ITSELF once no ring has come for
// `idle_ticks` of the 100 MHz s_memrealtime counter (200 us) or after `max_rings` — it is never left resident, the host
// relaunches it on demand.  Every loop in it has the time-out as its exit.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void stamp_kernel(volatile uint8_t *out) { out[63] = 1; }
__global__ void echo_kernel(const u32x4 *in, u32x4 *out) {
    if (threadIdx.x >= 4) return;
    u32x4 v = in[threadIdx.x];
    if (threadIdx.x == 3) v.w = (v.w & 0x00FFFFFFu) | 0x01000000u;           // byte 63 = the stamp, with the record
    out[threadIdx.x] = v;
}
__global__ void mailbox_kernel(const u32x4 *in, u32x4 *out, unsigned *doorbell, unsigned *answered, unsigned first_ring,
                               unsigned max_rings, unsigned long long idle_ticks) {
    unsigned want = first_ring;
    for (unsigned served = 0; served < max_rings; ++served) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        bool rung = false;
        while (!rung) {                                                       // exit: the ring, or the time-out
            // wave-uniform: every lane acts on lane 0's view of the doorbell, so the wave leaves or answers as one
            const unsigned seen = __builtin_amdgcn_readfirstlane(__hip_atomic_load(doorbell, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM));
            rung = seen == want;
            if (!rung && __builtin_amdgcn_s_memrealtime() - t0 > idle_ticks) return;
        }
        if (threadIdx.x < 4) {
            u32x4 v = __builtin_nontemporal_load(&in[threadIdx.x]);
            out[threadIdx.x] = v;
        }
        __threadfence_system();
        if (threadIdx.x == 0) __hip_atomic_store(answered, want, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        ++want;
    }
}

static int floors(hipStream_t s) {
    uint8_t *in, *out; unsigned *bell;
    CK(hipHostMalloc(&in, 64, hipHostMallocDefault)); CK(hipHostMalloc(&out, 64, hipHostMallocDefault));
    CK(hipHostMalloc(&bell, 128, hipHostMallocDefault));
    memset(in, 7, 64); memset(out, 0, 64); memset(bell, 0, 128);
    volatile uint8_t *vo = out;
    volatile unsigned *answered = bell + 16;
    const int N = 3000;
    for (int r = 0; r < 2; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) { vo[63] = 0; hipLaunchKernelGGL(stamp_kernel, dim3(1), dim3(64), 0, s, out); while (vo[63] == 0) {} }
        printf("floor: empty kernel, launch + poll its stamp                : %.2f us per call\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N);
    }
    for (int r = 0; r < 2; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) { vo[63] = 0; in[0] = (uint8_t)i; hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(64), 0, s, (const u32x4 *)in, (u32x4 *)out); while (vo[63] == 0) {} if (vo[0] != (uint8_t)i) return 3; }
        printf("floor: kernel echoing the 64-byte pinned record, launch + poll: %.2f us per call\n", std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N);
    }
    CK(hipStreamSynchronize(s));
    // the mailbox: rings are numbered from 1; a resident kernel serves at most 4096 of them and leaves after 200 us idle
    unsigned ring = 0, launches = 0;
    for (int r = 0; r < 2; ++r) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            in[0] = (uint8_t)i;
            ++ring;
            bool alive = hipStreamQuery(s) == hipErrorNotReady;               // (a real client tracks this itself; the query costs ~1 us)
            if (!alive) { hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(64), 0, s, (const u32x4 *)in, (u32x4 *)out, bell, bell + 16, ring, 4096u, 20000ull); ++launches; }
            __atomic_store_n(bell, ring, __ATOMIC_RELEASE);
            const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(50);
            while (*answered != ring) {
                if (std::chrono::steady_clock::now() > give_up) {              // the kernel timed out between our query and the ring: relaunch
                    hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(64), 0, s, (const u32x4 *)in, (u32x4 *)out, bell, bell + 16, ring, 4096u, 20000ull); ++launches;
                    break;
                }
            }
            while (*answered != ring) {}
            if (vo[0] != (uint8_t)i) return 4;
        }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        CK(hipStreamSynchronize(s));                                           // the mailbox leaves by itself (200 us idle)
        printf("mailbox: ring a resident wave's doorbell, echo of the record : %.2f us per call (%u kernel launches for %d calls so far)\n", us, launches, (r + 1) * N);
    }
    // and with the caller doing 100 us of something else between calls (the mailbox survives), and 400 us (it does not)
    for (int gap_us : {100, 400}) {
        double busy = 0;
        const int M = 300;
        for (int i = 0; i < M; ++i) {
            auto g0 = std::chrono::steady_clock::now();
            while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - g0).count() < gap_us) {}
            auto t0 = std::chrono::steady_clock::now();
            in[0] = (uint8_t)i;
            ++ring;
            if (hipStreamQuery(s) != hipErrorNotReady) { hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(64), 0, s, (const u32x4 *)in, (u32x4 *)out, bell, bell + 16, ring, 4096u, 20000ull); ++launches; }
            __atomic_store_n(bell, ring, __ATOMIC_RELEASE);
            const auto give_up = std::chrono::steady_clock::now() + std::chrono::milliseconds(50);
            while (*answered != ring) {
                if (std::chrono::steady_clock::now() > give_up) { hipLaunchKernelGGL(mailbox_kernel, dim3(1), dim3(64), 0, s, (const u32x4 *)in, (u32x4 *)out, bell, bell + 16, ring, 4096u, 20000ull); ++launches; break; }
            }
            while (*answered != ring) {}
            busy += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        }
        CK(hipStreamSynchronize(s));
        printf("mailbox with %3d us of host work between calls               : %.2f us per call (%u launches in total)\n", gap_us, busy / M, launches);
    }
    return 0;
}

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
    if (int rc = floors(s)) { fprintf(stderr, "floors() failed: %d\n", rc); return rc; }
    return 0;
}
