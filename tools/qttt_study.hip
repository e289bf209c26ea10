// tools/qttt_study.hip — the wave-per-board MAPPING STUDY, built as its own library
// (tools/libqttt_study.so) so that neither include/qttt.h nor the product libqttt_hip.so carries it.
// It includes the product's step function (qttt_step_core.h), so "same results" is tested against the very
// code the product runs; tests/test_step_parity_gpu.py and tools/stepbench load it with dlopen / ctypes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Iinclude -Iqtttgym_amd/csrc tools/qttt_study.hip -o tools/libqttt_study.so
#include "qttt_step_core.h"

namespace {

// Mapping study: ONE WAVEFRONT PER BOARD, the mapping BASELINE.json's north_star
// sketches.  A board's step is a chain of dependent operations on a 9-node graph (validity ->
// component lookup -> path walk -> collapse -> line test), so whatever the 64 lanes of a wave do
// with __shfl/__ballot, the wave cannot retire a board faster than one lane can run that chain.
// This kernel is that lower bound made concrete: lane 0 of every wave runs the same step_core, the
// other 63 lanes are idle, state is staged through LDS by the workgroup.  Same results as
// step_kernel (tested); measured beside it in tools/stepbench.
template <bool HAS_BITS, bool AUTO_RESET>
__global__ __launch_bounds__(256) void step_wave_per_board_kernel(
    u64 *__restrict__ pP, u64 *__restrict__ pQ, const uint16_t *__restrict__ actions,
    const uint8_t *__restrict__ bits, u32 key_fold, u32 id_base, u32 *__restrict__ reward_bits,
    uint8_t *__restrict__ terminated, int64_t n) {
    __shared__ __attribute__((aligned(16))) uint8_t lut[LINE_LUT_BYTES];
    __shared__ u64 sP[4], sQ[4];
    __shared__ u32 sAct[4], sBit[4];
    fill_line_lut_nosync<256>(lut);
    const int64_t i0 = (int64_t)blockIdx.x * 4;                   // 4 waves = 4 boards per workgroup
    if (threadIdx.x < 4 && i0 + threadIdx.x < n) {                 // cooperative tile load into LDS
        const int64_t i = i0 + threadIdx.x;
        sP[threadIdx.x] = pP[i];
        sQ[threadIdx.x] = pQ[i];
        sAct[threadIdx.x] = actions[i];
        sBit[threadIdx.x] = HAS_BITS ? bits[i] & 1u : collapse_bit_of((id_base + (u32)i) ^ key_fold);
    }
    __syncthreads();
    const u32 w = threadIdx.x >> 6;
    const int64_t i = i0 + w;
    if (i >= n || (threadIdx.x & 63u) != 0u) return;              // lane 0 of each wave owns the board
    u32 P0 = (u32)sP[w], P1 = (u32)(sP[w] >> 32), Q0 = (u32)sQ[w], Q1 = (u32)(sQ[w] >> 32);
    const u32 win = step_core<AUTO_RESET>(P0, P1, Q0, Q1, sAct[w], sBit[w], lut);
    pP[i] = (u64)P0 | ((u64)P1 << 32);
    pQ[i] = (u64)Q0 | ((u64)Q1 << 32);
    reward_bits[i] = 0x80000000u | (win << 23);
    terminated[i] = (uint8_t)(P1 >> 31);
}

}  // namespace

extern "C" int qttt_step_wave_per_board(void *state, const uint8_t *actions, const uint8_t *bits, uint64_t seed,
                                        uint32_t step_idx, int64_t board_offset, uint32_t flags, float *reward,
                                        uint8_t *terminated, int64_t n, void *stream) {
    if (n < 0 || board_offset < 0) return QTTT_ERR_SIZE;
    if (n == 0) return 0;
    if (!state || !actions || !reward || !terminated) return QTTT_ERR_NULL;
    if ((uintptr_t)actions & 1u) return QTTT_ERR_ACTION;
    const u64 first = (u64)board_offset;
    if ((first >> 32) != ((first + (u64)n - 1u) >> 32)) return QTTT_ERR_SIZE;   // study kernel: one id range
    Planes p = planes(state, n);
    const u32 key_fold = (u32)launch_key(seed, step_idx) ^ ((u32)(first >> 32) * 0x9E3779B9u);
    const bool ar = (flags & QTTT_FLAG_AUTO_RESET) != 0;
    dim3 g((unsigned)((n + 3) / 4)), b(256);
    hipStream_t s = (hipStream_t)stream;
    const uint16_t *a16 = reinterpret_cast<const uint16_t *>(actions);
    u32 *rb = reinterpret_cast<u32 *>(reward);
#define QTTT_WPB(HB, AR) hipLaunchKernelGGL((step_wave_per_board_kernel<HB, AR>), g, b, 0, s, p.P, p.Q, \
                                            a16, bits, key_fold, (u32)first, rb, terminated, n)
    if (bits) { if (ar) QTTT_WPB(true, true); else QTTT_WPB(true, false); }
    else      { if (ar) QTTT_WPB(false, true); else QTTT_WPB(false, false); }
#undef QTTT_WPB
    const hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}
