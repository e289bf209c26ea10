// tools/valu_rates.cpp — measures the issue cost of the VALU instructions the step kernel is
// made of, relative to v_xor_b32, with 8 waves per SIMD on every CU (diagnostic only).
//   hipcc --offload-arch=gfx950 -O3 tools/valu_rates.cpp -o tools/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e)); exit(1); } } while (0)

#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP32(x) REP16(x) REP16(x)

#define KERNEL(name, ASM)                                                            \
    __global__ __launch_bounds__(256) void name(unsigned *out, int iters) {          \
        unsigned a = threadIdx.x, b = threadIdx.x * 3 + 1, c = 5, d = 7;             \
        unsigned long long q = threadIdx.x, p = 3;                                   \
        for (int i = 0; i < iters; ++i) {                                            \
            asm volatile(REP32(ASM) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(q), "+v"(p)); \
        }                                                                            \
        if (a == 0x12345678u && q == 77) out[0] = a + b + c + d + (unsigned)p;       \
    }

KERNEL(k_xor, "v_xor_b32 %0, %1, %0\n v_xor_b32 %2, %3, %2\n")
KERNEL(k_and_or, "v_and_or_b32 %0, %1, %2, %0\n v_and_or_b32 %2, %3, %1, %2\n")
KERNEL(k_bitop3, "v_bitop3_b32 %0, %1, %2, %0 bitop3:0x6c\n v_bitop3_b32 %2, %3, %1, %2 bitop3:0x6c\n")
KERNEL(k_bfe, "v_bfe_u32 %0, %1, 4, 8\n v_bfe_u32 %2, %3, 4, 8\n")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %1, 4, %0\n v_lshl_or_b32 %2, %3, 4, %2\n")
KERNEL(k_lshlrev32, "v_lshlrev_b32 %0, %1, %0\n v_lshlrev_b32 %2, %3, %2\n")
KERNEL(k_lshl64, "v_lshlrev_b64 %4, %1, %4\n v_lshlrev_b64 %5, %3, %5\n")
KERNEL(k_lshr64, "v_lshrrev_b64 %4, %1, %4\n v_lshrrev_b64 %5, %3, %5\n")
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %1, %0\n v_mul_lo_u32 %2, %3, %2\n")
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %1, %0\n v_mul_u32_u24 %2, %3, %2\n")
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %1, %2, %0\n v_mad_u32_u24 %2, %3, %1, %2\n")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %1, %0, vcc\n v_cndmask_b32 %2, %3, %2, vcc\n")
KERNEL(k_ffbl, "v_ffbl_b32 %0, %1\n v_ffbl_b32 %2, %3\n")
KERNEL(k_bcnt, "v_bcnt_u32_b32 %0, %1, %0\n v_bcnt_u32_b32 %2, %3, %2\n")
KERNEL(k_perm, "v_perm_b32 %0, %1, %2, %0\n v_perm_b32 %2, %3, %1, %2\n")
KERNEL(k_alignbit, "v_alignbit_b32 %0, %1, %2, %0\n v_alignbit_b32 %2, %3, %1, %2\n")
KERNEL(k_cmp, "v_cmp_eq_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %2, %3\n")
KERNEL(k_cmp_sgpr, "v_cmp_eq_u32 s[10:11], %0, %1\n v_cmp_lt_u32 s[12:13], %2, %3\n")
KERNEL(k_sdwa, "v_and_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:DWORD\n v_and_b32_sdwa %2, %3, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD\n")
KERNEL(k_add_u64, "v_lshl_add_u64 %4, %4, 0, %5\n v_lshl_add_u64 %5, %5, 1, %4\n")
KERNEL(k_min_max, "v_min_u32 %0, %1, %0\n v_max_u32 %2, %3, %2\n")
KERNEL(k_add3, "v_add3_u32 %0, %1, %2, %0\n v_xad_u32 %2, %3, %1, %2\n")
KERNEL(k_mov, "v_mov_b32 %0, %1\n v_mov_b32 %2, %3\n")
KERNEL(k_dep_xor, "v_xor_b32 %0, %1, %0\n v_xor_b32 %0, %3, %0\n")
KERNEL(k_pk_mixed, "v_xor_b32 %0, %1, %0\n s_nop 0\n")

KERNEL(k_and, "v_and_b32 %0, %1, %0\n v_and_b32 %2, %3, %2\n")
KERNEL(k_or, "v_or_b32 %0, %1, %0\n v_or_b32 %2, %3, %2\n")
KERNEL(k_add, "v_add_u32 %0, %1, %0\n v_add_u32 %2, %3, %2\n")
KERNEL(k_sub, "v_sub_u32 %0, %1, %0\n v_sub_u32 %2, %3, %2\n")
KERNEL(k_shl_c, "v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %2, 5, %3\n")
KERNEL(k_shr_c, "v_lshrrev_b32 %0, 3, %1\n v_lshrrev_b32 %2, 5, %3\n")
KERNEL(k_bfi, "v_bfi_b32 %0, %1, %2, %0\n v_bfi_b32 %2, %3, %1, %2\n")
KERNEL(k_or3, "v_or3_b32 %0, %1, %2, %0\n v_or3_b32 %2, %3, %1, %2\n")
KERNEL(k_not, "v_not_b32 %0, %1\n v_not_b32 %2, %3\n")
KERNEL(k_and_lit, "v_and_b32 %0, 0x8040201, %0\n v_and_b32 %2, 0x1ff01ff, %2\n")
KERNEL(k_cndmask_s, "v_cndmask_b32 %0, %1, %0, s[10:11]\n v_cndmask_b32 %2, %3, %2, s[12:13]\n")
KERNEL(k_cmp_e32, "v_cmp_eq_u32_e32 vcc, %0, %1\n v_cmp_lt_u32_e32 vcc, %2, %3\n")
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %1, 2, %0\n v_lshl_add_u32 %2, %3, 2, %2\n")
KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %1, %0, 2\n v_add_lshl_u32 %2, %3, %2, 2\n")
KERNEL(k_xor_dpp, "v_xor_b32_dpp %0, %1, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_xor_b32_dpp %2, %3, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")
KERNEL(k_fma, "v_fma_f32 %0, %1, %2, %0\n v_fma_f32 %2, %3, %1, %2\n")
KERNEL(k_pk_add, "v_pk_add_u16 %0, %1, %0\n v_pk_add_u16 %2, %3, %2\n")
KERNEL(k_pk_lshl, "v_pk_lshlrev_b16 %0, %1, %0\n v_pk_lshlrev_b16 %2, %3, %2\n")
KERNEL(k_xnor, "v_xnor_b32 %0, %1, %0\n v_xnor_b32 %2, %3, %2\n")
KERNEL(k_xor4, "v_xor_b32 %0, %1, %0\n v_xor_b32 %2, %3, %2\n v_xor_b32 %1, %0, %1\n v_xor_b32 %3, %2, %3\n")
KERNEL(k_mix_xor_shl, "v_xor_b32 %0, %1, %0\n v_lshlrev_b32 %2, 5, %3\n")

KERNEL(k_shr_v, "v_lshrrev_b32 %0, %1, %0\n v_lshrrev_b32 %2, %3, %2\n")
KERNEL(k_ashr_v, "v_ashrrev_i32 %0, %1, %0\n v_ashrrev_i32 %2, %3, %2\n")
KERNEL(k_shr64_c, "v_lshrrev_b64 %4, 9, %4\n v_lshrrev_b64 %5, 28, %5\n")
KERNEL(k_addco, "v_add_co_u32 %0, vcc, %1, %0\n v_addc_co_u32 %2, vcc, %3, %2, vcc\n")
KERNEL(k_subrev, "v_subrev_u32 %0, %1, %0\n v_subrev_u32 %2, %3, %2\n")
KERNEL(k_and_s, "v_and_b32 %0, s10, %0\n v_and_b32 %2, s11, %2\n")
KERNEL(k_bitop3_s, "v_bitop3_b32 %0, %1, s10, %0 bitop3:0x6c\n v_bitop3_b32 %2, %3, s11, %2 bitop3:0x6c\n")
KERNEL(k_bitop3_c, "v_bitop3_b32 %0, %1, 15, %0 bitop3:0x6c\n v_bitop3_b32 %2, %3, 1, %2 bitop3:0x6c\n")
KERNEL(k_mov_s, "v_mov_b32 %0, s10\n v_mov_b32 %2, s11\n")
KERNEL(k_max_i, "v_max_i32 %0, %1, %0\n v_min_i32 %2, %3, %2\n")
KERNEL(k_cvt, "v_cvt_f32_u32 %0, %1\n v_cvt_u32_f32 %2, %3\n")
KERNEL(k_fadd, "v_add_f32 %0, %1, %0\n v_mul_f32 %2, %3, %2\n")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %1, %0\n v_ldexp_f32 %2, %3, %2\n")
KERNEL(k_frexp, "v_frexp_exp_i32_f32 %0, %1\n v_frexp_exp_i32_f32 %2, %3\n")
KERNEL(k_sad, "v_sad_u32 %0, %1, %2, %0\n v_sad_u32 %2, %3, %1, %2\n")
KERNEL(k_mad_u64, "v_mad_u64_u32 %4, vcc, %1, %3, %4\n v_mad_u64_u32 %5, vcc, %3, %1, %5\n")
KERNEL(k_mul_hi, "v_mul_hi_u32 %0, %1, %0\n v_mul_hi_u32 %2, %3, %2\n")
KERNEL(k_xor3x, "v_xor_b32 %0, %1, %0\n v_and_b32 %2, %3, %2\n v_or_b32 %1, %0, %1\n")

template <typename F>
double run(F kern, int iters, unsigned *out, hipStream_t s) {
    dim3 g(256 * 8), b(256);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, g, b, 0, s, out, 10);
    double best = 1e30;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(kern, g, b, 0, s, out, iters);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    unsigned *out; CK(hipMalloc(&out, 64));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int iters = 2000;
    double base = run(k_xor, iters, out, s);
    // per SIMD: 8 waves x iters x 64 instrs
    double inst = 8.0 * iters * 64;
    printf("v_xor_b32: %.3f ms -> %.2f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz)\n", base, base * 1e6 / inst, base * 1e6 / inst * 2.4);
#define T(k) { double t = run(k, iters, out, s); printf("%-14s %.3f ms  x%.2f vs v_xor_b32\n", #k, t, t / base); }
    T(k_mov) T(k_and_or) T(k_bitop3) T(k_bfe) T(k_lshl_or) T(k_lshlrev32) T(k_lshl64) T(k_lshr64) T(k_mul_lo) T(k_mul_u24)
    T(k_mad_u24) T(k_cndmask) T(k_ffbl) T(k_bcnt) T(k_perm) T(k_alignbit) T(k_cmp) T(k_cmp_sgpr) T(k_sdwa) T(k_add_u64)
    T(k_min_max) T(k_add3) T(k_dep_xor) T(k_pk_mixed)
    T(k_and) T(k_or) T(k_add) T(k_sub) T(k_shl_c) T(k_shr_c) T(k_bfi) T(k_or3) T(k_not) T(k_and_lit)
    T(k_cndmask_s) T(k_cmp_e32) T(k_lshl_add) T(k_add_lshl) T(k_xor_dpp) T(k_fma) T(k_pk_add) T(k_pk_lshl) T(k_xnor) T(k_xor4) T(k_mix_xor_shl)
    T(k_shr_v) T(k_ashr_v) T(k_shr64_c) T(k_addco) T(k_subrev) T(k_and_s) T(k_bitop3_s) T(k_bitop3_c) T(k_mov_s) T(k_max_i) T(k_cvt) T(k_fadd) T(k_ldexp) T(k_frexp) T(k_sad) T(k_xor3x) T(k_mad_u64) T(k_mul_hi)
    return 0;
}
