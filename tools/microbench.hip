// microbench.hip — per-instruction issue cost on gfx950 at 1/2/4/8 waves per SIMD.
// Build-and-measure tool (not part of the product): hipcc --offload-arch=gfx950 -O3 tools/microbench.hip -o /tmp/mb
// Each wave runs ITERS iterations of 16 INDEPENDENT instructions of one kind (8 accumulators x 2),
// timed with s_memtime; reports shader cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define ITERS 2000

#define BODY8(INS)                                                    \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) \
                 : "v"(x), "v"(y));

#define F32_KERNEL(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                   \
    __global__ void NAME(float* out, long long* cyc, float x, float y)                                      \
    {                                                                                                       \
        float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
              a7 = a0 + 7;                                                                                  \
        __syncthreads();                                                                                    \
        long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        for (int i = 0; i < ITERS; ++i) {                                                                   \
            asm volatile(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"                     \
                         I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
                         : "v"(x), "v"(y));                                                                 \
        }                                                                                                   \
        long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                  \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;            \
    }

F32_KERNEL(k_fma_f32, "v_fma_f32 %0, %8, %9, %0", "v_fma_f32 %1, %8, %9, %1", "v_fma_f32 %2, %8, %9, %2",
           "v_fma_f32 %3, %8, %9, %3", "v_fma_f32 %4, %8, %9, %4", "v_fma_f32 %5, %8, %9, %5",
           "v_fma_f32 %6, %8, %9, %6", "v_fma_f32 %7, %8, %9, %7")
F32_KERNEL(k_mul_f32, "v_mul_f32 %0, %8, %0", "v_mul_f32 %1, %8, %1", "v_mul_f32 %2, %8, %2", "v_mul_f32 %3, %8, %3",
           "v_mul_f32 %4, %8, %4", "v_mul_f32 %5, %8, %5", "v_mul_f32 %6, %8, %6", "v_mul_f32 %7, %8, %7")
F32_KERNEL(k_add_f32, "v_add_f32 %0, %8, %0", "v_add_f32 %1, %8, %1", "v_add_f32 %2, %8, %2", "v_add_f32 %3, %8, %3",
           "v_add_f32 %4, %8, %4", "v_add_f32 %5, %8, %5", "v_add_f32 %6, %8, %6", "v_add_f32 %7, %8, %7")
F32_KERNEL(k_rcp_f32, "v_rcp_f32 %0, %0", "v_rcp_f32 %1, %1", "v_rcp_f32 %2, %2", "v_rcp_f32 %3, %3",
           "v_rcp_f32 %4, %4", "v_rcp_f32 %5, %5", "v_rcp_f32 %6, %6", "v_rcp_f32 %7, %7")
F32_KERNEL(k_sqrt_f32, "v_sqrt_f32 %0, %0", "v_sqrt_f32 %1, %1", "v_sqrt_f32 %2, %2", "v_sqrt_f32 %3, %3",
           "v_sqrt_f32 %4, %4", "v_sqrt_f32 %5, %5", "v_sqrt_f32 %6, %6", "v_sqrt_f32 %7, %7")
F32_KERNEL(k_exp_f32, "v_exp_f32 %0, %0", "v_exp_f32 %1, %1", "v_exp_f32 %2, %2", "v_exp_f32 %3, %3",
           "v_exp_f32 %4, %4", "v_exp_f32 %5, %5", "v_exp_f32 %6, %6", "v_exp_f32 %7, %7")
F32_KERNEL(k_cvt_u32, "v_cvt_u32_f32 %0, %0", "v_cvt_u32_f32 %1, %1", "v_cvt_u32_f32 %2, %2", "v_cvt_u32_f32 %3, %3",
           "v_cvt_u32_f32 %4, %4", "v_cvt_u32_f32 %5, %5", "v_cvt_u32_f32 %6, %6", "v_cvt_u32_f32 %7, %7")
F32_KERNEL(k_and_b32, "v_and_b32 %0, %8, %0", "v_and_b32 %1, %8, %1", "v_and_b32 %2, %8, %2", "v_and_b32 %3, %8, %3",
           "v_and_b32 %4, %8, %4", "v_and_b32 %5, %8, %5", "v_and_b32 %6, %8, %6", "v_and_b32 %7, %8, %7")
F32_KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 3, %8", "v_lshl_add_u32 %1, %1, 3, %8", "v_lshl_add_u32 %2, %2, 3, %8",
           "v_lshl_add_u32 %3, %3, 3, %8", "v_lshl_add_u32 %4, %4, 3, %8", "v_lshl_add_u32 %5, %5, 3, %8",
           "v_lshl_add_u32 %6, %6, 3, %8", "v_lshl_add_u32 %7, %7, 3, %8")
F32_KERNEL(k_cndmask, "v_cndmask_b32 %0, %8, %0, s[20:21]", "v_cndmask_b32 %1, %8, %1, s[20:21]", "v_cndmask_b32 %2, %8, %2, s[20:21]",
           "v_cndmask_b32 %3, %8, %3, s[20:21]", "v_cndmask_b32 %4, %8, %4, s[20:21]", "v_cndmask_b32 %5, %8, %5, s[20:21]",
           "v_cndmask_b32 %6, %8, %6, s[20:21]", "v_cndmask_b32 %7, %8, %7, s[20:21]")
F32_KERNEL(k_dpp_mov, "v_mov_b32_dpp %0, %0 wave_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf",
           "v_mov_b32_dpp %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %3 wave_shr:1 row_mask:0xf bank_mask:0xf",
           "v_mov_b32_dpp %4, %4 wave_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %5, %5 wave_shr:1 row_mask:0xf bank_mask:0xf",
           "v_mov_b32_dpp %6, %6 wave_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %7, %7 wave_shr:1 row_mask:0xf bank_mask:0xf")

#define U8(fmt) fmt(0), fmt(1), fmt(2), fmt(3), fmt(4), fmt(5), fmt(6), fmt(7)
F32_KERNEL(k_cnd_vcc, "v_cndmask_b32 %0, %8, %0, vcc", "v_cndmask_b32 %1, %8, %1, vcc", "v_cndmask_b32 %2, %8, %2, vcc",
           "v_cndmask_b32 %3, %8, %3, vcc", "v_cndmask_b32 %4, %8, %4, vcc", "v_cndmask_b32 %5, %8, %5, vcc",
           "v_cndmask_b32 %6, %8, %6, vcc", "v_cndmask_b32 %7, %8, %7, vcc")
F32_KERNEL(k_add_u32, "v_add_u32 %0, %8, %0", "v_add_u32 %1, %8, %1", "v_add_u32 %2, %8, %2", "v_add_u32 %3, %8, %3",
           "v_add_u32 %4, %8, %4", "v_add_u32 %5, %8, %5", "v_add_u32 %6, %8, %6", "v_add_u32 %7, %8, %7")
F32_KERNEL(k_lshr, "v_lshrrev_b32 %0, 3, %0", "v_lshrrev_b32 %1, 3, %1", "v_lshrrev_b32 %2, 3, %2", "v_lshrrev_b32 %3, 3, %3",
           "v_lshrrev_b32 %4, 3, %4", "v_lshrrev_b32 %5, 3, %5", "v_lshrrev_b32 %6, 3, %6", "v_lshrrev_b32 %7, 3, %7")
F32_KERNEL(k_max_f32, "v_max_f32 %0, %8, %0", "v_max_f32 %1, %8, %1", "v_max_f32 %2, %8, %2", "v_max_f32 %3, %8, %3",
           "v_max_f32 %4, %8, %4", "v_max_f32 %5, %8, %5", "v_max_f32 %6, %8, %6", "v_max_f32 %7, %8, %7")
F32_KERNEL(k_med3_f32, "v_med3_f32 %0, %0, %8, %9", "v_med3_f32 %1, %1, %8, %9", "v_med3_f32 %2, %2, %8, %9", "v_med3_f32 %3, %3, %8, %9",
           "v_med3_f32 %4, %4, %8, %9", "v_med3_f32 %5, %5, %8, %9", "v_med3_f32 %6, %6, %8, %9", "v_med3_f32 %7, %7, %8, %9")
F32_KERNEL(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 7", "v_bfe_u32 %1, %1, 3, 7", "v_bfe_u32 %2, %2, 3, 7", "v_bfe_u32 %3, %3, 3, 7",
           "v_bfe_u32 %4, %4, 3, 7", "v_bfe_u32 %5, %5, 3, 7", "v_bfe_u32 %6, %6, 3, 7", "v_bfe_u32 %7, %7, 3, 7")
F32_KERNEL(k_cmp_f32, "v_cmp_lt_f32 vcc, %8, %0", "v_cmp_lt_f32 vcc, %8, %1", "v_cmp_lt_f32 vcc, %8, %2", "v_cmp_lt_f32 vcc, %8, %3",
           "v_cmp_lt_f32 vcc, %8, %4", "v_cmp_lt_f32 vcc, %8, %5", "v_cmp_lt_f32 vcc, %8, %6", "v_cmp_lt_f32 vcc, %8, %7")
F32_KERNEL(k_mov_b32, "v_mov_b32 %0, %8", "v_mov_b32 %1, %9", "v_mov_b32 %2, %8", "v_mov_b32 %3, %9",
           "v_mov_b32 %4, %8", "v_mov_b32 %5, %9", "v_mov_b32 %6, %8", "v_mov_b32 %7, %9")
F32_KERNEL(k_mul_sgpr, "v_mul_f32 %0, s20, %0", "v_mul_f32 %1, s20, %1", "v_mul_f32 %2, s20, %2", "v_mul_f32 %3, s20, %3",
           "v_mul_f32 %4, s20, %4", "v_mul_f32 %5, s20, %5", "v_mul_f32 %6, s20, %6", "v_mul_f32 %7, s20, %7")
F32_KERNEL(k_mul_lit, "v_mul_f32 %0, 0x3f9e3779, %0", "v_mul_f32 %1, 0x3f9e3779, %1", "v_mul_f32 %2, 0x3f9e3779, %2", "v_mul_f32 %3, 0x3f9e3779, %3",
           "v_mul_f32 %4, 0x3f9e3779, %4", "v_mul_f32 %5, 0x3f9e3779, %5", "v_mul_f32 %6, 0x3f9e3779, %6", "v_mul_f32 %7, 0x3f9e3779, %7")
F32_KERNEL(k_fma_sgpr, "v_fma_f32 %0, %8, s20, %0", "v_fma_f32 %1, %8, s20, %1", "v_fma_f32 %2, %8, s20, %2", "v_fma_f32 %3, %8, s20, %3",
           "v_fma_f32 %4, %8, s20, %4", "v_fma_f32 %5, %8, s20, %5", "v_fma_f32 %6, %8, s20, %6", "v_fma_f32 %7, %8, s20, %7")
F32_KERNEL(k_readlane, "v_readlane_b32 s20, %0, 3", "v_readlane_b32 s21, %1, 3", "v_readlane_b32 s22, %2, 3", "v_readlane_b32 s23, %3, 3",
           "v_readlane_b32 s24, %4, 3", "v_readlane_b32 s25, %5, 3", "v_readlane_b32 s26, %6, 3", "v_readlane_b32 s27, %7, 3")
F32_KERNEL(k_cvt_f32_u32, "v_cvt_f32_u32 %0, %0", "v_cvt_f32_u32 %1, %1", "v_cvt_f32_u32 %2, %2", "v_cvt_f32_u32 %3, %3",
           "v_cvt_f32_u32 %4, %4", "v_cvt_f32_u32 %5, %5", "v_cvt_f32_u32 %6, %6", "v_cvt_f32_u32 %7, %7")

F32_KERNEL(k_cnd_e32, "v_cndmask_b32_e32 %0, %8, %0, vcc", "v_cndmask_b32_e32 %1, %8, %1, vcc", "v_cndmask_b32_e32 %2, %8, %2, vcc", "v_cndmask_b32_e32 %3, %8, %3, vcc", "v_cndmask_b32_e32 %4, %8, %4, vcc", "v_cndmask_b32_e32 %5, %8, %5, vcc", "v_cndmask_b32_e32 %6, %8, %6, vcc", "v_cndmask_b32_e32 %7, %8, %7, vcc")
F32_KERNEL(k_cnd_e64vcc, "v_cndmask_b32_e64 %0, %8, %0, vcc", "v_cndmask_b32_e64 %1, %8, %1, vcc", "v_cndmask_b32_e64 %2, %8, %2, vcc", "v_cndmask_b32_e64 %3, %8, %3, vcc", "v_cndmask_b32_e64 %4, %8, %4, vcc", "v_cndmask_b32_e64 %5, %8, %5, vcc", "v_cndmask_b32_e64 %6, %8, %6, vcc", "v_cndmask_b32_e64 %7, %8, %7, vcc")
F32_KERNEL(k_cmp_cnd_vcc, "v_cmp_lt_f32_e32 vcc, %8, %0\nv_cndmask_b32_e32 %0, %9, %0, vcc", "v_cmp_lt_f32_e32 vcc, %8, %1\nv_cndmask_b32_e32 %1, %9, %1, vcc", "v_cmp_lt_f32_e32 vcc, %8, %2\nv_cndmask_b32_e32 %2, %9, %2, vcc", "v_cmp_lt_f32_e32 vcc, %8, %3\nv_cndmask_b32_e32 %3, %9, %3, vcc", "v_cmp_lt_f32_e32 vcc, %8, %4\nv_cndmask_b32_e32 %4, %9, %4, vcc", "v_cmp_lt_f32_e32 vcc, %8, %5\nv_cndmask_b32_e32 %5, %9, %5, vcc", "v_cmp_lt_f32_e32 vcc, %8, %6\nv_cndmask_b32_e32 %6, %9, %6, vcc", "v_cmp_lt_f32_e32 vcc, %8, %7\nv_cndmask_b32_e32 %7, %9, %7, vcc")
F32_KERNEL(k_cmp_cnd_sgpr, "v_cmp_lt_f32_e64 s[20:21], %8, %0\nv_cndmask_b32_e64 %0, %9, %0, s[20:21]", "v_cmp_lt_f32_e64 s[20:21], %8, %1\nv_cndmask_b32_e64 %1, %9, %1, s[20:21]", "v_cmp_lt_f32_e64 s[20:21], %8, %2\nv_cndmask_b32_e64 %2, %9, %2, s[20:21]", "v_cmp_lt_f32_e64 s[20:21], %8, %3\nv_cndmask_b32_e64 %3, %9, %3, s[20:21]", "v_cmp_lt_f32_e64 s[20:21], %8, %4\nv_cndmask_b32_e64 %4, %9, %4, s[20:21]", "v_cmp_lt_f32_e64 s[20:21], %8, %5\nv_cndmask_b32_e64 %5, %9, %5, s[20:21]", "v_cmp_lt_f32_e64 s[20:21], %8, %6\nv_cndmask_b32_e64 %6, %9, %6, s[20:21]", "v_cmp_lt_f32_e64 s[20:21], %8, %7\nv_cndmask_b32_e64 %7, %9, %7, s[20:21]")
F32_KERNEL(k_trunc_f32, "v_trunc_f32 %0, %0", "v_trunc_f32 %1, %1", "v_trunc_f32 %2, %2", "v_trunc_f32 %3, %3", "v_trunc_f32 %4, %4", "v_trunc_f32 %5, %5", "v_trunc_f32 %6, %6", "v_trunc_f32 %7, %7")
F32_KERNEL(k_and_or, "v_and_or_b32 %0, %0, %8, %9", "v_and_or_b32 %1, %1, %8, %9", "v_and_or_b32 %2, %2, %8, %9", "v_and_or_b32 %3, %3, %8, %9", "v_and_or_b32 %4, %4, %8, %9", "v_and_or_b32 %5, %5, %8, %9", "v_and_or_b32 %6, %6, %8, %9", "v_and_or_b32 %7, %7, %8, %9")
F32_KERNEL(k_min_u32, "v_min_u32 %0, %8, %0", "v_min_u32 %1, %8, %1", "v_min_u32 %2, %8, %2", "v_min_u32 %3, %8, %3", "v_min_u32 %4, %8, %4", "v_min_u32 %5, %8, %5", "v_min_u32 %6, %8, %6", "v_min_u32 %7, %8, %7")
F32_KERNEL(k_sub_f32_lit, "v_sub_f32 %0, 0x3f9e3779, %0", "v_sub_f32 %1, 0x3f9e3779, %1", "v_sub_f32 %2, 0x3f9e3779, %2", "v_sub_f32 %3, 0x3f9e3779, %3", "v_sub_f32 %4, 0x3f9e3779, %4", "v_sub_f32 %5, 0x3f9e3779, %5", "v_sub_f32 %6, 0x3f9e3779, %6", "v_sub_f32 %7, 0x3f9e3779, %7")
F32_KERNEL(k_fmac_f32, "v_fmac_f32 %0, %8, %9", "v_fmac_f32 %1, %8, %9", "v_fmac_f32 %2, %8, %9", "v_fmac_f32 %3, %8, %9", "v_fmac_f32 %4, %8, %9", "v_fmac_f32 %5, %8, %9", "v_fmac_f32 %6, %8, %9", "v_fmac_f32 %7, %8, %9")
F32_KERNEL(k_mad_lit, "v_fmaak_f32 %0, %8, %0, 0x3f9e3779", "v_fmaak_f32 %1, %8, %1, 0x3f9e3779", "v_fmaak_f32 %2, %8, %2, 0x3f9e3779", "v_fmaak_f32 %3, %8, %3, 0x3f9e3779", "v_fmaak_f32 %4, %8, %4, 0x3f9e3779", "v_fmaak_f32 %5, %8, %5, 0x3f9e3779", "v_fmaak_f32 %6, %8, %6, 0x3f9e3779", "v_fmaak_f32 %7, %8, %7, 0x3f9e3779")

F32_KERNEL(k_bfi_b32, "v_bfi_b32 %0, %8, %9, %0", "v_bfi_b32 %1, %8, %9, %1", "v_bfi_b32 %2, %8, %9, %2", "v_bfi_b32 %3, %8, %9, %3", "v_bfi_b32 %4, %8, %9, %4", "v_bfi_b32 %5, %8, %9, %5", "v_bfi_b32 %6, %8, %9, %6", "v_bfi_b32 %7, %8, %9, %7")
F32_KERNEL(k_xor_b32, "v_xor_b32 %0, %8, %0", "v_xor_b32 %1, %8, %1", "v_xor_b32 %2, %8, %2", "v_xor_b32 %3, %8, %3", "v_xor_b32 %4, %8, %4", "v_xor_b32 %5, %8, %5", "v_xor_b32 %6, %8, %6", "v_xor_b32 %7, %8, %7")
F32_KERNEL(k_perm_b32, "v_perm_b32 %0, %0, %8, %9", "v_perm_b32 %1, %1, %8, %9", "v_perm_b32 %2, %2, %8, %9", "v_perm_b32 %3, %3, %8, %9", "v_perm_b32 %4, %4, %8, %9", "v_perm_b32 %5, %5, %8, %9", "v_perm_b32 %6, %6, %8, %9", "v_perm_b32 %7, %7, %8, %9")
F32_KERNEL(k_add_f32_dpp, "v_add_f32_dpp %0, %0, %8 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %1, %1, %8 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %2, %2, %8 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %3, %3, %8 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %4, %4, %8 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %5, %5, %8 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %6, %6, %8 row_shr:1 row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %7, %7, %8 row_shr:1 row_mask:0xf bank_mask:0xf")
F32_KERNEL(k_add_f32_clamp, "v_add_f32_e64 %0, %8, %0 clamp", "v_add_f32_e64 %1, %8, %1 clamp", "v_add_f32_e64 %2, %8, %2 clamp", "v_add_f32_e64 %3, %8, %3 clamp", "v_add_f32_e64 %4, %8, %4 clamp", "v_add_f32_e64 %5, %8, %5 clamp", "v_add_f32_e64 %6, %8, %6 clamp", "v_add_f32_e64 %7, %8, %7 clamp")
F32_KERNEL(k_mul_f32_neg, "v_mul_f32_e64 %0, -%8, %0", "v_mul_f32_e64 %1, -%8, %1", "v_mul_f32_e64 %2, -%8, %2", "v_mul_f32_e64 %3, -%8, %3", "v_mul_f32_e64 %4, -%8, %4", "v_mul_f32_e64 %5, -%8, %5", "v_mul_f32_e64 %6, -%8, %6", "v_mul_f32_e64 %7, -%8, %7")
F32_KERNEL(k_max_u32, "v_max_u32 %0, %8, %0", "v_max_u32 %1, %8, %1", "v_max_u32 %2, %8, %2", "v_max_u32 %3, %8, %3", "v_max_u32 %4, %8, %4", "v_max_u32 %5, %8, %5", "v_max_u32 %6, %8, %6", "v_max_u32 %7, %8, %7")
F32_KERNEL(k_max_i32, "v_max_i32 %0, %8, %0", "v_max_i32 %1, %8, %1", "v_max_i32 %2, %8, %2", "v_max_i32 %3, %8, %3", "v_max_i32 %4, %8, %4", "v_max_i32 %5, %8, %5", "v_max_i32 %6, %8, %6", "v_max_i32 %7, %8, %7")
F32_KERNEL(k_sub_u32, "v_sub_u32 %0, %8, %0", "v_sub_u32 %1, %8, %1", "v_sub_u32 %2, %8, %2", "v_sub_u32 %3, %8, %3", "v_sub_u32 %4, %8, %4", "v_sub_u32 %5, %8, %5", "v_sub_u32 %6, %8, %6", "v_sub_u32 %7, %8, %7")
F32_KERNEL(k_add3_u32, "v_add3_u32 %0, %0, %8, %9", "v_add3_u32 %1, %1, %8, %9", "v_add3_u32 %2, %2, %8, %9", "v_add3_u32 %3, %3, %8, %9", "v_add3_u32 %4, %4, %8, %9", "v_add3_u32 %5, %5, %8, %9", "v_add3_u32 %6, %6, %8, %9", "v_add3_u32 %7, %7, %8, %9")
F32_KERNEL(k_alignbit, "v_alignbit_b32 %0, %0, %8, 7", "v_alignbit_b32 %1, %1, %8, 7", "v_alignbit_b32 %2, %2, %8, 7", "v_alignbit_b32 %3, %3, %8, 7", "v_alignbit_b32 %4, %4, %8, 7", "v_alignbit_b32 %5, %5, %8, 7", "v_alignbit_b32 %6, %6, %8, 7", "v_alignbit_b32 %7, %7, %8, 7")
F32_KERNEL(k_ldexp_f32, "v_ldexp_f32 %0, %0, 1", "v_ldexp_f32 %1, %1, 1", "v_ldexp_f32 %2, %2, 1", "v_ldexp_f32 %3, %3, 1", "v_ldexp_f32 %4, %4, 1", "v_ldexp_f32 %5, %5, 1", "v_ldexp_f32 %6, %6, 1", "v_ldexp_f32 %7, %7, 1")
F32_KERNEL(k_mul_u32_u24, "v_mul_u32_u24 %0, %8, %0", "v_mul_u32_u24 %1, %8, %1", "v_mul_u32_u24 %2, %8, %2", "v_mul_u32_u24 %3, %8, %3", "v_mul_u32_u24 %4, %8, %4", "v_mul_u32_u24 %5, %8, %5", "v_mul_u32_u24 %6, %8, %6", "v_mul_u32_u24 %7, %8, %7")
F32_KERNEL(k_or3, "v_or3_b32 %0, %0, %8, %9", "v_or3_b32 %1, %1, %8, %9", "v_or3_b32 %2, %2, %8, %9", "v_or3_b32 %3, %3, %8, %9", "v_or3_b32 %4, %4, %8, %9", "v_or3_b32 %5, %5, %8, %9", "v_or3_b32 %6, %6, %8, %9", "v_or3_b32 %7, %7, %8, %9")
F32_KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 3, %8", "v_lshl_or_b32 %1, %1, 3, %8", "v_lshl_or_b32 %2, %2, 3, %8", "v_lshl_or_b32 %3, %3, 3, %8", "v_lshl_or_b32 %4, %4, 3, %8", "v_lshl_or_b32 %5, %5, 3, %8", "v_lshl_or_b32 %6, %6, 3, %8", "v_lshl_or_b32 %7, %7, 3, %8")
F32_KERNEL(k_fma_f32_neg, "v_fma_f32 %0, -%8, %9, %0", "v_fma_f32 %1, -%8, %9, %1", "v_fma_f32 %2, -%8, %9, %2", "v_fma_f32 %3, -%8, %9, %3", "v_fma_f32 %4, -%8, %9, %4", "v_fma_f32 %5, -%8, %9, %5", "v_fma_f32 %6, -%8, %9, %6", "v_fma_f32 %7, -%8, %9, %7")
F32_KERNEL(k_cvt_ubyte, "v_cvt_f32_ubyte0 %0, %0", "v_cvt_f32_ubyte0 %1, %1", "v_cvt_f32_ubyte0 %2, %2", "v_cvt_f32_ubyte0 %3, %3", "v_cvt_f32_ubyte0 %4, %4", "v_cvt_f32_ubyte0 %5, %5", "v_cvt_f32_ubyte0 %6, %6", "v_cvt_f32_ubyte0 %7, %7")
F32_KERNEL(k_floor_f32, "v_floor_f32 %0, %0", "v_floor_f32 %1, %1", "v_floor_f32 %2, %2", "v_floor_f32 %3, %3", "v_floor_f32 %4, %4", "v_floor_f32 %5, %5", "v_floor_f32 %6, %6", "v_floor_f32 %7, %7")
F32_KERNEL(k_frexp_mant, "v_frexp_mant_f32 %0, %0", "v_frexp_mant_f32 %1, %1", "v_frexp_mant_f32 %2, %2", "v_frexp_mant_f32 %3, %3", "v_frexp_mant_f32 %4, %4", "v_frexp_mant_f32 %5, %5", "v_frexp_mant_f32 %6, %6", "v_frexp_mant_f32 %7, %7")

F32_KERNEL(k_or_b32, "v_or_b32 %0, %8, %0", "v_or_b32 %1, %8, %1", "v_or_b32 %2, %8, %2", "v_or_b32 %3, %8, %3", "v_or_b32 %4, %8, %4", "v_or_b32 %5, %8, %5", "v_or_b32 %6, %8, %6", "v_or_b32 %7, %8, %7")
F32_KERNEL(k_sub_f32, "v_sub_f32 %0, %8, %0", "v_sub_f32 %1, %8, %1", "v_sub_f32 %2, %8, %2", "v_sub_f32 %3, %8, %3", "v_sub_f32 %4, %8, %4", "v_sub_f32 %5, %8, %5", "v_sub_f32 %6, %8, %6", "v_sub_f32 %7, %8, %7")
F32_KERNEL(k_lshl_b32, "v_lshlrev_b32 %0, 3, %0", "v_lshlrev_b32 %1, 3, %1", "v_lshlrev_b32 %2, 3, %2", "v_lshlrev_b32 %3, 3, %3", "v_lshlrev_b32 %4, 3, %4", "v_lshlrev_b32 %5, 3, %5", "v_lshlrev_b32 %6, 3, %6", "v_lshlrev_b32 %7, 3, %7")
F32_KERNEL(k_ashr_i32, "v_ashrrev_i32 %0, 3, %0", "v_ashrrev_i32 %1, 3, %1", "v_ashrrev_i32 %2, 3, %2", "v_ashrrev_i32 %3, 3, %3", "v_ashrrev_i32 %4, 3, %4", "v_ashrrev_i32 %5, 3, %5", "v_ashrrev_i32 %6, 3, %6", "v_ashrrev_i32 %7, 3, %7")
F32_KERNEL(k_mul_e64, "v_mul_f32_e64 %0, %8, %0", "v_mul_f32_e64 %1, %8, %1", "v_mul_f32_e64 %2, %8, %2", "v_mul_f32_e64 %3, %8, %3", "v_mul_f32_e64 %4, %8, %4", "v_mul_f32_e64 %5, %8, %5", "v_mul_f32_e64 %6, %8, %6", "v_mul_f32_e64 %7, %8, %7")
F32_KERNEL(k_add_abs, "v_add_f32_e64 %0, |%8|, %0", "v_add_f32_e64 %1, |%8|, %1", "v_add_f32_e64 %2, |%8|, %2", "v_add_f32_e64 %3, |%8|, %3", "v_add_f32_e64 %4, |%8|, %4", "v_add_f32_e64 %5, |%8|, %5", "v_add_f32_e64 %6, |%8|, %6", "v_add_f32_e64 %7, |%8|, %7")
F32_KERNEL(k_mul_omod, "v_mul_f32_e64 %0, %8, %0 mul:2", "v_mul_f32_e64 %1, %8, %1 mul:2", "v_mul_f32_e64 %2, %8, %2 mul:2", "v_mul_f32_e64 %3, %8, %3 mul:2", "v_mul_f32_e64 %4, %8, %4 mul:2", "v_mul_f32_e64 %5, %8, %5 mul:2", "v_mul_f32_e64 %6, %8, %6 mul:2", "v_mul_f32_e64 %7, %8, %7 mul:2")
F32_KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %8, %9", "v_mad_u32_u24 %1, %1, %8, %9", "v_mad_u32_u24 %2, %2, %8, %9", "v_mad_u32_u24 %3, %3, %8, %9", "v_mad_u32_u24 %4, %4, %8, %9", "v_mad_u32_u24 %5, %5, %8, %9", "v_mad_u32_u24 %6, %6, %8, %9", "v_mad_u32_u24 %7, %7, %8, %9")
F32_KERNEL(k_sdwa_lshl, "v_lshlrev_b32_sdwa %0, %8, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "v_lshlrev_b32_sdwa %1, %8, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "v_lshlrev_b32_sdwa %2, %8, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "v_lshlrev_b32_sdwa %3, %8, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "v_lshlrev_b32_sdwa %4, %8, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "v_lshlrev_b32_sdwa %5, %8, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "v_lshlrev_b32_sdwa %6, %8, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0", "v_lshlrev_b32_sdwa %7, %8, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0")
F32_KERNEL(k_mov_sdwa, "v_mov_b32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_mov_b32_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_mov_b32_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_mov_b32_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_mov_b32_sdwa %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_mov_b32_sdwa %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_mov_b32_sdwa %6, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1", "v_mov_b32_sdwa %7, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
F32_KERNEL(k_cvt_pk_u16, "v_cvt_pknorm_u16_f32 %0, %0, %8", "v_cvt_pknorm_u16_f32 %1, %1, %8", "v_cvt_pknorm_u16_f32 %2, %2, %8", "v_cvt_pknorm_u16_f32 %3, %3, %8", "v_cvt_pknorm_u16_f32 %4, %4, %8", "v_cvt_pknorm_u16_f32 %5, %5, %8", "v_cvt_pknorm_u16_f32 %6, %6, %8", "v_cvt_pknorm_u16_f32 %7, %7, %8")
F32_KERNEL(k_fract, "v_fract_f32 %0, %0", "v_fract_f32 %1, %1", "v_fract_f32 %2, %2", "v_fract_f32 %3, %3", "v_fract_f32 %4, %4", "v_fract_f32 %5, %5", "v_fract_f32 %6, %6", "v_fract_f32 %7, %7")
F32_KERNEL(k_rndne, "v_rndne_f32 %0, %0", "v_rndne_f32 %1, %1", "v_rndne_f32 %2, %2", "v_rndne_f32 %3, %3", "v_rndne_f32 %4, %4", "v_rndne_f32 %5, %5", "v_rndne_f32 %6, %6", "v_rndne_f32 %7, %7")
F32_KERNEL(k_min3, "v_min3_f32 %0, %0, %8, %9", "v_min3_f32 %1, %1, %8, %9", "v_min3_f32 %2, %2, %8, %9", "v_min3_f32 %3, %3, %8, %9", "v_min3_f32 %4, %4, %8, %9", "v_min3_f32 %5, %5, %8, %9", "v_min3_f32 %6, %6, %8, %9", "v_min3_f32 %7, %7, %8, %9")
F32_KERNEL(k_xad, "v_xad_u32 %0, %0, %8, %9", "v_xad_u32 %1, %1, %8, %9", "v_xad_u32 %2, %2, %8, %9", "v_xad_u32 %3, %3, %8, %9", "v_xad_u32 %4, %4, %8, %9", "v_xad_u32 %5, %5, %8, %9", "v_xad_u32 %6, %6, %8, %9", "v_xad_u32 %7, %7, %8, %9")
F32_KERNEL(k_add_lshl, "v_add_lshl_u32 %0, %0, %8, 3", "v_add_lshl_u32 %1, %1, %8, 3", "v_add_lshl_u32 %2, %2, %8, 3", "v_add_lshl_u32 %3, %3, %8, 3", "v_add_lshl_u32 %4, %4, %8, 3", "v_add_lshl_u32 %5, %5, %8, 3", "v_add_lshl_u32 %6, %6, %8, 3", "v_add_lshl_u32 %7, %7, %8, 3")
F32_KERNEL(k_sad, "v_sad_u32 %0, %0, %8, %9", "v_sad_u32 %1, %1, %8, %9", "v_sad_u32 %2, %2, %8, %9", "v_sad_u32 %3, %3, %8, %9", "v_sad_u32 %4, %4, %8, %9", "v_sad_u32 %5, %5, %8, %9", "v_sad_u32 %6, %6, %8, %9", "v_sad_u32 %7, %7, %8, %9")
F32_KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %8", "v_mul_lo_u32 %1, %1, %8", "v_mul_lo_u32 %2, %2, %8", "v_mul_lo_u32 %3, %3, %8", "v_mul_lo_u32 %4, %4, %8", "v_mul_lo_u32 %5, %5, %8", "v_mul_lo_u32 %6, %6, %8", "v_mul_lo_u32 %7, %7, %8")
F32_KERNEL(k_cvt_i32, "v_cvt_i32_f32 %0, %0", "v_cvt_i32_f32 %1, %1", "v_cvt_i32_f32 %2, %2", "v_cvt_i32_f32 %3, %3", "v_cvt_i32_f32 %4, %4", "v_cvt_i32_f32 %5, %5", "v_cvt_i32_f32 %6, %6", "v_cvt_i32_f32 %7, %7")
F32_KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0", "v_cvt_f32_i32 %1, %1", "v_cvt_f32_i32 %2, %2", "v_cvt_f32_i32 %3, %3", "v_cvt_f32_i32 %4, %4", "v_cvt_f32_i32 %5, %5", "v_cvt_f32_i32 %6, %6", "v_cvt_f32_i32 %7, %7")
F32_KERNEL(k_dpp_quad, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
F32_KERNEL(k_dpp_rowshr, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf")
F32_KERNEL(k_mul_dpp, "v_mul_f32_dpp %0, %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %1, %1, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %2, %2, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %3, %3, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %4, %4, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %5, %5, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %6, %6, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mul_f32_dpp %7, %7, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
F32_KERNEL(k_subrev, "v_subrev_f32 %0, %8, %0", "v_subrev_f32 %1, %8, %1", "v_subrev_f32 %2, %8, %2", "v_subrev_f32 %3, %8, %3", "v_subrev_f32 %4, %8, %4", "v_subrev_f32 %5, %8, %5", "v_subrev_f32 %6, %8, %6", "v_subrev_f32 %7, %8, %7")
F32_KERNEL(k_mac_2lit, "v_fmamk_f32 %0, %8, 0x3f9e3779, %0", "v_fmamk_f32 %1, %8, 0x3f9e3779, %1", "v_fmamk_f32 %2, %8, 0x3f9e3779, %2", "v_fmamk_f32 %3, %8, 0x3f9e3779, %3", "v_fmamk_f32 %4, %8, 0x3f9e3779, %4", "v_fmamk_f32 %5, %8, 0x3f9e3779, %5", "v_fmamk_f32 %6, %8, 0x3f9e3779, %6", "v_fmamk_f32 %7, %8, 0x3f9e3779, %7")
F32_KERNEL(k_cmp_u32, "v_cmp_lt_u32 vcc, %8, %0", "v_cmp_lt_u32 vcc, %8, %1", "v_cmp_lt_u32 vcc, %8, %2", "v_cmp_lt_u32 vcc, %8, %3", "v_cmp_lt_u32 vcc, %8, %4", "v_cmp_lt_u32 vcc, %8, %5", "v_cmp_lt_u32 vcc, %8, %6", "v_cmp_lt_u32 vcc, %8, %7")
F32_KERNEL(k_cmp_class, "v_cmp_class_f32 vcc, %0, %8", "v_cmp_class_f32 vcc, %1, %8", "v_cmp_class_f32 vcc, %2, %8", "v_cmp_class_f32 vcc, %3, %8", "v_cmp_class_f32 vcc, %4, %8", "v_cmp_class_f32 vcc, %5, %8", "v_cmp_class_f32 vcc, %6, %8", "v_cmp_class_f32 vcc, %7, %8")
F32_KERNEL(k_swap, "v_swap_b32 %0, %0", "v_swap_b32 %1, %1", "v_swap_b32 %2, %2", "v_swap_b32 %3, %3", "v_swap_b32 %4, %4", "v_swap_b32 %5, %5", "v_swap_b32 %6, %6", "v_swap_b32 %7, %7")

// 64-bit operand kernels: accumulators are doubles / float2
#define F64_KERNEL(NAME, T, INIT, I0, I1, I2, I3, I4, I5, I6, I7)                                           \
    __global__ void NAME(float* out, long long* cyc, T x, T y)                                               \
    {                                                                                                       \
        T a0 = INIT(0), a1 = INIT(1), a2 = INIT(2), a3 = INIT(3), a4 = INIT(4), a5 = INIT(5), a6 = INIT(6),  \
          a7 = INIT(7);                                                                                     \
        __syncthreads();                                                                                    \
        long long t0 = __builtin_amdgcn_s_memtime();                                                        \
        for (int i = 0; i < ITERS; ++i) {                                                                   \
            asm volatile(I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"                     \
                         I0 "\n" I1 "\n" I2 "\n" I3 "\n" I4 "\n" I5 "\n" I6 "\n" I7 "\n"                     \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)    \
                         : "v"(x), "v"(y));                                                                 \
        }                                                                                                   \
        long long t1 = __builtin_amdgcn_s_memtime();                                                        \
        out[blockIdx.x * blockDim.x + threadIdx.x] = SUM8;                                                  \
        if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;            \
    }

#define DINIT(k) ((double)threadIdx.x + k)
#define SUM8 (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
F64_KERNEL(k_fma_f64, double, DINIT, "v_fma_f64 %0, %8, %9, %0", "v_fma_f64 %1, %8, %9, %1", "v_fma_f64 %2, %8, %9, %2",
           "v_fma_f64 %3, %8, %9, %3", "v_fma_f64 %4, %8, %9, %4", "v_fma_f64 %5, %8, %9, %5",
           "v_fma_f64 %6, %8, %9, %6", "v_fma_f64 %7, %8, %9, %7")
F64_KERNEL(k_mul_f64, double, DINIT, "v_mul_f64 %0, %8, %0", "v_mul_f64 %1, %8, %1", "v_mul_f64 %2, %8, %2",
           "v_mul_f64 %3, %8, %3", "v_mul_f64 %4, %8, %4", "v_mul_f64 %5, %8, %5", "v_mul_f64 %6, %8, %6",
           "v_mul_f64 %7, %8, %7")
F64_KERNEL(k_add_f64, double, DINIT, "v_add_f64 %0, %8, %0", "v_add_f64 %1, %8, %1", "v_add_f64 %2, %8, %2",
           "v_add_f64 %3, %8, %3", "v_add_f64 %4, %8, %4", "v_add_f64 %5, %8, %5", "v_add_f64 %6, %8, %6",
           "v_add_f64 %7, %8, %7")
F64_KERNEL(k_fma_f64_sgpr, double, DINIT, "v_fma_f64 %0, s[20:21], %9, %0", "v_fma_f64 %1, s[20:21], %9, %1", "v_fma_f64 %2, s[20:21], %9, %2",
           "v_fma_f64 %3, s[20:21], %9, %3", "v_fma_f64 %4, s[20:21], %9, %4", "v_fma_f64 %5, s[20:21], %9, %5",
           "v_fma_f64 %6, s[20:21], %9, %6", "v_fma_f64 %7, s[20:21], %9, %7")
F64_KERNEL(k_lshl_add_u64, double, DINIT, "v_lshl_add_u64 %0, %0, 1, %8", "v_lshl_add_u64 %1, %1, 1, %8", "v_lshl_add_u64 %2, %2, 1, %8",
           "v_lshl_add_u64 %3, %3, 1, %8", "v_lshl_add_u64 %4, %4, 1, %8", "v_lshl_add_u64 %5, %5, 1, %8",
           "v_lshl_add_u64 %6, %6, 1, %8", "v_lshl_add_u64 %7, %7, 1, %8")
#undef SUM8
typedef float f2 __attribute__((ext_vector_type(2)));
#define PINIT(k) (f2{(float)threadIdx.x + k, (float)threadIdx.x - k})
#define SUM8 (a0.x + a1.y + a2.x + a3.y + a4.x + a5.y + a6.x + a7.y)
F64_KERNEL(k_pk_fma_f32, f2, PINIT, "v_pk_fma_f32 %0, %8, %9, %0", "v_pk_fma_f32 %1, %8, %9, %1", "v_pk_fma_f32 %2, %8, %9, %2",
           "v_pk_fma_f32 %3, %8, %9, %3", "v_pk_fma_f32 %4, %8, %9, %4", "v_pk_fma_f32 %5, %8, %9, %5",
           "v_pk_fma_f32 %6, %8, %9, %6", "v_pk_fma_f32 %7, %8, %9, %7")
F64_KERNEL(k_pk_mul_f32, f2, PINIT, "v_pk_mul_f32 %0, %8, %0", "v_pk_mul_f32 %1, %8, %1", "v_pk_mul_f32 %2, %8, %2",
           "v_pk_mul_f32 %3, %8, %3", "v_pk_mul_f32 %4, %8, %4", "v_pk_mul_f32 %5, %8, %5", "v_pk_mul_f32 %6, %8, %6",
           "v_pk_mul_f32 %7, %8, %7")
F64_KERNEL(k_pk_add_f32, f2, PINIT, "v_pk_add_f32 %0, %8, %0", "v_pk_add_f32 %1, %8, %1", "v_pk_add_f32 %2, %8, %2",
           "v_pk_add_f32 %3, %8, %3", "v_pk_add_f32 %4, %8, %4", "v_pk_add_f32 %5, %8, %5", "v_pk_add_f32 %6, %8, %6",
           "v_pk_add_f32 %7, %8, %7")
#undef SUM8

// conversions: separate source/destination widths
__global__ void k_cvt_f64_f32(float* out, long long* cyc, float x, float y)
{
    float s0 = threadIdx.x, s1 = s0 + 1, s2 = s0 + 2, s3 = s0 + 3;
    double d0, d1, d2, d3;
    __syncthreads();
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; ++i) {
        asm volatile("v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n"
                     "v_cvt_f32_f64 %4, %0\nv_cvt_f32_f64 %5, %1\nv_cvt_f32_f64 %6, %2\nv_cvt_f32_f64 %7, %3\n"
                     "v_cvt_f64_f32 %0, %4\nv_cvt_f64_f32 %1, %5\nv_cvt_f64_f32 %2, %6\nv_cvt_f64_f32 %3, %7\n"
                     "v_cvt_f32_f64 %4, %0\nv_cvt_f32_f64 %5, %1\nv_cvt_f32_f64 %6, %2\nv_cvt_f32_f64 %7, %3\n"
                     : "=&v"(d0), "=&v"(d1), "=&v"(d2), "=&v"(d3), "+v"(s0), "+v"(s1), "+v"(s2), "+v"(s3));
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s0 + s1 + s2 + s3 + x + y;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// LDS gathers: ds_read_u16 / ds_read_b64 / ds_read_b128 at pseudo-random addresses
template <int BYTES>
__global__ void k_lds_gather(float* out, long long* cyc, float x, float y)
{
    extern __shared__ unsigned char lds[];
    for (int i = threadIdx.x; i < 32768; i += blockDim.x) reinterpret_cast<unsigned*>(lds)[i] = i * 2654435761u;
    __syncthreads();
    unsigned idx = threadIdx.x * 2654435761u + blockIdx.x;
    unsigned acc = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            idx = idx * 1664525u + 1013904223u;
            const unsigned off = (idx >> 15) & (131072 - BYTES) & ~(BYTES - 1);
            if (BYTES == 2) acc += *reinterpret_cast<const unsigned short*>(lds + off);
            if (BYTES == 8) { uint2 v = *reinterpret_cast<const uint2*>(lds + off); acc += v.x ^ v.y; }
            if (BYTES == 16) { uint4 v = *reinterpret_cast<const uint4*>(lds + off); acc += v.x ^ v.y ^ v.z ^ v.w; }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)acc + x + y;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

// k blocks of 256 threads (one wave per SIMD each) per CU: a dynamic-LDS claim of 160 KiB / k caps the
// residency at exactly k blocks per CU, and grid = k * CUs fills every slot.
template <class K, class... A>
static void run(const char* name, K kernel, size_t min_lds, A... args)
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    printf("%-14s", name);
    for (int wps : {1, 2, 4, 8}) {
        size_t lds = (size_t)(160 * 1024) / wps - 64;
        if (lds < min_lds) { printf("        n/a        "); continue; }
        const int blocks = cus * wps;
        float* out; long long* cyc;
        (void)hipMalloc(&out, sizeof(float) * blocks * 256);
        (void)hipMalloc(&cyc, sizeof(long long) * blocks * 4);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, 0, out, cyc, args...);  // warm
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), lds, 0, out, cyc, args...);
        (void)hipEventRecord(e1, 0);
        (void)hipDeviceSynchronize();
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(blocks * 4);
        (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += v;
        // s_memtime ticks per wave-instruction per SIMD | wall ns per wave-instruction per SIMD
        printf(" %6.2f|%5.2fns", s / h.size() / (ITERS * 16.0) / wps, ms * 1e6 / (ITERS * 16.0) / wps);
        (void)hipFree(out); (void)hipFree(cyc);
    }
    printf("\n");
}

__global__ void k_clock(long long* o)
{
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < 200000; ++i) a = a * 1.0001f + 0.5f;
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { o[0] = t1 - t0; o[1] = r1 - r0; o[2] = (long long)a; }
}

int main(int argc, char**)
{
    long long* dclk; long long hclk[3];
    (void)hipMalloc(&dclk, 24);
    hipLaunchKernelGGL(k_clock, dim3(1), dim3(64), 0, 0, dclk);
    (void)hipMemcpy(hclk, dclk, 24, hipMemcpyDeviceToHost);
    printf("s_memtime ticks per s_memrealtime tick (100 MHz): %.3f -> s_memtime at %.1f MHz\n", (double)hclk[0] / hclk[1], 100.0 * hclk[0] / hclk[1]);
    printf("per wave-instruction per SIMD: s_memtime ticks | wall ns, at waves/SIMD = 1, 2, 4, 8\n");
    if (argc > 3) {   // fourth batch (r03): which opcodes / encodings issue at the full rate
        run("v_or_b32", k_or_b32, 0, 1.0001f, 0.5f);
        run("v_sub_f32", k_sub_f32, 0, 1.0001f, 0.5f);
        run("v_subrev_f32", k_subrev, 0, 1.0001f, 0.5f);
        run("v_lshlrev_b32", k_lshl_b32, 0, 1.0001f, 0.5f);
        run("v_ashrrev_i32", k_ashr_i32, 0, 1.0001f, 0.5f);
        run("v_mul_f32 e64", k_mul_e64, 0, 1.0001f, 0.5f);
        run("v_add_f32 |abs|", k_add_abs, 0, 1.0001f, 0.5f);
        run("v_mul_f32 mul:2", k_mul_omod, 0, 1.0001f, 0.5f);
        run("v_fmamk_f32", k_mac_2lit, 0, 1.0001f, 0.5f);
        run("v_mad_u32_u24", k_mad_u24, 0, 1.0001f, 0.5f);
        run("v_lshl sdwa", k_sdwa_lshl, 0, 1.0001f, 0.5f);
        run("v_mov sdwa", k_mov_sdwa, 0, 1.0001f, 0.5f);
        run("v_cvt_pknorm", k_cvt_pk_u16, 0, 1.0001f, 0.5f);
        run("v_fract_f32", k_fract, 0, 1.0001f, 0.5f);
        run("v_rndne_f32", k_rndne, 0, 1.0001f, 0.5f);
        run("v_min3_f32", k_min3, 0, 1.0001f, 0.5f);
        run("v_xad_u32", k_xad, 0, 1.0001f, 0.5f);
        run("v_add_lshl_u32", k_add_lshl, 0, 1.0001f, 0.5f);
        run("v_sad_u32", k_sad, 0, 1.0001f, 0.5f);
        run("v_mul_lo_u32", k_mul_lo, 0, 1.0001f, 0.5f);
        run("v_cvt_i32_f32", k_cvt_i32, 0, 1.0001f, 0.5f);
        run("v_cvt_f32_i32", k_cvt_f32_i32, 0, 1.0001f, 0.5f);
        run("dpp quad_perm", k_dpp_quad, 0, 1.0001f, 0.5f);
        run("dpp row_shr", k_dpp_rowshr, 0, 1.0001f, 0.5f);
        run("v_mul_f32 dpp", k_mul_dpp, 0, 1.0001f, 0.5f);
        run("v_cmp_lt_u32", k_cmp_u32, 0, 1.0001f, 0.5f);
        run("v_cmp_class", k_cmp_class, 0, 1.0001f, 0.5f);
        run("v_swap_b32", k_swap, 0, 1.0001f, 0.5f);
        return 0;
    }
    if (argc > 2) {   // third batch (r02): candidates for replacing half-rate selects / shuffles
        run("v_bfi_b32", k_bfi_b32, 0, 1.0001f, 0.5f);
        run("v_xor_b32", k_xor_b32, 0, 1.0001f, 0.5f);
        run("v_perm_b32", k_perm_b32, 0, 1.0001f, 0.5f);
        run("v_add_f32 dpp", k_add_f32_dpp, 0, 1.0001f, 0.5f);
        run("v_add_f32 clamp", k_add_f32_clamp, 0, 1.0001f, 0.5f);
        run("v_mul_f32 neg", k_mul_f32_neg, 0, 1.0001f, 0.5f);
        run("v_fma_f32 neg", k_fma_f32_neg, 0, 1.0001f, 0.5f);
        run("v_max_u32", k_max_u32, 0, 1.0001f, 0.5f);
        run("v_max_i32", k_max_i32, 0, 1.0001f, 0.5f);
        run("v_sub_u32", k_sub_u32, 0, 1.0001f, 0.5f);
        run("v_add3_u32", k_add3_u32, 0, 1.0001f, 0.5f);
        run("v_alignbit_b32", k_alignbit, 0, 1.0001f, 0.5f);
        run("v_ldexp_f32", k_ldexp_f32, 0, 1.0001f, 0.5f);
        run("v_mul_u32_u24", k_mul_u32_u24, 0, 1.0001f, 0.5f);
        run("v_or3_b32", k_or3, 0, 1.0001f, 0.5f);
        run("v_lshl_or_b32", k_lshl_or, 0, 1.0001f, 0.5f);
        run("v_cvt_f32_ubyte0", k_cvt_ubyte, 0, 1.0001f, 0.5f);
        run("v_floor_f32", k_floor_f32, 0, 1.0001f, 0.5f);
        run("v_frexp_mant_f32", k_frexp_mant, 0, 1.0001f, 0.5f);
        return 0;
    }
    if (argc > 1) {   // second batch (r01 late): forms of select / compare, a few more classes
        run("cnd e32 vcc", k_cnd_e32, 0, 1.0001f, 0.5f);
        run("cnd e64 vcc", k_cnd_e64vcc, 0, 1.0001f, 0.5f);
        run("cnd e64 sgpr", k_cndmask, 0, 1.0001f, 0.5f);
        run("cmp+cnd vcc x2", k_cmp_cnd_vcc, 0, 1.0001f, 0.5f);
        run("cmp+cnd sgpr x2", k_cmp_cnd_sgpr, 0, 1.0001f, 0.5f);
        run("v_trunc_f32", k_trunc_f32, 0, 1.0001f, 0.5f);
        run("v_and_or_b32", k_and_or, 0, 1.0001f, 0.5f);
        run("v_min_u32", k_min_u32, 0, 1.0001f, 0.5f);
        run("v_sub_f32 lit", k_sub_f32_lit, 0, 1.0001f, 0.5f);
        run("v_fmac_f32", k_fmac_f32, 0, 1.0001f, 0.5f);
        run("v_fmaak_f32", k_mad_lit, 0, 1.0001f, 0.5f);
        run("v_fma_f64 sgpr", k_fma_f64_sgpr, 0, 1.0001, 0.5);
        run("v_fma_f64", k_fma_f64, 0, 1.0001, 0.5);
        run("v_lshl_add_u64", k_lshl_add_u64, 0, 1.0001, 0.5);
        return 0;
    }
    run("v_fma_f32", k_fma_f32, 0, 1.0001f, 0.5f);
    run("v_mul_f32", k_mul_f32, 0, 1.0001f, 0.5f);
    run("v_add_f32", k_add_f32, 0, 1.0001f, 0.5f);
    run("v_pk_fma_f32", k_pk_fma_f32, 0, f2{1.0001f, 1.0f}, f2{0.5f, 0.25f});
    run("v_pk_mul_f32", k_pk_mul_f32, 0, f2{1.0001f, 1.0f}, f2{0.5f, 0.25f});
    run("v_pk_add_f32", k_pk_add_f32, 0, f2{1.0001f, 1.0f}, f2{0.5f, 0.25f});
    run("v_fma_f64", k_fma_f64, 0, 1.0001, 0.5);
    run("v_mul_f64", k_mul_f64, 0, 1.0001, 0.5);
    run("v_add_f64", k_add_f64, 0, 1.0001, 0.5);
    run("v_cvt f64<>f32", k_cvt_f64_f32, 0, 1.0f, 0.5f);
    run("v_rcp_f32", k_rcp_f32, 0, 1.0001f, 0.5f);
    run("v_sqrt_f32", k_sqrt_f32, 0, 1.0001f, 0.5f);
    run("v_exp_f32", k_exp_f32, 0, 1.0001f, 0.5f);
    run("v_cvt_u32_f32", k_cvt_u32, 0, 1.0001f, 0.5f);
    run("v_and_b32", k_and_b32, 0, 1.0001f, 0.5f);
    run("v_lshl_add_u32", k_lshl_add, 0, 1.0001f, 0.5f);
    run("v_cndmask_b32", k_cndmask, 0, 1.0001f, 0.5f);
    run("v_mov_dpp", k_dpp_mov, 0, 1.0001f, 0.5f);
    run("cndmask vcc", k_cnd_vcc, 0, 1.0001f, 0.5f);
    run("v_add_u32", k_add_u32, 0, 1.0001f, 0.5f);
    run("v_lshrrev_b32", k_lshr, 0, 1.0001f, 0.5f);
    run("v_max_f32", k_max_f32, 0, 1.0001f, 0.5f);
    run("v_med3_f32", k_med3_f32, 0, 1.0001f, 0.5f);
    run("v_bfe_u32", k_bfe_u32, 0, 1.0001f, 0.5f);
    run("v_cmp_lt_f32", k_cmp_f32, 0, 1.0001f, 0.5f);
    run("v_mov_b32", k_mov_b32, 0, 1.0001f, 0.5f);
    run("v_mul_f32 sgpr", k_mul_sgpr, 0, 1.0001f, 0.5f);
    run("v_mul_f32 lit", k_mul_lit, 0, 1.0001f, 0.5f);
    run("v_fma_f32 sgpr", k_fma_sgpr, 0, 1.0001f, 0.5f);
    run("v_readlane", k_readlane, 0, 1.0001f, 0.5f);
    run("v_cvt_f32_u32", k_cvt_f32_u32, 0, 1.0001f, 0.5f);
    printf("LDS random gathers over 128 KiB (cycles per wave-instruction per CU-quarter; includes ~4 address VALU ops)\n");
    run("ds_read_u16", k_lds_gather<2>, 131072, 1.0f, 0.5f);  // k = 1 only
    run("ds_read_b64", k_lds_gather<8>, 131072, 1.0f, 0.5f);
    run("ds_read_b128", k_lds_gather<16>, 131072, 1.0f, 0.5f);
    return 0;
}
