// tools/ubench_valu.hip — the measured instruction-issue ceiling behind `roofline_valu` (bench.py) and DESIGN.md §5.
//
// For every instruction class the fused kernels execute, a loop of that ONE instruction — pinned with inline asm (one asm
// statement = 16 instructions on eight independent register chains, so the compiler's hazard recognizer cannot pad them
// with s_nop), no global-memory traffic — is run at 1, 2, 4, 5 and 8 resident waves per SIMD with every CU busy.
// Two time bases, because they answer different questions:
//   issue   s_memtime (shader cycles) around the loop of each wave, median over waves, / instructions per wave:
//           how often ONE wave gets to issue (>= 4-5 cycles per VALU instruction even on an otherwise idle SIMD);
//   SIMD    wall time (HIP events) x sustained clock x number of SIMDs / all instructions issued: SIMD cycles per wave64
//           instruction, independent of where the dispatcher put the blocks — THE CEILING a mix is priced with
//           (tools/valu_ceiling.py multiplies it with a kernel's dynamic SQ_INSTS_VALU_* counts).
// Sustained clock = s_memtime / s_memrealtime (100 MHz), median over waves.
//
// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o build/tools/ubench_valu && build/tools/ubench_valu
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float v4f __attribute__((ext_vector_type(4)));

enum Cls { FMA_F32, MUL_F32, ADD_F32, MAX_F32, CNDMASK, CMP_F32, CMP_CND, RCP_F32, RSQ_F32, SQRT_F32, SIN_F32, COS_F32, MUL_LO_U32, MUL_HI_U32, MAD_U64_U32, MUL_U32_U24, MAD_U32_U24, ADD_U32, LSHL_ADD, XOR_B32, AND_OR, CVT_F32_U32, CVT_U32_F32, FMA_F64, MUL_F64, ADD_F64, DIV_FIXUP, DIV_SCALE, DIV_FMAS, MOV_B32, MBCNT, READLANE, READFIRST, DPP_MOV, BPERMUTE, DS_READ_B32, DS_READ_B128_BCAST, DS_WRITE_B32, SALU, PK_FMA_F32, PK_MUL_F32, PK_ADD_F32, MAX3_F32, MIN3_F32, LSHL_ADD_U64, ALIGNBIT, BFE_U32, WRITELANE, SUB_F32, CMP_SAND, AND_B32, OR_B32, LSHL_B32, LSHR_B32, SUB_U32, MIN_F32, FMAC_F32, MIN_U32, ADD3_U32, CMP_VCC, CND_VCC, FFBL, BFI, NUM_CLS };
static const char* kNames[NUM_CLS] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_max_f32", "v_cndmask_b32 (mask in SGPRs)", "v_cmp_lt_f32 (to SGPR pair)", "v_cmp_lt_f32 + v_cndmask_b32 (vcc)", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mul_u32_u24", "v_mad_u32_u24", "v_add_u32", "v_lshl_add_u32", "v_xor_b32", "v_and_or_b32", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_fma_f64", "v_mul_f64", "v_add_f64", "v_div_fixup_f32", "v_div_scale_f32", "v_div_fmas_f32", "v_mov_b32", "v_mbcnt_lo_u32_b32", "v_readlane_b32", "v_readfirstlane_b32", "v_mov_b32 dpp row_shr:1", "ds_bpermute_b32", "ds_read_b32 (lane-consecutive)", "ds_read_b128 (uniform address)", "ds_write_b32 (lane-consecutive)", "s_and_b64 (SALU beside nothing)", "v_pk_fma_f32 (2 FMAs per lane)", "v_pk_mul_f32", "v_pk_add_f32", "v_max3_f32", "v_min3_f32", "v_lshl_add_u64", "v_alignbit_b32", "v_bfe_u32", "v_writelane_b32", "v_subrev_f32 / v_sub_f32", "v_cmp_lt_f32 -> s_and_b64 (VALU + SALU)", "v_and_b32", "v_or_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_sub_u32", "v_min_f32", "v_fmac_f32", "v_min_u32", "v_add3_u32", "v_cmp_lt_f32 (to vcc)", "v_cndmask_b32 (vcc)", "v_ffbl_b32", "v_bfi_b32"};

constexpr int kPerTrip = 64;  // 4 asm statements x 16 instructions (CMP_CND: 2 instructions per line, counted below)

template <int C>
__device__ __forceinline__ void body(float (&x)[8], double (&dd)[8], unsigned long long (&q)[8], v4f (&v4)[8], float a, float b) {
#define OPS_F "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])
#define OPS_D "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]), "+v"(dd[4]), "+v"(dd[5]), "+v"(dd[6]), "+v"(dd[7])
#define OPS_Q "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])
#define OPS_V4 "+v"(v4[0]), "+v"(v4[1]), "+v"(v4[2]), "+v"(v4[3]), "+v"(v4[4]), "+v"(v4[5]), "+v"(v4[6]), "+v"(v4[7])
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    if constexpr (C == FMA_F32) asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == MUL_F32) asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == ADD_F32) asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == MAX_F32) asm volatile("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]\n v_cndmask_b32 %0, %0, %8, s[20:21]\n v_cndmask_b32 %1, %1, %8, s[20:21]\n v_cndmask_b32 %2, %2, %8, s[20:21]\n v_cndmask_b32 %3, %3, %8, s[20:21]\n v_cndmask_b32 %4, %4, %8, s[20:21]\n v_cndmask_b32 %5, %5, %8, s[20:21]\n v_cndmask_b32 %6, %6, %8, s[20:21]\n v_cndmask_b32 %7, %7, %8, s[20:21]" : OPS_F : "v"(a), "v"(b) : "s20", "s21");
    if constexpr (C == CMP_F32) asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[20:21], %1, %8\n v_cmp_lt_f32 s[20:21], %2, %8\n v_cmp_lt_f32 s[20:21], %3, %8\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[20:21], %5, %8\n v_cmp_lt_f32 s[20:21], %6, %8\n v_cmp_lt_f32 s[20:21], %7, %8\n v_cmp_lt_f32 s[20:21], %0, %8\n v_cmp_lt_f32 s[20:21], %1, %8\n v_cmp_lt_f32 s[20:21], %2, %8\n v_cmp_lt_f32 s[20:21], %3, %8\n v_cmp_lt_f32 s[20:21], %4, %8\n v_cmp_lt_f32 s[20:21], %5, %8\n v_cmp_lt_f32 s[20:21], %6, %8\n v_cmp_lt_f32 s[20:21], %7, %8" : OPS_F : "v"(a), "v"(b) : "s20", "s21");
    if constexpr (C == CMP_CND) asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc\n v_cmp_lt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %9, vcc\n v_cmp_lt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %9, vcc\n v_cmp_lt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %9, vcc\n v_cmp_lt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %9, vcc\n v_cmp_lt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %9, vcc\n v_cmp_lt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %9, vcc\n v_cmp_lt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %9, vcc\n v_cmp_lt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %9, vcc" : OPS_F : "v"(a), "v"(b) : "vcc");
    if constexpr (C == RCP_F32) asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == RSQ_F32) asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == SQRT_F32) asm volatile("v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sqrt_f32 %2, %2\n v_sqrt_f32 %3, %3\n v_sqrt_f32 %4, %4\n v_sqrt_f32 %5, %5\n v_sqrt_f32 %6, %6\n v_sqrt_f32 %7, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == SIN_F32) asm volatile("v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n v_sin_f32 %4, %4\n v_sin_f32 %5, %5\n v_sin_f32 %6, %6\n v_sin_f32 %7, %7\n v_sin_f32 %0, %0\n v_sin_f32 %1, %1\n v_sin_f32 %2, %2\n v_sin_f32 %3, %3\n v_sin_f32 %4, %4\n v_sin_f32 %5, %5\n v_sin_f32 %6, %6\n v_sin_f32 %7, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == COS_F32) asm volatile("v_cos_f32 %0, %0\n v_cos_f32 %1, %1\n v_cos_f32 %2, %2\n v_cos_f32 %3, %3\n v_cos_f32 %4, %4\n v_cos_f32 %5, %5\n v_cos_f32 %6, %6\n v_cos_f32 %7, %7\n v_cos_f32 %0, %0\n v_cos_f32 %1, %1\n v_cos_f32 %2, %2\n v_cos_f32 %3, %3\n v_cos_f32 %4, %4\n v_cos_f32 %5, %5\n v_cos_f32 %6, %6\n v_cos_f32 %7, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == MUL_LO_U32) asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == MUL_HI_U32) asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == MAD_U64_U32) asm volatile("v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %8, %9, %1\n v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n v_mad_u64_u32 %3, s[20:21], %8, %9, %3\n v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n v_mad_u64_u32 %6, s[20:21], %8, %9, %6\n v_mad_u64_u32 %7, s[20:21], %8, %9, %7\n v_mad_u64_u32 %0, s[20:21], %8, %9, %0\n v_mad_u64_u32 %1, s[20:21], %8, %9, %1\n v_mad_u64_u32 %2, s[20:21], %8, %9, %2\n v_mad_u64_u32 %3, s[20:21], %8, %9, %3\n v_mad_u64_u32 %4, s[20:21], %8, %9, %4\n v_mad_u64_u32 %5, s[20:21], %8, %9, %5\n v_mad_u64_u32 %6, s[20:21], %8, %9, %6\n v_mad_u64_u32 %7, s[20:21], %8, %9, %7" : OPS_Q : "v"(a), "v"(b) : "s20", "s21");
    if constexpr (C == MUL_U32_U24) asm volatile("v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8\n v_mul_u32_u24 %0, %0, %8\n v_mul_u32_u24 %1, %1, %8\n v_mul_u32_u24 %2, %2, %8\n v_mul_u32_u24 %3, %3, %8\n v_mul_u32_u24 %4, %4, %8\n v_mul_u32_u24 %5, %5, %8\n v_mul_u32_u24 %6, %6, %8\n v_mul_u32_u24 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == MAD_U32_U24) asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == ADD_U32) asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %8\n v_lshl_add_u32 %1, %1, 3, %8\n v_lshl_add_u32 %2, %2, 3, %8\n v_lshl_add_u32 %3, %3, 3, %8\n v_lshl_add_u32 %4, %4, 3, %8\n v_lshl_add_u32 %5, %5, 3, %8\n v_lshl_add_u32 %6, %6, 3, %8\n v_lshl_add_u32 %7, %7, 3, %8\n v_lshl_add_u32 %0, %0, 3, %8\n v_lshl_add_u32 %1, %1, 3, %8\n v_lshl_add_u32 %2, %2, 3, %8\n v_lshl_add_u32 %3, %3, 3, %8\n v_lshl_add_u32 %4, %4, 3, %8\n v_lshl_add_u32 %5, %5, 3, %8\n v_lshl_add_u32 %6, %6, 3, %8\n v_lshl_add_u32 %7, %7, 3, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == XOR_B32) asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == AND_OR) asm volatile("v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9\n v_and_or_b32 %0, %0, %8, %9\n v_and_or_b32 %1, %1, %8, %9\n v_and_or_b32 %2, %2, %8, %9\n v_and_or_b32 %3, %3, %8, %9\n v_and_or_b32 %4, %4, %8, %9\n v_and_or_b32 %5, %5, %8, %9\n v_and_or_b32 %6, %6, %8, %9\n v_and_or_b32 %7, %7, %8, %9" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == CVT_F32_U32) asm volatile("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7\n v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n v_cvt_f32_u32 %4, %4\n v_cvt_f32_u32 %5, %5\n v_cvt_f32_u32 %6, %6\n v_cvt_f32_u32 %7, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == CVT_U32_F32) asm volatile("v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3\n v_cvt_u32_f32 %4, %4\n v_cvt_u32_f32 %5, %5\n v_cvt_u32_f32 %6, %6\n v_cvt_u32_f32 %7, %7\n v_cvt_u32_f32 %0, %0\n v_cvt_u32_f32 %1, %1\n v_cvt_u32_f32 %2, %2\n v_cvt_u32_f32 %3, %3\n v_cvt_u32_f32 %4, %4\n v_cvt_u32_f32 %5, %5\n v_cvt_u32_f32 %6, %6\n v_cvt_u32_f32 %7, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == FMA_F64) asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7\n v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n v_fma_f64 %4, %4, %4, %4\n v_fma_f64 %5, %5, %5, %5\n v_fma_f64 %6, %6, %6, %6\n v_fma_f64 %7, %7, %7, %7" : OPS_D : "v"(a), "v"(b));
    if constexpr (C == MUL_F64) asm volatile("v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3\n v_mul_f64 %4, %4, %4\n v_mul_f64 %5, %5, %5\n v_mul_f64 %6, %6, %6\n v_mul_f64 %7, %7, %7\n v_mul_f64 %0, %0, %0\n v_mul_f64 %1, %1, %1\n v_mul_f64 %2, %2, %2\n v_mul_f64 %3, %3, %3\n v_mul_f64 %4, %4, %4\n v_mul_f64 %5, %5, %5\n v_mul_f64 %6, %6, %6\n v_mul_f64 %7, %7, %7" : OPS_D : "v"(a), "v"(b));
    if constexpr (C == ADD_F64) asm volatile("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3\n v_add_f64 %4, %4, %4\n v_add_f64 %5, %5, %5\n v_add_f64 %6, %6, %6\n v_add_f64 %7, %7, %7\n v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3\n v_add_f64 %4, %4, %4\n v_add_f64 %5, %5, %5\n v_add_f64 %6, %6, %6\n v_add_f64 %7, %7, %7" : OPS_D : "v"(a), "v"(b));
    if constexpr (C == DIV_FIXUP) asm volatile("v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9\n v_div_fixup_f32 %0, %0, %8, %9\n v_div_fixup_f32 %1, %1, %8, %9\n v_div_fixup_f32 %2, %2, %8, %9\n v_div_fixup_f32 %3, %3, %8, %9\n v_div_fixup_f32 %4, %4, %8, %9\n v_div_fixup_f32 %5, %5, %8, %9\n v_div_fixup_f32 %6, %6, %8, %9\n v_div_fixup_f32 %7, %7, %8, %9" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == DIV_SCALE) asm volatile("v_div_scale_f32 %0, s[20:21], %0, %8, %9\n v_div_scale_f32 %1, s[20:21], %1, %8, %9\n v_div_scale_f32 %2, s[20:21], %2, %8, %9\n v_div_scale_f32 %3, s[20:21], %3, %8, %9\n v_div_scale_f32 %4, s[20:21], %4, %8, %9\n v_div_scale_f32 %5, s[20:21], %5, %8, %9\n v_div_scale_f32 %6, s[20:21], %6, %8, %9\n v_div_scale_f32 %7, s[20:21], %7, %8, %9\n v_div_scale_f32 %0, s[20:21], %0, %8, %9\n v_div_scale_f32 %1, s[20:21], %1, %8, %9\n v_div_scale_f32 %2, s[20:21], %2, %8, %9\n v_div_scale_f32 %3, s[20:21], %3, %8, %9\n v_div_scale_f32 %4, s[20:21], %4, %8, %9\n v_div_scale_f32 %5, s[20:21], %5, %8, %9\n v_div_scale_f32 %6, s[20:21], %6, %8, %9\n v_div_scale_f32 %7, s[20:21], %7, %8, %9" : OPS_F : "v"(a), "v"(b) : "s20", "s21");
    if constexpr (C == DIV_FMAS) asm volatile("v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n v_div_fmas_f32 %4, %4, %8, %9\n v_div_fmas_f32 %5, %5, %8, %9\n v_div_fmas_f32 %6, %6, %8, %9\n v_div_fmas_f32 %7, %7, %8, %9\n v_div_fmas_f32 %0, %0, %8, %9\n v_div_fmas_f32 %1, %1, %8, %9\n v_div_fmas_f32 %2, %2, %8, %9\n v_div_fmas_f32 %3, %3, %8, %9\n v_div_fmas_f32 %4, %4, %8, %9\n v_div_fmas_f32 %5, %5, %8, %9\n v_div_fmas_f32 %6, %6, %8, %9\n v_div_fmas_f32 %7, %7, %8, %9" : OPS_F : "v"(a), "v"(b) : "vcc");
    if constexpr (C == MOV_B32) asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == MBCNT) asm volatile("v_mbcnt_lo_u32_b32 %0, %8, %0\n v_mbcnt_lo_u32_b32 %1, %8, %1\n v_mbcnt_lo_u32_b32 %2, %8, %2\n v_mbcnt_lo_u32_b32 %3, %8, %3\n v_mbcnt_lo_u32_b32 %4, %8, %4\n v_mbcnt_lo_u32_b32 %5, %8, %5\n v_mbcnt_lo_u32_b32 %6, %8, %6\n v_mbcnt_lo_u32_b32 %7, %8, %7\n v_mbcnt_lo_u32_b32 %0, %8, %0\n v_mbcnt_lo_u32_b32 %1, %8, %1\n v_mbcnt_lo_u32_b32 %2, %8, %2\n v_mbcnt_lo_u32_b32 %3, %8, %3\n v_mbcnt_lo_u32_b32 %4, %8, %4\n v_mbcnt_lo_u32_b32 %5, %8, %5\n v_mbcnt_lo_u32_b32 %6, %8, %6\n v_mbcnt_lo_u32_b32 %7, %8, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == READLANE) asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s20, %1, 3\n v_readlane_b32 s20, %2, 3\n v_readlane_b32 s20, %3, 3\n v_readlane_b32 s20, %4, 3\n v_readlane_b32 s20, %5, 3\n v_readlane_b32 s20, %6, 3\n v_readlane_b32 s20, %7, 3\n v_readlane_b32 s20, %0, 3\n v_readlane_b32 s20, %1, 3\n v_readlane_b32 s20, %2, 3\n v_readlane_b32 s20, %3, 3\n v_readlane_b32 s20, %4, 3\n v_readlane_b32 s20, %5, 3\n v_readlane_b32 s20, %6, 3\n v_readlane_b32 s20, %7, 3" : OPS_F : "v"(a), "v"(b) : "s20");
    if constexpr (C == READFIRST) asm volatile("v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s20, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s20, %3\n v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s20, %5\n v_readfirstlane_b32 s20, %6\n v_readfirstlane_b32 s20, %7\n v_readfirstlane_b32 s20, %0\n v_readfirstlane_b32 s20, %1\n v_readfirstlane_b32 s20, %2\n v_readfirstlane_b32 s20, %3\n v_readfirstlane_b32 s20, %4\n v_readfirstlane_b32 s20, %5\n v_readfirstlane_b32 s20, %6\n v_readfirstlane_b32 s20, %7" : OPS_F : "v"(a), "v"(b) : "s20");
    if constexpr (C == DPP_MOV) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == BPERMUTE) asm volatile("ds_bpermute_b32 %0, %9, %0\n ds_bpermute_b32 %1, %9, %1\n ds_bpermute_b32 %2, %9, %2\n ds_bpermute_b32 %3, %9, %3\n ds_bpermute_b32 %4, %9, %4\n ds_bpermute_b32 %5, %9, %5\n ds_bpermute_b32 %6, %9, %6\n ds_bpermute_b32 %7, %9, %7\n ds_bpermute_b32 %0, %9, %0\n ds_bpermute_b32 %1, %9, %1\n ds_bpermute_b32 %2, %9, %2\n ds_bpermute_b32 %3, %9, %3\n ds_bpermute_b32 %4, %9, %4\n ds_bpermute_b32 %5, %9, %5\n ds_bpermute_b32 %6, %9, %6\n ds_bpermute_b32 %7, %9, %7\n s_waitcnt lgkmcnt(0)" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == DS_READ_B32) asm volatile("ds_read_b32 %0, %9 offset:0*256\n ds_read_b32 %1, %9 offset:1*256\n ds_read_b32 %2, %9 offset:2*256\n ds_read_b32 %3, %9 offset:3*256\n ds_read_b32 %4, %9 offset:4*256\n ds_read_b32 %5, %9 offset:5*256\n ds_read_b32 %6, %9 offset:6*256\n ds_read_b32 %7, %9 offset:7*256\n ds_read_b32 %0, %9 offset:0*256\n ds_read_b32 %1, %9 offset:1*256\n ds_read_b32 %2, %9 offset:2*256\n ds_read_b32 %3, %9 offset:3*256\n ds_read_b32 %4, %9 offset:4*256\n ds_read_b32 %5, %9 offset:5*256\n ds_read_b32 %6, %9 offset:6*256\n ds_read_b32 %7, %9 offset:7*256\n s_waitcnt lgkmcnt(0)" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == DS_READ_B128_BCAST) asm volatile("ds_read_b128 %0, %9 offset:0*16\n ds_read_b128 %1, %9 offset:1*16\n ds_read_b128 %2, %9 offset:2*16\n ds_read_b128 %3, %9 offset:3*16\n ds_read_b128 %4, %9 offset:4*16\n ds_read_b128 %5, %9 offset:5*16\n ds_read_b128 %6, %9 offset:6*16\n ds_read_b128 %7, %9 offset:7*16\n ds_read_b128 %0, %9 offset:0*16\n ds_read_b128 %1, %9 offset:1*16\n ds_read_b128 %2, %9 offset:2*16\n ds_read_b128 %3, %9 offset:3*16\n ds_read_b128 %4, %9 offset:4*16\n ds_read_b128 %5, %9 offset:5*16\n ds_read_b128 %6, %9 offset:6*16\n ds_read_b128 %7, %9 offset:7*16\n s_waitcnt lgkmcnt(0)" : OPS_V4 : "v"(a), "v"(b));
    if constexpr (C == DS_WRITE_B32) asm volatile("ds_write_b32 %9, %0 offset:0*256\n ds_write_b32 %9, %1 offset:1*256\n ds_write_b32 %9, %2 offset:2*256\n ds_write_b32 %9, %3 offset:3*256\n ds_write_b32 %9, %4 offset:4*256\n ds_write_b32 %9, %5 offset:5*256\n ds_write_b32 %9, %6 offset:6*256\n ds_write_b32 %9, %7 offset:7*256\n ds_write_b32 %9, %0 offset:0*256\n ds_write_b32 %9, %1 offset:1*256\n ds_write_b32 %9, %2 offset:2*256\n ds_write_b32 %9, %3 offset:3*256\n ds_write_b32 %9, %4 offset:4*256\n ds_write_b32 %9, %5 offset:5*256\n ds_write_b32 %9, %6 offset:6*256\n ds_write_b32 %9, %7 offset:7*256\n s_waitcnt lgkmcnt(0)" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == SALU) asm volatile("s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]\n s_and_b64 s[20:21], s[20:21], s[22:23]" : OPS_F : "v"(a), "v"(b) : "s20", "s21", "scc");
    if constexpr (C == PK_FMA_F32) asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7\n v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n v_pk_fma_f32 %4, %4, %4, %4\n v_pk_fma_f32 %5, %5, %5, %5\n v_pk_fma_f32 %6, %6, %6, %6\n v_pk_fma_f32 %7, %7, %7, %7" : OPS_D : "v"(a), "v"(b));
    if constexpr (C == PK_MUL_F32) asm volatile("v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n v_pk_mul_f32 %4, %4, %4\n v_pk_mul_f32 %5, %5, %5\n v_pk_mul_f32 %6, %6, %6\n v_pk_mul_f32 %7, %7, %7\n v_pk_mul_f32 %0, %0, %0\n v_pk_mul_f32 %1, %1, %1\n v_pk_mul_f32 %2, %2, %2\n v_pk_mul_f32 %3, %3, %3\n v_pk_mul_f32 %4, %4, %4\n v_pk_mul_f32 %5, %5, %5\n v_pk_mul_f32 %6, %6, %6\n v_pk_mul_f32 %7, %7, %7" : OPS_D : "v"(a), "v"(b));
    if constexpr (C == PK_ADD_F32) asm volatile("v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n v_pk_add_f32 %4, %4, %4\n v_pk_add_f32 %5, %5, %5\n v_pk_add_f32 %6, %6, %6\n v_pk_add_f32 %7, %7, %7\n v_pk_add_f32 %0, %0, %0\n v_pk_add_f32 %1, %1, %1\n v_pk_add_f32 %2, %2, %2\n v_pk_add_f32 %3, %3, %3\n v_pk_add_f32 %4, %4, %4\n v_pk_add_f32 %5, %5, %5\n v_pk_add_f32 %6, %6, %6\n v_pk_add_f32 %7, %7, %7" : OPS_D : "v"(a), "v"(b));
    if constexpr (C == MAX3_F32) asm volatile("v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == MIN3_F32) asm volatile("v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9\n v_min3_f32 %0, %0, %8, %9\n v_min3_f32 %1, %1, %8, %9\n v_min3_f32 %2, %2, %8, %9\n v_min3_f32 %3, %3, %8, %9\n v_min3_f32 %4, %4, %8, %9\n v_min3_f32 %5, %5, %8, %9\n v_min3_f32 %6, %6, %8, %9\n v_min3_f32 %7, %7, %8, %9" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == LSHL_ADD_U64) asm volatile("v_lshl_add_u64 %0, %0, 2, %0\n v_lshl_add_u64 %1, %1, 2, %1\n v_lshl_add_u64 %2, %2, 2, %2\n v_lshl_add_u64 %3, %3, 2, %3\n v_lshl_add_u64 %4, %4, 2, %4\n v_lshl_add_u64 %5, %5, 2, %5\n v_lshl_add_u64 %6, %6, 2, %6\n v_lshl_add_u64 %7, %7, 2, %7\n v_lshl_add_u64 %0, %0, 2, %0\n v_lshl_add_u64 %1, %1, 2, %1\n v_lshl_add_u64 %2, %2, 2, %2\n v_lshl_add_u64 %3, %3, 2, %3\n v_lshl_add_u64 %4, %4, 2, %4\n v_lshl_add_u64 %5, %5, 2, %5\n v_lshl_add_u64 %6, %6, 2, %6\n v_lshl_add_u64 %7, %7, 2, %7" : OPS_Q : "v"(a), "v"(b));
    if constexpr (C == ALIGNBIT) asm volatile("v_alignbit_b32 %0, %0, %8, 15\n v_alignbit_b32 %1, %1, %8, 15\n v_alignbit_b32 %2, %2, %8, 15\n v_alignbit_b32 %3, %3, %8, 15\n v_alignbit_b32 %4, %4, %8, 15\n v_alignbit_b32 %5, %5, %8, 15\n v_alignbit_b32 %6, %6, %8, 15\n v_alignbit_b32 %7, %7, %8, 15\n v_alignbit_b32 %0, %0, %8, 15\n v_alignbit_b32 %1, %1, %8, 15\n v_alignbit_b32 %2, %2, %8, 15\n v_alignbit_b32 %3, %3, %8, 15\n v_alignbit_b32 %4, %4, %8, 15\n v_alignbit_b32 %5, %5, %8, 15\n v_alignbit_b32 %6, %6, %8, 15\n v_alignbit_b32 %7, %7, %8, 15" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == BFE_U32) asm volatile("v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9\n v_bfe_u32 %4, %4, 3, 9\n v_bfe_u32 %5, %5, 3, 9\n v_bfe_u32 %6, %6, 3, 9\n v_bfe_u32 %7, %7, 3, 9\n v_bfe_u32 %0, %0, 3, 9\n v_bfe_u32 %1, %1, 3, 9\n v_bfe_u32 %2, %2, 3, 9\n v_bfe_u32 %3, %3, 3, 9\n v_bfe_u32 %4, %4, 3, 9\n v_bfe_u32 %5, %5, 3, 9\n v_bfe_u32 %6, %6, 3, 9\n v_bfe_u32 %7, %7, 3, 9" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == WRITELANE) asm volatile("v_writelane_b32 %0, s20, 5\n v_writelane_b32 %1, s20, 5\n v_writelane_b32 %2, s20, 5\n v_writelane_b32 %3, s20, 5\n v_writelane_b32 %4, s20, 5\n v_writelane_b32 %5, s20, 5\n v_writelane_b32 %6, s20, 5\n v_writelane_b32 %7, s20, 5\n v_writelane_b32 %0, s20, 5\n v_writelane_b32 %1, s20, 5\n v_writelane_b32 %2, s20, 5\n v_writelane_b32 %3, s20, 5\n v_writelane_b32 %4, s20, 5\n v_writelane_b32 %5, s20, 5\n v_writelane_b32 %6, s20, 5\n v_writelane_b32 %7, s20, 5" : OPS_F : "v"(a), "v"(b) : "s20");
    if constexpr (C == SUB_F32) asm volatile("v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8\n v_sub_f32 %0, %0, %8\n v_sub_f32 %1, %1, %8\n v_sub_f32 %2, %2, %8\n v_sub_f32 %3, %3, %8\n v_sub_f32 %4, %4, %8\n v_sub_f32 %5, %5, %8\n v_sub_f32 %6, %6, %8\n v_sub_f32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == CMP_SAND) asm volatile("v_cmp_lt_f32 s[20:21], %0, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %1, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %2, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %3, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %4, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %5, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %6, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %7, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %0, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %1, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %2, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %3, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %4, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %5, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %6, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]\n v_cmp_lt_f32 s[20:21], %7, %8\n s_and_b64 s[22:23], s[22:23], s[20:21]" : OPS_F : "v"(a), "v"(b) : "s20", "s21", "s22", "s23", "scc");
    if constexpr (C == AND_B32) asm volatile("v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == OR_B32) asm volatile("v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8\n v_or_b32 %0, %0, %8\n v_or_b32 %1, %1, %8\n v_or_b32 %2, %2, %8\n v_or_b32 %3, %3, %8\n v_or_b32 %4, %4, %8\n v_or_b32 %5, %5, %8\n v_or_b32 %6, %6, %8\n v_or_b32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == LSHL_B32) asm volatile("v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7\n v_lshlrev_b32 %0, 3, %0\n v_lshlrev_b32 %1, 3, %1\n v_lshlrev_b32 %2, 3, %2\n v_lshlrev_b32 %3, 3, %3\n v_lshlrev_b32 %4, 3, %4\n v_lshlrev_b32 %5, 3, %5\n v_lshlrev_b32 %6, 3, %6\n v_lshlrev_b32 %7, 3, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == LSHR_B32) asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7\n v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == SUB_U32) asm volatile("v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8\n v_sub_u32 %0, %0, %8\n v_sub_u32 %1, %1, %8\n v_sub_u32 %2, %2, %8\n v_sub_u32 %3, %3, %8\n v_sub_u32 %4, %4, %8\n v_sub_u32 %5, %5, %8\n v_sub_u32 %6, %6, %8\n v_sub_u32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == MIN_F32) asm volatile("v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == FMAC_F32) asm volatile("v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == MIN_U32) asm volatile("v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8\n v_min_u32 %0, %0, %8\n v_min_u32 %1, %1, %8\n v_min_u32 %2, %2, %8\n v_min_u32 %3, %3, %8\n v_min_u32 %4, %4, %8\n v_min_u32 %5, %5, %8\n v_min_u32 %6, %6, %8\n v_min_u32 %7, %7, %8" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == ADD3_U32) asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == CMP_VCC) asm volatile("v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8" : OPS_F : "v"(a), "v"(b) : "vcc");
    if constexpr (C == CND_VCC) asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : OPS_F : "v"(a), "v"(b) : "vcc");
    if constexpr (C == FFBL) asm volatile("v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7\n v_ffbl_b32 %0, %0\n v_ffbl_b32 %1, %1\n v_ffbl_b32 %2, %2\n v_ffbl_b32 %3, %3\n v_ffbl_b32 %4, %4\n v_ffbl_b32 %5, %5\n v_ffbl_b32 %6, %6\n v_ffbl_b32 %7, %7" : OPS_F : "v"(a), "v"(b));
    if constexpr (C == BFI) asm volatile("v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9\n v_bfi_b32 %0, %8, %0, %9\n v_bfi_b32 %1, %8, %1, %9\n v_bfi_b32 %2, %8, %2, %9\n v_bfi_b32 %3, %8, %3, %9\n v_bfi_b32 %4, %8, %4, %9\n v_bfi_b32 %5, %8, %5, %9\n v_bfi_b32 %6, %8, %6, %9\n v_bfi_b32 %7, %8, %7, %9" : OPS_F : "v"(a), "v"(b));
  }
}

template <int C>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int trips, float a, float b) {
  __shared__ float lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
  __syncthreads();
  float x[8];
  double dd[8];
  unsigned long long q[8];
  v4f v4[8];
  for (int i = 0; i < 8; ++i)
    x[i] = a + i + threadIdx.x * 1e-3f, dd[i] = 1.0 + 1e-9 * (i + threadIdx.x), q[i] = i + threadIdx.x, v4[i] = v4f{a, b, a, b};
  // the LDS classes take a byte address in operand %9: lane * 4 (conflict-free ds_read_b32 / identity permutation),
  // 0 for the wave-uniform (broadcast) 16-byte read
  const bool lds_cls = C == BPERMUTE || C == DS_READ_B32 || C == DS_WRITE_B32;
  const float bb = lds_cls ? __int_as_float((int)((threadIdx.x & 63) * 4)) : (C == DS_READ_B128_BCAST ? __int_as_float(0) : b);
  asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 s[22:23], exec" ::: "s20", "s21", "s22", "s23");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < trips; ++i) body<C>(x, dd, q, v4, a, bb);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += x[i] + (float)dd[i] + (float)q[i] + v4[i].x;
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  out[gid] = s + lds[threadIdx.x];
  if ((threadIdx.x & 63) == 0) {
    const int wave = gid >> 6;
    stamps[2 * wave] = t1 - t0;
    stamps[2 * wave + 1] = r1 - r0;
  }
}

template <int C>
void run(float* d_out, unsigned long long* d_st, int num_cus) {
  const int wps_list[5] = {1, 2, 4, 5, 8};
  const int per_line = (C == CMP_CND || C == CMP_SAND) ? 2 : 1;
  double issue[5], simd[5], mhz[5];
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int wi = 0; wi < 5; ++wi) {
    const int wps = wps_list[wi];  // 256-thread blocks per CU = waves per SIMD if the dispatcher spreads them evenly
    const int grid = num_cus * wps, trips = 6000;
    hipLaunchKernelGGL(k<C>, dim3(grid), dim3(256), 0, 0, d_out, d_st, 50, 1.0001f, 1e-6f);  // warm-up
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<C>, dim3(grid), dim3(256), 0, 0, d_out, d_st, trips, 1.0001f, 1e-6f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const int waves = grid * 4;
    std::vector<unsigned long long> st(2 * waves);
    (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> cyc(waves), clk(waves);
    for (int w = 0; w < waves; ++w) cyc[w] = (double)st[2 * w], clk[w] = st[2 * w + 1] ? (double)st[2 * w] / (double)st[2 * w + 1] * 100.0 : 0.0;
    std::nth_element(cyc.begin(), cyc.begin() + waves / 2, cyc.end());
    std::nth_element(clk.begin(), clk.begin() + waves / 2, clk.end());
    const double n_per_wave = (double)trips * kPerTrip * per_line;
    issue[wi] = cyc[waves / 2] / n_per_wave;
    mhz[wi] = clk[waves / 2];
    simd[wi] = (double)ms * 1e-3 * mhz[wi] * 1e6 * (num_cus * 4.0) / (n_per_wave * waves);
  }
  printf("%-36s |%6.2f %6.2f %6.2f %6.2f %6.2f |%6.2f %6.2f %6.2f %6.2f %6.2f | %5.0f %5.0f\n", kNames[C], issue[0], issue[1], issue[2],
         issue[3], issue[4], simd[0], simd[1], simd[2], simd[3], simd[4], mhz[2], mhz[4]);
  fflush(stdout);
}

int main() {
  hipDeviceProp_t p;
  (void)hipGetDeviceProperties(&p, 0);
  const int num_cus = p.multiProcessorCount;
  float* d_out;
  unsigned long long* d_st;
  (void)hipMalloc(&d_out, (size_t)num_cus * 8 * 256 * 4);
  (void)hipMalloc(&d_st, (size_t)num_cus * 8 * 4 * 16);
  printf("# %s, %d CUs.  issue = shader cycles between two instructions of ONE wave; SIMD = SIMD cycles per wave64 instruction from wall time\n",
         p.gcnArchName, num_cus);
  printf("# (the ceiling: 1 / SIMD is what a SIMD can retire per cycle with W resident waves); MHz = sustained clock at W = 4 / 8\n");
  printf("%-36s |%6s %6s %6s %6s %6s |%6s %6s %6s %6s %6s | %5s %5s\n", "instruction", "iss W1", "W2", "W4", "W5", "W8", "SIMD 1", "W2", "W4",
         "W5", "W8", "MHz 4", "MHz 8");
  run<FMA_F32>(d_out, d_st, num_cus);
  run<MUL_F32>(d_out, d_st, num_cus);
  run<ADD_F32>(d_out, d_st, num_cus);
  run<MAX_F32>(d_out, d_st, num_cus);
  run<CNDMASK>(d_out, d_st, num_cus);
  run<CMP_F32>(d_out, d_st, num_cus);
  run<CMP_CND>(d_out, d_st, num_cus);
  run<RCP_F32>(d_out, d_st, num_cus);
  run<RSQ_F32>(d_out, d_st, num_cus);
  run<SQRT_F32>(d_out, d_st, num_cus);
  run<SIN_F32>(d_out, d_st, num_cus);
  run<COS_F32>(d_out, d_st, num_cus);
  run<MUL_LO_U32>(d_out, d_st, num_cus);
  run<MUL_HI_U32>(d_out, d_st, num_cus);
  run<MAD_U64_U32>(d_out, d_st, num_cus);
  run<MUL_U32_U24>(d_out, d_st, num_cus);
  run<MAD_U32_U24>(d_out, d_st, num_cus);
  run<ADD_U32>(d_out, d_st, num_cus);
  run<LSHL_ADD>(d_out, d_st, num_cus);
  run<XOR_B32>(d_out, d_st, num_cus);
  run<AND_OR>(d_out, d_st, num_cus);
  run<CVT_F32_U32>(d_out, d_st, num_cus);
  run<CVT_U32_F32>(d_out, d_st, num_cus);
  run<FMA_F64>(d_out, d_st, num_cus);
  run<MUL_F64>(d_out, d_st, num_cus);
  run<ADD_F64>(d_out, d_st, num_cus);
  run<DIV_FIXUP>(d_out, d_st, num_cus);
  run<DIV_SCALE>(d_out, d_st, num_cus);
  run<DIV_FMAS>(d_out, d_st, num_cus);
  run<MOV_B32>(d_out, d_st, num_cus);
  run<MBCNT>(d_out, d_st, num_cus);
  run<READLANE>(d_out, d_st, num_cus);
  run<READFIRST>(d_out, d_st, num_cus);
  run<DPP_MOV>(d_out, d_st, num_cus);
  run<BPERMUTE>(d_out, d_st, num_cus);
  run<DS_READ_B32>(d_out, d_st, num_cus);
  run<DS_READ_B128_BCAST>(d_out, d_st, num_cus);
  run<DS_WRITE_B32>(d_out, d_st, num_cus);
  run<SALU>(d_out, d_st, num_cus);
  run<PK_FMA_F32>(d_out, d_st, num_cus);
  run<PK_MUL_F32>(d_out, d_st, num_cus);
  run<PK_ADD_F32>(d_out, d_st, num_cus);
  run<MAX3_F32>(d_out, d_st, num_cus);
  run<MIN3_F32>(d_out, d_st, num_cus);
  run<LSHL_ADD_U64>(d_out, d_st, num_cus);
  run<ALIGNBIT>(d_out, d_st, num_cus);
  run<BFE_U32>(d_out, d_st, num_cus);
  run<WRITELANE>(d_out, d_st, num_cus);
  run<SUB_F32>(d_out, d_st, num_cus);
  run<CMP_SAND>(d_out, d_st, num_cus);
  run<AND_B32>(d_out, d_st, num_cus);
  run<OR_B32>(d_out, d_st, num_cus);
  run<LSHL_B32>(d_out, d_st, num_cus);
  run<LSHR_B32>(d_out, d_st, num_cus);
  run<SUB_U32>(d_out, d_st, num_cus);
  run<MIN_F32>(d_out, d_st, num_cus);
  run<FMAC_F32>(d_out, d_st, num_cus);
  run<MIN_U32>(d_out, d_st, num_cus);
  run<ADD3_U32>(d_out, d_st, num_cus);
  run<CMP_VCC>(d_out, d_st, num_cus);
  run<CND_VCC>(d_out, d_st, num_cus);
  run<FFBL>(d_out, d_st, num_cus);
  run<BFI>(d_out, d_st, num_cus);
  return 0;
}
