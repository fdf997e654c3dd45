#!/usr/bin/env python3
"""Writes tools/ubench_valu.hip from the class table below: one asm statement = 16 instructions (two rounds over eight
independent register chains), so the compiler's hazard recognizer cannot pad the measured instruction with s_nop (it does
after every single-instruction asm statement that names vcc or an SGPR).  Operand %8 / %9 = two VGPR inputs (for the LDS
classes %9 is a byte address); {i} = the chain's register.  Kinds: f float x[8], d double dd[8] (64-bit pairs, also the
packed-f32 classes), q 64-bit integers, v4 128-bit vectors.  usage: python tools/gen_ubench_valu.py"""
import os

CLASSES = [('v_fma_f32', 'FMA_F32', 'v_fma_f32 %{i}, %{i}, %8, %9', 'f', ''),
    ('v_mul_f32', 'MUL_F32', 'v_mul_f32 %{i}, %{i}, %8', 'f', ''),
    ('v_add_f32', 'ADD_F32', 'v_add_f32 %{i}, %{i}, %8', 'f', ''),
    ('v_max_f32', 'MAX_F32', 'v_max_f32 %{i}, %{i}, %8', 'f', ''),
    ('v_cndmask_b32 (mask in SGPRs)', 'CNDMASK', 'v_cndmask_b32 %{i}, %{i}, %8, s[20:21]', 'f', '"s20", "s21"'),
    ('v_cmp_lt_f32 (to SGPR pair)', 'CMP_F32', 'v_cmp_lt_f32 s[20:21], %{i}, %8', 'f', '"s20", "s21"'),
    ('v_cmp_lt_f32 + v_cndmask_b32 (vcc)', 'CMP_CND', 'v_cmp_lt_f32 vcc, %{i}, %8\\n v_cndmask_b32 %{i}, %{i}, %9, vcc', 'f', '"vcc"'),
    ('v_rcp_f32', 'RCP_F32', 'v_rcp_f32 %{i}, %{i}', 'f', ''),
    ('v_rsq_f32', 'RSQ_F32', 'v_rsq_f32 %{i}, %{i}', 'f', ''),
    ('v_sqrt_f32', 'SQRT_F32', 'v_sqrt_f32 %{i}, %{i}', 'f', ''),
    ('v_sin_f32', 'SIN_F32', 'v_sin_f32 %{i}, %{i}', 'f', ''),
    ('v_cos_f32', 'COS_F32', 'v_cos_f32 %{i}, %{i}', 'f', ''),
    ('v_mul_lo_u32', 'MUL_LO_U32', 'v_mul_lo_u32 %{i}, %{i}, %8', 'f', ''),
    ('v_mul_hi_u32', 'MUL_HI_U32', 'v_mul_hi_u32 %{i}, %{i}, %8', 'f', ''),
    ('v_mad_u64_u32', 'MAD_U64_U32', 'v_mad_u64_u32 %{i}, s[20:21], %8, %9, %{i}', 'q', '"s20", "s21"'),
    ('v_mul_u32_u24', 'MUL_U32_U24', 'v_mul_u32_u24 %{i}, %{i}, %8', 'f', ''),
    ('v_mad_u32_u24', 'MAD_U32_U24', 'v_mad_u32_u24 %{i}, %{i}, %8, %9', 'f', ''),
    ('v_add_u32', 'ADD_U32', 'v_add_u32 %{i}, %{i}, %8', 'f', ''),
    ('v_lshl_add_u32', 'LSHL_ADD', 'v_lshl_add_u32 %{i}, %{i}, 3, %8', 'f', ''),
    ('v_xor_b32', 'XOR_B32', 'v_xor_b32 %{i}, %{i}, %8', 'f', ''),
    ('v_and_or_b32', 'AND_OR', 'v_and_or_b32 %{i}, %{i}, %8, %9', 'f', ''),
    ('v_cvt_f32_u32', 'CVT_F32_U32', 'v_cvt_f32_u32 %{i}, %{i}', 'f', ''),
    ('v_cvt_u32_f32', 'CVT_U32_F32', 'v_cvt_u32_f32 %{i}, %{i}', 'f', ''),
    ('v_fma_f64', 'FMA_F64', 'v_fma_f64 %{i}, %{i}, %{i}, %{i}', 'd', ''),
    ('v_mul_f64', 'MUL_F64', 'v_mul_f64 %{i}, %{i}, %{i}', 'd', ''),
    ('v_add_f64', 'ADD_F64', 'v_add_f64 %{i}, %{i}, %{i}', 'd', ''),
    ('v_div_fixup_f32', 'DIV_FIXUP', 'v_div_fixup_f32 %{i}, %{i}, %8, %9', 'f', ''),
    ('v_div_scale_f32', 'DIV_SCALE', 'v_div_scale_f32 %{i}, s[20:21], %{i}, %8, %9', 'f', '"s20", "s21"'),
    ('v_div_fmas_f32', 'DIV_FMAS', 'v_div_fmas_f32 %{i}, %{i}, %8, %9', 'f', '"vcc"'),
    ('v_mov_b32', 'MOV_B32', 'v_mov_b32 %{i}, %8', 'f', ''),
    ('v_mbcnt_lo_u32_b32', 'MBCNT', 'v_mbcnt_lo_u32_b32 %{i}, %8, %{i}', 'f', ''),
    ('v_readlane_b32', 'READLANE', 'v_readlane_b32 s20, %{i}, 3', 'f', '"s20"'),
    ('v_readfirstlane_b32', 'READFIRST', 'v_readfirstlane_b32 s20, %{i}', 'f', '"s20"'),
    ('v_mov_b32 dpp row_shr:1', 'DPP_MOV', 'v_mov_b32_dpp %{i}, %{i} row_shr:1 row_mask:0xf bank_mask:0xf', 'f', ''),
    ('ds_bpermute_b32', 'BPERMUTE', 'ds_bpermute_b32 %{i}, %9, %{i}', 'f', 'LDS'),
    ('ds_read_b32 (lane-consecutive)', 'DS_READ_B32', 'ds_read_b32 %{i}, %9 offset:{i}*256', 'f', 'LDS'),
    ('ds_read_b128 (uniform address)', 'DS_READ_B128_BCAST', 'ds_read_b128 %{i}, %9 offset:{i}*16', 'v4', 'LDS'),
    ('ds_write_b32 (lane-consecutive)', 'DS_WRITE_B32', 'ds_write_b32 %9, %{i} offset:{i}*256', 'f', 'LDS'),
    ('s_and_b64 (SALU beside nothing)', 'SALU', 's_and_b64 s[20:21], s[20:21], s[22:23]', 'f', '"s20", "s21", "scc"'),
    ('v_pk_fma_f32 (2 FMAs per lane)', 'PK_FMA_F32', 'v_pk_fma_f32 %{i}, %{i}, %{i}, %{i}', 'd', ''),
    ('v_pk_mul_f32', 'PK_MUL_F32', 'v_pk_mul_f32 %{i}, %{i}, %{i}', 'd', ''),
    ('v_pk_add_f32', 'PK_ADD_F32', 'v_pk_add_f32 %{i}, %{i}, %{i}', 'd', ''),
    ('v_max3_f32', 'MAX3_F32', 'v_max3_f32 %{i}, %{i}, %8, %9', 'f', ''),
    ('v_min3_f32', 'MIN3_F32', 'v_min3_f32 %{i}, %{i}, %8, %9', 'f', ''),
    ('v_lshl_add_u64', 'LSHL_ADD_U64', 'v_lshl_add_u64 %{i}, %{i}, 2, %{i}', 'q', ''),
    ('v_alignbit_b32', 'ALIGNBIT', 'v_alignbit_b32 %{i}, %{i}, %8, 15', 'f', ''),
    ('v_bfe_u32', 'BFE_U32', 'v_bfe_u32 %{i}, %{i}, 3, 9', 'f', ''),
    ('v_writelane_b32', 'WRITELANE', 'v_writelane_b32 %{i}, s20, 5', 'f', '"s20"'),
    ('v_subrev_f32 / v_sub_f32', 'SUB_F32', 'v_sub_f32 %{i}, %{i}, %8', 'f', ''),
    ('v_cmp_lt_f32 -> s_and_b64 (VALU + SALU)', 'CMP_SAND', 'v_cmp_lt_f32 s[20:21], %{i}, %8\\n s_and_b64 s[22:23], s[22:23], s[20:21]', 'f', '"s20", "s21", "s22", "s23", "scc"'),
    ('v_and_b32', 'AND_B32', 'v_and_b32 %{i}, %{i}, %8', 'f', ''),
    ('v_or_b32', 'OR_B32', 'v_or_b32 %{i}, %{i}, %8', 'f', ''),
    ('v_lshlrev_b32', 'LSHL_B32', 'v_lshlrev_b32 %{i}, 3, %{i}', 'f', ''),
    ('v_lshrrev_b32', 'LSHR_B32', 'v_lshrrev_b32 %{i}, 3, %{i}', 'f', ''),
    ('v_sub_u32', 'SUB_U32', 'v_sub_u32 %{i}, %{i}, %8', 'f', ''),
    ('v_min_f32', 'MIN_F32', 'v_min_f32 %{i}, %{i}, %8', 'f', ''),
    ('v_fmac_f32', 'FMAC_F32', 'v_fmac_f32 %{i}, %8, %9', 'f', ''),
    ('v_min_u32', 'MIN_U32', 'v_min_u32 %{i}, %{i}, %8', 'f', ''),
    ('v_add3_u32', 'ADD3_U32', 'v_add3_u32 %{i}, %{i}, %8, %9', 'f', ''),
    ('v_cmp_lt_f32 (to vcc)', 'CMP_VCC', 'v_cmp_lt_f32 vcc, %{i}, %8', 'f', '"vcc"'),
    ('v_cndmask_b32 (vcc)', 'CND_VCC', 'v_cndmask_b32 %{i}, %{i}, %8, vcc', 'f', '"vcc"'),
    ('v_ffbl_b32', 'FFBL', 'v_ffbl_b32 %{i}, %{i}', 'f', ''),
    ('v_bfi_b32', 'BFI', 'v_bfi_b32 %{i}, %8, %{i}, %9', 'f', ''),
]

TWO_PER_LINE = {"CMP_CND", "CMP_SAND"}  # two instructions per chain line: both are counted

HEAD = "// tools/ubench_valu.hip — the measured instruction-issue ceiling behind `roofline_valu` (bench.py) and DESIGN.md §5.\n//\n// For every instruction class the fused kernels execute, a loop of that ONE instruction — pinned with inline asm (one asm\n// statement = 16 instructions on eight independent register chains, so the compiler's hazard recognizer cannot pad them\n// with s_nop), no global-memory traffic — is run at 1, 2, 4, 5 and 8 resident waves per SIMD with every CU busy.\n// Two time bases, because they answer different questions:\n//   issue   s_memtime (shader cycles) around the loop of each wave, median over waves, / instructions per wave:\n//           how often ONE wave gets to issue (>= 4-5 cycles per VALU instruction even on an otherwise idle SIMD);\n//   SIMD    wall time (HIP events) x sustained clock x number of SIMDs / all instructions issued: SIMD cycles per wave64\n//           instruction, independent of where the dispatcher put the blocks — THE CEILING a mix is priced with\n//           (tools/valu_ceiling.py multiplies it with a kernel's dynamic SQ_INSTS_VALU_* counts).\n// Sustained clock = s_memtime / s_memrealtime (100 MHz), median over waves.\n//\n// Build + run (GPU box):  hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o build/tools/ubench_valu && build/tools/ubench_valu\n#include <hip/hip_runtime.h>\n\n#include <algorithm>\n#include <cstdio>\n#include <vector>\n\ntypedef float v4f __attribute__((ext_vector_type(4)));\n\n"
MID = 'constexpr int kPerTrip = 64;  // 4 asm statements x 16 instructions (CMP_CND: 2 instructions per line, counted below)\n\ntemplate <int C>\n__device__ __forceinline__ void body(float (&x)[8], double (&dd)[8], unsigned long long (&q)[8], v4f (&v4)[8], float a, float b) {\n#define OPS_F "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7])\n#define OPS_D "+v"(dd[0]), "+v"(dd[1]), "+v"(dd[2]), "+v"(dd[3]), "+v"(dd[4]), "+v"(dd[5]), "+v"(dd[6]), "+v"(dd[7])\n#define OPS_Q "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]), "+v"(q[4]), "+v"(q[5]), "+v"(q[6]), "+v"(q[7])\n#define OPS_V4 "+v"(v4[0]), "+v"(v4[1]), "+v"(v4[2]), "+v"(v4[3]), "+v"(v4[4]), "+v"(v4[5]), "+v"(v4[6]), "+v"(v4[7])\n#pragma unroll\n  for (int u = 0; u < 4; ++u) {\n'
TAIL_PRE = '  }\n}\n\ntemplate <int C>\n__global__ __launch_bounds__(256) void k(float* out, unsigned long long* stamps, int trips, float a, float b) {\n  __shared__ float lds[4096];\n  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;\n  __syncthreads();\n  float x[8];\n  double dd[8];\n  unsigned long long q[8];\n  v4f v4[8];\n  for (int i = 0; i < 8; ++i)\n    x[i] = a + i + threadIdx.x * 1e-3f, dd[i] = 1.0 + 1e-9 * (i + threadIdx.x), q[i] = i + threadIdx.x, v4[i] = v4f{a, b, a, b};\n  // the LDS classes take a byte address in operand %9: lane * 4 (conflict-free ds_read_b32 / identity permutation),\n  // 0 for the wave-uniform (broadcast) 16-byte read\n  const bool lds_cls = C == BPERMUTE || C == DS_READ_B32 || C == DS_WRITE_B32;\n  const float bb = lds_cls ? __int_as_float((int)((threadIdx.x & 63) * 4)) : (C == DS_READ_B128_BCAST ? __int_as_float(0) : b);\n  asm volatile("s_mov_b64 s[20:21], exec\\n s_mov_b64 s[22:23], exec" ::: "s20", "s21", "s22", "s23");\n  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();\n  for (int i = 0; i < trips; ++i) body<C>(x, dd, q, v4, a, bb);\n  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();\n  float s = 0.f;\n  for (int i = 0; i < 8; ++i) s += x[i] + (float)dd[i] + (float)q[i] + v4[i].x;\n  const int gid = blockIdx.x * blockDim.x + threadIdx.x;\n  out[gid] = s + lds[threadIdx.x];\n  if ((threadIdx.x & 63) == 0) {\n    const int wave = gid >> 6;\n    stamps[2 * wave] = t1 - t0;\n    stamps[2 * wave + 1] = r1 - r0;\n  }\n}\n\ntemplate <int C>\nvoid run(float* d_out, unsigned long long* d_st, int num_cus) {\n  const int wps_list[5] = {1, 2, 4, 5, 8};\n  const int per_line = C == CMP_CND ? 2 : 1;\n  double issue[5], simd[5], mhz[5];\n  hipEvent_t e0, e1;\n  (void)hipEventCreate(&e0);\n  (void)hipEventCreate(&e1);\n  for (int wi = 0; wi < 5; ++wi) {\n    const int wps = wps_list[wi];  // 256-thread blocks per CU = waves per SIMD if the dispatcher spreads them evenly\n    const int grid = num_cus * wps, trips = 6000;\n    hipLaunchKernelGGL(k<C>, dim3(grid), dim3(256), 0, 0, d_out, d_st, 50, 1.0001f, 1e-6f);  // warm-up\n    (void)hipEventRecord(e0);\n    hipLaunchKernelGGL(k<C>, dim3(grid), dim3(256), 0, 0, d_out, d_st, trips, 1.0001f, 1e-6f);\n    (void)hipEventRecord(e1);\n    (void)hipEventSynchronize(e1);\n    float ms = 0.f;\n    (void)hipEventElapsedTime(&ms, e0, e1);\n    const int waves = grid * 4;\n    std::vector<unsigned long long> st(2 * waves);\n    (void)hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost);\n    std::vector<double> cyc(waves), clk(waves);\n    for (int w = 0; w < waves; ++w) cyc[w] = (double)st[2 * w], clk[w] = st[2 * w + 1] ? (double)st[2 * w] / (double)st[2 * w + 1] * 100.0 : 0.0;\n    std::nth_element(cyc.begin(), cyc.begin() + waves / 2, cyc.end());\n    std::nth_element(clk.begin(), clk.begin() + waves / 2, clk.end());\n    const double n_per_wave = (double)trips * kPerTrip * per_line;\n    issue[wi] = cyc[waves / 2] / n_per_wave;\n    mhz[wi] = clk[waves / 2];\n    simd[wi] = (double)ms * 1e-3 * mhz[wi] * 1e6 * (num_cus * 4.0) / (n_per_wave * waves);\n  }\n  printf("%-36s |%6.2f %6.2f %6.2f %6.2f %6.2f |%6.2f %6.2f %6.2f %6.2f %6.2f | %5.0f %5.0f\\n", kNames[C], issue[0], issue[1], issue[2],\n         issue[3], issue[4], simd[0], simd[1], simd[2], simd[3], simd[4], mhz[2], mhz[4]);\n  fflush(stdout);\n}\n\nint main() {\n  hipDeviceProp_t p;\n  (void)hipGetDeviceProperties(&p, 0);\n  const int num_cus = p.multiProcessorCount;\n  float* d_out;\n  unsigned long long* d_st;\n  (void)hipMalloc(&d_out, (size_t)num_cus * 8 * 256 * 4);\n  (void)hipMalloc(&d_st, (size_t)num_cus * 8 * 4 * 16);\n  printf("# %s, %d CUs.  issue = shader cycles between two instructions of ONE wave; SIMD = SIMD cycles per wave64 instruction from wall time\\n",\n         p.gcnArchName, num_cus);\n  printf("# (the ceiling: 1 / SIMD is what a SIMD can retire per cycle with W resident waves); MHz = sustained clock at W = 4 / 8\\n");\n  printf("%-36s |%6s %6s %6s %6s %6s |%6s %6s %6s %6s %6s | %5s %5s\\n", "instruction", "iss W1", "W2", "W4", "W5", "W8", "SIMD 1", "W2", "W4",\n         "W5", "W8", "MHz 4", "MHz 8");\n'
TAIL_POST = '  return 0;\n}\n'


def main():
    body = []
    for name, en, tmpl, kind, clob in CLASSES:
        lines = [tmpl.replace("{i}", str(i)) for _ in range(2) for i in range(8)]
        s = "\\n ".join(lines)
        ops = {"f": "OPS_F", "d": "OPS_D", "q": "OPS_Q", "v4": "OPS_V4"}[kind]
        cl = clob
        if clob == "LDS":
            cl = ""
            s += "\\n s_waitcnt lgkmcnt(0)"
        body.append(f'    if constexpr (C == {en}) asm volatile("{s}" : {ops} : "v"(a), "v"(b){" : " + cl if cl else ""});')
    src = HEAD + "enum Cls { " + ", ".join(c[1] for c in CLASSES) + ", NUM_CLS };\n"
    src += "static const char* kNames[NUM_CLS] = {" + ", ".join('"' + c[0] + '"' for c in CLASSES) + "};\n\n"
    src += MID + "\n".join(body) + "\n" + TAIL_PRE
    src += "\n".join(f"  run<{c[1]}>(d_out, d_st, num_cus);" for c in CLASSES) + "\n" + TAIL_POST
    src = src.replace("C == CMP_CND ? 2 : 1", "(" + " || ".join(f"C == {c}" for c in sorted(TWO_PER_LINE)) + ") ? 2 : 1")
    open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench_valu.hip"), "w").write(src)


if __name__ == "__main__":
    main()
