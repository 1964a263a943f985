// issue_microbench.hip -- VALU issue-rate calibration for gfx950 (roofline.issue.peak of bench.py).
// One opcode per kernel, 8 independent register streams per wave, K waves per SIMD (K blocks of 256 threads per CU,
// pinned by LDS size: K blocks fit a CU, K + 1 do not).  Reported per opcode and K: chip-wide wave-instructions per
// second inside the kernel (first start stamp to last end stamp, s_memrealtime at 100 MHz), the same from hipEvents, and
// the shader clock the waves saw (s_memtime / s_memrealtime).  hipcc -O3 --offload-arch=gfx950.
//   tools/build/issue_microbench > profiles/r02_issue_microbench.json
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define ITERS 1024
// asm operands: %0 float stream, %1 uint stream, %2 64-bit pair stream (read-write); %3 / %4 loop-invariant floats, %5 invariant pair
// clang-format off
#define OPS(_) \
  _(0,  "v_fma_f32",            1, "v_fma_f32 %0, %0, %3, %4") \
  _(1,  "v_mul_f32",            1, "v_mul_f32 %0, %0, %3") \
  _(2,  "v_add_f32",            1, "v_add_f32 %0, %0, %3") \
  _(3,  "v_fmac_f32",           1, "v_fmac_f32 %0, %3, %4") \
  _(4,  "v_max_f32",            1, "v_max_f32 %0, %0, %3") \
  _(5,  "v_min_f32",            1, "v_min_f32 %0, %0, %3") \
  _(6,  "v_max3_f32",           1, "v_max3_f32 %0, %0, %3, %4") \
  _(7,  "v_min3_f32",           1, "v_min3_f32 %0, %0, %3, %4") \
  _(8,  "v_med3_f32",           1, "v_med3_f32 %0, %0, %3, %4") \
  _(9,  "v_cvt_f32_ubyte1",     1, "v_cvt_f32_ubyte1 %0, %1") \
  _(10, "v_cvt_f32_ubyte0",     1, "v_cvt_f32_ubyte0 %0, %1") \
  _(11, "v_cvt_f32_u32",        1, "v_cvt_f32_u32 %0, %1") \
  _(12, "v_cvt_f32_u32_sdwa_byte1", 1, "v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1") \
  _(13, "v_cmp_lt_f32+v_cndmask_b32", 2, "v_cmp_lt_f32 %2, %3, %4\n v_cndmask_b32 %0, %0, %1, %2") \
  _(14, "v_cmp_lt_f32",         1, "v_cmp_lt_f32 %2, %0, %3") \
  _(15, "v_cndmask_b32",        1, "v_cndmask_b32 %0, %0, %1, vcc") \
  _(16, "v_mov_b32",            1, "v_mov_b32 %0, %1") \
  _(17, "v_and_b32",            1, "v_and_b32 %1, %1, %3") \
  _(18, "v_or_b32",             1, "v_or_b32 %1, %1, %3") \
  _(19, "v_lshrrev_b32",        1, "v_lshrrev_b32 %1, 1, %1") \
  _(20, "v_add_u32",            1, "v_add_u32 %1, %1, %3") \
  _(21, "v_bfe_u32",            1, "v_bfe_u32 %1, %1, 8, 8") \
  _(22, "v_and_or_b32",         1, "v_and_or_b32 %1, %1, %3, %4") \
  _(23, "v_lshl_or_b32",        1, "v_lshl_or_b32 %1, %1, 1, %3") \
  _(24, "v_perm_b32",           1, "v_perm_b32 %1, %1, %3, %4") \
  _(25, "v_bcnt_u32_b32",       1, "v_bcnt_u32_b32 %1, %1, %3") \
  _(26, "v_pk_fma_f32",         1, "v_pk_fma_f32 %2, %2, %5, %5") \
  _(27, "v_pk_mul_f32",         1, "v_pk_mul_f32 %2, %2, %5") \
  _(28, "v_rcp_f32",            1, "v_rcp_f32 %0, %0") \
  _(29, "v_sqrt_f32",           1, "v_sqrt_f32 %0, %0") \
  _(30, "v_mul_lo_u32",         1, "v_mul_lo_u32 %1, %1, %3") \
  _(31, "v_mad_u32_u24",        1, "v_mad_u32_u24 %1, %1, %3, %4") \
  _(32, "v_mov_b32_sdwa_byte1_preserve", 1, "v_mov_b32_sdwa %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2") \
  _(33, "v_or_b32_sdwa_byte",   1, "v_or_b32_sdwa %0, %1, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:DWORD") \
  _(34, "v_lshlrev_b32_sdwa",   1, "v_lshlrev_b32_sdwa %1, %3, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_0 src1_sel:BYTE_1") \
  _(35, "v_max_i32",            1, "v_max_i32 %1, %1, %3") \
  _(36, "v_min_u32",            1, "v_min_u32 %1, %1, %3") \
  _(37, "v_max3_i32",           1, "v_max3_i32 %1, %1, %3, %4") \
  _(38, "v_sub_f32",            1, "v_sub_f32 %0, %0, %3") \
  _(39, "v_xor_b32",            1, "v_xor_b32 %1, %1, %3") \
  _(40, "v_lshlrev_b32_var",    1, "v_lshlrev_b32 %1, %3, %1") \
  _(41, "v_bitop3_b32",         1, "v_bitop3_b32 %1, %1, %3, %4 bitop3:0x6c") \
  _(42, "v_or3_b32",            1, "v_or3_b32 %1, %1, %3, %4") \
  _(43, "v_add3_u32",           1, "v_add3_u32 %1, %1, %3, %4") \
  _(44, "v_cvt_f32_f16",        1, "v_cvt_f32_f16 %0, %1") \
  _(45, "v_cvt_u32_f32",        1, "v_cvt_u32_f32 %1, %0") \
  _(46, "v_ldexp_f32",          1, "v_ldexp_f32 %0, %0, %3") \
  _(47, "v_fma_mix_f32",        1, "v_fma_mix_f32 %0, %0, %3, %4") \
  _(48, "v_pk_fma_f16",         1, "v_pk_fma_f16 %1, %1, %3, %4") \
  _(49, "v_pk_mul_f16",         1, "v_pk_mul_f16 %1, %1, %3") \
  _(50, "v_pk_add_f16",         1, "v_pk_add_f16 %1, %1, %3") \
  _(51, "v_pk_max_f16",         1, "v_pk_max_f16 %1, %1, %3") \
  _(52, "v_pk_min_f16",         1, "v_pk_min_f16 %1, %1, %3") \
  _(53, "v_pk_maximum3_f16",    1, "v_pk_maximum3_f16 %1, %1, %3, %4") \
  _(54, "v_pk_minimum3_f16",    1, "v_pk_minimum3_f16 %1, %1, %3, %4") \
  _(55, "v_maximum3_f32",       1, "v_maximum3_f32 %0, %0, %3, %4") \
  _(56, "v_minimum3_f32",       1, "v_minimum3_f32 %0, %0, %3, %4") \
  _(57, "v_pk_lshrrev_b16",     1, "v_pk_lshrrev_b16 %1, 8, %1") \
  _(58, "v_pk_ashrrev_i16",     1, "v_pk_ashrrev_i16 %1, 15, %1") \
  _(59, "v_pk_mad_u16",         1, "v_pk_mad_u16 %1, %1, %3, %4") \
  _(60, "v_pk_max_u16",         1, "v_pk_max_u16 %1, %1, %3") \
  _(61, "v_pk_sub_i16",         1, "v_pk_sub_i16 %1, %1, %3") \
  _(62, "v_pk_add_u16",         1, "v_pk_add_u16 %1, %1, %3") \
  _(63, "v_bfe_i32",            1, "v_bfe_i32 %1, %1, 15, 1") \
  _(64, "v_ashrrev_i32",        1, "v_ashrrev_i32 %1, 31, %1") \
  _(65, "v_cvt_pkrtz_f16_f32",  1, "v_cvt_pkrtz_f16_f32 %1, %0, %3") \
  _(66, "v_cvt_pk_f16_f32",     1, "v_cvt_pk_f16_f32 %1, %0, %3") \
  _(67, "v_pk_fma_f16_denormal_operands", 1, "v_pk_fma_f16 %1, %1, %1, %1") \
  _(68, "v_cmp_lt_i16",         1, "v_cmp_lt_i16 %2, %1, %3") \
  _(69, "v_lshlrev_b32_imm",    1, "v_lshlrev_b32 %1, 16, %1") \
  _(70, "v_and_b32_literal",    1, "v_and_b32 %1, 0x00ff00ff, %1") \
  _(71, "v_sub_u32",            1, "v_sub_u32 %1, %1, %3") \
  _(72, "v_dot4_u32_u8",        1, "v_dot4_u32_u8 %1, %1, %3, %4") \
  _(73, "v_cvt_f16_f32",        1, "v_cvt_f16_f32 %1, %0") \
  _(74, "v_pk_fma_f16_opsel",   1, "v_pk_fma_f16 %1, %1, %3, %4 op_sel:[0,1,0] op_sel_hi:[1,0,1]") \
  _(75, "v_pk_fma_f16_subnormal_x_normal", 1, "v_pk_fma_f16 %0, %1, %3, %4") \
  _(76, "v_fma_mixlo_f16",      1, "v_fma_mixlo_f16 %1, %0, %3, 0") \
  _(77, "v_pk_maximum3_f16_subnormal", 1, "v_pk_maximum3_f16 %0, %1, %3, %4") \
  _(78, "v_pk_add_f16_subnormal", 1, "v_pk_add_f16 %0, %1, %3")
// clang-format on
template <int OP> __global__ __launch_bounds__(256) void k_issue(float* out, unsigned long long* cyc, float s)
{
  extern __shared__ int pin[];
  float v[8]; unsigned u[8]; double p[4], q[4]; unsigned long long m[8];
  for(int i = 0; i < 8; i++) { v[i] = s + threadIdx.x + i; u[i] = threadIdx.x * 2654435761u + i; }
  for(int i = 0; i < 4; i++) { p[i] = s + i; q[i] = s * 0.25 + i; }
  const float a = s * 0.5f, b = s + 0.25f;
  if(OP == 67 || OP == 75 || OP == 77 || OP == 78) for(int i = 0; i < 8; i++) u[i] &= 0x00ff00ffu;  // binary16 subnormals in both halves: x * x + x stays subnormal
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), t0 = __builtin_amdgcn_s_memtime();
  for(int it = 0; it < ITERS; it++)
  {
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
#define GEN(id, name, n, text)                                                                                                        \
  if(OP == id)                                                                                                                        \
  {                                                                                                                                   \
    _Pragma("unroll") for(int i = 0; i < 8; i++)                                                                                      \
      if(id == 13 || id == 14 || id == 68)                                                                                              \
        asm volatile(text : "+v"(v[i]), "+v"(u[i]), "=s"(m[i]) : "v"(a), "v"(b));                                                     \
      else if(id == 26 || id == 27) /* the 64-bit pair stream only where it is used (declared pair writes get s_nop padding) */        \
        asm volatile(text : "+v"(v[i]), "+v"(u[i]), "+v"(p[i & 3]) : "v"(a), "v"(b), "v"(q[i & 3]));                                   \
      else /* no vcc clobber either: a declared vcc write makes the compiler pad every statement with s_nop */                        \
        asm volatile(text : "+v"(v[i]), "+v"(u[i]) : "v"(p[i & 3]), "v"(a), "v"(b));                                                   \
  }
      OPS(GEN)
#undef GEN
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0; for(int i = 0; i < 8; i++) acc += v[i] + (float)u[i] + (float)p[i & 3] + ((OP == 13 || OP == 14 || OP == 68) ? (float)m[i] : 0.0f);
  out[blockIdx.x * 256 + threadIdx.x] = acc + (float)pin[0];
  if((threadIdx.x & 63) == 0)
  {
    unsigned long long* c = cyc + (blockIdx.x * 4 + (threadIdx.x >> 6)) * 3;
    c[0] = t1 - t0; c[1] = r0; c[2] = r1;
  }
}
template <int OP> static void run(const char* name, int perBody, bool last)
{
  const int CU = 256; float* out; unsigned long long* cyc;
  (void)hipMalloc(&out, CU * 8 * 256 * 4); (void)hipMalloc(&cyc, CU * 8 * 4 * 8 * 3);
  printf("  \"%s\": {", name);
  const int ks[4] = {1, 2, 5, 8};
  for(int q = 0; q < 4; q++)
  {
    const int k = ks[q]; const size_t lds = (150 * 1024 / k) & ~1023;  // k blocks fit one CU (<= 150 KB), k + 1 do not (> 160 KB)
    (void)hipFuncSetAttribute((const void*)k_issue<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k_issue<OP><<<CU * k, 256, lds>>>(out, cyc, 1.0f);
    (void)hipDeviceSynchronize();  // the timed launch starts on an idle chip: every block is resident from the start
    (void)hipEventRecord(e0); k_issue<OP><<<CU * k, 256, lds>>>(out, cyc, 1.0f); (void)hipEventRecord(e1); (void)hipDeviceSynchronize();
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const size_t nw = (size_t)CU * k * 4;
    std::vector<unsigned long long> h(nw * 3); (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0, real = 0; unsigned long long s0 = ~0ull, s1 = 0, re = 0;
    for(size_t w = 0; w < nw; w++)
    {
      mean += (double)h[3 * w]; real += (double)(h[3 * w + 2] - h[3 * w + 1]);
      s0 = std::min(s0, h[3 * w + 1]); s1 = std::max(s1, h[3 * w + 1]); re = std::max(re, h[3 * w + 2]);
    }
    const double instr = (double)ITERS * 4 * 8 * perBody;  // wave-instructions per wave
    const double rate = instr * nw / ((double)(re - s0) * 1e-8) / 1e9, clk = mean / real * 0.1;
    printf("%s\"%d\": {\"G_wave_instr_per_s\": %.1f, \"G_wave_instr_per_s_hipevent\": %.1f, \"clock_GHz\": %.3f, \"cyc_per_instr_per_simd\": %.2f, "
           "\"start_spread_us\": %.1f, \"kernel_us\": %.1f}", q ? ", " : "", k, rate, instr * nw / (ms * 1e-3) / 1e9, clk, 1024.0 * clk / rate,
           (double)(s1 - s0) * 0.01, (double)(re - s0) * 0.01);
  }
  printf("}%s\n", last ? "" : ",");
  (void)hipFree(out); (void)hipFree(cyc);
}
int main()
{
  printf("{\n \"note\": \"per opcode: waves/SIMD -> chip-wide G wave64-instructions/s (in-kernel stamps; hipEvent), shader clock, cycles per instruction per SIMD\",\n");
#define RUN(id, name, n, text) run<id>(name, n, id == 78);
  OPS(RUN)
  printf("}\n");
  return 0;
}
