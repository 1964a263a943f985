// issue_microbench.hip -- VALU issue-rate calibration for gfx950 (roofline.issue.peak of bench.py).
// Independent instruction streams of one opcode class, K waves per SIMD (K blocks of 256 threads per CU, pinned by
// LDS size: K blocks fit a CU, K+1 do not), cycles per wave-instruction per SIMD from s_memtime stamps around the loop,
// shader clock from s_memtime / s_memrealtime (100 MHz), co-residency check from the spread of the start stamps.
// hipcc --offload-arch=gfx950.
//   ./issue_microbench > profiles/r02_issue_microbench.json
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define ITERS 2048
#define UNROLL 32  // wave-instructions per loop body (4 x 8 independent registers)
#define X8(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
template <int OP> __global__ __launch_bounds__(256) void k_issue(float* out, unsigned long long* cyc, float s)
{
  extern __shared__ int pin[];
  float v[8]; unsigned u[8];
  for(int i = 0; i < 8; i++) { v[i] = s + threadIdx.x + i; u[i] = threadIdx.x * 2654435761u + i; }
  const float a = s * 0.5f, b = s + 0.25f;
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for(int it = 0; it < ITERS; it++)
  {
#pragma unroll
    for(int r = 0; r < 4; r++)
    {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(a), "v"(b));
#define CVT(i) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(v[i]) : "v"(u[i]));
#define CMPSEL(i) asm volatile("v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc" : "+v"(v[i]) : "v"(a), "v"(b), "v"(u[i]) : "vcc");
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
#define INTOP(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(a));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(*(double*)&v[i & 6]) : "v"(*(const double*)&u[(i + 2) & 6]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(a));
      if(OP == 0) { X8(FMA) } if(OP == 1) { X8(CVT) } if(OP == 2) { X8(CMPSEL) } if(OP == 3) { X8(MAXF) }
      if(OP == 4) { X8(INTOP) } if(OP == 5) { X8(PKFMA) } if(OP == 6) { X8(RCP) } if(OP == 7) { X8(MUL) }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float acc = 0; for(int i = 0; i < 8; i++) acc += v[i] + (float)u[i];
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
  hipMalloc(&out, CU * 8 * 256 * 4); hipMalloc(&cyc, CU * 8 * 4 * 8 * 3);
  printf("  \"%s\": {", name);
  const int ks[6] = {1, 2, 3, 4, 5, 8};
  for(int q = 0; q < 6; q++)
  {
    const int k = ks[q]; const size_t lds = (150 * 1024 / k) & ~1023;  // k blocks fit one CU (<= 150 KB), k + 1 do not (> 160 KB)
    hipFuncSetAttribute((const void*)k_issue<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_issue<OP><<<CU * k, 256, lds>>>(out, cyc, 1.0f);
    hipEventRecord(e0); k_issue<OP><<<CU * k, 256, lds>>>(out, cyc, 1.0f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const size_t nw = (size_t)CU * k * 4;
    std::vector<unsigned long long> h(nw * 3); hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0, real = 0; unsigned long long s0 = ~0ull, s1 = 0, re = 0;
    for(size_t w = 0; w < nw; w++)
    {
      mean += (double)h[3 * w]; real += (double)(h[3 * w + 2] - h[3 * w + 1]);
      s0 = std::min(s0, h[3 * w + 1]); s1 = std::max(s1, h[3 * w + 1]); re = std::max(re, h[3 * w + 2]);
    }
    mean /= nw; real /= nw;
    const double instr = (double)ITERS * 4 * 8 * perBody;  // wave-instructions per wave
    printf("%s\"%d\": {\"cyc_per_instr_per_simd\": %.3f, \"G_wave_instr_per_s\": %.1f, \"G_wave_instr_per_s_in_kernel\": %.1f, \"clock_GHz\": %.3f, "
           "\"start_spread_us\": %.1f, \"kernel_us\": %.1f}", q ? ", " : "", k, mean / (instr * k), instr * nw / (ms * 1e-3) / 1e9,
           instr * nw / ((double)(re - s0) * 1e-8) / 1e9, mean / real * 0.1, (double)(s1 - s0) * 0.01, (double)(re - s0) * 0.01);
  }
  printf("}%s\n", last ? "" : ",");
  hipFree(out); hipFree(cyc);
}
int main()
{
  printf("{\n \"note\": \"waves/SIMD -> cycles (s_memtime) per wave64 instruction per SIMD and chip-wide rate (hipEvent wall time)\",\n");
  run<0>("v_fma_f32", 1, false); run<7>("v_mul_f32", 1, false); run<1>("v_cvt_f32_ubyte1", 1, false); run<2>("v_cmp_lt_f32+v_cndmask_b32", 2, false);
  run<3>("v_max_f32", 1, false); run<4>("v_and_or_b32", 1, false); run<5>("v_pk_fma_f32", 1, false); run<6>("v_rcp_f32", 1, true);
  printf("}\n");
  return 0;
}
