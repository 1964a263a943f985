// What does a wave's node fetch cost in the vector L1 / texture-address path of a gfx950 CU?  (profiles/r03_experiments.md #96)
//
// The traversal kernel fetches one 80-B node per lane with five global_load_dwordx4: 64 lanes, 64 different cache lines per
// instruction.  This microbenchmark times that access shape against alternatives that move the same bytes, on a table the size
// of the bench scene's tree (L2-resident), at the traversal kernel's occupancy (5 waves per SIMD):
//   mode 0  per-lane node, 5 x dwordx4 per lane (the product's shape)
//   mode 1  cooperative: lanes 5k..5k+4 fetch the five quads of node k (12 nodes per instruction, 60 lanes active), five
//           instructions per iteration as in mode 0 (so 60 nodes per iteration instead of 64)
//   mode 2  contiguous: lane L reads 16 B at base + 16 L of a random 1-KiB-aligned block (the data-path floor)
//   mode 3  per-lane node of 128 B, 8 x dwordx4 (experiment #95's node)
//   mode 4  per-lane node, 5 x dwordx4, but 16 lanes share each node (coherent rays: 4 distinct nodes per instruction)
// Output: ns per wave-instruction per CU and the implied cycles at the clock the kernel itself measures (s_memrealtime is 100 MHz;
// wall_clock64 ticks, so the clock is taken from the VALU-side calibration of tools/issue_microbench.hip when comparing).
//
// build: hipcc --offload-arch=gfx950 -O3 -o gather_microbench tools/gather_microbench.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if(e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while(0)

template <int MODE>
__global__ __launch_bounds__(64) void k_gather(const uint4* __restrict__ table, unsigned nodeCount, int iters, unsigned* sink)
{
  __shared__ unsigned pad[1600];  // 6.4 KB per one-wave workgroup: 5 workgroups per SIMD fit the 160-KB LDS, like the traversal kernel
  const unsigned lane = threadIdx.x;
  unsigned rng = (blockIdx.x * 64u + lane) * 2654435761u + 12345u;
  unsigned acc = 0;
  for(int it = 0; it < iters; it++)
  {
    rng = rng * 1664525u + 1013904223u;
    if(MODE == 0)
    {
      const unsigned node = (rng >> 8) % nodeCount;
      const uint4* p = table + (size_t)node * 5;
#pragma unroll
      for(int q = 0; q < 5; q++) { const uint4 v = p[q]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    else if(MODE == 1)
    {
      // the wave agrees on 60 nodes: node of slot s = hash(it, s); lane L serves slot (L / 5) of each of the five instructions
      const unsigned slot = lane / 5u, quad = lane % 5u;
#pragma unroll
      for(int i = 0; i < 5; i++)
      {
        unsigned h = (blockIdx.x * 977u + (unsigned)it * 131u + (unsigned)i * 12u + slot) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const unsigned node = h % nodeCount;
        if(lane < 60u) { const uint4 v = table[(size_t)node * 5 + quad]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
      }
    }
    else if(MODE == 2)
    {
#pragma unroll
      for(int i = 0; i < 5; i++)
      {
        unsigned h = (blockIdx.x * 977u + (unsigned)it * 131u + (unsigned)i) * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        const unsigned blk = h % (nodeCount * 5u / 64u);
        const uint4 v = table[(size_t)blk * 64 + lane];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
      }
    }
    else if(MODE == 3)
    {
      const unsigned node = (rng >> 8) % (nodeCount * 5u / 8u);
      const uint4* p = table + (size_t)node * 8;
#pragma unroll
      for(int q = 0; q < 8; q++) { const uint4 v = p[q]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
    else
    {
      unsigned h = (blockIdx.x * 977u + (unsigned)it * 131u + (lane >> 4)) * 2654435761u;
      h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
      const unsigned node = h % nodeCount;
      const uint4* p = table + (size_t)node * 5;
#pragma unroll
      for(int q = 0; q < 5; q++) { const uint4 v = p[q]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    }
  }
  pad[lane] = acc;
  if(acc == 0x12345678u) sink[0] = pad[(lane + 1) & 63];
}

template <int MODE>
static double run(const uint4* table, unsigned nodeCount, int iters, unsigned* sink, int blocks)
{
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  k_gather<MODE><<<blocks, 64>>>(table, nodeCount, iters / 4, sink);  // warm the caches
  CHECK(hipEventRecord(a));
  k_gather<MODE><<<blocks, 64>>>(table, nodeCount, iters, sink);
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main(int argc, char** argv)
{
  const unsigned nodeCount = argc > 1 ? (unsigned)atoi(argv[1]) : 28343u;  // the bench scene's tree: 28,343 nodes x 80 B = 2.27 MB
  const int iters = argc > 2 ? atoi(argv[2]) : 2000;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus * 20;  // 5 one-wave workgroups per SIMD
  std::vector<uint4> h((size_t)nodeCount * 5 + 64);
  for(size_t i = 0; i < h.size(); i++) h[i] = make_uint4((unsigned)i * 2654435761u, (unsigned)i, ~(unsigned)i, 7u);
  uint4* table; unsigned* sink;
  CHECK(hipMalloc(&table, h.size() * sizeof(uint4)));
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMemcpy(table, h.data(), h.size() * sizeof(uint4), hipMemcpyHostToDevice));
  const char* names[5] = {"per-lane 80-B node, 5 x dwordx4", "cooperative: 5 lanes per node, 12 nodes per instruction", "contiguous 1 KiB per instruction",
                          "per-lane 128-B node, 8 x dwordx4", "per-lane 80-B node, 16 lanes share a node"};
  const int instr[5] = {5, 5, 5, 8, 5};
  double ms[5] = {run<0>(table, nodeCount, iters, sink, blocks), run<1>(table, nodeCount, iters, sink, blocks), run<2>(table, nodeCount, iters, sink, blocks),
                  run<3>(table, nodeCount, iters, sink, blocks), run<4>(table, nodeCount, iters, sink, blocks)};
  printf("{\"device\": \"%s\", \"cus\": %d, \"waves_per_cu\": 20, \"table_bytes\": %zu, \"iters\": %d, \"modes\": [", prop.name, cus, (size_t)nodeCount * 80, iters);
  for(int m = 0; m < 5; m++)
  {
    const double waveInstrPerCu = 20.0 * iters * instr[m];
    const double ns = ms[m] * 1e6 / waveInstrPerCu;
    const double lanes = m == 1 ? 60.0 : 64.0;
    printf("%s{\"mode\": %d, \"shape\": \"%s\", \"ms\": %.3f, \"ns_per_wave_instr_per_cu\": %.2f, \"GBps_per_cu\": %.1f, \"TBps_chip\": %.2f}", m ? ", " : "", m, names[m],
           ms[m], ns, lanes * 16.0 / ns, lanes * 16.0 / ns * cus / 1e3);
  }
  printf("]}\n");
  return 0;
}
