// queue_probe.hip -- how many kernels of different HIP streams run at the same time on this device, and what a kernel boundary costs
// when several streams run chains of dependent kernels (the shape of the wavefront pipeline's lanes: wavefront.hip).
//   hipcc --offload-arch=gfx950 -O2 -o tools/build/queue_probe tools/queue_probe.hip && tools/build/queue_probe
// Part 1: K streams, ONE spinning kernel each (64 workgroups x 256 threads for `us` microseconds: 1/16 of the CUs): wall time / us
//         = 1 when all K overlap, = ceil(K / M) when at most M kernels run at once.
// Part 2: K streams, a CHAIN of n kernels each; every kernel keeps `wgs` workgroups busy for `us` microseconds.  wall time against
//         n * us: the per-boundary cost and whether chains of different streams hide each other's boundaries.
// Part 3: as part 2 with full-GPU kernels (wgs = 5120 workgroups of 64 threads, the traversal kernel's shape) whose waves end at
//         staggered times (tail), K = 1..6: what overlap between streams recovers of the tail.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin(unsigned long long ticks, int tail, unsigned* sink)
{
  // s_memrealtime: 100 MHz constant clock; `tail`: workgroup b spins (1 + (b % 8) / 8) x as long, like waves with walks of different lengths
  unsigned long long want = ticks;
  if(tail) want += ticks * (blockIdx.x % 8u) / 8u;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned x = 0;
  while(__builtin_amdgcn_s_memrealtime() - t0 < want) x++;
  if(x == 0xffffffffu) *sink = x;
}

static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main()
{
  const int maxK = 8;
  hipStream_t st[maxK];
  for(int k = 0; k < maxK; k++) (void)hipStreamCreateWithFlags(&st[k], hipStreamNonBlocking);
  unsigned* sink;
  (void)hipMalloc(&sink, 4);
  auto run = [&](int K, int n, unsigned wgs, unsigned threads, double us, int tail) {
    const unsigned long long ticks = (unsigned long long)(us * 100.0);
    double best = 1e30;
    for(int rep = 0; rep < 3; rep++)
    {
      (void)hipDeviceSynchronize();
      const double t0 = now();
      for(int i = 0; i < n; i++)
        for(int k = 0; k < K; k++) hipLaunchKernelGGL(spin, dim3(wgs), dim3(threads), 0, st[k], ticks, tail, sink);
      (void)hipDeviceSynchronize();
      best = std::min(best, now() - t0);
    }
    return best;
  };
  (void)run(1, 4, 64, 256, 100, 0);
  printf("{\"part1_one_kernel_per_stream_1000us_64wg\": {");
  for(int K = 1; K <= maxK; K++) printf("\"%d\": %.0f%s", K, run(K, 1, 64, 256, 1000, 0), K < maxK ? ", " : "},\n");
  for(double us : {20.0, 100.0})
  {
    printf(" \"part2_chain_of_200_kernels_%.0fus_64wg\": {", us);
    for(int K = 1; K <= 6; K++) printf("\"%d\": %.0f%s", K, run(K, 200, 64, 256, us, 0), K < 6 ? ", " : "},\n");
  }
  for(unsigned wgs : {5120u, 20480u})
  {
    printf(" \"part3_chain_of_100_full_gpu_kernels_50us_tail_%uwg\": {", wgs);
    for(int K = 1; K <= 6; K++) printf("\"%d\": %.0f%s", K, run(K, 100, wgs, 64, 50, 1), K < 6 ? ", " : (wgs == 20480u ? "}}\n" : "},\n"));
  }
  return 0;
}
