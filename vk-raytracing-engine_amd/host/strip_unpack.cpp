// strip_unpack.cpp -- the un-interleave kernel of the multi-GPU gather (strip_gather.h); no RCCL dependency, part of
// libvkrt_host.so so that the CPU tests can reach the row map through the C API.
#include "strip_gather.h"

namespace vkrt_host {

__global__ void k_unpack_strips(const float4* __restrict__ gathered, float4* __restrict__ full, StripLayout L, uint32_t cap)
{
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if(x >= L.width)
    return;
  uint32_t rank, local;
  L.source(y, rank, local);
  full[(size_t)y * L.width + x] = gathered[((size_t)rank * cap + local) * L.width + x];
}

hipError_t unpackStrips(const float* gathered, float* full, const StripLayout& L, hipStream_t stream)
{
  if(L.width == 0 || L.height == 0)
    return hipSuccess;
  const dim3 block(256), grid((L.width + 255) / 256, L.height);
  hipLaunchKernelGGL(k_unpack_strips, grid, block, 0, stream, (const float4*)gathered, (float4*)full, L, L.capRows());
  return hipGetLastError();
}

}  // namespace vkrt_host
