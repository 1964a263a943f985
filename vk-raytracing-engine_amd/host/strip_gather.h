// strip_gather.h -- multi-GPU side of the C++ host: one process per GPU renders the image strips of its vkrt_shard
// (include/vkrt.h) and the strips of all ranks are gathered with ONE RCCL all-gather per frame over xGMI, then
// un-interleaved into the full image by a small HIP kernel.  The reference renders on a single device (main.cpp:200) with
// one vkCmdTraceRaysKHR(W,H,1) (hello_vulkan.cpp:1446); this is what stands in its place on an 8-GPU node.
//
//   StripLayout      pure index math of the strip deal (shared by the kernel, the host code and the CPU tests)
//   unpackStrips     HIP kernel launch: [world][capRows][W] gathered buffer -> [H][W] image       (libvkrt_host.so)
//   StripGather      RCCL communicator + buffers + gather(); bootstrap through a shared id file   (libvkrt_gather.so)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>

namespace vkrt_host {

struct StripLayout
{
  uint32_t width = 0, height = 0, stripRows = 16, world = 1;
  uint32_t strips() const { return (height + stripRows - 1) / stripRows; }
  // rows of `rank`'s buffer (= vkrt_shard_rows)
  uint32_t rowsOf(uint32_t rank) const
  {
    uint32_t rows = 0;
    for(uint32_t s = rank; s < strips(); s += world)
      rows += (s + 1) * stripRows <= height ? stripRows : height - s * stripRows;
    return rows;
  }
  // rows every rank's send buffer is padded to (rank 0 always holds the most)
  uint32_t capRows() const { return rowsOf(0); }
  // global row y lives in rank `rank`, local row `local` (strip s = y / stripRows belongs to rank s % world)
  __host__ __device__ void source(uint32_t y, uint32_t& rank, uint32_t& local) const
  {
    const uint32_t s = y / stripRows;
    rank = s % world;
    local = (s / world) * stripRows + (y - s * stripRows);
  }
};

// full[y][x] = gathered[rank(y)][local(y)][x] for every pixel (rgba32f); one thread per pixel
hipError_t unpackStrips(const float* gathered, float* full, const StripLayout& L, hipStream_t stream);

class StripGather
{
public:
  // Collective over all ranks: rank 0 creates the RCCL unique id and publishes it in `idFile` (written to a temporary name and
  // renamed), the others wait for the file.  `device` must be the HIP device this process renders on.
  StripGather(const StripLayout& L, uint32_t rank, int device, const std::string& idFile);
  ~StripGather();
  StripGather(const StripGather&) = delete;
  StripGather& operator=(const StripGather&) = delete;
  // localStrips: this rank's rowsOf(rank) x W rgba32f rows (the image vkrt_pathtrace wrote).  Enqueues copy + ncclAllGather +
  // unpack on `stream`; fullImage() is complete when the stream reaches this point.
  void gather(const float* localStrips, hipStream_t stream);
  const float* fullImage() const { return m_full; }
  const StripLayout& layout() const { return m_layout; }

private:
  StripLayout m_layout;
  uint32_t m_rank;
  void* m_comm = nullptr;  // ncclComm_t
  float *m_send = nullptr, *m_recv = nullptr, *m_full = nullptr;
};

}  // namespace vkrt_host
