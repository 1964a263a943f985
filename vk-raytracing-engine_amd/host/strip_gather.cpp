// strip_gather.cpp -- RCCL side of the multi-GPU gather (strip_gather.h): libvkrt_gather.so, linked by vkrt_render.
#include "strip_gather.h"

#include <fcntl.h>
#include <unistd.h>

#include <rccl/rccl.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <stdexcept>
#include <thread>

namespace vkrt_host {

namespace {
void hipOk(hipError_t e, const char* what)
{
  if(e != hipSuccess)
    throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
void ncclOk(ncclResult_t r, const char* what)
{
  if(r != ncclSuccess)
    throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}
}  // namespace

StripGather::StripGather(const StripLayout& L, uint32_t rank, int device, const std::string& idFile) : m_layout(L), m_rank(rank)
{
  if(L.world == 0 || rank >= L.world)
    throw std::runtime_error("StripGather: bad rank / world");
  hipOk(hipSetDevice(device), "hipSetDevice");
  ncclUniqueId id;
  if(rank == 0)
  {
    ncclOk(ncclGetUniqueId(&id), "ncclGetUniqueId");
    // written under a temporary name and renamed, so a reader never sees a partial id; the temporary file is created exclusively
    // and without following a symbolic link (a pre-planted name makes the open fail instead of redirecting the write)
    const std::string tmp = idFile + ".tmp";
    const int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_EXCL | O_NOFOLLOW, 0600);
    bool ok = fd >= 0 && write(fd, &id, sizeof id) == (ssize_t)sizeof id;
    if(fd >= 0) ok = (close(fd) == 0) && ok;
    if(!ok || rename(tmp.c_str(), idFile.c_str()) != 0)
    {
      if(fd >= 0) unlink(tmp.c_str());
      throw std::runtime_error("StripGather: cannot publish the RCCL id in " + idFile);
    }
  }
  else
  {
    bool ok = false;
    for(int tries = 0; tries < 6000 && !ok; tries++)  // up to 60 s for rank 0 to come up
    {
      std::ifstream f(idFile, std::ios::binary);
      if(f && f.read((char*)&id, sizeof id) && f.gcount() == (std::streamsize)sizeof id)
        ok = true;
      else
        std::this_thread::sleep_for(std::chrono::milliseconds(10));
    }
    if(!ok)
      throw std::runtime_error("StripGather: timed out waiting for the RCCL id in " + idFile);
  }
  ncclComm_t comm;
  ncclOk(ncclCommInitRank(&comm, (int)L.world, id, (int)rank), "ncclCommInitRank");
  m_comm = comm;
  const size_t rowBytes = (size_t)L.width * 16, cap = L.capRows();
  hipOk(hipMalloc((void**)&m_send, std::max<size_t>(cap * rowBytes, 16)), "hipMalloc(send)");
  hipOk(hipMalloc((void**)&m_recv, std::max<size_t>((size_t)L.world * cap * rowBytes, 16)), "hipMalloc(recv)");
  hipOk(hipMalloc((void**)&m_full, std::max<size_t>((size_t)L.height * rowBytes, 16)), "hipMalloc(full)");
  hipOk(hipMemset(m_send, 0, std::max<size_t>(cap * rowBytes, 16)), "hipMemset(send)");
}

StripGather::~StripGather()
{
  if(m_comm) (void)ncclCommDestroy((ncclComm_t)m_comm);
  if(m_send) (void)hipFree(m_send);
  if(m_recv) (void)hipFree(m_recv);
  if(m_full) (void)hipFree(m_full);
}

void StripGather::gather(const float* localStrips, hipStream_t stream)
{
  const StripLayout& L = m_layout;
  const size_t rowFloats = (size_t)L.width * 4, cap = L.capRows();
  hipOk(hipMemcpyAsync(m_send, localStrips, (size_t)L.rowsOf(m_rank) * rowFloats * 4, hipMemcpyDeviceToDevice, stream), "hipMemcpyAsync(strips)");
  // ring all-gather over xGMI: every rank contributes capRows x W x rgba32f (16.6 MB per rank for 3840x2160 over 8 GPUs)
  ncclOk(ncclAllGather(m_send, m_recv, cap * rowFloats, ncclFloat, (ncclComm_t)m_comm, stream), "ncclAllGather");
  hipOk(unpackStrips(m_recv, m_full, L, stream), "unpackStrips");
}

}  // namespace vkrt_host
