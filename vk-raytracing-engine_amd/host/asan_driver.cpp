// asan_driver.cpp -- command-line front end of the host layer's PARSERS for the sanitizer build (`make asan`, CPU only).
// The glTF / GLB loader, the PNG inflate path, the JPEG decoder and json_mini.h read untrusted files (they stand where tinygltf and
// stb_image stand in the reference, hello_vulkan.cpp:329-342,482-485); this program links them -- and nothing that needs the GPU --
// with -fsanitize=address,undefined, so the CPU suite can run their ordinary, corrupt-input and mutation cases under the sanitizers
// (tests/test_host_asan.py).  A sanitizer report ends the process with a non-zero status; a refused input is an ordinary "ERR" line.
//   vkrt_host_asan load FILE            -> OK <7 counts> <fnv of the flat arrays and textures>
//   vkrt_host_asan decode FILE          -> OK <w> <h> <fnv of the RGBA8 texels>
//   vkrt_host_asan json FILE            -> OK <type> <size>
//   vkrt_host_asan mutate KIND SEED N FILE   KIND = image | json | gltf: N random corruptions of FILE (byte flips, runs of 0x00 /
//                                            0xff, truncations, length-field edits) through the same entry points -> OK <accepted> <refused>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "gltf_scene.h"
#include "json_mini.h"

using namespace vkrt_host;

static uint64_t fnv(uint64_t h, const void* p, size_t n)
{
  const uint8_t* b = (const uint8_t*)p;
  for(size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
  return h;
}

static std::vector<uint8_t> slurp(const char* path)
{
  std::ifstream f(path, std::ios::binary);
  if(!f) throw std::runtime_error(std::string("cannot open ") + path);
  std::stringstream ss;
  ss << f.rdbuf();
  const std::string s = ss.str();
  return std::vector<uint8_t>(s.begin(), s.end());
}

static uint64_t sceneHash(const GltfScene& s)
{
  uint64_t h = 1469598103934665603ull;
  h = fnv(h, s.m_positions.data(), s.m_positions.size() * 4);
  h = fnv(h, s.m_normals.data(), s.m_normals.size() * 4);
  h = fnv(h, s.m_tangents.data(), s.m_tangents.size() * 4);
  h = fnv(h, s.m_texcoords0.data(), s.m_texcoords0.size() * 4);
  h = fnv(h, s.m_indices.data(), s.m_indices.size() * 4);
  h = fnv(h, s.m_primMeshes.data(), s.m_primMeshes.size() * sizeof(vkrt_prim_mesh));
  h = fnv(h, s.m_nodes.data(), s.m_nodes.size() * sizeof(vkrt_node));
  h = fnv(h, s.m_materials.data(), s.m_materials.size() * sizeof(GltfPBRMaterial));
  h = fnv(h, s.m_lights.data(), s.m_lights.size() * sizeof(GltfLight));
  for(const auto& t : s.m_textures)
  {
    const uint32_t whs[3] = {t.width, t.height, t.srgb ? 1u : 0u};
    h = fnv(h, whs, sizeof whs);
    h = fnv(h, t.rgba.data(), t.rgba.size());
  }
  return h;
}

static int cmdLoad(const char* path)
{
  try
  {
    const GltfScene s = loadGltf(path);
    printf("OK %u %zu %zu %zu %zu %zu %zu %016llx\n", s.vertexCount(), s.m_indices.size(), s.m_primMeshes.size(), s.m_nodes.size(),
           s.m_materials.size(), s.m_lights.size(), s.m_textures.size(), (unsigned long long)sceneHash(s));
  }
  catch(const std::exception& e) { printf("ERR %s\n", e.what()); }
  return 0;
}

static bool decodeBytes(const std::vector<uint8_t>& d, uint32_t& w, uint32_t& h, uint64_t& crc, std::string& why)
{
  TextureImage t;
  if(!decodeImageMemory(d.data(), d.size(), t, why)) return false;
  if(t.rgba.size() != (size_t)t.width * t.height * 4) throw std::logic_error("decoder returned a short image");
  w = t.width; h = t.height;
  crc = fnv(1469598103934665603ull, t.rgba.data(), t.rgba.size());
  return true;
}

static int cmdDecode(const char* path)
{
  const std::vector<uint8_t> d = slurp(path);
  uint32_t w, h; uint64_t crc; std::string why;
  if(decodeBytes(d, w, h, crc, why)) printf("OK %u %u %016llx\n", w, h, (unsigned long long)crc);
  else printf("ERR %s\n", why.c_str());
  return 0;
}

static int cmdJson(const char* path)
{
  const std::vector<uint8_t> d = slurp(path);
  try
  {
    const Json j = Json::parse(std::string(d.begin(), d.end()));
    printf("OK %d %zu\n", (int)j.type, j.size());
  }
  catch(const std::exception& e) { printf("ERR %s\n", e.what()); }
  return 0;
}

// xorshift64*: the corruptions are a pure function of (seed, case)
struct Rng
{
  uint64_t s;
  uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 2685821657736338717ull; }
  uint32_t below(uint32_t n) { return n ? (uint32_t)(next() % n) : 0u; }
};

static std::vector<uint8_t> corrupt(const std::vector<uint8_t>& src, Rng& r)
{
  std::vector<uint8_t> d = src;
  const uint32_t kind = r.below(6);
  const uint32_t n = (uint32_t)d.size();
  if(n == 0) return d;
  if(kind == 0)  // a few single-byte edits
    for(uint32_t k = 0, m = 1 + r.below(8); k < m; k++) d[r.below(n)] = (uint8_t)r.next();
  else if(kind == 1)  // truncation
    d.resize(r.below(n));
  else if(kind == 2)  // a run of one value
  {
    const uint32_t a = r.below(n), len = 1 + r.below(64);
    const uint8_t v = (r.next() & 1) ? 0xff : 0x00;
    for(uint32_t i = a; i < n && i < a + len; i++) d[i] = v;
  }
  else if(kind == 3)  // a 16- or 32-bit field set to an extreme (lengths, sizes, counts)
  {
    const uint32_t a = r.below(n);
    const uint8_t v = (r.next() & 1) ? 0xff : 0x7f;
    for(uint32_t i = a; i < n && i < a + 2 + 2 * (uint32_t)(r.next() & 1); i++) d[i] = v;
  }
  else if(kind == 4)  // a block copied over another place
  {
    const uint32_t a = r.below(n), b = r.below(n), len = 1 + r.below(256);
    for(uint32_t i = 0; i < len && a + i < n && b + i < n; i++) d[b + i] = src[a + i];
  }
  else  // bytes inserted
  {
    const uint32_t a = r.below(n), len = 1 + r.below(16);
    std::vector<uint8_t> ins(len);
    for(auto& x : ins) x = (uint8_t)r.next();
    d.insert(d.begin() + a, ins.begin(), ins.end());
  }
  return d;
}

static int cmdMutate(const char* kind, uint64_t seed, int cases, const char* path)
{
  const std::vector<uint8_t> src = slurp(path);
  Rng r{seed * 0x9e3779b97f4a7c15ull + 1};
  int ok = 0, refused = 0;
  const std::string k = kind;
  std::string tmp;
  if(k == "gltf")
  {
    const char* dir = getenv("TMPDIR");
    const std::string p = path;
    const std::string ext = p.size() > 4 && p.substr(p.size() - 4) == ".glb" ? ".glb" : ".gltf";
    tmp = std::string(dir ? dir : "/tmp") + "/vkrt_asan_mut_" + std::to_string((unsigned long long)seed) + ext;
  }
  for(int c = 0; c < cases; c++)
  {
    const std::vector<uint8_t> d = corrupt(src, r);
    if(k == "image")
    {
      uint32_t w, h; uint64_t crc; std::string why;
      (decodeBytes(d, w, h, crc, why) ? ok : refused)++;
    }
    else if(k == "json")
    {
      try { (void)Json::parse(std::string(d.begin(), d.end())); ok++; }
      catch(const std::exception&) { refused++; }
    }
    else if(k == "gltf")
    {
      // (a .gltf's external buffers and images resolve against the directory of the mutated copy: the caller puts it beside them)
      std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
      f.write((const char*)d.data(), (std::streamsize)d.size());
      f.close();
      try { const GltfScene s = loadGltf(tmp); (void)sceneHash(s); ok++; }
      catch(const std::exception&) { refused++; }
    }
    else { fprintf(stderr, "unknown kind %s\n", kind); return 2; }
  }
  if(!tmp.empty()) remove(tmp.c_str());
  printf("OK %d %d\n", ok, refused);
  return 0;
}

int main(int argc, char** argv)
{
  try
  {
    if(argc == 3 && !strcmp(argv[1], "load")) return cmdLoad(argv[2]);
    if(argc == 3 && !strcmp(argv[1], "decode")) return cmdDecode(argv[2]);
    if(argc == 3 && !strcmp(argv[1], "json")) return cmdJson(argv[2]);
    if(argc == 6 && !strcmp(argv[1], "mutate")) return cmdMutate(argv[2], strtoull(argv[3], nullptr, 10), atoi(argv[4]), argv[5]);
  }
  catch(const std::exception& e)
  {
    printf("ERR %s\n", e.what());
    return 0;
  }
  fprintf(stderr, "usage: vkrt_host_asan load|decode|json FILE | mutate image|json|gltf SEED N FILE\n");
  return 2;
}
