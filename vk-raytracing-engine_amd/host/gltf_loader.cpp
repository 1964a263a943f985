// gltf_loader.cpp -- glTF 2.0 ingest of the host layer: file -> the flat arrays the ray-tracing path
// consumes.  Stands in for tinygltf + nvh::GltfScene::importMaterials/importDrawableNodes as called by
// HelloVulkan::loadGltfScene (reference hello_vulkan.cpp:327-346) and for loadGltfMaterials (:207-224),
// loadGltfLights (:226-325) and createTextureImages/getImageFormat (:417-513).
//
// nvpro_core/tinygltf are not part of the reference tree; their behaviour is restated from
// default-scene node hierarchy flattened to world matrices; one primMesh per
// TRIANGLES primitive in mesh order; primitives sharing one attribute set share vertices; indices
// widened to u32; missing NORMAL -> per-face normals; missing TEXCOORD_0 -> 0; missing TANGENT ->
// per-vertex tangents from UV derivatives, Gram-Schmidt against the normal, w = handedness.
#include <zlib.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

#include "gltf_scene.h"
#include "json_mini.h"

namespace vkrt_host {

namespace {

std::vector<uint8_t> readFile(const std::string& path)
{
  std::ifstream f(path, std::ios::binary);
  if(!f)
    throw std::runtime_error("cannot open " + path);
  f.seekg(0, std::ios::end);
  const std::streamoff n = f.tellg();
  f.seekg(0);
  std::vector<uint8_t> d((size_t)n);
  if(n)
    f.read((char*)d.data(), n);
  return d;
}

std::string dirOf(const std::string& p)
{
  const size_t s = p.find_last_of("/\\");
  return s == std::string::npos ? std::string(".") : p.substr(0, s);
}

std::vector<uint8_t> base64Decode(const std::string& s, size_t from)
{
  static int8_t T[256];
  static bool init = false;
  if(!init)
  {
    memset(T, -1, sizeof T);
    const char* A = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    for(int i = 0; i < 64; i++) T[(uint8_t)A[i]] = (int8_t)i;
    init = true;
  }
  std::vector<uint8_t> o;
  uint32_t acc = 0;
  int bits = 0;
  for(size_t i = from; i < s.size(); i++)
  {
    const int8_t v = T[(uint8_t)s[i]];
    if(v < 0)
      continue;
    acc = (acc << 6) | (uint32_t)v;
    bits += 6;
    if(bits >= 8)
    {
      bits -= 8;
      o.push_back((uint8_t)((acc >> bits) & 0xFF));
    }
  }
  return o;
}

// ---- column-major 4x4 helpers --------------------------------------------------------------------------
struct M4
{
  float m[16];
};
M4 ident()
{
  M4 r;
  memset(&r, 0, sizeof r);
  r.m[0] = r.m[5] = r.m[10] = r.m[15] = 1.f;
  return r;
}
M4 mul(const M4& A, const M4& B)
{
  M4 R;
  for(int c = 0; c < 4; c++)
    for(int r = 0; r < 4; r++)
    {
      float s = 0.f;
      for(int k = 0; k < 4; k++)
        s += A.m[k * 4 + r] * B.m[c * 4 + k];
      R.m[c * 4 + r] = s;
    }
  return R;
}
M4 localMatrix(const Json& node)
{
  M4 T = ident(), R = ident(), S = ident(), M = ident();
  if(node.has("translation"))
    for(int k = 0; k < 3; k++) T.m[12 + k] = (float)node["translation"][k].number();
  if(node.has("rotation"))
  {
    const float x = (float)node["rotation"][0].number(), y = (float)node["rotation"][1].number(),
                z = (float)node["rotation"][2].number(), w = (float)node["rotation"][3].number(1.0);
    R.m[0] = 1 - 2 * (y * y + z * z); R.m[4] = 2 * (x * y - z * w); R.m[8] = 2 * (x * z + y * w);
    R.m[1] = 2 * (x * y + z * w); R.m[5] = 1 - 2 * (x * x + z * z); R.m[9] = 2 * (y * z - x * w);
    R.m[2] = 2 * (x * z - y * w); R.m[6] = 2 * (y * z + x * w); R.m[10] = 1 - 2 * (x * x + y * y);
  }
  if(node.has("scale"))
  {
    S.m[0] = (float)node["scale"][0].number(1.0);
    S.m[5] = (float)node["scale"][1].number(1.0);
    S.m[10] = (float)node["scale"][2].number(1.0);
  }
  if(node.has("matrix"))
    for(int k = 0; k < 16; k++) M.m[k] = (float)node["matrix"][k].number();
  return mul(mul(mul(T, R), S), M);
}

struct Doc
{
  Json g;
  std::vector<std::vector<uint8_t>> buffers;
  std::string base;
};

int compSize(int ct)
{
  switch(ct)
  {
    case 5120: case 5121: return 1;
    case 5122: case 5123: return 2;
    case 5125: case 5126: return 4;
  }
  throw std::runtime_error("gltf: unsupported componentType " + std::to_string(ct));
}
int typeCount(const std::string& t)
{
  if(t == "SCALAR") return 1;
  if(t == "VEC2") return 2;
  if(t == "VEC3") return 3;
  if(t == "VEC4") return 4;
  if(t == "MAT4") return 16;
  throw std::runtime_error("gltf: unsupported accessor type " + t);
}

// Reads accessor `idx` as doubles-free floats (ncomp per element); integers converted (normalised if flagged).
void readAccessorFloat(const Doc& d, int idx, int wantComp, std::vector<float>& out, size_t& count)
{
  const Json& a = d.g["accessors"][(size_t)idx];
  const int ct = a["componentType"].integer(), nc = typeCount(a["type"].string());
  if(nc != wantComp)
    throw std::runtime_error("gltf: accessor " + std::to_string(idx) + " has " + std::to_string(nc) + " components, expected " + std::to_string(wantComp));
  count = (size_t)a["count"].integer();
  out.assign(count * nc, 0.f);
  if(!a.has("bufferView"))
    return;
  const Json& bv = d.g["bufferViews"][(size_t)a["bufferView"].integer()];
  const std::vector<uint8_t>& buf = d.buffers.at((size_t)bv["buffer"].integer());
  const size_t off = (size_t)bv["byteOffset"].integer(0) + (size_t)a["byteOffset"].integer(0);
  const int cs = compSize(ct);
  const size_t item = (size_t)cs * nc;
  const size_t stride = bv.has("byteStride") && bv["byteStride"].integer() > 0 ? (size_t)bv["byteStride"].integer() : item;
  if(count && off + (count - 1) * stride + item > buf.size())
    throw std::runtime_error("gltf: accessor " + std::to_string(idx) + " overruns its buffer");
  const bool norm = a["normalized"].boolean(false);
  for(size_t i = 0; i < count; i++)
  {
    const uint8_t* p = buf.data() + off + i * stride;
    for(int c = 0; c < nc; c++)
    {
      float v;
      switch(ct)
      {
        case 5126: memcpy(&v, p + 4 * c, 4); break;
        case 5120: { int8_t x; memcpy(&x, p + c, 1); v = norm ? std::max((float)x / 127.f, -1.f) : (float)x; break; }
        case 5121: { uint8_t x = p[c]; v = norm ? (float)x / 255.f : (float)x; break; }
        case 5122: { int16_t x; memcpy(&x, p + 2 * c, 2); v = norm ? std::max((float)x / 32767.f, -1.f) : (float)x; break; }
        case 5123: { uint16_t x; memcpy(&x, p + 2 * c, 2); v = norm ? (float)x / 65535.f : (float)x; break; }
        default: { uint32_t x; memcpy(&x, p + 4 * c, 4); v = (float)x; break; }
      }
      out[i * nc + c] = v;
    }
  }
}

void readIndices(const Doc& d, int idx, std::vector<uint32_t>& out)
{
  const Json& a = d.g["accessors"][(size_t)idx];
  const int ct = a["componentType"].integer();
  const size_t count = (size_t)a["count"].integer();
  out.assign(count, 0u);
  if(!a.has("bufferView"))
    return;
  const Json& bv = d.g["bufferViews"][(size_t)a["bufferView"].integer()];
  const std::vector<uint8_t>& buf = d.buffers.at((size_t)bv["buffer"].integer());
  const size_t off = (size_t)bv["byteOffset"].integer(0) + (size_t)a["byteOffset"].integer(0);
  const int cs = compSize(ct);
  const size_t stride = bv.has("byteStride") && bv["byteStride"].integer() > 0 ? (size_t)bv["byteStride"].integer() : (size_t)cs;
  if(count && off + (count - 1) * stride + cs > buf.size())
    throw std::runtime_error("gltf: index accessor overruns its buffer");
  for(size_t i = 0; i < count; i++)
  {
    const uint8_t* p = buf.data() + off + i * stride;
    if(cs == 1) out[i] = p[0];
    else if(cs == 2) { uint16_t x; memcpy(&x, p, 2); out[i] = x; }
    else { uint32_t x; memcpy(&x, p, 4); out[i] = x; }
  }
}

int textureIndex(const Json& obj, const char* key)
{
  const Json& t = obj[key];
  return (t.isObject() && t.has("index")) ? t["index"].integer() : -1;
}

struct V3
{
  float x, y, z;
};
inline V3 sub3(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 cross3(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot3(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

// Per-vertex tangents from UV derivatives (Lengyel), accumulate per triangle, orthonormalise.
void generateTangents(const float* pos, const float* nrm, const float* uv, size_t vcount, const uint32_t* idx, size_t icount,
                      std::vector<float>& out4)
{
  std::vector<V3> tan(vcount, V3{0, 0, 0}), bit(vcount, V3{0, 0, 0});
  for(size_t i = 0; i + 2 < icount; i += 3)
  {
    const uint32_t i0 = idx[i], i1 = idx[i + 1], i2 = idx[i + 2];
    const V3 p0{pos[3 * i0], pos[3 * i0 + 1], pos[3 * i0 + 2]}, p1{pos[3 * i1], pos[3 * i1 + 1], pos[3 * i1 + 2]},
        p2{pos[3 * i2], pos[3 * i2 + 1], pos[3 * i2 + 2]};
    const V3 e1 = sub3(p1, p0), e2 = sub3(p2, p0);
    const float du1 = uv[2 * i1] - uv[2 * i0], dv1 = uv[2 * i1 + 1] - uv[2 * i0 + 1];
    const float du2 = uv[2 * i2] - uv[2 * i0], dv2 = uv[2 * i2 + 1] - uv[2 * i0 + 1];
    const float a = du1 * dv2 - du2 * dv1;
    const float r = std::fabs(a) > 0.f ? 1.0f / a : 1.0f;
    const V3 t{(e1.x * dv2 - e2.x * dv1) * r, (e1.y * dv2 - e2.y * dv1) * r, (e1.z * dv2 - e2.z * dv1) * r};
    const V3 b{(e2.x * du1 - e1.x * du2) * r, (e2.y * du1 - e1.y * du2) * r, (e2.z * du1 - e1.z * du2) * r};
    for(uint32_t v : {i0, i1, i2})
    {
      tan[v].x += t.x; tan[v].y += t.y; tan[v].z += t.z;
      bit[v].x += b.x; bit[v].y += b.y; bit[v].z += b.z;
    }
  }
  out4.resize(vcount * 4);
  for(size_t v = 0; v < vcount; v++)
  {
    const V3 n{nrm[3 * v], nrm[3 * v + 1], nrm[3 * v + 2]}, t = tan[v];
    const float nd = dot3(n, t);
    V3 o{t.x - nd * n.x, t.y - nd * n.y, t.z - nd * n.z};
    const float l = std::sqrt(dot3(o, o));
    if(l > 0.f && std::isfinite(l))
      o = V3{o.x / l, o.y / l, o.z / l};
    else
    {  // degenerate UVs: axis fallback (same construction as random.glsl:47-54)
      if(std::fabs(n.x) > std::fabs(n.y))
      {
        const float s = std::sqrt(n.x * n.x + n.z * n.z);
        o = V3{n.z / s, 0.f, -n.x / s};
      }
      else
      {
        const float s = std::sqrt(n.y * n.y + n.z * n.z);
        o = V3{0.f, -n.z / s, n.y / s};
      }
      if(!std::isfinite(o.x) || !std::isfinite(o.y) || !std::isfinite(o.z) || (o.x == 0.f && o.y == 0.f && o.z == 0.f))
        o = V3{1.f, 0.f, 0.f};
    }
    const float hand = dot3(cross3(n, t), bit[v]) < 0.f ? 1.0f : -1.0f;
    out4[4 * v] = o.x; out4[4 * v + 1] = o.y; out4[4 * v + 2] = o.z; out4[4 * v + 3] = hand;
  }
}

GltfLight makeLight(float x, float y, float z, float r, float g, float b)
{
  GltfLight l;
  l.position[0] = x; l.position[1] = y; l.position[2] = z;
  l.color[0] = r; l.color[1] = g; l.color[2] = b;
  l.intensity = 50.0f;
  l.type = 0;
  return l;
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// PNG (8/16-bit, colour types 0/2/3/4/6, non-interlaced) -> RGBA8
// ---------------------------------------------------------------------------------------------------------
bool decodePngMemory(const uint8_t* data, size_t size, TextureImage& out, std::string& why)
{
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if(size < 8 || memcmp(data, sig, 8) != 0) { why = "not a PNG"; return false; }
  auto be32 = [](const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; };
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat, plte, trns;
  size_t i = 8;
  while(i + 12 <= size)
  {
    const uint32_t len = be32(data + i);
    const char* tag = (const char*)data + i + 4;
    const uint8_t* body = data + i + 8;
    if(i + 12 + (size_t)len > size) { why = "truncated chunk"; return false; }
    if(!memcmp(tag, "IHDR", 4))
    {
      w = be32(body); h = be32(body + 4); depth = body[8]; ctype = body[9]; interlace = body[12];
    }
    else if(!memcmp(tag, "PLTE", 4)) plte.assign(body, body + len);
    else if(!memcmp(tag, "tRNS", 4)) trns.assign(body, body + len);
    else if(!memcmp(tag, "IDAT", 4)) idat.insert(idat.end(), body, body + len);
    else if(!memcmp(tag, "IEND", 4)) break;
    i += 12 + (size_t)len;
  }
  if(w == 0 || h == 0) { why = "missing IHDR"; return false; }
  if(w > 65535u || h > 65535u || (uint64_t)w * h > (1ull << 28)) { why = "PNG larger than the texture limits (65535 per side, 2^28 texels)"; return false; }
  if(interlace) { why = "interlaced PNG not supported"; return false; }
  int channels;
  switch(ctype)
  {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default: why = "bad colour type"; return false;
  }
  if(!(depth == 8 || depth == 16 || ((ctype == 0 || ctype == 3) && (depth == 1 || depth == 2 || depth == 4)))) { why = "unsupported bit depth"; return false; }
  const size_t bitsPerPixel = (size_t)channels * depth;
  const size_t rowBytes = (w * bitsPerPixel + 7) / 8;
  const size_t bpp = std::max<size_t>(1, bitsPerPixel / 8);
  // (deflate expands at most ~1032 : 1: a header that promises more than the data can hold is corrupt -- do not allocate for it)
  if((rowBytes + 1) * (size_t)h > idat.size() * 1040 + 1024) { why = "PNG data too short for its header"; return false; }
  std::vector<uint8_t> raw((rowBytes + 1) * (size_t)h);
  uLongf rawLen = (uLongf)raw.size();
  if(uncompress(raw.data(), &rawLen, idat.data(), (uLong)idat.size()) != Z_OK || rawLen != raw.size()) { why = "zlib inflate failed"; return false; }
  std::vector<uint8_t> prev(rowBytes, 0), cur(rowBytes);
  out.width = w; out.height = h;
  out.rgba.assign((size_t)w * h * 4, 255);
  for(uint32_t y = 0; y < h; y++)
  {
    const uint8_t* src = raw.data() + (size_t)y * (rowBytes + 1);
    const int filter = src[0];
    for(size_t x = 0; x < rowBytes; x++)
    {
      const int a = x >= bpp ? cur[x - bpp] : 0, b = prev[x], c = x >= bpp ? prev[x - bpp] : 0;
      int v = src[1 + x];
      switch(filter)
      {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4:
        {
          const int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
          v += (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
          break;
        }
        default: why = "bad filter"; return false;
      }
      cur[x] = (uint8_t)v;
    }
    uint8_t* dst = &out.rgba[(size_t)y * w * 4];
    auto sample = [&](size_t px, int ch) -> uint32_t {
      if(depth == 8) return cur[px * channels + ch];
      if(depth == 16) return cur[(px * channels + ch) * 2];  // high byte
      const size_t bit = px * depth;
      return (cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1u << depth) - 1u);
    };
    for(uint32_t x = 0; x < w; x++)
    {
      uint8_t* o = dst + 4 * x;
      if(ctype == 3)
      {
        const uint32_t k = sample(x, 0);
        if(3 * k + 2 < plte.size()) { o[0] = plte[3 * k]; o[1] = plte[3 * k + 1]; o[2] = plte[3 * k + 2]; }
        o[3] = k < trns.size() ? trns[k] : 255;
      }
      else if(ctype == 0 || ctype == 4)
      {
        uint32_t g = sample(x, 0);
        if(depth < 8) g = g * 255u / ((1u << depth) - 1u);
        o[0] = o[1] = o[2] = (uint8_t)g;
        o[3] = ctype == 4 ? (uint8_t)sample(x, 1) : 255;
      }
      else
      {
        o[0] = (uint8_t)sample(x, 0); o[1] = (uint8_t)sample(x, 1); o[2] = (uint8_t)sample(x, 2);
        o[3] = ctype == 6 ? (uint8_t)sample(x, 3) : 255;
      }
    }
    std::swap(prev, cur);
  }
  return true;
}

bool decodeImageMemory(const uint8_t* data, size_t size, TextureImage& out, std::string& why)
{
  if(size >= 2 && data[0] == 0xff && data[1] == 0xd8)
    return decodeJpegMemory(data, size, out, why);
  return decodePngMemory(data, size, out, why);
}

bool decodeImageFile(const std::string& path, TextureImage& out, std::string& why)
{
  // a raw sidecar wins when present (tools/decode_textures.py: formats this loader does not read, or a pinned decode)
  {
    std::ifstream f(path + ".rgba8", std::ios::binary);
    if(f)
    {
      uint32_t wh[2];
      f.read((char*)wh, 8);
      if(f && wh[0] && wh[1] && (uint64_t)wh[0] * wh[1] < (1ull << 28))
      {
        out.width = wh[0]; out.height = wh[1];
        out.rgba.resize((size_t)wh[0] * wh[1] * 4);
        f.read((char*)out.rgba.data(), (std::streamsize)out.rgba.size());
        if(f)
          return true;
      }
      why = "bad .rgba8 sidecar";
      return false;
    }
  }
  std::vector<uint8_t> d;
  try { d = readFile(path); } catch(const std::exception& e) { why = e.what(); return false; }
  return decodeImageMemory(d.data(), d.size(), out, why);
}

// ---------------------------------------------------------------------------------------------------------
GltfScene loadGltf(const std::string& filename)
{
  Doc doc;
  doc.base = dirOf(filename);
  std::vector<uint8_t> file = readFile(filename);
  std::vector<uint8_t> glbBin;
  if(file.size() >= 12 && !memcmp(file.data(), "glTF", 4))
  {  // .glb (LoadBinaryFromFile, hello_vulkan.cpp:340)
    size_t i = 12;
    std::string jsonText;
    while(i + 8 <= file.size())
    {
      uint32_t len, type;
      memcpy(&len, &file[i], 4);
      memcpy(&type, &file[i + 4], 4);
      if(i + 8 + (size_t)len > file.size())
        throw std::runtime_error("glb: truncated chunk");
      if(type == 0x4E4F534Au) jsonText.assign((const char*)&file[i + 8], len);
      else if(type == 0x004E4942u) glbBin.assign(file.begin() + (long)i + 8, file.begin() + (long)i + 8 + len);
      i += 8 + (size_t)len;
    }
    doc.g = Json::parse(jsonText);
  }
  else
    doc.g = Json::parse(std::string((const char*)file.data(), file.size()));
  const Json& g = doc.g;

  for(size_t b = 0; b < g["buffers"].size(); b++)
  {
    const Json& bj = g["buffers"][b];
    if(!bj.has("uri"))
      doc.buffers.push_back(glbBin);
    else
    {
      const std::string uri = bj["uri"].string();
      if(uri.compare(0, 5, "data:") == 0)
        doc.buffers.push_back(base64Decode(uri, uri.find(',') + 1));
      else
        doc.buffers.push_back(readFile(doc.base + "/" + uri));
    }
  }

  GltfScene sc;
  // ---- materials (importMaterials + loadGltfMaterials) ---------------------------------------------------
  auto defaultMat = []() {
    GltfPBRMaterial m;
    memset(&m, 0, sizeof m);
    for(int k = 0; k < 4; k++) m.pbrBaseColorFactor[k] = 1.f;
    m.metallicFactor = 1.f; m.roughnessFactor = 1.f;
    m.pbrBaseColorTexture = m.metallicRoughnessTexture = m.normalTexture = m.emissiveTexture = -1;
    return m;
  };
  for(size_t i = 0; i < g["materials"].size(); i++)
  {
    const Json& mj = g["materials"][i];
    const Json& pbr = mj["pbrMetallicRoughness"];
    GltfPBRMaterial m = defaultMat();
    if(pbr.has("baseColorFactor"))
      for(int k = 0; k < 4; k++) m.pbrBaseColorFactor[k] = (float)pbr["baseColorFactor"][(size_t)k].number(1.0);
    m.pbrBaseColorTexture = textureIndex(pbr, "baseColorTexture");
    m.metallicFactor = (float)pbr["metallicFactor"].number(1.0);
    m.roughnessFactor = (float)pbr["roughnessFactor"].number(1.0);
    m.metallicRoughnessTexture = textureIndex(pbr, "metallicRoughnessTexture");
    m.normalTexture = textureIndex(mj, "normalTexture");
    if(mj.has("emissiveFactor"))
      for(int k = 0; k < 3; k++) m.emissiveFactor[k] = (float)mj["emissiveFactor"][(size_t)k].number(0.0);
    m.emissiveTexture = textureIndex(mj, "emissiveTexture");
    sc.m_materials.push_back(m);
  }
  if(sc.m_materials.empty())
    sc.m_materials.push_back(defaultMat());

  // ---- primitive meshes (processMesh, in mesh order; shared attribute sets are cached) -------------------
  std::vector<std::vector<uint32_t>> meshToPrims(g["meshes"].size());
  std::map<std::string, std::pair<uint32_t, uint32_t>> cache;  // attribute key -> (vertexOffset, vertexCount)
  for(size_t mi = 0; mi < g["meshes"].size(); mi++)
  {
    const Json& prims = g["meshes"][mi]["primitives"];
    for(size_t pi = 0; pi < prims.size(); pi++)
    {
      const Json& prim = prims[pi];
      if(prim.has("mode") && prim["mode"].integer() != 4)
        continue;
      const Json& attr = prim["attributes"];
      if(!attr.has("POSITION"))
        throw std::runtime_error("gltf: primitive without POSITION");
      std::vector<std::pair<std::string, int>> keyv;
      for(const auto& kv : attr.obj) keyv.emplace_back(kv.first, kv.second.integer());
      std::sort(keyv.begin(), keyv.end());
      std::ostringstream key;
      for(const auto& kv : keyv) key << kv.first << kv.second << ";";

      std::vector<float> pos;
      size_t vcount = 0;
      readAccessorFloat(doc, attr["POSITION"].integer(), 3, pos, vcount);
      std::vector<uint32_t> idx;
      if(prim.has("indices"))
        readIndices(doc, prim["indices"].integer(), idx);
      else
      {
        idx.resize(vcount);
        for(size_t k = 0; k < vcount; k++) idx[k] = (uint32_t)k;
      }
      idx.resize((idx.size() / 3) * 3);
      for(uint32_t v : idx)
        if(v >= vcount)
          throw std::runtime_error("gltf: index out of range");

      vkrt_prim_mesh pm{};
      pm.firstIndex = (uint32_t)sc.m_indices.size();
      pm.indexCount = (uint32_t)idx.size();
      pm.materialIndex = prim.has("material") ? prim["material"].integer() : -1;
      auto hit = cache.find(key.str());
      if(hit != cache.end())
      {
        pm.vertexOffset = hit->second.first;
        pm.vertexCount = hit->second.second;
      }
      else
      {
        pm.vertexOffset = sc.vertexCount();
        pm.vertexCount = (uint32_t)vcount;
        cache[key.str()] = {pm.vertexOffset, pm.vertexCount};
        std::vector<float> nrm, uv, tan;
        size_t n2 = 0;
        if(attr.has("NORMAL"))
        {
          readAccessorFloat(doc, attr["NORMAL"].integer(), 3, nrm, n2);
          if(n2 != vcount) throw std::runtime_error("gltf: NORMAL count mismatch");
        }
        else
        {
          nrm.assign(vcount * 3, 0.f);
          for(size_t k = 0; k + 2 < idx.size(); k += 3)
          {
            const uint32_t i0 = idx[k], i1 = idx[k + 1], i2 = idx[k + 2];
            const V3 p0{pos[3 * i0], pos[3 * i0 + 1], pos[3 * i0 + 2]}, p1{pos[3 * i1], pos[3 * i1 + 1], pos[3 * i1 + 2]},
                p2{pos[3 * i2], pos[3 * i2 + 1], pos[3 * i2 + 2]};
            V3 n = cross3(sub3(p1, p0), sub3(p2, p0));
            const float l = std::sqrt(dot3(n, n));
            n = V3{n.x / l, n.y / l, n.z / l};
            for(uint32_t v : {i0, i1, i2}) { nrm[3 * v] = n.x; nrm[3 * v + 1] = n.y; nrm[3 * v + 2] = n.z; }
          }
        }
        if(attr.has("TEXCOORD_0"))
        {
          readAccessorFloat(doc, attr["TEXCOORD_0"].integer(), 2, uv, n2);
          if(n2 != vcount) throw std::runtime_error("gltf: TEXCOORD_0 count mismatch");
        }
        else
          uv.assign(vcount * 2, 0.f);
        if(attr.has("TANGENT"))
        {
          readAccessorFloat(doc, attr["TANGENT"].integer(), 4, tan, n2);
          if(n2 != vcount) throw std::runtime_error("gltf: TANGENT count mismatch");
        }
        else
          generateTangents(pos.data(), nrm.data(), uv.data(), vcount, idx.data(), idx.size(), tan);
        sc.m_positions.insert(sc.m_positions.end(), pos.begin(), pos.end());
        sc.m_normals.insert(sc.m_normals.end(), nrm.begin(), nrm.end());
        sc.m_texcoords0.insert(sc.m_texcoords0.end(), uv.begin(), uv.end());
        sc.m_tangents.insert(sc.m_tangents.end(), tan.begin(), tan.end());
      }
      sc.m_indices.insert(sc.m_indices.end(), idx.begin(), idx.end());
      meshToPrims[mi].push_back((uint32_t)sc.m_primMeshes.size());
      sc.m_primMeshes.push_back(pm);
    }
  }

  // ---- node hierarchy (processNode) -------------------------------------------------------------------
  struct LightRef { M4 world; int light; };
  std::vector<LightRef> lightRefs;
  std::vector<std::pair<int, M4>> stack;
  const Json& scenes = g["scenes"];
  const size_t sceneIdx = (size_t)g["scene"].integer(0);
  if(scenes.size() > sceneIdx)
  {
    const Json& roots = scenes[sceneIdx]["nodes"];
    for(size_t k = roots.size(); k-- > 0;)
      stack.emplace_back(roots[k].integer(), ident());
  }
  size_t guard = 0;
  while(!stack.empty())
  {
    if(++guard > 10000000) throw std::runtime_error("gltf: node graph too large or cyclic");
    const auto [ni, parent] = stack.back();
    stack.pop_back();
    const Json& node = g["nodes"][(size_t)ni];
    if(node.isNull()) throw std::runtime_error("gltf: bad node index");
    const M4 world = mul(parent, localMatrix(node));
    if(node.has("mesh"))
    {
      const size_t mi = (size_t)node["mesh"].integer();
      if(mi >= meshToPrims.size()) throw std::runtime_error("gltf: bad mesh index");
      for(uint32_t pmi : meshToPrims[mi])
      {
        vkrt_node n;
        memcpy(n.worldMatrix, world.m, sizeof n.worldMatrix);
        n.primMesh = (int32_t)pmi;
        sc.m_nodes.push_back(n);
      }
    }
    const Json& lext = node["extensions"]["KHR_lights_punctual"];
    if(lext.isObject() && lext.has("light"))
      lightRefs.push_back(LightRef{world, lext["light"].integer()});
    const Json& ch = node["children"];
    for(size_t k = ch.size(); k-- > 0;)
      stack.emplace_back(ch[k].integer(), world);
  }

  // ---- lights (loadGltfLights, hello_vulkan.cpp:226-325) ---------------------------------------------------
  const Json& glights = g["extensions"]["KHR_lights_punctual"]["lights"];
  for(const LightRef& lr : lightRefs)
  {
    const Json& lj = glights[(size_t)lr.light];
    GltfLight l;
    l.position[0] = lr.world.m[12]; l.position[1] = lr.world.m[13]; l.position[2] = lr.world.m[14];
    for(int k = 0; k < 3; k++) l.color[k] = lj.has("color") ? (float)lj["color"][(size_t)k].number(1.0) : 1.f;
    l.intensity = (float)lj["intensity"].number(1.0);
    const std::string t = lj["type"].string("point");
    l.type = t == "point" ? 0 : t == "directional" ? 1 : t == "spot" ? 2 : 0;
    sc.m_lights.push_back(l);
  }
  if(sc.m_lights.empty())
  {  // the reference's 8 hard-coded point lights, intensity 50 (hello_vulkan.cpp:255-320)
    sc.m_lights.push_back(makeLight(1.0f, 5.0f, -1.33f, 1.f, 1.f, 1.f));
    sc.m_lights.push_back(makeLight(0.f, 3.f, 67.f, 1.0f, 0.01f, 0.1f));
    sc.m_lights.push_back(makeLight(-1.3f, 7.62f, 59.f, 1.f, 1.f, 1.f));
    sc.m_lights.push_back(makeLight(2.4f, 2.05f, 40.6f, 1.f, 1.f, 1.f));
    sc.m_lights.push_back(makeLight(-0.33f, 6.85f, 30.f, 1.f, 1.f, 1.f));
    sc.m_lights.push_back(makeLight(-6.2f, 9.6f, 20.18f, 1.f, 1.f, 1.f));
    sc.m_lights.push_back(makeLight(-0.23f, 6.93f, 12.21f, 1.0f, 1.0f, 0.0f));
    sc.m_lights.push_back(makeLight(0.24f, 3.03f, 49.94f, 0.0f, 0.0f, 1.0f));
  }

  // ---- textures (createTextureImages + getImageFormat, hello_vulkan.cpp:417-513) ----------------------------
  const Json& images = g["images"];
  const Json& textures = g["textures"];
  if(images.size() > 0 && textures.size() > 0)
  {
    std::vector<TextureImage> decoded(images.size());
    std::vector<bool> ok(images.size(), false);
    for(size_t i = 0; i < images.size(); i++)
    {
      const Json& im = images[i];
      std::string why;
      if(im.has("uri"))
      {
        const std::string uri = im["uri"].string();
        if(uri.compare(0, 5, "data:") == 0)
        {
          const std::vector<uint8_t> d = base64Decode(uri, uri.find(',') + 1);
          ok[i] = decodeImageMemory(d.data(), d.size(), decoded[i], why);
        }
        else
          ok[i] = decodeImageFile(doc.base + "/" + uri, decoded[i], why);
      }
      else if(im.has("bufferView"))
      {
        const Json& bv = g["bufferViews"][(size_t)im["bufferView"].integer()];
        const std::vector<uint8_t>& buf = doc.buffers.at((size_t)bv["buffer"].integer());
        const size_t off = (size_t)bv["byteOffset"].integer(0), len = (size_t)bv["byteLength"].integer(0);
        if(off + len <= buf.size())
          ok[i] = decodeImageMemory(buf.data() + off, len, decoded[i], why);
      }
      if(!ok[i])
      {  // addDefaultTexture: 1x1 white (hello_vulkan.cpp:458-472,487-491)
        decoded[i].width = decoded[i].height = 1;
        decoded[i].rgba.assign(4, 255);
        sc.warnings += "image " + std::to_string(i) + ": " + why + " -> 1x1 white; ";
      }
      // getImageFormat: sRGB iff the first texture using this image is some material's base-colour or emissive map
      int texId = -1;
      for(size_t j = 0; j < textures.size(); j++)
        if((size_t)textures[j]["source"].integer(0) == i) { texId = (int)j; break; }
      bool srgb = false;
      if(texId > -1)
        for(size_t m = 0; m < g["materials"].size(); m++)
        {
          const Json& mj = g["materials"][m];
          if(textureIndex(mj["pbrMetallicRoughness"], "baseColorTexture") == texId || textureIndex(mj, "emissiveTexture") == texId)
          {
            srgb = true;
            break;
          }
        }
      decoded[i].srgb = srgb;
    }
    for(size_t t = 0; t < textures.size(); t++)
    {
      const size_t src = (size_t)textures[t]["source"].integer(0);
      if(src >= decoded.size())
        throw std::runtime_error("gltf: texture source out of range");
      sc.m_textures.push_back(decoded[src]);
    }
  }
  return sc;
}

}  // namespace vkrt_host
