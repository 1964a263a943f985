// jpeg_decode.cpp -- JPEG (JFIF / Exif, 8-bit Huffman: baseline, extended-sequential and progressive) -> RGBA8.
//
// The reference gets its texels from tinygltf -> stb_image (hello_vulkan.cpp:24-26 pulls in the implementation, :482-485 reads
// gltfimage.image).  stb_image is a third-party dependency that is not in the reference tree (it lives in nvpro_core's
// third_party), so this file restates the decoder from the JPEG standard (ITU-T T.81) and follows stb_image's published
// choices where the standard leaves the arithmetic open, so that texel values come out the same:
//   * inverse DCT: the 13-multiply integer LL&M scheme in 12-bit fixed point, columns rounded to 10 fractional bits dropped
//     (+512 >> 10), rows with +65536 + (128 << 17) >> 17, then clamped;
//   * chroma upsampling: 2x horizontally and/or vertically by the 3/4 - 1/4 triangle filter (+8 >> 4 for both directions,
//     +2 >> 2 for one), any other ratio by replication;
//   * YCbCr -> RGB in 20-bit fixed point (coefficients rounded to 12 bits, then << 8), the Cb term of green truncated to its
//     upper 16 bits;
//   * grey images replicate Y; alpha is 255.
// CMYK / YCCK (4 components), 12-bit samples, arithmetic coding and lossless / hierarchical modes are refused with a message
// (the loader then binds the 1x1 white dummy, as the reference does for an image tinygltf could not read).
// Pinned by tests/test_host_layer.py against Pillow's decoder on generated files (baseline / progressive, 4:4:4 / 4:2:2 /
// 4:2:0 / 4:4:0, grey, restart intervals, odd sizes): the two decoders differ in rounding only (max 3-4 levels on chroma edges).
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

#include "gltf_scene.h"

namespace vkrt_host {
namespace {

const uint8_t kZigzag[64 + 15] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,
                                   6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31,
                                   39, 46, 53, 60, 61, 54, 47, 55, 62, 63,
                                   // a corrupt run can step past the block: let it land on the last coefficient
                                   63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63, 63};

struct HuffTable
{
  bool present = false;
  uint8_t lookup[512];      // 9-bit prefix -> symbol index (255 = longer code)
  uint8_t lookupLen[512];
  uint16_t code[256];
  uint8_t len[256];
  uint8_t value[256];
  int count = 0;
  int maxcode[18];          // maxcode[l] = (largest code of length l + 1) << (16 - l)
  int delta[17];            // index of the first symbol of length l minus its code

  bool build(const uint8_t* counts16, const uint8_t* values, std::string& why)
  {
    count = 0;
    for(int l = 1; l <= 16; l++)
      for(int k = 0; k < counts16[l - 1]; k++)
      {
        if(count >= 256) { why = "jpeg: bad Huffman table"; return false; }
        len[count++] = (uint8_t)l;
      }
    int c = 0, k = 0;
    for(int l = 1; l <= 16; l++)
    {
      delta[l] = k - c;
      while(k < count && len[k] == l)
        code[k++] = (uint16_t)c++;
      if(c - 1 >= (1 << l)) { why = "jpeg: bad Huffman code lengths"; return false; }
      maxcode[l] = c << (16 - l);
      c <<= 1;
    }
    maxcode[17] = 0x7fffffff;
    memset(lookup, 255, sizeof lookup);
    memset(lookupLen, 0, sizeof lookupLen);
    for(int i = 0; i < count; i++)
    {
      value[i] = values[i];
      if(len[i] <= 9)
      {
        const int first = code[i] << (9 - len[i]), n = 1 << (9 - len[i]);
        for(int j = 0; j < n; j++)
        {
          lookup[first + j] = (uint8_t)i;
          lookupLen[first + j] = len[i];
        }
      }
    }
    present = true;
    return true;
  }
};

struct Component
{
  int id = 0, h = 1, v = 1, tq = 0;
  int hd = 0, ha = 0;             // Huffman tables of the current scan
  int dcPred = 0;
  int x = 0, y = 0;               // size in samples
  int w2 = 0, h2 = 0;             // allocated size (whole MCUs)
  int blocksW = 0, blocksH = 0;   // blocks covering (x, y): the grid of a non-interleaved scan
  std::vector<uint8_t> data;      // w2 x h2 samples
  std::vector<short> coeff;       // progressive: 64 per block of the allocated grid
};

struct Decoder
{
  const uint8_t* p;
  const uint8_t* end;
  std::string& why;
  HuffTable dc[4], ac[4];
  uint16_t dequant[4][64];
  bool haveQ[4] = {false, false, false, false};
  Component comp[4];
  int nComp = 0, width = 0, height = 0;
  int hMax = 1, vMax = 1, mcuW = 0, mcuH = 0, mcusX = 0, mcusY = 0;
  bool progressive = false, sawFrame = false;
  int restartInterval = 0, todo = 0;
  int adobeTransform = -1;
  bool jfif = false;
  // scan state
  int scanN = 0, order[4];
  int specStart = 0, specEnd = 63, succHigh = 0, succLow = 0, eobRun = 0;
  // bit reader
  uint32_t bits = 0;
  int nbits = 0;
  uint8_t marker = 0xff;  // a marker met inside the entropy-coded data (0xff = none)
  bool noMore = false;

  Decoder(const uint8_t* d, size_t n, std::string& w) : p(d), end(d + n), why(w) { memset(dequant, 0, sizeof dequant); }

  bool fail(const char* m) { why = std::string("jpeg: ") + m; return false; }
  int get8() { return p < end ? *p++ : 0; }
  int get16() { const int a = get8(); return (a << 8) | get8(); }

  void grow()
  {
    do
    {
      unsigned b = noMore ? 0u : (unsigned)get8();
      if(b == 0xff)
      {
        int c = get8();
        while(c == 0xff) c = get8();  // fill bytes
        if(c != 0)
        {
          marker = (uint8_t)c;
          noMore = true;
          b = 0;  // (the rest of the segment reads as zeros)
        }
      }
      bits |= b << (24 - nbits);
      nbits += 8;
    } while(nbits <= 24);
  }
  int getBits(int n)
  {
    if(n == 0) return 0;
    if(nbits < n) grow();
    const int v = (int)(bits >> (32 - n));
    bits <<= n;
    nbits -= n;
    return v;
  }
  int getBit() { return getBits(1); }
  // n-bit magnitude category value with JPEG's sign extension (T.81 F.2.2.1 EXTEND)
  int receiveExtend(int n)
  {
    if(n == 0) return 0;
    const int v = getBits(n);
    return v < (1 << (n - 1)) ? v - (1 << n) + 1 : v;
  }
  int decodeSymbol(const HuffTable& h)
  {
    if(nbits < 16) grow();
    const int top9 = (int)(bits >> 23);
    if(h.lookup[top9] != 255)
    {
      const int l = h.lookupLen[top9];
      bits <<= l;
      nbits -= l;
      return h.value[h.lookup[top9]];
    }
    const int top16 = (int)(bits >> 16);
    int l = 10;
    while(l <= 16 && top16 >= h.maxcode[l]) l++;
    if(l > 16) { nbits -= 16; bits <<= 16; return -1; }
    const int idx = (int)(bits >> (32 - l)) + h.delta[l];
    if(idx < 0 || idx >= h.count) return -1;
    bits <<= l;
    nbits -= l;
    return h.value[idx];
  }
  void resetEntropy()
  {
    bits = 0; nbits = 0; marker = 0xff; noMore = false;
    for(int i = 0; i < 4; i++) comp[i].dcPred = 0;
    eobRun = 0;
    todo = restartInterval ? restartInterval : 0x7fffffff;
  }

  // ---- blocks -------------------------------------------------------------------------------------------------------
  bool decodeBlockSequential(short* data, Component& c)
  {
    const HuffTable &hd = dc[c.hd], &ha = ac[c.ha];
    const uint16_t* dq = dequant[c.tq];
    int t = decodeSymbol(hd);
    if(t < 0 || t > 15) return fail("bad DC code");
    memset(data, 0, 64 * sizeof(short));
    const int diff = t ? receiveExtend(t) : 0;
    c.dcPred = (int)((unsigned)c.dcPred + (unsigned)diff);  // (wraps on corrupt data instead of overflowing)
    data[0] = (short)((unsigned)c.dcPred * dq[0]);
    int k = 1;
    do
    {
      const int rs = decodeSymbol(ha);
      if(rs < 0) return fail("bad AC code");
      const int s = rs & 15, r = rs >> 4;
      if(s == 0)
      {
        if(rs != 0xf0) break;  // end of block
        k += 16;
      }
      else
      {
        k += r;
        const unsigned zig = kZigzag[k++];
        data[zig] = (short)(receiveExtend(s) * dq[zig]);
      }
    } while(k < 64);
    return true;
  }
  bool decodeBlockProgDC(short* data, Component& c)
  {
    if(specEnd != 0) return fail("DC scan with spectral end");
    if(succHigh == 0)
    {
      memset(data, 0, 64 * sizeof(short));
      const int t = decodeSymbol(dc[c.hd]);
      if(t < 0 || t > 15) return fail("bad DC code");
      const int diff = t ? receiveExtend(t) : 0;
      c.dcPred = (int)((unsigned)c.dcPred + (unsigned)diff);
      data[0] = (short)((unsigned)c.dcPred << succLow);
    }
    else if(getBit())
      data[0] = (short)(data[0] + (short)(1 << succLow));
    return true;
  }
  bool decodeBlockProgAC(short* data, Component& c)
  {
    if(specStart == 0) return fail("AC scan starting at DC");
    const HuffTable& ha = ac[c.ha];
    if(succHigh == 0)
    {
      const int shift = succLow;
      if(eobRun)
      {
        eobRun--;
        return true;
      }
      int k = specStart;
      do
      {
        const int rs = decodeSymbol(ha);
        if(rs < 0) return fail("bad AC code");
        const int s = rs & 15, r = rs >> 4;
        if(s == 0)
        {
          if(r < 15)
          {
            eobRun = 1 << r;
            if(r) eobRun += getBits(r);
            eobRun--;
            break;
          }
          k += 16;
        }
        else
        {
          k += r;
          const unsigned zig = kZigzag[k++];
          data[zig] = (short)(receiveExtend(s) * (1 << shift));
        }
      } while(k <= specEnd);
    }
    else
    {
      const short bit = (short)(1 << succLow);
      auto refine = [&](short& v) {
        if(getBit() && (v & bit) == 0)
          v = (short)(v > 0 ? v + bit : v - bit);
      };
      if(eobRun)
      {
        eobRun--;
        for(int k = specStart; k <= specEnd; k++)
        {
          short& v = data[kZigzag[k]];
          if(v != 0) refine(v);
        }
      }
      else
      {
        int k = specStart;
        do
        {
          const int rs = decodeSymbol(ha);
          if(rs < 0) return fail("bad AC code");
          int s = rs & 15, r = rs >> 4;
          if(s == 0)
          {
            if(r < 15)
            {
              eobRun = (1 << r) - 1;
              if(r) eobRun += getBits(r);
              r = 64;  // refine to the end of the band
            }
          }
          else
          {
            if(s != 1) return fail("bad refinement code");
            s = getBit() ? bit : -bit;
          }
          while(k <= specEnd)
          {
            short& v = data[kZigzag[k++]];
            if(v != 0)
              refine(v);
            else
            {
              if(r == 0)
              {
                v = (short)s;
                break;
              }
              r--;
            }
          }
        } while(k <= specEnd);
      }
    }
    return true;
  }

  // ---- inverse DCT (see the header) --------------------------------------------------------------------------------------
  static inline int f2f(double x) { return (int)(x * 4096 + 0.5); }
  static inline uint8_t clamp8(long long x) { return (uint8_t)(x < 0 ? 0 : (x > 255 ? 255 : x)); }
  // (64-bit intermediates: the values of a well-formed file fit 32 bits with room to spare; a corrupt one must not run into
  //  signed overflow)
  typedef long long wide;
  struct Idct1D
  {
    wide x0, x1, x2, x3, t0, t1, t2, t3;
    Idct1D(wide s0, wide s1, wide s2, wide s3, wide s4, wide s5, wide s6, wide s7)
    {
      wide p1, p2, p3, p4, p5;
      p2 = s2; p3 = s6;
      p1 = (p2 + p3) * f2f(0.5411961);
      t2 = p1 + p3 * f2f(-1.847759065);
      t3 = p1 + p2 * f2f(0.765366865);
      p2 = s0; p3 = s4;
      t0 = (p2 + p3) * 4096;
      t1 = (p2 - p3) * 4096;
      x0 = t0 + t3; x3 = t0 - t3; x1 = t1 + t2; x2 = t1 - t2;
      t0 = s7; t1 = s5; t2 = s3; t3 = s1;
      p3 = t0 + t2; p4 = t1 + t3; p1 = t0 + t3; p2 = t1 + t2;
      p5 = (p3 + p4) * f2f(1.175875602);
      t0 = t0 * f2f(0.298631336);
      t1 = t1 * f2f(2.053119869);
      t2 = t2 * f2f(3.072711026);
      t3 = t3 * f2f(1.501321110);
      p1 = p5 + p1 * f2f(-0.899976223);
      p2 = p5 + p2 * f2f(-2.562915447);
      p3 = p3 * f2f(-1.961570560);
      p4 = p4 * f2f(-0.390180644);
      t3 += p1 + p4; t2 += p2 + p3; t1 += p2 + p4; t0 += p1 + p3;
    }
  };
  static void idctBlock(uint8_t* out, int stride, const short* d)
  {
    wide val[64];
    for(int i = 0; i < 8; i++)
    {
      const short* c = d + i;
      wide* v = val + i;
      if(c[8] == 0 && c[16] == 0 && c[24] == 0 && c[32] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0)
      {
        const wide dcterm = c[0] * 4;
        v[0] = v[8] = v[16] = v[24] = v[32] = v[40] = v[48] = v[56] = dcterm;
      }
      else
      {
        Idct1D k(c[0], c[8], c[16], c[24], c[32], c[40], c[48], c[56]);
        const wide x0 = k.x0 + 512, x1 = k.x1 + 512, x2 = k.x2 + 512, x3 = k.x3 + 512;
        v[0] = (x0 + k.t3) >> 10; v[56] = (x0 - k.t3) >> 10;
        v[8] = (x1 + k.t2) >> 10; v[48] = (x1 - k.t2) >> 10;
        v[16] = (x2 + k.t1) >> 10; v[40] = (x2 - k.t1) >> 10;
        v[24] = (x3 + k.t0) >> 10; v[32] = (x3 - k.t0) >> 10;
      }
    }
    for(int i = 0; i < 8; i++)
    {
      const wide* v = val + 8 * i;
      uint8_t* o = out + (size_t)i * stride;
      Idct1D k(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]);
      const wide bias = 65536 + (128 << 17);
      const wide x0 = k.x0 + bias, x1 = k.x1 + bias, x2 = k.x2 + bias, x3 = k.x3 + bias;
      o[0] = clamp8((x0 + k.t3) >> 17); o[7] = clamp8((x0 - k.t3) >> 17);
      o[1] = clamp8((x1 + k.t2) >> 17); o[6] = clamp8((x1 - k.t2) >> 17);
      o[2] = clamp8((x2 + k.t1) >> 17); o[5] = clamp8((x2 - k.t1) >> 17);
      o[3] = clamp8((x3 + k.t0) >> 17); o[4] = clamp8((x3 - k.t0) >> 17);
    }
  }

  // ---- segments ---------------------------------------------------------------------------------------------------------
  bool readDQT(int len)
  {
    while(len > 0)
    {
      const int q = get8(), prec = q >> 4, t = q & 15;
      if(prec > 1 || t > 3) return fail("bad DQT");
      for(int i = 0; i < 64; i++)
        dequant[t][kZigzag[i]] = (uint16_t)(prec ? get16() : get8());
      haveQ[t] = true;
      len -= prec ? 129 : 65;
    }
    return len == 0 ? true : fail("bad DQT length");
  }
  bool readDHT(int len)
  {
    while(len > 0)
    {
      const int q = get8(), tc = q >> 4, th = q & 15;
      if(tc > 1 || th > 3) return fail("bad DHT");
      uint8_t counts[16], values[256];
      int n = 0;
      for(int i = 0; i < 16; i++) { counts[i] = (uint8_t)get8(); n += counts[i]; }
      if(n > 256) return fail("bad DHT");
      for(int i = 0; i < n; i++) values[i] = (uint8_t)get8();
      if(!(tc ? ac[th] : dc[th]).build(counts, values, why)) return false;
      len -= 17 + n;
    }
    return len == 0 ? true : fail("bad DHT length");
  }
  bool readSOF(int marker_, int len)
  {
    if(sawFrame) return fail("more than one frame");
    if(marker_ != 0xc0 && marker_ != 0xc1 && marker_ != 0xc2)
      return fail(marker_ == 0xc9 || marker_ == 0xca || marker_ == 0xcb ? "arithmetic coding is not supported" : "lossless / hierarchical JPEG is not supported");
    progressive = marker_ == 0xc2;
    if(get8() != 8) return fail("only 8-bit samples are supported");
    height = get16(); width = get16();
    nComp = get8();
    if(width <= 0 || height <= 0) return fail("empty image");
    if((uint64_t)width * (uint64_t)height > (1ull << 28)) return fail("image too large");
    if(nComp != 1 && nComp != 3) return fail(nComp == 4 ? "CMYK / YCCK JPEG is not supported" : "bad component count");
    if(len != 8 + 3 * nComp) return fail("bad SOF length");
    for(int i = 0; i < nComp; i++)
    {
      Component& c = comp[i];
      c.id = get8();
      const int q = get8();
      c.h = q >> 4; c.v = q & 15;
      c.tq = get8();
      if(c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return fail("bad component");
      hMax = c.h > hMax ? c.h : hMax;
      vMax = c.v > vMax ? c.v : vMax;
    }
    for(int i = 0; i < nComp; i++)
      if(hMax % comp[i].h || vMax % comp[i].v) return fail("fractional sampling ratio");
    mcuW = 8 * hMax; mcuH = 8 * vMax;
    mcusX = (width + mcuW - 1) / mcuW; mcusY = (height + mcuH - 1) / mcuH;
    for(int i = 0; i < nComp; i++)
    {
      Component& c = comp[i];
      c.x = (width * c.h + hMax - 1) / hMax;
      c.y = (height * c.v + vMax - 1) / vMax;
      c.w2 = mcusX * c.h * 8; c.h2 = mcusY * c.v * 8;
      c.blocksW = (c.x + 7) >> 3; c.blocksH = (c.y + 7) >> 3;
      c.data.assign((size_t)c.w2 * c.h2, 0);
      if(progressive)
        c.coeff.assign((size_t)c.w2 * c.h2, 0);
    }
    sawFrame = true;
    return true;
  }
  bool readSOS(int len)
  {
    if(!sawFrame) return fail("scan before frame");
    scanN = get8();
    if(scanN < 1 || scanN > nComp || len != 6 + 2 * scanN) return fail("bad SOS");
    for(int i = 0; i < scanN; i++)
    {
      const int id = get8(), q = get8();
      int which = -1;
      for(int k = 0; k < nComp; k++)
        if(comp[k].id == id) which = k;
      if(which < 0) return fail("scan names an unknown component");
      comp[which].hd = q >> 4; comp[which].ha = q & 15;
      if(comp[which].hd > 3 || comp[which].ha > 3) return fail("bad table selector");
      order[i] = which;
    }
    specStart = get8(); specEnd = get8();
    const int a = get8();
    succHigh = a >> 4; succLow = a & 15;
    if(progressive)
    {
      if(specStart > 63 || specEnd > 63 || specStart > specEnd || succHigh > 13 || succLow > 13) return fail("bad progressive scan");
    }
    else
    {
      if(specStart != 0 || succHigh != 0 || succLow != 0) return fail("bad sequential scan");
      specEnd = 63;
    }
    for(int i = 0; i < scanN; i++)
    {
      const Component& c = comp[order[i]];
      const bool needDC = !progressive || specStart == 0, needAC = !progressive || specStart > 0;
      if((needDC && !(progressive && succHigh) && !dc[c.hd].present) || (needAC && !ac[c.ha].present)) return fail("missing Huffman table");
      if(!progressive && !haveQ[c.tq]) return fail("missing quantisation table");
    }
    return true;
  }

  // after every MCU (or block of a single-component scan): restart interval bookkeeping; false = the scan's data ends here
  // (an interval is over and no restart marker follows)
  bool restartCheck()
  {
    if(--todo > 0)
      return true;
    if(nbits < 24) grow();
    if(marker < 0xd0 || marker > 0xd7)
      return false;
    resetEntropy();
    return true;
  }

  bool decodeScan()
  {
    resetEntropy();
    short block[64];
    if(scanN == 1)
    {
      Component& c = comp[order[0]];
      // non-interleaved: the component's own block grid (T.81 A.2.2), not the MCU-padded one
      for(int by = 0; by < c.blocksH; by++)
        for(int bx = 0; bx < c.blocksW; bx++)
        {
          if(progressive)
          {
            short* d = &c.coeff[64 * ((size_t)bx + (size_t)by * (c.w2 >> 3))];
            if(!(specStart == 0 ? decodeBlockProgDC(d, c) : decodeBlockProgAC(d, c))) return false;
          }
          else
          {
            if(!decodeBlockSequential(block, c)) return false;
            idctBlock(&c.data[(size_t)c.w2 * by * 8 + (size_t)bx * 8], c.w2, block);
          }
          if(!restartCheck())
            return true;
        }
      return true;
    }
    for(int my = 0; my < mcusY; my++)
      for(int mx = 0; mx < mcusX; mx++)
      {
        for(int i = 0; i < scanN; i++)
        {
          Component& c = comp[order[i]];
          for(int y = 0; y < c.v; y++)
            for(int x = 0; x < c.h; x++)
            {
              const int bx = mx * c.h + x, by = my * c.v + y;
              if(progressive)
              {
                if(specStart != 0) return fail("interleaved AC scan");
                if(!decodeBlockProgDC(&c.coeff[64 * ((size_t)bx + (size_t)by * (c.w2 >> 3))], c)) return false;
              }
              else
              {
                if(!decodeBlockSequential(block, c)) return false;
                idctBlock(&c.data[(size_t)c.w2 * by * 8 + (size_t)bx * 8], c.w2, block);
              }
            }
        }
        if(!restartCheck())
          return true;
      }
    return true;
  }

  void finishProgressive()
  {
    short block[64];
    for(int i = 0; i < nComp; i++)
    {
      Component& c = comp[i];
      if(!haveQ[c.tq])
        continue;
      for(int by = 0; by < c.blocksH; by++)
        for(int bx = 0; bx < c.blocksW; bx++)
        {
          const short* d = &c.coeff[64 * ((size_t)bx + (size_t)by * (c.w2 >> 3))];
          for(int k = 0; k < 64; k++) block[k] = (short)(d[k] * dequant[c.tq][k]);
          idctBlock(&c.data[(size_t)c.w2 * by * 8 + (size_t)bx * 8], c.w2, block);
        }
    }
  }

  bool run()
  {
    if(get8() != 0xff || get8() != 0xd8) return fail("not a JPEG");
    bool done = false, sawScan = false;
    while(!done)
    {
      int m;
      if(marker != 0xff)
      {
        m = marker;  // met while decoding the previous scan
        marker = 0xff;
      }
      else
      {
        int b = get8();
        while(b != 0xff && p < end) b = get8();
        if(p >= end) break;
        m = get8();
        while(m == 0xff) m = get8();
      }
      if(m == 0xd9) break;
      if(m == 0 || (m >= 0xd0 && m <= 0xd7) || m == 0x01) continue;  // stuffed zero / stray restart / TEM: no payload
      const int len = get16();
      if(len < 2 || p + (len - 2) > end) return fail("truncated segment");
      const uint8_t* next = p + (len - 2);
      switch(m)
      {
        case 0xdb: if(!readDQT(len - 2)) return false; break;
        case 0xc4: if(!readDHT(len - 2)) return false; break;
        case 0xdd: if(len != 4) return fail("bad DRI"); restartInterval = get16(); break;
        case 0xe0: jfif = len >= 7 && memcmp(p, "JFIF", 5) == 0; break;
        case 0xee: if(len >= 14 && memcmp(p, "Adobe", 5) == 0) adobeTransform = p[11]; break;
        case 0xda:
          if(!readSOS(len)) return false;
          if(!decodeScan()) return false;
          sawScan = true;
          next = p;
          if(marker == 0xff)
          {
            // skip what is left of the entropy-coded segment up to the next marker
            while(next + 1 < end && !(next[0] == 0xff && next[1] != 0 && !(next[1] >= 0xd0 && next[1] <= 0xd7) && next[1] != 0xff)) next++;
          }
          break;
        default:
          if((m >= 0xc0 && m <= 0xcf) && m != 0xc4 && m != 0xc8 && m != 0xcc)
          {
            if(!readSOF(m, len)) return false;
          }
          break;
      }
      p = next;
    }
    if(!sawFrame || !sawScan) return fail("no image data");
    if(progressive) finishProgressive();
    return true;
  }

  // ---- output -----------------------------------------------------------------------------------------------------------
  void upsampleRow(const Component& c, int ratioH, int ratioV, int row, std::vector<uint8_t>& tmp, const uint8_t*& out)
  {
    // sample rows around output row `row` (the state machine of an incremental resampler, written as a function of the row)
    const int wLo = (width + ratioH - 1) / ratioH;
    const uint8_t* base = c.data.data();
    auto line = [&](int y) { return base + (size_t)c.w2 * (y < 0 ? 0 : (y >= c.y ? c.y - 1 : y)); };
    const uint8_t *nearL, *farL;
    if(ratioV == 1)
      nearL = farL = line(row);
    else if(ratioV == 2)
    {
      const int src = row >> 1;
      nearL = line(src);
      farL = (row & 1) ? line(src + 1) : line(src - 1);
    }
    else
      nearL = farL = line(row / ratioV);
    if(ratioH == 1 && ratioV == 1)
    {
      out = nearL;
      return;
    }
    tmp.resize((size_t)wLo * ratioH + 8);
    uint8_t* o = tmp.data();
    if(ratioH == 1 && ratioV == 2)
    {
      for(int i = 0; i < wLo; i++) o[i] = (uint8_t)((3 * nearL[i] + farL[i] + 2) >> 2);
    }
    else if(ratioH == 2 && ratioV == 1)
    {
      const uint8_t* in = nearL;
      if(wLo == 1)
        o[0] = o[1] = in[0];
      else
      {
        o[0] = in[0];
        o[1] = (uint8_t)((in[0] * 3 + in[1] + 2) >> 2);
        int i;
        for(i = 1; i < wLo - 1; i++)
        {
          const int n = 3 * in[i] + 2;
          o[2 * i] = (uint8_t)((n + in[i - 1]) >> 2);
          o[2 * i + 1] = (uint8_t)((n + in[i + 1]) >> 2);
        }
        o[2 * i] = (uint8_t)((in[wLo - 2] * 3 + in[wLo - 1] + 2) >> 2);
        o[2 * i + 1] = in[wLo - 1];
      }
    }
    else if(ratioH == 2 && ratioV == 2)
    {
      if(wLo == 1)
        o[0] = o[1] = (uint8_t)((3 * nearL[0] + farL[0] + 2) >> 2);
      else
      {
        int t0, t1 = 3 * nearL[0] + farL[0];
        o[0] = (uint8_t)((t1 + 2) >> 2);
        for(int i = 1; i < wLo; i++)
        {
          t0 = t1;
          t1 = 3 * nearL[i] + farL[i];
          o[2 * i - 1] = (uint8_t)((3 * t0 + t1 + 8) >> 4);
          o[2 * i] = (uint8_t)((3 * t1 + t0 + 8) >> 4);
        }
        o[2 * wLo - 1] = (uint8_t)((t1 + 2) >> 2);
      }
    }
    else
    {
      for(int i = 0; i < wLo; i++)
        for(int j = 0; j < ratioH; j++) o[i * ratioH + j] = nearL[i];
    }
    out = o;
  }

  void output(TextureImage& img)
  {
    img.width = (uint32_t)width; img.height = (uint32_t)height;
    img.rgba.resize((size_t)width * height * 4);
    // three components are YCbCr unless the file says otherwise: Adobe transform 0, or component ids 'R','G','B' without JFIF
    bool rgb = false;
    if(nComp == 3)
    {
      if(adobeTransform == 0) rgb = true;
      else if(adobeTransform < 0 && !jfif && comp[0].id == 'R' && comp[1].id == 'G' && comp[2].id == 'B') rgb = true;
    }
    std::vector<uint8_t> tmp[3];
    auto fixed = [](double x) { return ((int)(x * 4096.0 + 0.5)) << 8; };
    const int crR = fixed(1.40200), crG = fixed(0.71414), cbG = fixed(0.34414), cbB = fixed(1.77200);
    for(int y = 0; y < height; y++)
    {
      const uint8_t* row[3] = {nullptr, nullptr, nullptr};
      for(int k = 0; k < nComp; k++)
        upsampleRow(comp[k], hMax / comp[k].h, vMax / comp[k].v, y, tmp[k], row[k]);
      uint8_t* o = &img.rgba[(size_t)y * width * 4];
      if(nComp == 1)
        for(int x = 0; x < width; x++)
        {
          o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = row[0][x];
          o[4 * x + 3] = 255;
        }
      else if(rgb)
        for(int x = 0; x < width; x++)
        {
          o[4 * x] = row[0][x]; o[4 * x + 1] = row[1][x]; o[4 * x + 2] = row[2][x];
          o[4 * x + 3] = 255;
        }
      else
        for(int x = 0; x < width; x++)
        {
          const int yf = (row[0][x] << 20) + (1 << 19);
          const int cb = row[1][x] - 128, cr = row[2][x] - 128;
          const int r = yf + cr * crR;
          const int g = yf + cr * -crG + (int)(((unsigned)(cb * -cbG)) & 0xffff0000u);
          const int b = yf + cb * cbB;
          o[4 * x] = clamp8(r >> 20); o[4 * x + 1] = clamp8(g >> 20); o[4 * x + 2] = clamp8(b >> 20);
          o[4 * x + 3] = 255;
        }
    }
  }
};

}  // namespace

bool decodeJpegMemory(const uint8_t* data, size_t size, TextureImage& out, std::string& why)
{
  if(size < 4 || data[0] != 0xff || data[1] != 0xd8) { why = "not a JPEG"; return false; }
  Decoder d(data, size, why);
  if(!d.run())
    return false;
  d.output(out);
  return true;
}

}  // namespace vkrt_host
