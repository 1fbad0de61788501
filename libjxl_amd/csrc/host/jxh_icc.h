// libjxl_amd — host front-end: the embedded ICC profile of an image (ImageMetadata.color_encoding.want_icc).
//
// Replaces lib/jxl/icc_codec.cc:306-428 (ICCReader: the profile is coded as a byte stream in a predicted form, entropy
// coded with 41 contexts chosen from the two previous bytes, icc_codec_common.cc:20-48,172-176) and :128-304
// (UnpredictICC: header bytes as differences to a predicted header, the tag table as commands with implied offsets and
// sizes, the tag data as inserts / shuffles / Nth-order predictions). The profile has to be decoded even by a decoder
// that only wants pixels: it is entropy coded, so its length is known only afterwards. Pinned by the reference's own
// vector (lib/jxl/icc_codec_test.cc:52-211: kEncodedTestProfile -> kTestProfile) in tests/test_kats.py.
#ifndef JXH_ICC_H_
#define JXH_ICC_H_

#include <cstdint>
#include <cstring>
#include <vector>

#include "jxh_entropy.h"

namespace jxh {

namespace icc {

constexpr size_t kHeaderSize = 128;
constexpr size_t kNumContexts = 41;

static inline int ByteKind1(uint8_t b) {
  if (('a' <= b && b <= 'z') || ('A' <= b && b <= 'Z')) return 0;
  if (('0' <= b && b <= '9') || b == '.' || b == ',') return 1;
  if (b == 0) return 2;
  if (b == 1) return 3;
  if (b < 16) return 4;
  if (b == 255) return 6;
  if (b > 240) return 5;
  return 7;
}
static inline int ByteKind2(uint8_t b) {
  if (('a' <= b && b <= 'z') || ('A' <= b && b <= 'Z')) return 0;
  if (('0' <= b && b <= '9') || b == '.' || b == ',') return 1;
  if (b < 16) return 2;
  if (b > 240) return 3;
  return 4;
}
static inline size_t Context(size_t i, uint8_t b1, uint8_t b2) { return i <= 128 ? 0 : size_t(1 + ByteKind1(b1) + ByteKind2(b2) * 8); }

static inline uint64_t VarInt(const uint8_t* in, size_t size, size_t* pos) {
  uint64_t v = 0;
  for (int i = 0; i < 10; i++) {
    JXH_CHECK(*pos < size, "ICC: truncated varint");
    const uint8_t b = in[(*pos)++];
    if (i == 9) JXH_CHECK((b & 0xFE) == 0, "ICC: varint too long");
    v |= uint64_t(b & 0x7F) << (7 * i);
    if (!(b & 0x80)) return v;
  }
  throw Error("ICC: varint too long");
}
static inline uint32_t Check32(uint64_t v) {
  JXH_CHECK(v <= 0xFFFFFFFFull, "ICC: 32-bit value expected");
  return uint32_t(v);
}
static inline void PutU32(std::vector<uint8_t>* out, uint32_t v) {
  for (int s = 24; s >= 0; s -= 8) out->push_back(uint8_t(v >> s));
}
static inline void PutTag(std::vector<uint8_t>* out, const char* t) { out->insert(out->end(), t, t + 4); }
static inline uint32_t GetU32(const uint8_t* d, size_t size, size_t pos) {
  return pos + 4 > size ? 0u : (uint32_t(d[pos]) << 24) | (uint32_t(d[pos + 1]) << 16) | (uint32_t(d[pos + 2]) << 8) | d[pos + 3];
}
// Interleaves `width` runs back into scanline order (icc_codec.cc:34-54).
static inline void Unshuffle(uint8_t* data, size_t size, size_t width) {
  const size_t height = (size + width - 1) / width;
  std::vector<uint8_t> r(size);
  size_t s = 0, j = 0;
  for (size_t i = 0; i < size; i++) {
    r[i] = data[j];
    j += height;
    if (j >= size) j = ++s;
  }
  if (size) memcpy(data, r.data(), size);
}
template <typename T>
static inline T Extrapolate(T p1, T p2, T p3, int order) {
  return order == 0 ? p1 : (order == 1 ? T(2 * p1 - p2) : (order == 2 ? T(3 * p1 - 3 * p2 + p3) : T(0)));
}
static inline uint8_t LinearPredict(const uint8_t* d, size_t start, size_t i, size_t stride, size_t width, int order) {
  const size_t pos = start + i;
  if (width == 1) return Extrapolate<uint8_t>(d[pos - stride], d[pos - 2 * stride], d[pos - 3 * stride], order);
  if (width == 2) {
    const size_t p = start + (i & ~size_t(1));
    auto at = [&](size_t q) { return uint16_t((d[q] << 8) + d[q + 1]); };
    const uint16_t v = Extrapolate<uint16_t>(at(p - stride), at(p - 2 * stride), at(p - 3 * stride), order);
    return (i & 1) ? uint8_t(v & 255) : uint8_t(v >> 8);
  }
  const size_t p = start + (i & ~size_t(3));
  const uint32_t v = Extrapolate<uint32_t>(GetU32(d, pos, p - stride), GetU32(d, pos, p - 2 * stride), GetU32(d, pos, p - 3 * stride), order);
  return uint8_t(v >> ((3 - (i & 3)) * 8));
}

// The predicted form back to the profile.
static inline void Unpredict(const uint8_t* enc, size_t size, std::vector<uint8_t>* out) {
  static const char* const kTagStrings[17] = {"cprt", "wtpt", "bkpt", "rXYZ", "gXYZ", "bXYZ", "kXYZ", "rTRC", "gTRC",
                                              "bTRC", "kTRC", "chad", "desc", "chrm", "dmnd", "dmdd", "lumi"};
  static const char* const kTypeStrings[8] = {"XYZ ", "desc", "text", "mluc", "para", "curv", "sf32", "gbd "};
  out->clear();
  size_t pos = 0;
  const uint32_t osize = Check32(VarInt(enc, size, &pos));
  const uint32_t csize = Check32(VarInt(enc, size, &pos));
  JXH_CHECK(osize <= (1u << 28), "ICC: decoded profile too large");
  size_t cpos = pos;
  JXH_CHECK(uint64_t(pos) + csize <= size, "ICC: commands out of bounds");
  const size_t cend = cpos + csize;
  pos = cend;
  auto done = [&]() {
    JXH_CHECK(cpos == cend && pos == size, "ICC: data left over");
  };
  // header: differences to a prediction that follows the bytes already known (icc_codec_common.cc:96-140)
  uint8_t header[kHeaderSize] = {0};
  {
    static const uint8_t kInit[kHeaderSize] = {
        0, 0, 0, 0, 0, 0, 0, 0, 4, 0, 0, 0, 'm', 'n', 't', 'r', 'R', 'G', 'B', ' ', 'X', 'Y', 'Z', ' ', 0, 0, 0, 0, 0, 0, 0, 0,
        0, 0, 0, 0, 'a', 'c', 's', 'p', 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
        0, 0, 0, 0, 0, 0, 246, 214, 0, 1, 0, 0, 0, 0, 211, 45};
    memcpy(header, kInit, sizeof(kInit));
    for (int s = 0; s < 4; s++) header[s] = uint8_t(osize >> (24 - 8 * s));
  }
  for (size_t i = 0; i <= kHeaderSize; i++) {
    if (out->size() == osize) return done();
    if (i == kHeaderSize) break;
    const std::vector<uint8_t>& o = *out;
    if (i == 8 && o.size() >= 8) memcpy(header + 80, o.data() + 4, 4);
    if (i == 41 && o.size() >= 41) {
      if (o[40] == 'A') memcpy(header + 41, "PPL", 3);
      if (o[40] == 'M') memcpy(header + 41, "SFT", 3);
    }
    if (i == 42 && o.size() >= 42) {
      if (o[40] == 'S' && o[41] == 'G') memcpy(header + 42, "I ", 2);
      if (o[40] == 'S' && o[41] == 'U') memcpy(header + 42, "NW", 2);
    }
    JXH_CHECK(pos < size, "ICC: header out of bounds");
    out->push_back(uint8_t(enc[pos++] + header[i]));
  }
  JXH_CHECK(cpos < cend, "ICC: commands out of bounds");
  // tag table
  uint64_t numtags = VarInt(enc, cend, &cpos);
  if (numtags != 0) {
    numtags--;
    PutU32(out, Check32(numtags));
    uint64_t prev_start = kHeaderSize + numtags * 12, prev_size = 0;
    for (;;) {
      JXH_CHECK(out->size() <= osize, "ICC: result too large");
      if (cpos == cend) break;
      const uint8_t command = enc[cpos++];
      const uint8_t code = command & 63;
      if (code == 0) break;
      char tag[4];
      if (code == 1) {
        JXH_CHECK(uint64_t(pos) + 4 <= size, "ICC: tag out of bounds");
        memcpy(tag, enc + pos, 4);
        pos += 4;
      } else if (code == 2) {
        memcpy(tag, "rTRC", 4);
      } else if (code == 3) {
        memcpy(tag, "rXYZ", 4);
      } else {
        JXH_CHECK(size_t(code - 4) < 17, "ICC: unknown tag code");
        memcpy(tag, kTagStrings[code - 4], 4);
      }
      out->insert(out->end(), tag, tag + 4);
      uint64_t start, tsize = prev_size;
      for (const char* t : {"rXYZ", "gXYZ", "bXYZ", "kXYZ", "wtpt", "bkpt", "lumi"})
        if (!memcmp(tag, t, 4)) tsize = 20;
      if (command & 64) start = VarInt(enc, cend, &cpos);
      else start = Check32(prev_start) + prev_size;
      PutU32(out, Check32(start));
      if (command & 128) tsize = VarInt(enc, cend, &cpos);
      PutU32(out, Check32(tsize));
      prev_start = start;
      prev_size = tsize;
      if (code == 2) {
        PutTag(out, "gTRC"); PutU32(out, uint32_t(start)); PutU32(out, uint32_t(tsize));
        PutTag(out, "bTRC"); PutU32(out, uint32_t(start)); PutU32(out, uint32_t(tsize));
      }
      if (code == 3) {
        (void)Check32(start + tsize * 2);
        PutTag(out, "gXYZ"); PutU32(out, uint32_t(start + tsize)); PutU32(out, uint32_t(tsize));
        PutTag(out, "bXYZ"); PutU32(out, uint32_t(start + tsize * 2)); PutU32(out, uint32_t(tsize));
      }
    }
  }
  // tag data
  for (;;) {
    JXH_CHECK(out->size() <= osize, "ICC: result too large");
    if (cpos == cend) break;
    const uint8_t command = enc[cpos++];
    if (command == 1) {  // insert
      const uint64_t num = VarInt(enc, cend, &cpos);
      JXH_CHECK(num <= size - pos, "ICC: insert out of bounds");
      out->insert(out->end(), enc + pos, enc + pos + num);
      pos += num;
    } else if (command == 2 || command == 3) {  // 2- / 4-byte values stored as byte planes
      const uint64_t num = VarInt(enc, cend, &cpos);
      JXH_CHECK(num <= size - pos, "ICC: shuffle out of bounds");
      std::vector<uint8_t> t(enc + pos, enc + pos + num);
      Unshuffle(t.data(), t.size(), command == 2 ? 2 : 4);
      out->insert(out->end(), t.begin(), t.end());
      pos += num;
    } else if (command == 4) {  // Nth-order prediction of 1-, 2- or 4-byte values at a stride
      JXH_CHECK(cpos + 2 <= cend, "ICC: predict out of bounds");
      const uint8_t flags = enc[cpos++];
      const size_t width = (flags & 3) + 1;
      const int order = (flags & 12) >> 2;
      JXH_CHECK(width != 3 && order != 3, "ICC: invalid prediction");
      uint64_t stride = width;
      if (flags & 16) {
        stride = VarInt(enc, cend, &cpos);
        JXH_CHECK(stride >= width, "ICC: invalid stride");
      }
      JXH_CHECK(!out->empty() && ((out->size() - 1) >> 2) >= stride, "ICC: invalid stride");
      const uint64_t num = VarInt(enc, cend, &cpos);
      JXH_CHECK(num <= size - pos, "ICC: predict out of bounds");
      std::vector<uint8_t> t(enc + pos, enc + pos + num);
      if (width > 1) Unshuffle(t.data(), t.size(), width);
      const size_t start = out->size();
      out->reserve(start + num);
      for (size_t i = 0; i < num; i++) {
        out->push_back(0);
        (*out)[start + i] = uint8_t(LinearPredict(out->data(), start, i, size_t(stride), width, order) + t[i]);
      }
      pos += num;
    } else if (command == 10) {  // an XYZ triple
      PutTag(out, "XYZ ");
      out->insert(out->end(), 4, 0);
      JXH_CHECK(pos + 12 <= size, "ICC: XYZ out of bounds");
      out->insert(out->end(), enc + pos, enc + pos + 12);
      pos += 12;
    } else if (command >= 16 && command < 24) {  // a type signature + 4 reserved bytes
      PutTag(out, kTypeStrings[command - 16]);
      out->insert(out->end(), 4, 0);
    } else {
      throw Error("ICC: unknown command");
    }
  }
  JXH_CHECK(pos == size && out->size() == osize, "ICC: size mismatch");
}

}  // namespace icc

// Reads the coded profile at the reader's position (icc_codec.cc:306-428).
static inline void ReadIcc(BitReader& br, std::vector<uint8_t>* profile) {
  const uint64_t enc_size = ReadU64(br);
  JXH_CHECK(enc_size <= (uint64_t(1) << 28), "ICC: encoded profile too large");
  EntropyCode code;
  DecodeHistograms(br, icc::kNumContexts, &code);
  SymbolReader rd(&code, &br);
  std::vector<uint8_t> enc(enc_size);
  for (size_t i = 0; i < enc_size; i++) {
    const uint32_t v = rd.Read(icc::Context(i, i > 0 ? enc[i - 1] : 0, i > 1 ? enc[i - 2] : 0));
    JXH_CHECK(v < 256, "ICC: invalid byte");
    enc[i] = uint8_t(v);
    JXH_CHECK(!br.Overread(), "ICC: truncated");
  }
  JXH_CHECK(rd.FinalStateOk(), "ICC: bad ANS final state");
  icc::Unpredict(enc.data(), enc.size(), profile);
}

}  // namespace jxh
#endif  // JXH_ICC_H_
