// Known-answer tests of the host bitstream layers, compiled by the test suite once against the PRODUCT's host front-end
// (libjxl_amd/csrc/host/jxh_*.h) and once, with -DKAT_ORACLE, against the oracle's headers. The expectations come from
// the reference's own tests and from reference-ENCODER output, never from the other decoder:
//   alias    every symbol's offsets enumerate 0 .. freq-1 exactly once over the 4096 slots (ans_common_test.cc:26-44)
//   hybrid   the worked examples of lib/jxl/dec_ans.h:47-67 and the round trip of entropy_coder_test.cc:38-59
//   lehmer   random permutations -> Lehmer code -> DecodeLehmer (lehmer_code_test.cc)
//   fjxl F.jxl F.raw W H C   a Modular (lossless) file written by the reference's enc_fast_lossless.cc decodes, through
//            this front-end's bit reader / headers / TOC / prefix + LZ77 entropy coder / MA-tree channel decode / inverse
//            transforms, to exactly the raw interleaved 8-bit samples it was made from.
//   icc ENC ICC   (product only) the reference's own ICC codec vector (icc_codec_test.cc:52-211): the coded stream ENC must
//            decode to the profile ICC byte for byte, and damaged copies of it must be rejected, not crash.
//   fastmath the error bars of the reference's own fast_math_test.cc:46-111 on this front-end's FastLog2f / FastPow2f /
//            FastPowf (they shape the dequantisation tables, quant_weights.cc) and the splines' FastCosf, over the ranges
//            the reference samples (2^20 draws each).
//   bits     bit_reader_test.cc:28-50,152-255: the byte-order known answers of TestOrder (bytes 1F FC, F8 3F, 3F F8, BD 8D
//            and the fields they were written from), zero extension past the end, the consumed-bit counter.
//   fields   fields_test.cc:57-209: every U32 / U64 value the reference tests, written here by an independent writer of
//            the wire format (fields.cc:444-452, 494-521), must read back AND consume exactly the number of bits the
//            reference's test states; the F16 values it lists decode exactly (fields.cc:550-575).
//   quant    quant_weights_test.cc:185-271 (DCTUniform): with every table coded as DCT bands {1/4, 0} the table builder
//            must produce 4.0 in every entry of every one of the 17 tables (the reference asserts the transforms agree
//            with a uniformly quantised slow DCT to 1e-4 and that every matrix starts with 4: 1e-6).
// usage: host_kats {alias|hybrid|lehmer|fastmath|bits|fields|quant|fjxl ...|icc ...};  exit code 0 = pass, message on stderr otherwise.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifdef KAT_ORACLE
#include "../../oracle/jxlo_bits.h"
#include "../../oracle/jxlo_entropy.h"
#include "../../oracle/jxlo_headers.h"
#include "../../oracle/jxlo_modular.h"
#include "../../oracle/jxlo_splines.h"
#include "../../oracle/jxlo_vardct.h"
namespace H = jxlo;
#else
#include "../../libjxl_amd/csrc/host/jxh_bits.h"
#include "../../libjxl_amd/csrc/host/jxh_entropy.h"
#include "../../libjxl_amd/csrc/host/jxh_headers.h"
#include "../../libjxl_amd/csrc/host/jxh_modular.h"
#include "../../libjxl_amd/csrc/host/jxh_splines.h"
#include "../../libjxl_amd/csrc/host/jxh_vardct.h"
namespace H = jxh;
#endif

#define REQUIRE(cond, ...)                \
  do {                                    \
    if (!(cond)) {                        \
      fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
      fprintf(stderr, __VA_ARGS__);       \
      fprintf(stderr, "\n");              \
      return 1;                           \
    }                                     \
  } while (0)

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint32_t Rnd(uint32_t n) {
  g_rng ^= g_rng << 13;
  g_rng ^= g_rng >> 7;
  g_rng ^= g_rng << 17;
  return uint32_t((g_rng >> 11) % n);
}

static int VerifyAlias(const std::vector<int32_t>& dist, int log_alpha) {
  std::vector<H::AliasEntry> table(size_t(1) << log_alpha);
  H::InitAliasTable(dist, log_alpha, table.data());
  const int log_entry = 12 - log_alpha;
  std::vector<std::vector<uint32_t>> offsets(dist.size());
  for (uint32_t slot = 0; slot < 4096; slot++) {
    const H::AliasEntry& e = table[slot >> log_entry];
    const uint32_t pos = slot & ((1u << log_entry) - 1);
    uint32_t sym, off, freq;
    if (pos >= e.cutoff) {
      sym = e.right_value;
      off = e.offsets1 + pos;
      freq = e.freq1;
    } else {
      sym = slot >> log_entry;
      off = pos;
      freq = e.freq0;
    }
    REQUIRE(sym < dist.size(), "slot %u maps to symbol %u outside the alphabet", slot, sym);
    REQUIRE(freq == uint32_t(dist[sym]) || (dist[sym] == 4096 && (freq & 0xFFF) == 0), "slot %u: frequency %u of symbol %u, expected %d", slot,
            freq, sym, dist[sym]);
    offsets[sym].push_back(off);
  }
  for (size_t s = 0; s < dist.size(); s++) {
    REQUIRE(offsets[s].size() == size_t(dist[s]), "symbol %zu owns %zu slots, expected %d", s, offsets[s].size(), dist[s]);
    std::vector<bool> seen(offsets[s].size(), false);
    for (uint32_t o : offsets[s]) {
      REQUIRE(o < seen.size() && !seen[o], "symbol %zu: offset %u repeated or out of range", s, o);
      seen[o] = true;
    }
  }
  return 0;
}
static int TestAlias() {
  if (VerifyAlias({2048, 2048}, 8)) return 1;  // AliasDistributionSmoke
  if (VerifyAlias({4096}, 8)) return 1;
  if (VerifyAlias({0, 0, 0, 4096, 0}, 8)) return 1;
  for (int log_alpha = 5; log_alpha <= 8; log_alpha++)
    for (int rep = 0; rep < 200; rep++) {
      const size_t n = 1 + Rnd(1u << log_alpha);
      std::vector<int32_t> d(n, 0);
      int32_t left = 4096;
      // random composition of 4096, some symbols empty
      for (size_t i = 0; i + 1 < n && left > 0; i++) {
        if (Rnd(4) == 0) continue;
        const int32_t v = int32_t(Rnd(uint32_t(std::min<int32_t>(left, 1 + 8192 / int32_t(n)))));
        d[i] = v;
        left -= v;
      }
      d[n - 1] += left;
      if (VerifyAlias(d, log_alpha)) {
        fprintf(stderr, "  (log_alpha %d, alphabet %zu, repetition %d)\n", log_alpha, n, rep);
        return 1;
      }
    }
  return 0;
}

// value -> (token, nbits, bits) as the reference's encoder side defines it (dec_ans.h:47-67, HybridUintConfig::Encode)
static void HybridEncode(uint32_t se, uint32_t msb, uint32_t lsb, uint32_t value, uint32_t* token, uint32_t* nbits, uint32_t* bits) {
  if (value < (1u << se)) {
    *token = value;
    *nbits = *bits = 0;
    return;
  }
  uint32_t n = 31;
  while (!(value >> n)) n--;
  const uint32_t m = value - (1u << n);
  *token = (1u << se) + ((n - se) << (msb + lsb)) + ((m >> (n - msb)) << lsb) + (m & ((1u << lsb) - 1));
  *nbits = n - msb - lsb;
  *bits = (value >> lsb) & ((1u << *nbits) - 1);
}
static int TestHybrid() {
  struct Ex { uint32_t n, token, nbits, bits; };
  static const Ex kExamples[] = {{0, 0, 0, 0},   {15, 15, 0, 0}, {16, 16, 2, 0}, {17, 16, 2, 1}, {20, 17, 2, 0},
                                 {24, 18, 2, 0}, {28, 19, 2, 0}, {32, 20, 3, 0}, {65535, 63, 13, 0x1FFF}};  // dec_ans.h:58-67
  H::HybridCfg c;
  c.split_exp = 4;
  c.split_token = 16;
  c.msb = 2;
  c.lsb = 0;
  for (const Ex& e : kExamples) {
    uint8_t buf[8] = {uint8_t(e.bits), uint8_t(e.bits >> 8), 0, 0, 0, 0, 0, 0};
    H::BitReader br(buf, sizeof(buf));
    const uint32_t v = H::SymbolReader::ReadHybrid(c, e.token, &br);
    REQUIRE(v == e.n, "token %u + bits %u decodes to %u, the reference's example says %u", e.token, e.bits, v, e.n);
    REQUIRE(br.BitPos() == e.nbits, "token %u consumed %zu bits, expected %u", e.token, size_t(br.BitPos()), e.nbits);
  }
  static const uint32_t kConfigs[][3] = {{0, 0, 0}, {4, 1, 1}, {4, 2, 0}, {4, 1, 0}, {5, 2, 1}, {8, 0, 0}, {15, 0, 0}, {3, 3, 0}};
  for (const auto& k : kConfigs) {
    c.split_exp = k[0];
    c.split_token = 1u << k[0];
    c.msb = k[1];
    c.lsb = k[2];
    for (int i = 0; i < 200000; i++) {
      const uint32_t value = i < 70000 ? uint32_t(i) : Rnd(1u << 24);
      uint32_t token, nbits, bits;
      HybridEncode(k[0], k[1], k[2], value, &token, &nbits, &bits);
      uint8_t buf[8] = {uint8_t(bits), uint8_t(bits >> 8), uint8_t(bits >> 16), uint8_t(bits >> 24), 0, 0, 0, 0};
      H::BitReader br(buf, sizeof(buf));
      const uint32_t v = H::SymbolReader::ReadHybrid(c, token, &br);
      REQUIRE(v == value && br.BitPos() == nbits, "config %u/%u/%u: %u -> token %u -> %u", k[0], k[1], k[2], value, token, v);
    }
  }
  for (int32_t i = -1000; i < 1000; i++) {  // PackUnpack (entropy_coder_test.cc:20-27)
    const uint32_t packed = i >= 0 ? uint32_t(i) * 2 : uint32_t(-(i + 1)) * 2 + 1;
    REQUIRE(H::UnpackSigned(packed) == i, "UnpackSigned(%u) != %d", packed, i);
  }
  return 0;
}

static int TestLehmer() {
  for (int rep = 0; rep < 300; rep++) {
    const size_t n = 1 + Rnd(rep < 200 ? 64 : 5000);
    std::vector<uint32_t> perm(n);
    for (size_t i = 0; i < n; i++) perm[i] = uint32_t(i);
    for (size_t i = n; i > 1; i--) std::swap(perm[i - 1], perm[Rnd(uint32_t(i))]);
    // Lehmer code by definition: code[i] = number of later elements smaller than perm[i]
    std::vector<uint32_t> code(n);
    for (size_t i = 0; i < n; i++) {
      uint32_t c = 0;
      for (size_t j = i + 1; j < n; j++) c += perm[j] < perm[i];
      code[i] = c;
    }
    std::vector<uint32_t> back;
    H::DecodeLehmer(code, &back);
    REQUIRE(back == perm, "permutation of %zu entries does not round-trip", n);
  }
  return 0;
}

// A whole Modular frame with this front-end's pieces (frame walk: dec_frame.cc:135-434 for frame_header.encoding ==
// kModular; dec_modular.cc:209-425).
static int TestFjxl(const char* jxl_path, const char* raw_path, size_t W, size_t Hh, size_t C) {
  auto slurp = [](const char* p, std::vector<uint8_t>* v) {
    FILE* f = fopen(p, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    v->resize(size_t(ftell(f)));
    fseek(f, 0, SEEK_SET);
    const bool ok = fread(v->data(), 1, v->size(), f) == v->size();
    fclose(f);
    return ok;
  };
  std::vector<uint8_t> data, raw;
  REQUIRE(slurp(jxl_path, &data) && slurp(raw_path, &raw), "cannot read the fixture");
  REQUIRE(raw.size() == W * Hh * C, "raw size");
  try {
    REQUIRE(data.size() > 2 && data[0] == 0xFF && data[1] == 0x0A, "not a bare codestream");
    H::BitReader br(data.data(), data.size());
    br.Skip(16);
    H::ImageHeader ih;
    H::ReadImageHeader(br, &ih);
    REQUIRE(ih.xsize == W && ih.ysize == Hh, "image size %ux%u", ih.xsize, ih.ysize);
    const size_t ncolor = ih.gray ? 1 : 3;
    REQUIRE(ncolor + ih.extra.size() == C, "channel count %zu + %zu", ncolor, ih.extra.size());
    REQUIRE(!ih.xyb_encoded && ih.bits == 8, "fjxl writes non-XYB 8-bit images");
    H::FrameHeader fh;
    H::ReadFrameHeader(br, ih, &fh);
    REQUIRE(fh.modular && fh.is_last && fh.num_passes == 1 && fh.upsampling == 1, "frame header");
    const H::FrameDim d = H::MakeFrameDim(fh);
    const size_t entries = d.num_groups == 1 ? 1 : 2 + d.num_dc_groups + d.num_groups;
    H::Toc toc;
    H::ReadToc(br, entries, &toc);
    const size_t base = br.BitPos() / 8;
    REQUIRE(base + toc.total == data.size(), "TOC: sections end at %zu, file has %zu bytes", size_t(base + toc.total), data.size());
    H::MGlobal mg;
    H::MImage full;
    full.bitdepth = 8;
    for (size_t c = 0; c < C; c++) full.ch.emplace_back(d.xsize, d.ysize);
    auto dc_global = [&](H::BitReader& r) {
      REQUIRE(r.ReadBool(), "DC dequant must be default in a Modular frame written by fjxl");
      if (r.ReadBool()) {
        H::DecodeTree(r, &mg.tree, 1 << 20);
        H::DecodeHistograms(r, (mg.tree.size() + 1) / 2, &mg.code);
        mg.have = true;
      }
      H::ModularDecode(r, &full, 0, &mg, d.group_dim, /*undo_transforms=*/false);
      return 0;
    };
    auto group = [&](H::BitReader& r, size_t x0, size_t y0, size_t xs, size_t ys, int min_shift, int max_shift, int stream) {
      H::MImage part;
      part.bitdepth = 8;
      std::vector<size_t> which, px, py;
      size_t c = full.nb_meta;
      while (c < full.ch.size() && full.ch[c].w <= d.group_dim && full.ch[c].h <= d.group_dim) c++;
      for (; c < full.ch.size(); c++) {
        H::MChannel& fc = full.ch[c];
        const int shift = std::min(fc.hshift, fc.vshift);
        if (shift < min_shift || shift > max_shift) continue;
        const size_t rx = x0 >> fc.hshift, ry = y0 >> fc.vshift;
        if (rx >= fc.w || ry >= fc.h) continue;
        const size_t rw = std::min(xs >> fc.hshift, fc.w - rx), rh = std::min(ys >> fc.vshift, fc.h - ry);
        if (!rw || !rh) continue;
        part.ch.emplace_back(rw, rh, fc.hshift, fc.vshift);
        which.push_back(c);
        px.push_back(rx);
        py.push_back(ry);
      }
      if (part.ch.empty()) return;
      H::ModularDecode(r, &part, stream, &mg);
      for (size_t i = 0; i < which.size(); i++)
        for (size_t y = 0; y < part.ch[i].h; y++)
          memcpy(full.ch[which[i]].Row(py[i] + y) + px[i], part.ch[i].Row(y), part.ch[i].w * sizeof(int32_t));
    };
    const uint8_t* sec = data.data() + base;
    if (entries == 1) {
      H::BitReader r(sec + toc.offset[0], toc.size[0]);
      if (dc_global(r)) return 1;
      group(r, 0, 0, d.dc_group_dim, d.dc_group_dim, 3, 1000, int(1 + d.num_dc_groups));
      group(r, 0, 0, d.group_dim, d.group_dim, 0, 2, int(1 + 3 * d.num_dc_groups + 17));
      REQUIRE(!r.Overread(), "section over-read");
    } else {
      {
        H::BitReader r(sec + toc.offset[0], toc.size[0]);
        if (dc_global(r)) return 1;
        REQUIRE(!r.Overread(), "DC global over-read");
      }
      for (size_t g = 0; g < d.num_dc_groups; g++) {
        H::BitReader r(sec + toc.offset[1 + g], toc.size[1 + g]);
        const size_t gx = g % d.xsize_dc_groups, gy = g / d.xsize_dc_groups;
        group(r, gx * d.dc_group_dim, gy * d.dc_group_dim, d.dc_group_dim, d.dc_group_dim, 3, 1000, int(1 + d.num_dc_groups + g));
        REQUIRE(!r.Overread(), "DC group over-read");
      }
      for (size_t g = 0; g < d.num_groups; g++) {
        const size_t i = 2 + d.num_dc_groups + g;
        H::BitReader r(sec + toc.offset[i], toc.size[i]);
        const size_t gx = g % d.xsize_groups, gy = g / d.xsize_groups;
        group(r, gx * d.group_dim, gy * d.group_dim, d.group_dim, d.group_dim, 0, 2, int(1 + 3 * d.num_dc_groups + 17 + g));
        REQUIRE(!r.Overread(), "AC group %zu over-read", g);
      }
    }
    for (size_t i = full.transforms.size(); i-- > 0;) H::InverseTransform(&full, full.transforms[i]);
    REQUIRE(full.ch.size() == C, "channels after the inverse transforms: %zu", full.ch.size());
    size_t bad = 0;
    for (size_t y = 0; y < Hh; y++)
      for (size_t x = 0; x < W; x++)
        for (size_t c = 0; c < C; c++) bad += full.ch[c].Row(y)[x] != int32_t(raw[(y * W + x) * C + c]);
    REQUIRE(bad == 0, "%zu of %zu samples differ from the encoder's input", bad, W * Hh * C);
  } catch (const std::exception& e) {
    fprintf(stderr, "FAIL: %s\n", e.what());
    return 1;
  }
  return 0;
}

#ifndef KAT_ORACLE
static bool Slurp(const char* path, std::vector<uint8_t>* out) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  uint8_t buf[4096];
  size_t n;
  while ((n = fread(buf, 1, sizeof(buf), f)) > 0) out->insert(out->end(), buf, buf + n);
  fclose(f);
  return true;
}
static int TestIcc(const char* enc_path, const char* icc_path) {
  std::vector<uint8_t> enc, want;
  REQUIRE(Slurp(enc_path, &enc) && Slurp(icc_path, &want), "cannot read the fixtures");
  {
    std::vector<uint8_t> padded = enc;
    padded.resize(padded.size() + 16, 0);
    H::BitReader br(padded.data(), padded.size());
    std::vector<uint8_t> got;
    try {
      H::ReadIcc(br, &got);
    } catch (const std::exception& e) {
      REQUIRE(false, "the reference's coded profile was rejected: %s", e.what());
    }
    REQUIRE(got == want, "decoded profile differs from the reference's (%zu vs %zu bytes)", got.size(), want.size());
    REQUIRE((br.BitPos() + 7) / 8 <= enc.size(), "read past the coded profile");
  }
  // damaged streams: an error or a (different) profile, never a crash or an endless loop
  size_t rejected = 0;
  for (size_t trial = 0; trial < 400; trial++) {
    std::vector<uint8_t> bad = enc;
    bad[Rnd(uint32_t(bad.size()))] ^= uint8_t(1u << Rnd(8));
    if (trial & 1) bad.resize(Rnd(uint32_t(bad.size())) + 1);
    bad.resize(bad.size() + 16, 0);
    H::BitReader br(bad.data(), bad.size() - 16);
    std::vector<uint8_t> got;
    try {
      H::ReadIcc(br, &got);
    } catch (const std::exception&) {
      rejected++;
    }
  }
  REQUIRE(rejected > 200, "only %zu of 400 damaged streams were rejected", rejected);
  return 0;
}
#endif

static float Uniform(float lo, float hi) { return lo + (hi - lo) * float(Rnd(1u << 24)) / float(1u << 24); }
static int FastMathKat() {
  for (int i = 0; i < (1 << 20); i++) {
    const float f = Uniform(1e-7f, 1e3f);
    REQUIRE(std::fabs(std::log2(f) - H::FastLog2f(f)) < 3.1e-6f, "FastLog2f(%g)", f);
    const float g = Uniform(-100.0f, 100.0f), e2 = std::pow(2.0f, g);
    REQUIRE(std::fabs(e2 - H::FastPow2f(g)) / e2 < 3.1e-6f, "FastPow2f(%g)", g);
    const float b = Uniform(1e-3f, 1e3f), e = Uniform(-10.0f, 10.0f), ex = std::pow(b, e);
    REQUIRE(std::fabs(ex - H::FastPowf(b, e)) / ex < 3e-5f, "FastPowf(%g, %g)", b, e);
    const float c = Uniform(-1e3f, 1e3f);
    REQUIRE(std::fabs(std::cos(c) - H::SplineFastCos(c)) < 7e-5f, "FastCosf(%g)", c);
#ifdef KAT_ORACLE  // (the product draws splines on the device: its FastErff is compared with this one to 1e-6 by the GPU tests)
    const float r = Uniform(-5.0f, 5.0f);
    REQUIRE(std::fabs(std::erf(r) - H::SplineFastErf(r)) < 7e-4f, "FastErff(%g)", r);
#endif
  }
  return 0;
}

// LSB-first bit writer of the KATs' own (fields.cc / enc_bit_writer.cc wire format: bits fill a byte from its low end)
struct KatBitWriter {
  std::vector<uint8_t> buf;
  size_t pos = 0;
  void Write(unsigned nbits, uint64_t v) {
    for (unsigned i = 0; i < nbits; i++, pos++) {
      if ((pos & 7) == 0) buf.push_back(0);
      buf.back() |= uint8_t(((v >> i) & 1) << (pos & 7));
    }
  }
  void ZeroPad() { pos = (pos + 7) & ~size_t(7); }
  size_t BitPos() const { return pos; }
  std::vector<uint8_t>& bytes() { return buf; }
};

static int TestBits() {
  {  // TestOrder: LSB-first within a byte, bytes in stream order, multi-byte fields little-endian
    const uint8_t a[2] = {0x1F, 0xFC};
    H::BitReader r(a, 2);
    for (int i = 0; i < 5; i++) REQUIRE(r.Read(1) == 1, "order a");
    for (int i = 0; i < 5; i++) REQUIRE(r.Read(1) == 0, "order a");
    for (int i = 0; i < 6; i++) REQUIRE(r.Read(1) == 1, "order a");
    const uint8_t b[2] = {0x3F, 0xF8};
    H::BitReader rb(b, 2);
    REQUIRE(rb.Read(16) == 0xF83F, "u16 is little-endian");
    const uint8_t c[2] = {0xBD, 0x8D};
    H::BitReader rc(c, 2);
    REQUIRE(rc.Read(1) == 1 && rc.Read(3) == 6 && rc.Read(8) == 0xDB && rc.Read(4) == 8, "mixed sizes");
    REQUIRE(rc.BitPos() == 16 && !rc.Overread(), "consumed");
  }
  for (size_t size = 4; size < 32; size++) {  // ExtendsWithZeroes
    std::vector<uint8_t> data(size, 0xFF);
    for (size_t n = 0; n < size; n++) {
      H::BitReader r(data.data(), n);
      for (size_t i = 0; i < n * 8; i++) REQUIRE(r.Read(1) == 1, "n=%zu i=%zu", n, i);
      for (unsigned i = 0; i <= 56; i++) REQUIRE(r.Peek(i) == 0, "bits past the end are zero (n=%zu, %u bits)", n, i);
      REQUIRE(!r.Overread(), "peeking does not consume");
    }
  }
  {  // TotalCountersTest
    const uint8_t buf[8] = {1, 2, 3, 4};
    H::BitReader r(buf, 8);
    REQUIRE(r.BitPos() == 0, "counter");
    r.Read(1);
    REQUIRE(r.BitPos() == 1, "counter");
    r.Read(10);
    REQUIRE(r.BitPos() == 11, "counter");
    r.Read(4);
    r.Read(1);
    REQUIRE(r.BitPos() == 16, "counter");
    REQUIRE(r.Read(16) == 0x0403 && r.BitPos() == 32, "counter");
  }
  for (size_t skip = 0; skip < 128; skip++) {  // TestSkip: skipping equals reading
    for (size_t ones = 0; ones < 96; ones += 7) {
      KatBitWriter w;
      for (size_t i = 0; i < ones; i++) w.Write(1, 1);
      for (size_t i = 0; i < skip; i++) w.Write(1, 0);
      w.Write(3, 5);
      w.ZeroPad();
      H::BitReader r(w.bytes().data(), w.bytes().size());
      for (size_t i = 0; i < ones; i++) REQUIRE(r.Read(1) == 1, "skip");
      r.Skip(skip);
      REQUIRE(r.Read(3) == 5 && r.BitPos() == ones + skip + 3, "skip=%zu", skip);
    }
  }
  return 0;
}

// An independent writer of the U64 wire format (fields.cc:494-521), for the KAT only.
static void KatWriteU64(KatBitWriter& w, uint64_t v) {
  if (v == 0) {
    w.Write(2, 0);
  } else if (v <= 16) {
    w.Write(2, 1);
    w.Write(4, v - 1);
  } else if (v <= 272) {
    w.Write(2, 2);
    w.Write(8, v - 17);
  } else {
    w.Write(2, 3);
    w.Write(12, v & 4095);
    v >>= 12;
    int shift = 12;
    while (v > 0 && shift < 60) {
      w.Write(1, 1);
      w.Write(8, v & 255);
      v >>= 8;
      shift += 8;
    }
    if (v > 0) {
      w.Write(1, 1);
      w.Write(4, v & 15);
    } else {
      w.Write(1, 0);
    }
  }
}

static int TestFields() {
  {  // U32CoderTest: enc = Val(0), Bits(4), Val(0x7FFFFFFF), Bits(32); {value, bits, selector}
    const struct {
      uint32_t value, bits, sel;
    } cases[] = {{0, 2, 0}, {1, 6, 1}, {15, 6, 1}, {0x7FFFFFFF, 2, 2}, {128, 34, 3}, {0x7FFFFFFEu, 34, 3}, {0x80000000u, 34, 3}, {0xFFFFFFFFu, 34, 3}};
    for (const auto& t : cases) {
      KatBitWriter w;
      w.Write(2, t.sel);
      if (t.sel == 1) w.Write(4, t.value);
      if (t.sel == 3) w.Write(32, t.value);
      REQUIRE(w.BitPos() == t.bits, "U32 %u: the writer used %zu bits, the reference's test says %u", t.value, w.BitPos(), t.bits);
      w.ZeroPad();
      H::BitReader r(w.bytes().data(), w.bytes().size());
      const uint32_t got = H::ReadU32(r, H::Val(0), H::Bits(4), H::Val(0x7FFFFFFF), H::Bits(32));
      REQUIRE(got == t.value && r.BitPos() == t.bits, "U32 %u: read %u in %zu bits (expected %u bits)", t.value, got, r.BitPos(), t.bits);
    }
  }
  {  // U64CoderTest
    const struct {
      uint64_t value;
      unsigned bits;
    } cases[] = {{0, 2}, {1, 6}, {2, 6}, {8, 6}, {15, 6}, {16, 6}, {17, 10}, {18, 10}, {100, 10}, {271, 10}, {272, 10}, {273, 15}, {274, 15}, {1000, 15},
                 {4094, 15}, {4095, 15}, {4096, 24}, {4097, 24}, {10000, 24}, {1048574, 24}, {1048575, 24}, {1048576, 33}, {1048577, 33},
                 {10000000, 33}, {268435454, 33}, {268435455, 33}, {268435456ull, 42}, {268435457ull, 42}, {1000000000ull, 42},
                 {68719476734ull, 42}, {68719476735ull, 42}, {68719476736ull, 51}, {68719476737ull, 51}, {1000000000000ull, 51},
                 {17592186044414ull, 51}, {17592186044415ull, 51}, {17592186044416ull, 60}, {17592186044417ull, 60},
                 {100000000000000ull, 60}, {4503599627370494ull, 60}, {4503599627370495ull, 60}, {4503599627370496ull, 69},
                 {4503599627370497ull, 69}, {10000000000000000ull, 69}, {1152921504606846974ull, 69}, {1152921504606846975ull, 69},
                 {1152921504606846976ull, 73}, {1152921504606846977ull, 73}, {10000000000000000000ull, 73},
                 {18446744073709551614ull, 73}, {18446744073709551615ull, 73}};
    for (const auto& t : cases) {
      KatBitWriter w;
      KatWriteU64(w, t.value);
      REQUIRE(w.BitPos() == t.bits, "U64 %llu: the writer used %zu bits, the reference's test says %u", (unsigned long long)t.value, w.BitPos(), t.bits);
      w.ZeroPad();
      H::BitReader r(w.bytes().data(), w.bytes().size());
      const uint64_t got = H::ReadU64(r);
      REQUIRE(got == t.value && r.BitPos() == t.bits, "U64 %llu: read %llu in %zu bits", (unsigned long long)t.value, (unsigned long long)got, r.BitPos());
    }
  }
  {  // F16CoderTest: IEEE binary16 bit patterns of the values the reference lists (all exactly representable)
    const struct {
      float value;
      uint16_t bits;
    } cases[] = {{0.0f, 0x0000}, {0.5f, 0x3800}, {1.0f, 0x3C00}, {2.0f, 0x4000}, {2.5f, 0x4100}, {16.015625f, 0x4C01},
                 {1.0f / 4096, 0x0C00}, {1.0f / 16384, 0x0400}, {65504.0f, 0x7BFF}};
    for (int sign = 0; sign < 2; sign++)
      for (const auto& t : cases) {
        KatBitWriter w;
        w.Write(16, t.bits | (sign << 15));
        H::BitReader r(w.bytes().data(), w.bytes().size());
        const float got = H::ReadF16(r);
        REQUIRE(got == (sign ? -t.value : t.value) && r.BitPos() == 16, "F16 %g: read %g", t.value, got);
      }
    // subnormals (fields.cc:565-569): 2^-24 * mantissa
    KatBitWriter w;
    w.Write(16, 0x0001);
    w.Write(16, 0x83FF);
    H::BitReader r(w.bytes().data(), w.bytes().size());
    REQUIRE(H::ReadF16(r) == 1.0f / 16777216 && H::ReadF16(r) == -1023.0f / 16777216, "F16 subnormals");
  }
  return 0;
}

static int TestQuantUniform() {
  H::DequantTables t;
  for (int k = 0; k < 17; k++) {
    H::QuantEncoding e;
    e.mode = 6;  // QuantEncoding::DCT(dct_params) for EVERY table, like the reference's test
    e.nb = 2;
    for (int c = 0; c < 3; c++) {
      e.bands[c][0] = 1.0f / 4;
      e.bands[c][1] = 0.0f;
    }
    t.enc[k] = e;
  }
  for (int strategy = 0; strategy < 27; strategy++)
    for (int c = 0; c < 3; c++) {
      const float* m = t.Matrix(strategy, c);
      const size_t n = t.table[H::kStrategyQuantTable[strategy]].size() / 3;
      REQUIRE(n == size_t(64) * H::kQTReqX[H::kStrategyQuantTable[strategy]] * H::kQTReqY[H::kStrategyQuantTable[strategy]], "table size");
      // the first entry exactly like the reference asserts it (1e-6); the others pass through FastPowf(1, t), whose own
      // error bar is 3e-5 relative (fast_math_test.cc) and which the round-trip part of the reference's test holds to 1e-4
      REQUIRE(std::fabs(m[0] - 4.0f) < 1e-6f, "strategy %d channel %d: first entry %.9g, expected 4", strategy, c, m[0]);
      for (size_t i = 0; i < n; i++) REQUIRE(std::fabs(m[i] - 4.0f) < 4.0f * 3e-5f, "strategy %d channel %d entry %zu = %.9g, expected 4", strategy, c, i, m[i]);
    }
  // a second, non-trivial closed form of the band interpolation (quant_weights.cc:37-90 GetQuantWeights): bands {1, -1}
  // halve from distance 0 to the far corner, weight(d) = 1 * (1/2)^(d / dmax) along the scaled distance, so the
  // table (1 / weight) grows from 1 to 2 monotonically with the distance of (x, y) from the origin
  for (int k : {0, 4, 5}) {
    H::DequantTables u;
    H::QuantEncoding e;
    e.mode = 6;
    e.nb = 2;
    for (int c = 0; c < 3; c++) {
      e.bands[c][0] = 1.0f;
      e.bands[c][1] = -1.0f;
    }
    u.enc[k] = e;
    u.Compute(k);
    const size_t rows = 8 * H::kQTReqX[k], cols = 8 * H::kQTReqY[k];
    const float* m = u.table[k].data();
    REQUIRE(std::fabs(m[0] - 1.0f) < 1e-5f && std::fabs(m[rows * cols - 1] - 2.0f) < 2e-4f, "band ends: %g %g", m[0], m[rows * cols - 1]);
    for (size_t y = 0; y < rows; y++)
      for (size_t x = 0; x + 1 < cols; x++) REQUIRE(m[y * cols + x + 1] >= m[y * cols + x] - 1e-6f, "monotone along a row (%zu, %zu)", y, x);
  }
  return 0;
}

// "defaults": the floating-point constants this front-end carries (default-constructed headers, the .inc tables next to
// it), one "name v0 v1 ..." line each, for tests/test_kats.py to compare with the numbers extracted from the reference's
// source (tests/golden/ref_constant_floats.json).
namespace kat_tables {
#ifdef KAT_ORACLE
#include "../../oracle/upsampling_weights.inc"
#include "../../oracle/dither.inc"
#else
#include "../../libjxl_amd/csrc/host/upsampling_weights.inc"
#include "../../libjxl_amd/csrc/host/dither.inc"
#include "../../libjxl_amd/csrc/host/afv_basis.inc"
#endif
}  // namespace kat_tables
static void Dump(const char* name, const float* v, size_t n) {
  printf("%s", name);
  for (size_t i = 0; i < n; i++) printf(" %.9g", double(v[i]));
  printf("\n");
}
static int DumpDefaults() {
  H::ImageHeader ih;
  Dump("inverse_opsin", ih.inv_opsin, 9);
  Dump("opsin_bias", ih.opsin_bias, 3);
  Dump("quant_bias", ih.quant_bias, 4);
  H::LoopFilter lf;
  Dump("gab_weights", &lf.gab_w[0][0], 6);
  Dump("epf_sharp_lut", lf.epf_sharp_lut, 8);
  Dump("epf_channel_scale", lf.epf_channel_scale, 3);
  const float epf[4] = {lf.epf_quant_mul, lf.epf_pass0_sigma_scale, lf.epf_pass2_sigma_scale, lf.epf_border_sad_mul};
  Dump("epf_scalars", epf, 4);
  H::DequantTables dq;
  Dump("dc_quant", dq.dc_quant, 3);
  Dump("upsampling_weights2", kat_tables::kUpsamplingWeights2, 15);
  Dump("upsampling_weights4", kat_tables::kUpsamplingWeights4, 55);
  Dump("upsampling_weights8", kat_tables::kUpsamplingWeights8, 210);
  Dump("dither32", &kat_tables::kDither32[0][0], 1024);
  {  // the default dequantisation-table parameters, per kind in the reference's own order of writing them down
    for (int k = 0; k < 17; k++) {
      std::vector<float> v;
      if (k == 1) {
        for (int c = 0; c < 3; c++)
          for (int i = 0; i < 3; i++) v.push_back(float(H::kQLIdentity[c][i]));
      } else if (k == 2) {
        for (int c = 0; c < 3; c++)
          for (int i = 0; i < 6; i++) v.push_back(float(H::kQLDct2[c][i]));
      } else if (k == 10) {  // (its bands are DCT4X8's and DCT4X4's: the line carries the 27 weights)
        for (int c = 0; c < 3; c++)
          for (int i = 0; i < 9; i++) v.push_back(float(H::kQLAfv[c][i]));
        for (int c = 0; c < 3; c++)
          for (int i = 0; i < 4; i++)
            if (H::kQLBands[10][c][i] != H::kQLBands[9][c][i]) return 1;
      } else {
        for (int c = 0; c < 3; c++)
          for (int i = 0; i < H::kQLNumBands[k]; i++) v.push_back(float(H::kQLBands[k][c][i]));
        if (k == 3)
          for (int c = 0; c < 3; c++)
            for (int i = 0; i < 2; i++) v.push_back(float(H::kQLDct4Mul[c][i]));
        if (k == 9)
          for (int c = 0; c < 3; c++) v.push_back(float(H::kQLDct4x8Mul[c]));
      }
      Dump(("quant_library_" + std::to_string(k)).c_str(), v.data(), v.size());
    }
  }
#ifdef KAT_ORACLE
  Dump("afv_basis", &H::kAfvBasis[0][0], 256);
#else
  Dump("afv_basis", &kat_tables::kAfvBasis[0][0], 256);
#endif
  return 0;
}

// tables OUT [STREAM]: what the host-only rows compute, for the independent NumPy readings of tests/host_tables_np.py
// (tests/test_host_tables.py): the 17 dequantisation tables (3 channels each, float32) from the default library or from a
// coded DequantMatrices section (quant_weights.cc:497-511: one all_default bit, then 17 encodings) read with THIS
// front-end's reader; the natural coefficient order of the 13 order buckets (ac_strategy.cc:28-79); samples of the context
// arithmetic (ac_context.h:63-143). A flat binary file: uint32 count, then float32 / uint32 words, per item.
static int DumpTables(const char* out_path, const char* stream_path) {
  FILE* f = fopen(out_path, "wb");
  REQUIRE(f != nullptr, "cannot write %s", out_path);
  auto put = [&](const void* p, size_t words) {
    const uint32_t n = uint32_t(words);
    fwrite(&n, 4, 1, f);
    fwrite(p, 4, words, f);
  };
  H::DequantTables dq;
  if (stream_path) {
    std::vector<uint8_t> bytes;
    FILE* g = fopen(stream_path, "rb");
    REQUIRE(g != nullptr, "cannot read %s", stream_path);
    int ch;
    while ((ch = fgetc(g)) != EOF) bytes.push_back(uint8_t(ch));
    fclose(g);
    bytes.resize(bytes.size() + 16, 0);
    try {
      H::BitReader br(bytes.data(), bytes.size());
      if (!br.ReadBool())
        for (int k = 0; k < 17; k++) H::ReadQuantEncoding(br, k, &dq.enc[k]);
    } catch (const std::exception& e) {
      fclose(f);
      REQUIRE(false, "reading the coded tables: %s", e.what());
    }
  }
  for (int k = 0; k < 17; k++) {
    try {
      dq.Compute(k);
    } catch (const std::exception& e) {
      fclose(f);
      REQUIRE(false, "table %d: %s", k, e.what());
    }
    put(dq.table[k].data(), dq.table[k].size());
  }
  for (int ord = 0; ord < 13; ord++) {
    std::vector<uint32_t> o;
    H::NaturalOrder(H::OrderBucketStrategy(ord), &o);
    put(o.data(), o.size());
  }
  {  // ZeroDensityContext over (non-zeros left, k) for every covered-block count that occurs, prev = 0 and 1
    std::vector<uint32_t> z;
    for (int log2c = 0; log2c <= 10; log2c++) {
      const size_t covered = size_t(1) << log2c, size = covered * 64;
      for (size_t k = covered; k < size; k += (size / 64 > 7 ? size / 61 : 1))
        for (size_t nz = 1; nz <= size - k && nz <= 4096; nz += (nz < 70 ? 1 : 37))
          for (size_t prev = 0; prev < 2; prev++) {
            z.push_back(uint32_t(log2c));
            z.push_back(uint32_t(k));
            z.push_back(uint32_t(nz));
            z.push_back(uint32_t(prev));
            z.push_back(uint32_t(H::ZeroDensityContext(nz, k, covered, size_t(log2c), prev)));
          }
    }
    put(z.data(), z.size());
  }
  fclose(f);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string t = argv[1];
  if (t == "defaults") return DumpDefaults();
  if (t == "tables" && (argc == 3 || argc == 4)) return DumpTables(argv[2], argc == 4 ? argv[3] : nullptr);
  if (t == "alias") return TestAlias();
  if (t == "hybrid") return TestHybrid();
  if (t == "lehmer") return TestLehmer();
  if (t == "fastmath") return FastMathKat();
  if (t == "bits") return TestBits();
  if (t == "fields") return TestFields();
  if (t == "quant") return TestQuantUniform();
#ifndef KAT_ORACLE
  if (t == "icc" && argc == 4) return TestIcc(argv[2], argv[3]);
#endif
  if (t == "fjxl" && argc == 7) return TestFjxl(argv[2], argv[3], size_t(atol(argv[4])), size_t(atol(argv[5])), size_t(atol(argv[6])));
  return 2;
}
