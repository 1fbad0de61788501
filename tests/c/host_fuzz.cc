// Sanitizer harness of the product's host front-end (libjxl_amd/csrc/host): the bytes of a codestream come from the
// outside, so every parser must turn damage into jxh::Error, never into an out-of-bounds access, an overflow the
// sanitizers see, or a hang. Built by tests/test_kats.py with -fsanitize=address,undefined (CPU only; GPU sanitizers are
// not available on the pool) and run over streams of the synthetic writer: headers (+ ICC, animation, orientation), the
// VarDCT frame plan (TOC, DC groups through the host Modular decoder, quant tables, context maps, histograms) and the
// Modular frame plan, each stream damaged `rounds` times (byte flips, bit flips, truncation, splices), seeded.
// usage: host_fuzz rounds file...   exit code 0 = no sanitizer report, no crash.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "../../libjxl_amd/csrc/host/jxh_frame.h"
#include "../../libjxl_amd/csrc/host/jxh_modframe.h"

static uint64_t g_rng = 0x2545F4914F6CDD1Dull;
static uint32_t Rnd(uint32_t n) {
  g_rng ^= g_rng << 13;
  g_rng ^= g_rng >> 7;
  g_rng ^= g_rng << 17;
  return uint32_t((g_rng >> 11) % (n ? n : 1));
}

// Parses every frame the stream holds; returns the number of frames parsed, counts refusals.
static int ParseAll(const std::vector<uint8_t>& v, size_t* refused) {
  int frames = 0;
  try {
    jxh::FrameParser parser(v.data(), v.size());
    jxh::ImageHeader ih;
    size_t pos = parser.ParseImageHeader(&ih);
    for (size_t index = 0; index < 16; index++) {
      bool last = true;
      size_t end = 0;
      try {
        jxh::FramePlan plan;
        parser.ParseFrame(pos, ih, &plan, jxh::SerialFor, index);
        last = plan.fh.is_last;
        end = plan.frame_end;
      } catch (const std::exception& e) {
        if (!strstr(e.what(), "Modular frames")) throw;
        jxh::ModFramePlan plan;
        jxh::ModFrameParser mp(v.data(), v.size());
        mp.ParseFrame(pos, ih, &plan, index);
        last = plan.fh.is_last;
        end = plan.frame_end;
      }
      frames++;
      if (last || end <= pos || end >= v.size()) break;
      pos = end;
    }
  } catch (const std::exception&) {
    ++*refused;
  }
  return frames;
}

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int rounds = atoi(argv[1]);
  size_t refused = 0, parsed = 0, total = 0;
  for (int a = 2; a < argc; a++) {
    FILE* f = fopen(argv[a], "rb");
    if (!f) return 2;
    std::vector<uint8_t> orig;
    uint8_t buf[65536];
    size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) orig.insert(orig.end(), buf, buf + n);
    fclose(f);
    size_t r0 = 0;
    if (ParseAll(orig, &r0) < 1 || r0) {
      fprintf(stderr, "FAIL: the undamaged stream %s does not parse\n", argv[a]);
      return 1;
    }
    for (int r = 0; r < rounds; r++) {
      std::vector<uint8_t> v = orig;
      const uint32_t kind = Rnd(6);
      // damage is biased towards the front: headers, TOC and the global sections decide what every later read does
      const size_t span = Rnd(3) ? std::min<size_t>(v.size(), 64 + Rnd(1024)) : v.size();
      if (kind == 0) {
        for (uint32_t i = 0, k = 1 + Rnd(4); i < k; i++) v[Rnd(uint32_t(span))] = uint8_t(Rnd(256));
      } else if (kind == 1) {
        for (uint32_t i = 0, k = 1 + Rnd(3); i < k; i++) v[Rnd(uint32_t(span))] ^= uint8_t(1u << Rnd(8));
      } else if (kind == 2) {
        v.resize(Rnd(uint32_t(v.size())));
      } else if (kind == 3) {  // a run of one value
        const size_t at = Rnd(uint32_t(span)), len = std::min<size_t>(v.size() - at, 1 + Rnd(32));
        memset(v.data() + at, Rnd(2) ? 0xFF : 0, len);
      } else if (kind == 4) {  // a piece of the stream copied over another place
        const size_t len = 1 + Rnd(64), from = Rnd(uint32_t(v.size())), to = Rnd(uint32_t(span));
        for (size_t i = 0; i < len && from + i < v.size() && to + i < v.size(); i++) v[to + i] = orig[from + i];
      } else {  // bytes removed: everything behind shifts
        const size_t at = Rnd(uint32_t(span)), len = std::min<size_t>(v.size() - at, 1 + Rnd(8));
        v.erase(v.begin() + at, v.begin() + at + len);
      }
      if (v.empty()) continue;
      total++;
      if (ParseAll(v, &refused) > 0) parsed++;
    }
  }
  printf("host_fuzz: %zu damaged streams, %zu refused, %zu still gave at least one frame\n", total, refused, parsed);
  return 0;
}
