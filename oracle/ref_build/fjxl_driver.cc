// Test-infrastructure driver around the REFERENCE's standalone lossless encoder (lib/jxl/enc_fast_lossless.cc,
// compiled in place from /root/reference with -DFJXL_STANDALONE=1; see oracle/Makefile target `ref`).
// Usage: fjxl_enc <in.raw> <width> <height> <channels> <bitdepth> <effort> <out.jxl>
// in.raw = interleaved samples, 8-bit or 16-bit little-endian.
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "lib/jxl/enc_fast_lossless.h"

int main(int argc, char** argv) {
  if (argc != 8) {
    fprintf(stderr, "usage: %s in.raw width height channels bitdepth effort out.jxl\n", argv[0]);
    return 2;
  }
  size_t w = atol(argv[2]), h = atol(argv[3]), nc = atol(argv[4]), bits = atol(argv[5]);
  int effort = atoi(argv[6]);
  size_t bps = bits > 8 ? 2 : 1;
  std::vector<unsigned char> in(w * h * nc * bps);
  FILE* f = fopen(argv[1], "rb");
  if (!f || fread(in.data(), 1, in.size(), f) != in.size()) {
    fprintf(stderr, "cannot read %s\n", argv[1]);
    return 1;
  }
  fclose(f);
  unsigned char* out = nullptr;
  size_t n = JxlFastLosslessEncode(in.data(), w, w * nc * bps, h, nc, bits, /*big_endian=*/false, effort, &out,
                                   nullptr, nullptr);
  if (!n) return 1;
  f = fopen(argv[7], "wb");
  fwrite(out, 1, n, f);
  fclose(f);
  free(out);
  return 0;
}
