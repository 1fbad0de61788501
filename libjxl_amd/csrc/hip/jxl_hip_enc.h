// Forward VarDCT path on the device (first slice): sRGB8 -> XYB, inverse-Gaborish sharpening, block
// activity, transform selection + quant field, forward DCT + DC extraction + quantisation with chroma-from-luma.
// What the reference does in enc_xyb.cc:83-174 (SRGBToXYB), enc_gaborish.cc:21-70, enc_group.cc:380-533
// (ComputeCoefficients: TransformFromPixels, DCFromLowestFrequencies, QuantizeBlockAC / QuantizeRoundtripYBlockAC) and
// enc_transforms-inl.h. The selection heuristics are the ones of this repo's CPU stream writer (csrc/enc), NOT the
// reference's AC-strategy search / adaptive quantisation / Butteraugli loop (enc_ac_strategy.cc:827-1068,
// enc_adaptive_quantization.cc:664-1115): those stay out of this slice. Every float expression keeps the operation order
// of the CPU writer with contraction off, so that the two agree except where cbrtf / log2f differ in the last place.
#ifndef JXL_HIP_ENC_H_
#define JXL_HIP_ENC_H_

namespace jxlhip {

struct EncFwd {
  const uint8_t* rgb;     // interleaved RGB8, `rgb_stride` bytes per row
  const float* srgb_lut;  // 256 floats: sRGB byte -> linear
  float* planes;          // [3][yp][xp] XYB (X, Y, B), padded to whole blocks by edge replication
  float* planes_in;       // ping-pong partners of the sharpening iterations
  float* planes_orig;
  float* act;             // [yb][xb] mean absolute deviation of Y
  uint8_t* acs;           // [yb][xb] (strategy << 1) | first
  int32_t* qf;            // [yb][xb] quant field at first blocks
  uint32_t* coef_off;     // [yb][xb] coefficient offset of a first block inside its group
  const float* basis_t;   // [n][k] DCT matrices of 1..256 points (level l at (4^l - 1) / 3)
  const float* dequant;
  uint32_t dq_offset[17], dq_size[17];
  int32_t* dc;      // [3][yb][xb] quantised DC, stored X, Y, B
  int32_t* coeffs;  // [group][3][65536]
  uint32_t xs, ys, xb, yb, xp, yp, xg, yg;
  size_t rgb_stride;
  float distance, quant_ac, inv_gs, x_dm, b_dm;
  float dc_step[3];
  uint32_t strategy_mode;
  // chroma-from-luma fit per 64x64 tile (enc_chroma_from_luma.cc:128-151, 204-352): factors out, one int8 per tile
  uint32_t cfl_fit;
  float scale;  // Quantizer::Scale() = global_scale / 65536
  int8_t* ytox;
  int8_t* ytob;
};

// enc_xyb.cc:50-104: opsin absorbance matrix + bias, cube root, X = (L - M) / 2, Y = (L + M) / 2, B = S.
__global__ void k_enc_xyb(EncFwd P) {
#pragma clang fp contract(off)
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= P.xp) return;
  const uint32_t sx = min(x, P.xs - 1), sy = min(y, P.ys - 1);
  const uint8_t* p = P.rgb + size_t(sy) * P.rgb_stride + size_t(sx) * 3;
  const float r = P.srgb_lut[p[0]], g = P.srgb_lut[p[1]], b = P.srgb_lut[p[2]];
  const float bias = 0.0037930732552754493f, cb = cbrtf(bias);
  const float mr = 0.30f * r + (1.0f - 0.078f - 0.30f) * g + 0.078f * b + bias;
  const float mg = 0.23f * r + (1.0f - 0.078f - 0.23f) * g + 0.078f * b + bias;
  const float mb = 0.24342268924547819f * r + 0.20476744424496821f * g + (1.0f - 0.24342268924547819f - 0.20476744424496821f) * b + bias;
  const float gr = cbrtf(mr) - cb, gg = cbrtf(mg) - cb, gb = cbrtf(mb) - cb;
  const size_t plane = size_t(P.xp) * P.yp, i = size_t(y) * P.xp + x;
  P.planes[i] = 0.5f * (gr - gg);
  P.planes[plane + i] = 0.5f * (gr + gg);
  P.planes[2 * plane + i] = gb;
}

// One round of y <- y + (x - K * y), K the decoder's 3x3 Gaborish with the default weights (the encoder-side sharpening
// enc_gaborish.cc:21-70 obtains with one tuned 5x5 kernel).
__global__ void k_enc_sharpen(const float* __restrict__ orig, const float* __restrict__ in, float* __restrict__ out, uint32_t xp, uint32_t yp) {
#pragma clang fp contract(off)
  const uint32_t xx = blockIdx.x * blockDim.x + threadIdx.x, yy = blockIdx.y, c = blockIdx.z;
  if (xx >= xp) return;
  const size_t plane = size_t(xp) * yp * c;
  const float* y = in + plane;
  const uint32_t y0 = yy ? yy - 1 : 0, y1 = yy + 1 < yp ? yy + 1 : yp - 1, x0 = xx ? xx - 1 : 0, x1 = xx + 1 < xp ? xx + 1 : xp - 1;
  const float w1 = 1.1f * 0.104699568f, w2 = 1.1f * 0.055680538f, nrm = 1.0f / (1.0f + 4 * (w1 + w2));
  const float side = y[size_t(yy) * xp + x0] + y[size_t(yy) * xp + x1] + y[size_t(y0) * xp + xx] + y[size_t(y1) * xp + xx];
  const float corner = y[size_t(y0) * xp + x0] + y[size_t(y0) * xp + x1] + y[size_t(y1) * xp + x0] + y[size_t(y1) * xp + x1];
  const float centre = y[size_t(yy) * xp + xx];
  const float blur = (centre + w1 * side + w2 * corner) * nrm;
  out[plane + size_t(yy) * xp + xx] = centre + (orig[plane + size_t(yy) * xp + xx] - blur);
}

// The four rounds in one launch: a 64x32 tile with a 4-sample apron in LDS; round k is computed on the tile grown by
// 4 - k samples, so the last round has every neighbour it needs. Neighbour coordinates are clamped in IMAGE coordinates
// exactly like the one-round kernel does, so positions outside the image are never read and each sample sees the same
// operands in the same order as there.
__global__ __launch_bounds__(256) void k_enc_sharpen4(const float* __restrict__ orig, float* __restrict__ out, uint32_t xp, uint32_t yp) {
#pragma clang fp contract(off)
  constexpr int TW = 64, TH = 32, H = 4, LW = TW + 2 * H, LH = TH + 2 * H;
  __shared__ float s_o[LW * LH], s_a[LW * LH], s_b[LW * LH];
  const int tx0 = int(blockIdx.x) * TW - H, ty0 = int(blockIdx.y) * TH - H, tid = threadIdx.x;
  const size_t plane = size_t(xp) * yp * blockIdx.z;
  for (int i = tid; i < LW * LH; i += 256) {
    const int gx = tx0 + i % LW, gy = ty0 + i / LW;
    if (gx < 0 || gy < 0 || gx >= int(xp) || gy >= int(yp)) continue;
    const float v = orig[plane + size_t(gy) * xp + gx];
    s_o[i] = v;
    s_a[i] = v;
  }
  __syncthreads();
  const float w1 = 1.1f * 0.104699568f, w2 = 1.1f * 0.055680538f, nrm = 1.0f / (1.0f + 4 * (w1 + w2));
  float* y = s_a;
  float* t = s_b;
  // a tile whose apron lies inside the image needs no clamping: the neighbours are the LDS neighbours
  const bool interior = tx0 >= 0 && ty0 >= 0 && tx0 + LW <= int(xp) && ty0 + LH <= int(yp);
  for (int it = 1; it <= 4; it++) {
    for (int i = tid; i < LW * LH; i += 256) {
      const int lx = i % LW, ly = i / LW, gx = tx0 + lx, gy = ty0 + ly;  // (constant divisors)
      if (lx < it || ly < it || lx >= LW - it || ly >= LH - it) continue;
      int x0, x1, y0, y1;
      const int yc = ly * LW;
      if (interior) {
        x0 = lx - 1;
        x1 = lx + 1;
        y0 = yc - LW;
        y1 = yc + LW;
      } else {
        if (gx < 0 || gy < 0 || gx >= int(xp) || gy >= int(yp)) continue;
        x0 = (gx ? gx - 1 : 0) - tx0;
        x1 = (gx + 1 < int(xp) ? gx + 1 : int(xp) - 1) - tx0;
        y0 = ((gy ? gy - 1 : 0) - ty0) * LW;
        y1 = ((gy + 1 < int(yp) ? gy + 1 : int(yp) - 1) - ty0) * LW;
      }
      const float side = y[yc + x0] + y[yc + x1] + y[y0 + lx] + y[y1 + lx];
      const float corner = y[y0 + x0] + y[y0 + x1] + y[y1 + x0] + y[y1 + x1];
      const float centre = y[yc + lx];
      const float blur = (centre + w1 * side + w2 * corner) * nrm;
      const float v = centre + (s_o[yc + lx] - blur);
      if (it == 4) out[plane + size_t(gy) * xp + gx] = v;
      else t[yc + lx] = v;
    }
    __syncthreads();
    float* sw = y;
    y = t;
    t = sw;
  }
}

// The four rounds without LDS: a wave owns 64 adjacent columns (the middle 56 are written, four of halo either side) and
// walks down a strip of rows. Round k of row r needs rows r - 1 .. r + 1 of round k - 1, so when source row s arrives
// round 1 forms row s - 1, round 2 row s - 2, round 3 row s - 3 and round 4, the output, row s - 4: per round a window of
// three rows x (left, centre, right) lives in registers, the horizontal neighbours come from the adjacent lanes
// (wave_shr / wave_shl). Rows and columns outside the image are replaced as k_enc_sharpen does it (the neighbour index is
// clamped in image coordinates: the sample itself at an image edge); every sample sees the same operands in the same
// order as there. Lanes / rows of the halo that lack a neighbour hold garbage that never reaches a written sample.
constexpr int kSharpenCols = 56, kSharpenRows = 64;
__device__ __forceinline__ float EncFromLeft(float v) {  // the value of the lane of column x - 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xF, 0xF, true));
}
__device__ __forceinline__ float EncFromRight(float v) {  // ... of column x + 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xF, 0xF, true));
}
struct SharpenRow {  // one row of each level j (0 = the source, j = after round j): left / centre / right neighbours
  float L[4], C[4], R[4];
};
// One step: source row s enters; `c` (the oldest row's storage) receives the entering row of every level, `a` and `b` hold
// rows r - 1 and r of the level. The three row sets take the roles in turn, so nothing is copied from step to step.
// EDGE = false: no sample of the step lies on an image edge (no clamped neighbour to substitute).
template <bool EDGE>
__device__ __forceinline__ void SharpenStep(SharpenRow& a, SharpenRow& b, SharpenRow& c, float v0, const float (&osrc)[4], int s, int last,
                                            bool left_edge, bool right_edge, bool writes, int Y0, int Y1, float* dst, uint32_t xp) {
#pragma clang fp contract(off)
  const float w1 = 1.1f * 0.104699568f, w2 = 1.1f * 0.055680538f, nrm = 1.0f / (1.0f + 4 * (w1 + w2));
  float nC = v0;  // the new row of level j as it enters level j's window
#pragma unroll
  for (int j = 0; j < 4; j++) {
    const float nL = EncFromLeft(nC), nR = EncFromRight(nC);
    c.L[j] = (EDGE && left_edge) ? nC : nL;
    c.C[j] = nC;
    c.R[j] = (EDGE && right_edge) ? nC : nR;
    // round j + 1 forms row r = s - j - 1
    const int r = s - j - 1;
    if (EDGE && r == 0) {  // no row above: the row itself
      a.L[j] = b.L[j]; a.C[j] = b.C[j]; a.R[j] = b.R[j];
    }
    if (EDGE && r == last) {  // no row below
      c.L[j] = b.L[j]; c.C[j] = b.C[j]; c.R[j] = b.R[j];
    }
    const float side = b.L[j] + b.R[j] + a.C[j] + c.C[j];
    const float corner = a.L[j] + a.R[j] + c.L[j] + c.R[j];
    const float centre = b.C[j];
    const float blur = (centre + w1 * side + w2 * corner) * nrm;
    nC = centre + (osrc[j] - blur);  // row r of level j + 1
    if (j == 3 && writes && r >= Y0 && r < Y1) dst[size_t(r) * xp] = nC;
  }
}
__global__ __launch_bounds__(256) void k_enc_sharpen_rows(const float* __restrict__ orig, float* __restrict__ out, uint32_t xp, uint32_t yp) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x >> 6));
  const int strip_x0 = (int(blockIdx.x) * 4 + wave) * kSharpenCols - 4, gx = strip_x0 + lane;
  const int Y0 = int(blockIdx.y) * kSharpenRows, Y1 = min(Y0 + kSharpenRows, int(yp));
  if (strip_x0 + 4 >= int(xp)) return;  // (a wave beyond the last column strip)
  const size_t plane = size_t(xp) * yp * blockIdx.z;
  const bool in_x = gx >= 0 && gx < int(xp), left_edge = gx == 0, right_edge = gx == int(xp) - 1;
  const bool x_edge = strip_x0 <= 0 || strip_x0 + 63 >= int(xp) - 1;  // (wave-uniform: the strip holds an image edge column)
  const bool writes = lane >= 4 && lane < 4 + kSharpenCols && in_x;
  const float* src = orig + plane + (in_x ? gx : 0);
  float* dst = out + plane + (in_x ? gx : 0);
  SharpenRow P, Q, R;
#pragma unroll
  for (int j = 0; j < 4; j++)
    P.L[j] = P.C[j] = P.R[j] = Q.L[j] = Q.C[j] = Q.R[j] = R.L[j] = R.C[j] = R.R[j] = 0.0f;
  float osrc[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // the source's centre values of rows s - 1 .. s - 4
  const int s_begin = max(Y0 - 4, 0), s_end = Y1 - 1 + 4, last = int(yp) - 1;
  float next = in_x ? src[size_t(s_begin) * xp] : 0.0f;
  // (the row sets rotate through the roles in a fixed order of three steps; up to two steps beyond s_end write nothing)
  auto step = [&](SharpenRow& a, SharpenRow& b, SharpenRow& c, int s) {
    const float v0 = next;  // source row s (stale beyond the image: replaced by the edge rule before anything reads it)
    if (s + 1 <= last) next = in_x ? src[size_t(s + 1) * xp] : 0.0f;
    // rounds 1..4 form rows s - 1 .. s - 4: one of them is the image's first or last row
    if (x_edge || s <= 4 || s > last) SharpenStep<true>(a, b, c, v0, osrc, s, last, left_edge, right_edge, writes, Y0, Y1, dst, xp);
    else SharpenStep<false>(a, b, c, v0, osrc, s, last, left_edge, right_edge, writes, Y0, Y1, dst, xp);
    osrc[3] = osrc[2]; osrc[2] = osrc[1]; osrc[1] = osrc[0]; osrc[0] = v0;
  };
  for (int s = s_begin; s <= s_end; s += 3) {
    step(Q, R, P, s);
    step(R, P, Q, s + 1);
    step(P, Q, R, s + 2);
  }
}

// Mean absolute deviation of Y from the block mean, per 8x8 block: a wave per 8 blocks (lane = block * 8 + row).
__global__ void k_enc_activity(EncFwd P) {
#pragma clang fp contract(off)
  const uint32_t lane = threadIdx.x, row = lane & 7;
  const uint32_t bx = blockIdx.x * 8 + (lane >> 3), by = blockIdx.y;
  const bool live = bx < P.xb;
  const float* s = P.planes + size_t(P.xp) * P.yp + size_t(by * 8 + row) * P.xp + size_t(live ? bx : 0) * 8;
  float v[8];
  for (int x = 0; x < 8; x++) v[x] = s[x];
  // the CPU writer sums the 64 samples in raster order: row sums cannot be combined without changing the rounding, so
  // lane (block, 0) walks the rows of its block through cross-lane reads
  float mean = 0.0f;
  for (int y = 0; y < 8; y++)
    for (int x = 0; x < 8; x++) mean += __shfl(v[x], (lane & ~7u) + y, 64);
  mean /= 64;
  float a = 0.0f;
  for (int y = 0; y < 8; y++)
    for (int x = 0; x < 8; x++) a += fabsf(__shfl(v[x], (lane & ~7u) + y, 64) - mean);
  if (live && row == 0) P.act[size_t(by) * P.xb + bx] = a / 64;
}

// Transform selection and quant field of one 64x64 tile (8x8 blocks) per wave: every candidate is aligned to its own
// size, so the greedy raster scan of the CPU writer never looks outside the tile it is in. The activity tests of every
// (position, candidate) pair are evaluated by the lanes in parallel; only the occupancy bookkeeping of the scan is serial.
__global__ __launch_bounds__(64) void k_enc_select(EncFwd P) {
#pragma clang fp contract(off)
  const uint32_t tiles_x = (P.xb + 7) / 8;
  const uint32_t t = blockIdx.x, lane = threadIdx.x, x = lane & 7, y = lane >> 3;
  __shared__ float s_act[64];
  __shared__ uint32_t s_pass[64];
  __shared__ uint8_t s_acs[64];
  const uint32_t bx0 = (t % tiles_x) * 8, by0 = (t / tiles_x) * 8;
  const uint32_t w = min(8u, P.xb - bx0), h = min(8u, P.yb - by0);
  const bool inside = x < w && y < h;
  s_act[lane] = inside ? P.act[size_t(by0 + y) * P.xb + bx0 + x] : 0.0f;
  s_acs[lane] = 0xFF;
  __syncthreads();
  const float T64 = 0.004f * P.distance, T32 = 0.008f * P.distance, T16 = 0.016f * P.distance, TR = 0.011f * P.distance;
  const uint8_t cand[11] = {18, 20, 19, 5, 11, 10, 4, 9, 8, 7, 6};  // in the order the scan tries them
  const float thr[11] = {T64, T64 * 1.3f, T64 * 1.3f, T32, TR, TR, T16, T16 * 0.8f, T16 * 0.8f, T16 * 1.5f, T16 * 1.5f};
  auto region_max = [&](uint32_t bx, uint32_t by, uint32_t cx, uint32_t cy) {
    float m = 0.0f;
    for (uint32_t yy = 0; yy < cy; yy++)
      for (uint32_t xx = 0; xx < cx; xx++) m = fmaxf(m, s_act[(by + yy) * 8 + bx + xx]);
    return m;
  };
  uint32_t pass = 0;
  if (inside && P.strategy_mode == 1)
    for (int s = 0; s < 11; s++) {
      const uint32_t cx = c_covered_x[cand[s]], cy = c_covered_y[cand[s]];
      if (x % cx || y % cy || x + cx > w || y + cy > h) continue;
      if (region_max(x, y, cx, cy) < thr[s]) pass |= 1u << s;
    }
  s_pass[lane] = pass;
  __syncthreads();
  if (lane == 0) {
    uint64_t occupied = 0;
    for (uint32_t by = 0; by < h; by++)
      for (uint32_t bx = 0; bx < w; bx++) {
        if (occupied >> (by * 8 + bx) & 1) continue;
        int st = 0;
        uint64_t mask = uint64_t(1) << (by * 8 + bx);
        for (uint32_t ps = s_pass[by * 8 + bx]; ps; ps &= ps - 1) {
          const int s = __builtin_ctz(ps);
          const uint32_t cx = c_covered_x[cand[s]], cy = c_covered_y[cand[s]];
          const uint64_t row = ((uint64_t(1) << cx) - 1) << bx;  // cx <= 8
          uint64_t m = 0;
          for (uint32_t yy = 0; yy < cy; yy++) m |= row << ((by + yy) * 8);
          if (occupied & m) continue;
          st = cand[s];
          mask = m;
          break;
        }
        occupied |= mask;
        const uint32_t cx = c_covered_x[st], cy = c_covered_y[st];
        for (uint32_t yy = 0; yy < cy; yy++)
          for (uint32_t xx = 0; xx < cx; xx++) s_acs[(by + yy) * 8 + bx + xx] = uint8_t((st << 1) | ((xx | yy) == 0));
      }
  }
  __syncthreads();
  if (!inside) return;
  const uint8_t a = s_acs[lane];
  const size_t bi = size_t(by0 + y) * P.xb + bx0 + x;
  P.acs[bi] = a;
  int32_t q = 0;
  if (a & 1) {
    const int st = a >> 1;
    const float m = region_max(x, y, c_covered_x[st], c_covered_y[st]);
    float mul = 1.35f - 0.12f * log2f(1.0f + m * 400.0f);
    mul = fmaxf(0.8f, fminf(1.4f, mul));
    q = max(1, min(256, int(P.quant_ac * mul * P.inv_gs + 0.5f)));
  }
  P.qf[bi] = q;
}

// Coefficient offsets of the first blocks of one 256x256 group, in the raster order the bitstream uses: one wave per
// group, 16 consecutive raster positions per lane, exclusive scan across the lanes.
__global__ __launch_bounds__(64) void k_enc_offsets(EncFwd P) {
  const uint32_t g = blockIdx.x, lane = threadIdx.x;
  const uint32_t bx0 = (g % P.xg) * 32, by0 = (g / P.xg) * 32;
  const uint32_t gw = min(32u, P.xb - bx0), gh = min(32u, P.yb - by0), n = gw * gh;
  uint32_t sizes[16], sum = 0;
  for (uint32_t j = 0; j < 16; j++) {
    const uint32_t i = lane * 16 + j;
    uint32_t sz = 0;
    if (i < n) {
      const uint8_t a = P.acs[size_t(by0 + i / gw) * P.xb + bx0 + i % gw];
      if (a & 1) sz = 64u << c_log2_covered[a >> 1];
    }
    sizes[j] = sz;
    sum += sz;
  }
  uint32_t incl = sum;
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t o = __shfl_up(incl, d, 64);
    if (int(lane) >= d) incl += o;
  }
  uint32_t offset = incl - sum;
  for (uint32_t j = 0; j < 16; j++) {
    const uint32_t i = lane * 16 + j;
    if (i < n && sizes[j]) P.coef_off[size_t(by0 + i / gw) * P.xb + bx0 + i % gw] = offset;
    offset += sizes[j];
  }
}

__device__ __forceinline__ int32_t EncQuant(float v) {
  const float r = rintf(v);
  return fabsf(v) < 0.58f ? 0 : int32_t(r);
}

// One workgroup per transform: forward scaled DCT (matrix form, rows then columns), DC of the covered blocks from the
// lowest-frequency corner (the inverse of LowestFrequenciesFromDC, enc_transforms-inl.h DCFromLowestFrequencies), then
// quantisation: Y first, X and B as residuals of the chroma-from-luma prediction from the DEQUANTISED Y
// (enc_group.cc:455-520, QuantizeRoundtripYBlockAC :329-378).
template <int kThreads>
__global__ __launch_bounds__(kThreads) void k_enc_transform(EncFwd P) {
#pragma clang fp contract(off)
  const uint32_t bi = blockIdx.x, abx = bi % P.xb, aby = bi / P.xb, tid = threadIdx.x;
  const uint8_t a = P.acs[bi];
  if (!(a & 1)) return;
  const int st = a >> 1;
  const int cx = c_covered_x[st], cy = c_covered_y[st], R = cy * 8, C = cx * 8, size = R * C;
  const int cstride = max(cx, cy) * 8, lrows = min(cx, cy), lcols = max(cx, cy);
  __shared__ float s_a[4096], s_t[4096], s_yd[4096], s_ydc[64];
  const float* bc = P.basis_t + (size_t(C) * C - 1) / 3;
  const float* br = P.basis_t + (size_t(R) * R - 1) / 3;
  const float* bcc = P.basis_t + (size_t(cx) * cx - 1) / 3;
  const float* brr = P.basis_t + (size_t(cy) * cy - 1) / 3;
  const size_t plane = size_t(P.xp) * P.yp;
  const uint32_t g = (aby / 32) * P.xg + abx / 32, off = P.coef_off[bi];
  const float scaled = P.inv_gs / float(P.qf[bi]);
  const float norm = 1.0f / (float(R) * float(C));
  const int kind = c_strategy_qtable[st];
  const float biases1 = 1.0f - 0.07005449891748593f, biases3 = 0.145f;
  for (int ci = 0; ci < 3; ci++) {
    const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    const float* src = P.planes + plane * c + size_t(aby) * 8 * P.xp + size_t(abx) * 8;
    for (int i = tid; i < size; i += kThreads) s_a[i] = src[size_t(i / C) * P.xp + i % C];
    __syncthreads();
    for (int o = tid; o < size; o += kThreads) {
      const int y = o / C, kx = o % C;
      float s = 0;
      for (int x = 0; x < C; x++) s += s_a[y * C + x] * bc[x * C + kx];
      s_t[o] = s;
    }
    __syncthreads();
    for (int o = tid; o < size; o += kThreads) {
      const int ky = o / C, kx = o % C;
      float s = 0;
      for (int y = 0; y < R; y++) s += s_t[y * C + kx] * br[y * R + ky];
      s *= norm;
      s_a[R < C ? ky * C + kx : kx * R + ky] = s;
    }
    __syncthreads();
    if (int(tid) < cx * cy) {
      const int y = tid / cx, x = tid % cx;
      float s = 0;
      for (int ky = 0; ky < cy; ky++)
        for (int kx = 0; kx < cx; kx++) {
          const float cf = (R < C) ? s_a[ky * cstride + kx] : s_a[kx * cstride + ky];
          const float llf = cf / (c_resample[cy - 1 + ky] * c_resample[cx - 1 + kx]);
          s += llf * brr[y * cy + ky] * bcc[x * cx + kx];
        }
      const size_t di = size_t(aby + y) * P.xb + abx + x;
      const size_t nb = size_t(P.xb) * P.yb;
      if (c == 1) {
        const int32_t qy = int32_t(lroundf(s / P.dc_step[1]));
        s_ydc[tid] = float(qy) * P.dc_step[1];
        P.dc[nb + di] = qy;
      } else if (c == 0) {
        P.dc[di] = int32_t(lroundf((s - 0.0f * s_ydc[tid]) / P.dc_step[0]));
      } else {
        P.dc[2 * nb + di] = int32_t(lroundf((s - 1.0f * s_ydc[tid]) / P.dc_step[2]));
      }
    }
    const float mulc = c == 0 ? scaled * P.x_dm : (c == 1 ? scaled : scaled * P.b_dm);
    const float cc = c == 0 ? 0.0f : 1.0f;  // chroma-from-luma factors of the default colour correlation (0 and 1)
    const float* m = P.dequant + P.dq_offset[kind] + size_t(c) * P.dq_size[kind];
    int32_t* dst = P.coeffs + (size_t(g) * 3 + c) * 65536 + off;
    for (int k = tid; k < size; k += kThreads) {
      const int row = k / cstride, col = k % cstride;
      int32_t q = 0;
      if (!(row < lrows && col < lcols)) {
        if (c == 1) {
          q = EncQuant(s_a[k] / (m[k] * mulc));
          const float ybias = q == 0 ? 0.0f : (q == 1 ? biases1 : (q == -1 ? -biases1 : float(q) - biases3 / float(q)));
          s_yd[k] = ybias * (m[k] * mulc);
        } else {
          q = EncQuant((s_a[k] - cc * s_yd[k]) / (m[k] * mulc));
        }
      }
      dst[k] = q;
    }
    __syncthreads();
  }
}

// The same work for one 64x64 tile (8x8 blocks) per workgroup, every transform of the tile at once: the tile's samples,
// the row-pass result and the DCT matrices of 1..64 points sit in LDS; thread (column, 16-row band) accumulates 8 rows
// against one matrix element at a time, so that a matrix read serves 8 multiply-adds and the sample reads are
// broadcasts. Sums run in the CPU writer's order (x = 0.., y = 0..), products are not contracted.
__global__ __launch_bounds__(256) void k_enc_transform_tile(EncFwd P) {
#pragma clang fp contract(off)
  // the DCT matrices of 1..32 points in LDS ((4^6 - 1) / 3 floats: levels 0..5); the 64-point one (16 KB, only the three
  // largest transforms use it) is read in place, which lets four workgroups share a CU instead of two
  constexpr int kBasisLds = 1365;
  // (the level of N points starts at (N * N - 1) / 3, which is 1 mod 4: shifted by 3 floats so that rows are 16-byte aligned)
  __shared__ __attribute__((aligned(16))) float s_px[64 * 64], s_t[64 * 64], s_basis_raw[kBasisLds + 3];
  __shared__ float s_ydc[64];
  float* const s_basis = s_basis_raw + 3;
  __shared__ uint32_t s_info[64], s_off[64];
  __shared__ int32_t s_qf[64];
  __shared__ float s_red[8], s_cc;
  __shared__ uint32_t s_cells;  // blocks of the tile that lie inside the frame
  const uint32_t tiles_x = (P.xb + 7) / 8;
  const uint32_t bx0 = (blockIdx.x % tiles_x) * 8, by0 = (blockIdx.x / tiles_x) * 8, tid = threadIdx.x;
  const uint32_t w = min(8u, P.xb - bx0), h = min(8u, P.yb - by0);
  for (int i = tid; i < kBasisLds; i += 256) s_basis[i] = P.basis_t[i];
  if (tid < 64) s_info[tid] = 0xFFFFFFFFu;
  __syncthreads();
  if (tid < 64) {
    const uint32_t x = tid & 7, y = tid >> 3;
    if (x < w && y < h) {
      const size_t bi = size_t(by0 + y) * P.xb + bx0 + x;
      const uint8_t a = P.acs[bi];
      if (a & 1) {
        const uint32_t st = a >> 1, cx = c_covered_x[st], cy = c_covered_y[st], off = P.coef_off[bi];
        const int32_t q = P.qf[bi];
        for (uint32_t dy = 0; dy < cy; dy++)
          for (uint32_t dx = 0; dx < cx; dx++) {
            const uint32_t cell = (y + dy) * 8 + x + dx;
            s_info[cell] = x | y << 3 | st << 6;
            s_off[cell] = off;
            s_qf[cell] = q;
          }
      }
    }
  }
  __syncthreads();
  if (tid < 64) {
    const unsigned long long m = __ballot(s_info[tid] != 0xFFFFFFFFu);
    if (tid == 0) s_cells = uint32_t(__popcll(m));
  }
  const uint32_t col = tid & 63, band = tid >> 6;  // this thread: tile column `col`, cell rows 2 * band and 2 * band + 1
  const uint32_t g = (by0 / 32) * P.xg + bx0 / 32;
  const size_t plane = size_t(P.xp) * P.yp, nb = size_t(P.xb) * P.yb;
  const float biases1 = 1.0f - 0.07005449891748593f, biases3 = 0.145f;
  float ydeq[2][8], ycoef[2][8];
  for (int ci = 0; ci < 3; ci++) {
    const int c = ci == 0 ? 1 : (ci == 1 ? 0 : 2);
    __syncthreads();
    {
      // (all sixteen loads of a thread in flight before the first one is used; requesting the next channel's during this
      // one's transform was measured slower: 0.208 -> 0.255 ms per 4K frame)
      const float* src = P.planes + plane * c + size_t(by0) * 8 * P.xp + size_t(bx0) * 8;
      float px[16];
#pragma unroll
      for (uint32_t u = 0; u < 16; u++) {
        const uint32_t i = tid + u * 256, y = i >> 6, x = i & 63;
        px[u] = (x < w * 8 && y < h * 8) ? src[size_t(y) * P.xp + x] : 0.0f;
      }
#pragma unroll
      for (uint32_t u = 0; u < 16; u++) s_px[tid + u * 256] = px[u];
    }
    __syncthreads();
    // rows: s_t[y][kx] = sum_x px[y][x0 + x] * B_C[x][kx]
    for (uint32_t half = 0; half < 2; half++) {
      const uint32_t cr = band * 2 + half, info = s_info[cr * 8 + (col >> 3)];
      if (info == 0xFFFFFFFFu) continue;
      const uint32_t ox = info & 7, C = uint32_t(c_covered_x[info >> 6]) * 8, kx = col - ox * 8;
      const float* px = s_px + cr * 8 * 64 + ox * 8;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      auto rows = [&](const float* B) {
        for (uint32_t x = 0; x < C; x += 4) {  // four samples of each of the 8 rows per LDS read (a broadcast within a block)
          float4 v[8];
          for (int j = 0; j < 8; j++) v[j] = *reinterpret_cast<const float4*>(px + j * 64 + x);
          const float b0 = B[x * C], b1 = B[(x + 1) * C], b2 = B[(x + 2) * C], b3 = B[(x + 3) * C];
          for (int j = 0; j < 8; j++) {
            acc[j] += v[j].x * b0;
            acc[j] += v[j].y * b1;
            acc[j] += v[j].z * b2;
            acc[j] += v[j].w * b3;
          }
        }
      };
      if (C == 64) rows(P.basis_t + 1365 + kx);
      else rows(s_basis + (C * C - 1) / 3 + kx);
      for (int j = 0; j < 8; j++) s_t[(cr * 8 + j) * 64 + col] = acc[j];
    }
    __syncthreads();
    // columns: coefficient (ky, kx) at s_px[y0 + ky][col], natural layout
    for (uint32_t half = 0; half < 2; half++) {
      const uint32_t cr = band * 2 + half, info = s_info[cr * 8 + (col >> 3)];
      if (info == 0xFFFFFFFFu) continue;
      const uint32_t st = info >> 6, oy = (info >> 3) & 7, R = uint32_t(c_covered_y[st]) * 8, C = uint32_t(c_covered_x[st]) * 8;
      const uint32_t ky0 = (cr - oy) * 8;
      const float* tc = s_t + oy * 8 * 64 + col;
      float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      if (R == 64) {  // (the table in place: its rows are 4-byte aligned only)
        const float* B = P.basis_t + 1365 + ky0;
        for (uint32_t y = 0; y < R; y++) {
          const float v = tc[y * 64];
          for (int j = 0; j < 8; j++) acc[j] += v * B[y * R + j];
        }
      } else {
        const float* B = s_basis + (R * R - 1) / 3 + ky0;
        for (uint32_t y = 0; y < R; y++) {
          const float v = tc[y * 64];
          const float4 ba = *reinterpret_cast<const float4*>(B + y * R), bb = *reinterpret_cast<const float4*>(B + y * R + 4);
          acc[0] += v * ba.x;
          acc[1] += v * ba.y;
          acc[2] += v * ba.z;
          acc[3] += v * ba.w;
          acc[4] += v * bb.x;
          acc[5] += v * bb.y;
          acc[6] += v * bb.z;
          acc[7] += v * bb.w;
        }
      }
      const float norm = 1.0f / (float(R) * float(C));
      for (int j = 0; j < 8; j++) s_px[(cr * 8 + j) * 64 + col] = acc[j] * norm;
    }
    __syncthreads();
    // DC of every covered block from the lowest-frequency corner
    if (tid < 64 && s_info[tid] != 0xFFFFFFFFu) {
      const uint32_t info = s_info[tid], st = info >> 6, ox = info & 7, oy = (info >> 3) & 7;
      const int cx = c_covered_x[st], cy = c_covered_y[st], x = int(tid & 7) - int(ox), y = int(tid >> 3) - int(oy);
      const float* brr = s_basis + (cy * cy - 1) / 3;
      const float* bcc = s_basis + (cx * cx - 1) / 3;
      float s = 0;
      for (int ky = 0; ky < cy; ky++)
        for (int kx = 0; kx < cx; kx++) {
          const float llf = s_px[(oy * 8 + ky) * 64 + ox * 8 + kx] / (c_resample[cy - 1 + ky] * c_resample[cx - 1 + kx]);
          s += llf * brr[y * cy + ky] * bcc[x * cx + kx];
        }
      const size_t di = size_t(by0 + (tid >> 3)) * P.xb + bx0 + (tid & 7);
      if (c == 1) {
        const int32_t qy = int32_t(lroundf(s / P.dc_step[1]));
        s_ydc[tid] = float(qy) * P.dc_step[1];
        P.dc[nb + di] = qy;
      } else if (c == 0) {
        P.dc[di] = int32_t(lroundf((s - 0.0f * s_ydc[tid]) / P.dc_step[0]));
      } else {
        P.dc[2 * nb + di] = int32_t(lroundf((s - 1.0f * s_ydc[tid]) / P.dc_step[2]));
      }
    }
    // chroma-from-luma factor of the tile for this channel: the least squares of the reference's fast FindBestMultiplier
    // over the tile's AC coefficients (a = Y * w / 84, b = base * Y * w - chroma * w, w = Scale * 128 * qf / matrix)
    float cc = c == 0 ? 0.0f : 1.0f;  // (the default factors: 0 on X, the base correlation 1 on B)
    if (c == 1 || P.cfl_fit) {
      float sa2 = 0.0f, sab = 0.0f;
      for (uint32_t half = 0; half < 2; half++) {
        const uint32_t cr = band * 2 + half, cell = cr * 8 + (col >> 3), info = s_info[cell];
        if (info == 0xFFFFFFFFu) continue;
        const uint32_t st = info >> 6, ox = info & 7, oy = (info >> 3) & 7;
        const uint32_t cx = c_covered_x[st], cy = c_covered_y[st], R = cy * 8, C = cx * 8;
        const uint32_t kx = col - ox * 8, ky0 = (cr - oy) * 8;
        if (c == 1) {
          for (uint32_t j = 0; j < 8; j++) ycoef[half][j] = s_px[(cr * 8 + j) * 64 + col];
          continue;
        }
        const int kind = c_strategy_qtable[st];
        const float* m = P.dequant + P.dq_offset[kind] + size_t(c) * P.dq_size[kind];
        const float q = P.scale * 128.0f * float(s_qf[cell]);
        float mk[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) mk[j] = m[R < C ? (ky0 + j) * C + kx : kx * R + ky0 + j];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
          const uint32_t ky = ky0 + j;
          if (ky < cy && kx < cx) continue;
          const float w = q / mk[j], yw = ycoef[half][j] * w;
          const float a = (1.0f / 84) * yw, b = cc * yw - s_px[(cr * 8 + j) * 64 + col] * w;
          sa2 += a * a;
          sab += a * b;
        }
      }
      if (c != 1) {
        for (int d = 32; d > 0; d >>= 1) {
          sa2 += __shfl_down(sa2, d, 64);
          sab += __shfl_down(sab, d, 64);
        }
        if ((tid & 63) == 0) {
          s_red[(tid >> 6) * 2] = sa2;
          s_red[(tid >> 6) * 2 + 1] = sab;
        }
        __syncthreads();
        if (tid == 0) {
          const float ta2 = (s_red[0] + s_red[2]) + (s_red[4] + s_red[6]), tab = (s_red[1] + s_red[3]) + (s_red[5] + s_red[7]);
          const uint32_t cells = s_cells;
          const float num = float(cells * 64);
          float x = cells ? -tab / (ta2 + num * 1e-9f * 0.5f) : 0.0f;
          x = x >= 2.6f ? x - 2.6f : (x <= -2.6f ? x + 2.6f : 0.0f);
          const float v = fmaxf(-128.0f, fminf(127.0f, roundf(x)));
          (c == 0 ? P.ytox : P.ytob)[blockIdx.x] = int8_t(v);
          s_cc = cc + v / 84.0f;
        }
        __syncthreads();
        cc = s_cc;
      }
    }
    // quantisation of this thread's 2 x 8 coefficients
    for (uint32_t half = 0; half < 2; half++) {
      const uint32_t cr = band * 2 + half, cell = cr * 8 + (col >> 3), info = s_info[cell];
      if (info == 0xFFFFFFFFu) continue;
      const uint32_t st = info >> 6, ox = info & 7, oy = (info >> 3) & 7;
      const uint32_t cx = c_covered_x[st], cy = c_covered_y[st], R = cy * 8, C = cx * 8;
      const uint32_t kx = col - ox * 8, ky0 = (cr - oy) * 8;
      const int kind = c_strategy_qtable[st];
      const float scaled = P.inv_gs / float(s_qf[cell]);
      const float mulc = c == 0 ? scaled * P.x_dm : (c == 1 ? scaled : scaled * P.b_dm);
      const float* m = P.dequant + P.dq_offset[kind] + size_t(c) * P.dq_size[kind];
      int32_t* dst = P.coeffs + (size_t(g) * 3 + c) * 65536 + s_off[cell];
      int32_t q[8];
      float mk[8];  // (the table entries first: eight independent loads)
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) mk[j] = m[R < C ? (ky0 + j) * C + kx : kx * R + ky0 + j];
#pragma unroll
      for (uint32_t j = 0; j < 8; j++) {
        const uint32_t ky = ky0 + j;
        q[j] = 0;
        if (ky < cy && kx < cx) continue;  // lowest frequencies: carried by the DC image
        const float v = s_px[(cr * 8 + j) * 64 + col], step = mk[j] * mulc;
        if (c == 1) {
          q[j] = EncQuant(v / step);
          const float ybias = q[j] == 0 ? 0.0f : (q[j] == 1 ? biases1 : (q[j] == -1 ? -biases1 : float(q[j]) - biases3 / float(q[j])));
          ydeq[half][j] = ybias * step;
        } else {
          q[j] = EncQuant((v - cc * ydeq[half][j]) / step);
        }
      }
      if (R < C) {
        for (uint32_t j = 0; j < 8; j++) dst[(ky0 + j) * C + kx] = q[j];
      } else {  // 8 consecutive coefficients of column kx
        int4* d4 = reinterpret_cast<int4*>(dst + kx * R + ky0);
        d4[0] = make_int4(q[0], q[1], q[2], q[3]);
        d4[1] = make_int4(q[4], q[5], q[6], q[7]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------- tokenisation
// The AC coefficient tokens of a frame on the device (reference lib/jxl/enc_entropy_coder.cc:153-255 TokenizeCoefficients;
// contexts lib/jxl/ac_context.h:63-143, entropy_coder.h:25-35): the quantised coefficients of the forward path never leave
// HBM, the host entropy coder receives (context, value) pairs in bitstream order. Unlike the decoder's walk nothing here
// is serial: a (varblock, channel)'s non-zero count is a popcount, the predicted count comes from the finished map of
// counts, the running "non-zeros left" of coefficient k is the count minus a prefix popcount, and a token's place in its
// group's stream is a prefix sum of the per-(varblock, channel) token counts.
// Two launches, one workgroup of sixteen waves per 256x256 group, a wave per (varblock, channel):
//   k_enc_tok_count  list of the group's varblocks in raster order, per (varblock, channel) the non-zero count and the last
//                    non-zero scan position, the map of counts per 8x8 block, token counts and their exclusive scan
//   k_enc_tok_emit   the tokens, at the group's base (a host-side prefix sum of the 135 group totals) + that offset
// Single pass, natural coefficient order, a block context map without thresholds (the default one): what the forward path's
// callers use; anything else is tokenised by the host as before.
struct EncTok {
  const uint8_t* acs;        // per block: strategy << 1 | first (k_enc_select)
  const uint32_t* coef_off;  // per first block: offset of its coefficients in the group's planes (k_enc_offsets)
  const int32_t* coeffs;     // [group][3][65536]
  uint32_t xb, yb, xg;
  const uint16_t* orders;    // natural coefficient orders of the 13 order buckets, concatenated
  uint32_t order_offset[13];
  uint8_t ctx_map[39];       // block context of (channel in stream order Y, X, B -> c < 2 ? c ^ 1 : 2; order bucket)
  uint32_t num_ctxs, num_hist, nctx;
  // per group (kTokPerGroup = 3072 entries: 1024 varblocks x 3 channels)
  uint16_t* blk;             // [group][1024] raster positions of the first blocks
  uint32_t* nblk;            // [group]
  uint32_t* info;            // [group][3072]: non-zero count | (last non-zero scan position + 1) << 16
  uint32_t* off;             // [group][3072]: first token of the (varblock, channel) in its group's stream
  uint8_t* nzmap;            // [group][3][1024]: ceil(count / covered blocks) per 8x8 block (the decoder's nzeros map)
  uint32_t* total;           // [group]
  const uint32_t* base;      // [group] (emit): first token of the group in `tokens`
  uint2* tokens;             // (emit) {context, value}
};
constexpr uint32_t kTokPerGroup = 3072;

constexpr uint32_t kTokThreads = 1024, kTokWaves = kTokThreads / 64;  // sixteen waves per group: the walk is latency-bound

__global__ __launch_bounds__(kTokThreads) void k_enc_tok_count(EncTok P) {
  __shared__ uint16_t l_blk[1024];
  __shared__ uint32_t l_cnt[kTokPerGroup];
  __shared__ uint32_t l_scan[256];
  const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool scanner = tid < 256;  // the two scans are the first four waves' work
  const uint32_t bx0 = (g % P.xg) * 32, by0 = (g / P.xg) * 32;
  const uint32_t gw = min(32u, P.xb - bx0), gh = min(32u, P.yb - by0), n = gw * gh;
  // ---- the group's first blocks in raster order: four consecutive raster positions per thread, exclusive scan
  uint32_t first[4] = {0, 0, 0, 0}, cnt = 0;
  if (scanner) {
    for (uint32_t j = 0; j < 4; j++) {
      const uint32_t i = tid * 4 + j;
      first[j] = (i < n && (P.acs[size_t(by0 + i / gw) * P.xb + bx0 + i % gw] & 1)) ? 1u : 0u;
      cnt += first[j];
    }
    l_scan[tid] = cnt;
  }
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    const uint32_t v = (scanner && tid >= d) ? l_scan[tid - d] : 0u;
    __syncthreads();
    if (scanner) l_scan[tid] += v;
    __syncthreads();
  }
  const uint32_t nblk = l_scan[255];
  if (scanner) {
    uint32_t idx = l_scan[tid] - cnt;
    for (uint32_t j = 0; j < 4; j++)
      if (first[j]) {
        l_blk[idx] = uint16_t(tid * 4 + j);
        P.blk[size_t(g) * 1024 + idx] = uint16_t(tid * 4 + j);
        idx++;
      }
  }
  if (tid == 0) P.nblk[g] = nblk;
  __syncthreads();
  // ---- a wave per (varblock, channel): non-zero count beyond the lowest-frequency corner, last non-zero scan position
  for (uint32_t e = wave; e < nblk * 3; e += kTokWaves) {
    const uint32_t b = e / 3, ci = e % 3, c = ci == 0 ? 1u : (ci == 1 ? 0u : 2u);
    const uint32_t i = l_blk[b], lbx = i % gw, lby = i / gw;
    const size_t cell = size_t(by0 + lby) * P.xb + bx0 + lbx;
    const uint32_t st = P.acs[cell] >> 1, log2c = c_log2_covered[st], covered = 1u << log2c, size = covered * 64;
    const uint16_t* order = P.orders + P.order_offset[c_strategy_order[st]];
    const int32_t* co = P.coeffs + (size_t(g) * 3 + c) * 65536 + P.coef_off[cell];
    uint32_t nz = 0, last = 0;
    for (uint32_t k0 = 0; k0 < size; k0 += 64) {
      const uint32_t k = k0 + lane;
      const bool f = k >= covered && co[order[k]] != 0;
      const unsigned long long m = __ballot(f);
      nz += uint32_t(__popcll(m));
      if (m) last = k0 + 64 - uint32_t(__clzll(m));  // last non-zero scan position + 1
    }
    if (lane == 0) {
      P.info[size_t(g) * kTokPerGroup + e] = nz | last << 16;
      l_cnt[e] = 1 + (nz ? last - covered : 0u);
    }
    const uint32_t cx = c_covered_x[st], cy = c_covered_y[st];
    if (lane < cx * cy) P.nzmap[(size_t(g) * 3 + c) * 1024 + (lby + lane / cx) * 32 + lbx + lane % cx] = uint8_t((nz + covered - 1) >> log2c);
  }
  __syncthreads();
  // ---- exclusive scan of the token counts in stream order (twelve per thread)
  const uint32_t ne = nblk * 3;
  uint32_t sum = 0;
  if (scanner) {
    for (uint32_t j = 0; j < 12; j++) {
      const uint32_t e = tid * 12 + j;
      sum += e < ne ? l_cnt[e] : 0u;
    }
    l_scan[tid] = sum;
  }
  __syncthreads();
  for (uint32_t d = 1; d < 256; d <<= 1) {
    const uint32_t v = (scanner && tid >= d) ? l_scan[tid - d] : 0u;
    __syncthreads();
    if (scanner) l_scan[tid] += v;
    __syncthreads();
  }
  if (scanner) {
    uint32_t o = l_scan[tid] - sum;
    for (uint32_t j = 0; j < 12; j++) {
      const uint32_t e = tid * 12 + j;
      if (e < ne) {
        P.off[size_t(g) * kTokPerGroup + e] = o;
        o += l_cnt[e];
      }
    }
  }
  if (tid == 255) P.total[g] = l_scan[255];
}

__global__ __launch_bounds__(kTokThreads) void k_enc_tok_emit(EncTok P) {
  const uint32_t g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t bx0 = (g % P.xg) * 32, by0 = (g / P.xg) * 32;
  const uint32_t gw = min(32u, P.xb - bx0);
  const uint32_t nblk = P.nblk[g];
  const uint32_t hist_off = (g % P.num_hist) * P.nctx;
  uint2* const out = P.tokens + P.base[g];
  for (uint32_t e = wave; e < nblk * 3; e += kTokWaves) {
    const uint32_t b = e / 3, ci = e % 3, c = ci == 0 ? 1u : (ci == 1 ? 0u : 2u);
    const uint32_t i = P.blk[size_t(g) * 1024 + b], lbx = i % gw, lby = i / gw;
    const size_t cell = size_t(by0 + lby) * P.xb + bx0 + lbx;
    const uint32_t st = P.acs[cell] >> 1, log2c = c_log2_covered[st], covered = 1u << log2c, size = covered * 64;
    const uint32_t ord = c_strategy_order[st];
    const uint16_t* order = P.orders + P.order_offset[ord];
    const int32_t* co = P.coeffs + (size_t(g) * 3 + c) * 65536 + P.coef_off[cell];
    const uint32_t inf = P.info[size_t(g) * kTokPerGroup + e], nz = inf & 0xFFFFu, last = inf >> 16;
    const uint32_t o = P.off[size_t(g) * kTokPerGroup + e];
    const uint32_t bc = P.ctx_map[(c < 2 ? c ^ 1 : 2) * 13 + ord];
    if (lane == 0) {
      // entropy_coder.h:25-35 PredictFromTopAndLeft over the map of counts (every cell read here belongs to a varblock
      // that comes earlier in raster order), ac_context.h:131-143 NonZeroContext
      const uint8_t* m = P.nzmap + (size_t(g) * 3 + c) * 1024;
      uint32_t pred;
      if (lbx == 0) pred = lby ? m[(lby - 1) * 32] : 32u;
      else if (lby == 0) pred = m[lbx - 1];
      else pred = (uint32_t(m[(lby - 1) * 32 + lbx]) + m[lby * 32 + lbx - 1] + 1) >> 1;
      uint32_t nzb = pred >= 64 ? 64 : pred;
      nzb = nzb < 8 ? nzb : 4 + nzb / 2;
      out[o] = make_uint2(hist_off + nzb * P.num_ctxs + bc, nz);
    }
    if (!nz) continue;
    // ac_context.h:63-83 ZeroDensityContext for scan positions covered .. last - 1: non-zeros left = count - non-zeros before k
    const uint32_t hoff = hist_off + P.num_ctxs * 37 + 458 * bc;
    uint32_t before = 0;                            // non-zeros at scan positions [covered, k0)
    uint32_t carry = nz > size / 16 ? 0u : 1u;      // "previous coefficient was non-zero" for the chunk's first lane
    for (uint32_t k0 = covered & ~63u; k0 < last; k0 += 64) {
      const uint32_t k = k0 + lane;
      const bool in = k >= covered && k < last;
      const int32_t v = in ? co[order[k]] : 0;
      const unsigned long long mz = __ballot(in && v != 0);
      if (in) {
        const uint32_t left = nz - before - uint32_t(__popcll(mz & ((1ull << lane) - 1)));
        const uint32_t prev = (lane == 0 || k == covered) ? carry : uint32_t((mz >> (lane - 1)) & 1);
        const uint32_t nzl = (left + covered - 1) >> log2c;
        const uint32_t ctx = hoff + (uint32_t(c_coeff_nnz_ctx[nzl & 63]) + c_coeff_freq_ctx[(k >> log2c) & 63]) * 2 + prev;
        const uint32_t val = v >= 0 ? uint32_t(v) * 2 : uint32_t(-(v + 1)) * 2 + 1;
        out[o + 1 + (k - covered)] = make_uint2(ctx, val);
      }
      before += uint32_t(__popcll(mz));
      carry = uint32_t((mz >> 63) & 1);
    }
  }
}

}  // namespace jxlhip
#endif  // JXL_HIP_ENC_H_
