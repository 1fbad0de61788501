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

// Transform selection and quant field of one 64x64 tile (8x8 blocks): every candidate is aligned to its own size, so the
// greedy raster scan of the CPU writer never looks outside the tile it is in.
__global__ void k_enc_select(EncFwd P) {
#pragma clang fp contract(off)
  const uint32_t tiles_x = (P.xb + 7) / 8, tiles_y = (P.yb + 7) / 8;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= tiles_x * tiles_y) return;
  const uint32_t bx0 = (t % tiles_x) * 8, by0 = (t / tiles_x) * 8;
  const uint32_t w = min(8u, P.xb - bx0), h = min(8u, P.yb - by0);
  uint8_t acs[64];
  float act[64];
  for (uint32_t i = 0; i < 64; i++) {
    acs[i] = 0xFF;
    const uint32_t x = i & 7, y = i >> 3;
    act[i] = (x < w && y < h) ? P.act[size_t(by0 + y) * P.xb + bx0 + x] : 0.0f;
  }
  const float T64 = 0.004f * P.distance, T32 = 0.008f * P.distance, T16 = 0.016f * P.distance, TR = 0.011f * P.distance;
  auto region_max = [&](uint32_t bx, uint32_t by, uint32_t cx, uint32_t cy) {
    float m = 0.0f;
    for (uint32_t y = 0; y < cy; y++)
      for (uint32_t x = 0; x < cx; x++) m = fmaxf(m, act[(by + y) * 8 + bx + x]);
    return m;
  };
  auto ok = [&](uint32_t bx, uint32_t by, int cand, float thr) {
    const uint32_t cx = c_covered_x[cand], cy = c_covered_y[cand];
    if (bx % cx || by % cy || bx + cx > w || by + cy > h) return false;
    for (uint32_t y = 0; y < cy; y++)
      for (uint32_t x = 0; x < cx; x++)
        if (acs[(by + y) * 8 + bx + x] != 0xFF) return false;
    return region_max(bx, by, cx, cy) < thr;
  };
  for (uint32_t by = 0; by < h; by++)
    for (uint32_t bx = 0; bx < w; bx++) {
      if (acs[by * 8 + bx] != 0xFF) continue;
      int st = 0;
      if (P.strategy_mode == 1) {
        if (ok(bx, by, 18, T64)) st = 18;
        else if (ok(bx, by, 20, T64 * 1.3f)) st = 20;
        else if (ok(bx, by, 19, T64 * 1.3f)) st = 19;
        else if (ok(bx, by, 5, T32)) st = 5;
        else if (ok(bx, by, 11, TR)) st = 11;
        else if (ok(bx, by, 10, TR)) st = 10;
        else if (ok(bx, by, 4, T16)) st = 4;
        else if (ok(bx, by, 9, T16 * 0.8f)) st = 9;
        else if (ok(bx, by, 8, T16 * 0.8f)) st = 8;
        else if (ok(bx, by, 7, T16 * 1.5f)) st = 7;
        else if (ok(bx, by, 6, T16 * 1.5f)) st = 6;
      }
      const uint32_t cx = c_covered_x[st], cy = c_covered_y[st];
      for (uint32_t y = 0; y < cy; y++)
        for (uint32_t x = 0; x < cx; x++) acs[(by + y) * 8 + bx + x] = uint8_t((st << 1) | ((x | y) == 0));
    }
  for (uint32_t by = 0; by < h; by++)
    for (uint32_t bx = 0; bx < w; bx++) {
      const uint8_t a = acs[by * 8 + bx];
      const size_t bi = size_t(by0 + by) * P.xb + bx0 + bx;
      P.acs[bi] = a;
      int32_t q = 0;
      if (a & 1) {
        const int st = a >> 1;
        const float m = region_max(bx, by, c_covered_x[st], c_covered_y[st]);
        float mul = 1.35f - 0.12f * log2f(1.0f + m * 400.0f);
        mul = fmaxf(0.8f, fminf(1.4f, mul));
        q = max(1, min(256, int(P.quant_ac * mul * P.inv_gs + 0.5f)));
      }
      P.qf[bi] = q;
    }
}

// Coefficient offsets of the first blocks of one 256x256 group, in the raster order the bitstream uses.
__global__ void k_enc_offsets(EncFwd P) {
  const uint32_t g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= P.xg * P.yg) return;
  const uint32_t bx0 = (g % P.xg) * 32, by0 = (g / P.xg) * 32;
  const uint32_t gw = min(32u, P.xb - bx0), gh = min(32u, P.yb - by0);
  uint32_t offset = 0;
  for (uint32_t by = 0; by < gh; by++)
    for (uint32_t bx = 0; bx < gw; bx++) {
      const size_t bi = size_t(by0 + by) * P.xb + bx0 + bx;
      const uint8_t a = P.acs[bi];
      if (!(a & 1)) continue;
      P.coef_off[bi] = offset;
      offset += 64u << c_log2_covered[a >> 1];
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

}  // namespace jxlhip
#endif  // JXL_HIP_ENC_H_
