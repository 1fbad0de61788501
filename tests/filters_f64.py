"""A third reading of the loop filters, in float64 NumPy, written from the reference's formulas and from nothing in this
repository: Gaborish (lib/jxl/render_pipeline/stage_gaborish.cc:33-100, weights loop_filter.cc:28-51) and the three
edge-preserving filter stages (stage_epf.cc:47-50 Weight, :82-181 EPF0, :200-367 EPF1, :385-494 EPF2; constants
loop_filter.cc:60-92, epf.h kMinSigma). Both the oracle (oracle/jxlo_render.h) and the HIP kernels (jxl_hip_filter_fused.h)
are held to it on the planes of decoded frames: a misreading shared by those two would show here.

Inputs: the inverse-transform output (3 planes, the frame's first ysize rows and xsize columns), 1 / sigma per 8x8 block
as the decoder computed it (the sigma computation itself is not restated here), the frame's gab / epf_iters flags. The
loop-filter parameters are the codestream defaults (all_default, what the streams of the test suite carry).

Every stage reads its input mirrored about the frame (lib/jxl/image_ops.h:184-196: -1 -> 0, -2 -> 1, size -> size - 1)."""
import numpy as np

GAB_W1 = 1.1 * 0.104699568
GAB_W2 = 1.1 * 0.055680538
CHANNEL_SCALE = (40.0, 5.0, 3.5)
PASS0_SIGMA_SCALE = 0.9
PASS2_SIGMA_SCALE = 6.5
BORDER_SAD_MUL = 2.0 / 3.0
K_MIN_SIGMA = -3.90524291751269967465540850526868  # epf.h: 1 / sigma below this leaves the pixel alone


def _mirror_pad(p, r):
    """planes [3, H, W] -> [3, H + 2r, W + 2r] with the reference's mirroring (edge sample repeated)."""
    return np.pad(p, ((0, 0), (r, r), (r, r)), mode="symmetric")


def _shift(pp, r, dy, dx, h, w):
    return pp[:, r + dy:r + dy + h, r + dx:r + dx + w]


def gaborish(p):
    p = np.asarray(p, np.float64)
    _, h, w = p.shape
    pp = _mirror_pad(p, 1)
    div = 1.0 + 4.0 * (GAB_W1 + GAB_W2)
    w0, w1, w2 = 1.0 / div, GAB_W1 / div, GAB_W2 / div
    s1 = sum(_shift(pp, 1, dy, dx, h, w) for dy, dx in ((0, -1), (0, 1), (-1, 0), (1, 0)))
    s2 = sum(_shift(pp, 1, dy, dx, h, w) for dy, dx in ((-1, -1), (-1, 1), (1, -1), (1, 1)))
    return p * w0 + s1 * w1 + s2 * w2


def _sigma_per_pixel(inv_sigma_blocks, h, w, scale):
    """1 / sigma of every pixel's block times the stage's multiplier, the border multiplier on the block's outer ring."""
    yy, xx = np.mgrid[0:h, 0:w]
    s = inv_sigma_blocks[yy >> 3, xx >> 3].astype(np.float64)
    border = ((xx & 7) == 0) | ((xx & 7) == 7) | ((yy & 7) == 0) | ((yy & 7) == 7)
    return s, s * np.where(border, scale * BORDER_SAD_MUL, scale)


_PLUS = ((0, 0), (-1, 0), (0, -1), (1, 0), (0, 1))
_NB12 = ((-2, 0), (-1, -1), (-1, 0), (-1, 1), (0, -2), (0, -1), (0, 1), (0, 2), (1, -1), (1, 0), (1, 1), (2, 0))
_NB4 = ((-1, 0), (0, -1), (0, 1), (1, 0))


def _epf_stage(p, inv_sigma_blocks, stage):
    p = np.asarray(p, np.float64)
    _, h, w = p.shape
    scale = 1.65 * (PASS0_SIGMA_SCALE if stage == 0 else (PASS2_SIGMA_SCALE if stage == 2 else 1.0))
    raw, inv = _sigma_per_pixel(inv_sigma_blocks, h, w, scale)
    r = 3 if stage == 0 else (2 if stage == 1 else 1)
    pp = _mirror_pad(p, r)
    cs = np.asarray(CHANNEL_SCALE, np.float64)[:, None, None]
    wsum = np.ones((h, w))
    acc = p.copy()
    for dy, dx in (_NB12 if stage == 0 else _NB4):
        if stage == 2:  # single-sample difference
            sad = (np.abs(_shift(pp, r, dy, dx, h, w) - p) * cs).sum(axis=0)
        else:  # plus-shaped sum of absolute differences between the two neighbourhoods
            sad = np.zeros((h, w))
            for oy, ox in _PLUS:
                sad += (np.abs(_shift(pp, r, oy, ox, h, w) - _shift(pp, r, dy + oy, dx + ox, h, w)) * cs).sum(axis=0)
        weight = np.maximum(sad * inv + 1.0, 0.0)
        wsum += weight
        acc += weight[None] * _shift(pp, r, dy, dx, h, w)
    out = acc / wsum[None]
    return np.where((raw < K_MIN_SIGMA)[None], p, out)


def loop_filters(xyb_idct, inv_sigma_blocks, xsize, ysize, gab, epf_iters):
    """The filter stages of dec_cache.cc:151-170 in order: Gaborish, then EPF0 (3 iterations only), EPF1 (>= 1), EPF2 (>= 2)."""
    p = np.asarray(xyb_idct, np.float64)[:, :ysize, :xsize]
    if gab:
        p = gaborish(p)
    if epf_iters >= 3:
        p = _epf_stage(p, inv_sigma_blocks, 0)
    if epf_iters >= 1:
        p = _epf_stage(p, inv_sigma_blocks, 1)
    if epf_iters >= 2:
        p = _epf_stage(p, inv_sigma_blocks, 2)
    return p


# ---- the DC path's two stencils, same rules: float64, written from the reference's formulas only.
def dc_smoothing(dc, step):
    """AdaptiveDCSmoothing, lib/jxl/compressed_dc.cc:50-52 (weights) and :64-128: per block a 3x3 weighted mean of each of
    the three DC planes; gap = max over the channels of |dc - mean| in units of the channel's DC quantisation step, at
    least 0.5; the result moves from dc towards the mean by max(0, 3 - 4 * gap). Border blocks keep their value (:141-175);
    images of at most 2 blocks in either direction are left alone (:134)."""
    dc = np.asarray(dc, np.float64)
    _, h, w = dc.shape
    if h <= 2 or w <= 2:
        return dc.copy()
    w1, w2 = float(np.float32(0.20345139757231578)), float(np.float32(0.0334829185968739))
    w0 = float(np.float32(1.0) - np.float32(4.0) * (np.float32(w1) + np.float32(w2)))
    c = dc[:, 1:-1, 1:-1]
    side = dc[:, 1:-1, :-2] + dc[:, 1:-1, 2:] + dc[:, :-2, 1:-1] + dc[:, 2:, 1:-1]
    corner = dc[:, :-2, :-2] + dc[:, :-2, 2:] + dc[:, 2:, :-2] + dc[:, 2:, 2:]
    sm = corner * w2 + side * w1 + c * w0
    gap = np.maximum(0.5, (np.abs(c - sm) / np.asarray(step, np.float64)[:, None, None]).max(axis=0))
    factor = np.maximum(0.0, 3.0 - 4.0 * gap)
    out = dc.copy()
    out[:, 1:-1, 1:-1] = (sm - c) * factor + c
    return out


def inv_sigma_blocks(acs, quant, sharpness, quant_scale, epf_quant_mul, sharp_lut):
    """ComputeSigma, lib/jxl/epf.cc:39-81: for every varblock (acs = strategy << 1 | first block, quant = raw quant field
    at first blocks) sigma = epf_quant_mul / (quant_scale * quant * kInvSigmaNum) * sharp_lut[sharpness of the block],
    at most -1e-4; stored as 1 / sigma for each 8x8 block the varblock covers (ac_strategy.h:130-167)."""
    cx = [1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32]
    cy = [1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16]
    k_inv_sigma_num = -1.1715728752538099024
    h, w = acs.shape
    out = np.zeros((h, w), np.float64)
    lut = np.asarray(sharp_lut, np.float64)
    for by, bx in zip(*np.nonzero(acs & 1)):
        st = int(acs[by, bx]) >> 1
        sq = epf_quant_mul / (quant_scale * float(quant[by, bx]) * k_inv_sigma_num)
        sl = (slice(by, by + cy[st]), slice(bx, bx + cx[st]))
        out[sl] = 1.0 / np.minimum(-1e-4, sq * lut[sharpness[sl]])
    return out
