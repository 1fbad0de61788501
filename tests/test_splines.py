"""Splines (lib/jxl/splines.cc, render_pipeline/stage_splines.cc). The dictionary parse is checked on a
stream libjxl itself made (tools/wasm_demo/jxl_decoder_test.js:25-31, written by jxl_from_tree: a 320x320 Modular frame
whose only content is a spline), the rendering against the closed form of the splat, and the GPU path against the
oracle on that stream and on streams of the synthetic writer (VarDCT and lossless)."""
import math
import os

import numpy as np
import pytest

import replay_util as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "golden", "ref_wasm_splines.jxl")


def _test_splines():
    return [dict(points=[(20, 30), (120, 90), (220, 40), (280, 160)], color=[[40] + [0] * 31, [300, 10] + [0] * 30, [0] * 32],
                 sigma=[12] + [0] * 31),
            dict(points=[(50, 180), (150, 120)], color=[[0] * 32, [-200] + [0] * 31, [100] + [0] * 31], sigma=[6, 2] + [0] * 30)]


def test_reference_spline_stream_decodes_with_the_oracle():
    """The dictionary ends on the ANS final state and the rest of the frame (global tree, one all-zero Modular group) parses
    behind it: the spline is the picture. It is one thick-to-thin stroke: bright, connected, and nothing else is drawn."""
    import jxlo
    o = jxlo.Decoded(open(REF, "rb").read())
    img = o.rgb8.astype(int)
    assert img.shape == (320, 320, 3) and not np.any(o.buffer("modular"))
    lit = img.sum(axis=2) > 30
    assert 0.02 < lit.mean() < 0.25 and img.max() > 100
    # connected: every lit row / column range is contiguous over the stroke's extent
    rows = np.flatnonzero(lit.any(axis=1))
    assert rows[-1] - rows[0] + 1 == len(rows)
    o.close()


def test_spline_splat_closed_form(built):
    """A straight two-point spline with constant colour and sigma: far from its ends every sample of the centre line is
    the sum over unit-spaced centres of colour * sigma / 4 * (erf((d / 2 + 2^-1.5) / sigma) - erf((d / 2 - 2^-1.5) / sigma))^2
    (splines.cc:84-113), with the dequantisation of :484-489: coefficient 0 scaled by sqrt(1/2) * channel weight, and
    B += 1.0 * Y for the default colour correlation. The oracle's FastErff / FastCosf differ from erf / cos by < 1e-3."""
    import jxlo
    J = built
    y_coef, sigma_coef = 400, 9
    J.set_splines([dict(points=[(40, 60), (260, 60)], color=[[0] * 32, [y_coef] + [0] * 31, [0] * 32], sigma=[sigma_coef] + [0] * 31)])
    try:
        data = J.encode_lossless(np.zeros((120, 300, 3), np.uint8))
    finally:
        J.set_splines(None)
    o = jxlo.Decoded(data)
    got = o.planes("rgbf")  # R, G, B planes = the X, Y, B of the spline stage on a non-XYB frame
    o.close()
    # ContinuousIDCT of {c, 0, ...} is sqrt(2) * c * cos(0) with c already scaled by sqrt(1/2): the constant c0 * weight
    y_val = y_coef * 0.075
    sigma = sigma_coef * 0.3333
    k = 2 ** -1.5
    def splat(d):
        f = math.erf((d / 2 + k) / sigma) - math.erf((d / 2 - k) / sigma)
        return 0.25 * sigma * f * f
    for dy in (0, 1, 3):
        want = sum(splat(math.hypot(dx, dy)) for dx in range(-60, 61)) * y_val
        row = got[1][60 + dy]
        assert abs(row[150] - want) < 4e-3 * want, (dy, row[150], want)
        assert abs(got[2][60 + dy][150] - want) < 4e-3 * want  # B = 0 + 1.0 * Y
        assert abs(got[0][60 + dy][150]) < 1e-6               # X = 0 + 0.0 * Y
    assert np.abs(got[1][:20]).max() == 0.0  # beyond the maximum distance nothing is touched


def test_host_builds_the_draw_cache_of_both_frame_kinds(built):
    J = built
    img = J.synth_image(300, 200, seed=5)
    J.set_splines(_test_splines(), quantization_adjustment=-2)
    try:
        vardct, lossless = J.encode_rgb8(img), J.encode_lossless(img)
    finally:
        J.set_splines(None)
    J.Frame(vardct).close()
    J.ModFrame(lossless).close()
    J.ModFrame(open(REF, "rb").read()).close()
    # an upsampled frame's splines are drawn at the frame's own resolution, before the upsampling (dec_cache.cc:198-212)
    J.set_splines(_test_splines())
    try:
        up = J.encode_rgb8(img, upsampling=2)
    finally:
        J.set_splines(None)
    J.Frame(up).close()
    # identical successive control points have no direction: refused like splines.cc:676-683
    J.set_splines([dict(points=[(10, 10), (10, 10), (50, 50)], color=[[0] * 32, [100] + [0] * 31, [0] * 32], sigma=[9] + [0] * 31)])
    try:
        bad = J.encode_rgb8(img)
    finally:
        J.set_splines(None)
    with pytest.raises(J.JxlAmdError, match="identical"):
        J.Frame(bad)


@pytest.mark.gpu
def test_reference_spline_stream_through_the_gpu(built, tmp_path):
    import jxlo
    data = open(REF, "rb").read()
    o = jxlo.Decoded(data)
    want8, wantf = o.rgb8.copy(), o.planes("rgbf").transpose(1, 2, 0).copy()
    o.close()
    rc, events, out, px = R.run(data, tmp_path, "f32", 3)
    assert rc == 0 and events[-2:] == ["FULL_IMAGE", "SUCCESS"], out
    got = np.frombuffer(px, np.float32).reshape(320, 320, 3)
    assert np.abs(got - wantf).max() < 1e-6  # same operations in the same order (fused multiply-adds on both sides)
    rc, events, out, px = R.run(data, tmp_path, "u8", 3)
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.uint8).reshape(320, 320, 3).astype(int) - want8).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["vardct", "lossless", "vardct_noise_epf2", "vardct_upsampled2", "vardct_upsampled4"])
def test_splines_on_synthetic_frames_match_the_oracle(built, tmp_path, kind):
    """(upsampled2 / 4: the frame is coded at half / a quarter of the image's size; its splines -- coordinates in frame pixels
    -- are drawn before the upsampling, dec_cache.cc:198-212, so the strokes come out two / four times as wide)"""
    import jxlo
    J = built
    img = J.synth_image(300, 200, seed=5)
    J.set_splines(_test_splines() if "upsampled" not in kind else
                  [dict(points=[(10, 15), (60, 45), (110, 20)], color=[[40] + [0] * 31, [300, 10] + [0] * 30, [0] * 32], sigma=[12] + [0] * 31)]
                  if kind.endswith("2") else [dict(points=[(5, 8), (30, 22), (60, 10)], color=[[0] * 32, [250] + [0] * 31, [0] * 32], sigma=[8] + [0] * 31)],
                  quantization_adjustment=1)
    try:
        data = (J.encode_lossless(img) if kind == "lossless" else
                J.encode_rgb8(img, **(dict(noise=60, epf_iters=2) if kind.endswith("epf2") else
                                      (dict(upsampling=int(kind[-1])) if "upsampled" in kind else {}))))
    finally:
        J.set_splines(None)
    o = jxlo.Decoded(data)
    want8, wantf = o.rgb8.copy(), o.planes("rgbf").transpose(1, 2, 0).copy()
    o.close()
    rc, events, out, px = R.run(data, tmp_path, "f32", 3)
    assert rc == 0, out
    got = np.frombuffer(px, np.float32).reshape(200, 300, 3)
    assert np.abs(got - wantf).max() < (1e-6 if kind == "lossless" else 1e-4)
    rc, events, out, px = R.run(data, tmp_path, "u8", 3, "callback")
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.uint8).reshape(200, 300, 3).astype(int) - want8).max() <= 1
