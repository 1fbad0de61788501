"""CPU tests of the oracle (oracle/): pinned against output of the REFERENCE's own encoder and against the
closed-form properties the reference's unit tests assert.  No GPU needed."""
import hashlib
import json
import os
import subprocess

import numpy as np
import pytest

from golden.make_fjxl_golden import CASES, golden_image

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _expected(name):
    img = golden_image(name)
    nc = img.shape[2]
    if nc == 1:
        return np.repeat(img, 3, -1)
    if nc == 2:
        return np.concatenate([np.repeat(img[..., :1], 3, -1), img[..., 1:]], -1)
    return img


@pytest.mark.parametrize("name", sorted(CASES))
def test_fjxl_golden_bit_exact(built, name):
    """Streams written by the reference's standalone lossless encoder (lib/jxl/enc_fast_lossless.cc) must decode
    bit-exactly: pins bit reader, field coders, headers, TOC, prefix codes, hybrid uint, LZ77, MA-tree Modular decode,
    RCT and Palette, multi-group layout (what the reference's jxl_test.cc lossless round trips assert)."""
    import jxlo
    manifest = json.load(open(os.path.join(GOLDEN, "fjxl_manifest.json")))
    img = golden_image(name)
    assert hashlib.sha256(img.tobytes()).hexdigest() == manifest[name]["pixels_sha256"], "fixture generator drifted"
    data = open(os.path.join(GOLDEN, name + ".jxl"), "rb").read()
    assert len(data) == manifest[name]["jxl_bytes"]
    out = jxlo.Decoded(data, dumps=False).rgb8
    exp = _expected(name)
    assert out.shape == exp.shape
    assert np.array_equal(out, exp)


def test_fjxl_live_when_reference_encoder_present(built, tmp_path):
    """More sizes / efforts through oracle/_ref/fjxl_enc where it exists (authoring container and GPU box)."""
    import jxlo
    enc = os.path.join(ROOT, "oracle", "_ref", "fjxl_enc")
    if not os.path.exists(enc):
        pytest.skip("oracle/_ref/fjxl_enc not built (reference sources absent)")
    rng = np.random.default_rng(7)
    for (w, h, nc, effort) in [(33, 17, 3, 1), (257, 300, 3, 2), (600, 520, 4, 5), (256, 1, 3, 2), (1, 300, 3, 2), (512, 512, 1, 3)]:
        y, x = np.mgrid[0:h, 0:w]
        img = np.stack([128 + 90 * np.sin(x / 9.0 + c) * np.cos(y / 13.0) for c in range(nc)], -1) + rng.integers(-2, 3, (h, w, nc))
        img = np.clip(img, 0, 255).astype(np.uint8)
        raw = tmp_path / "in.raw"
        img.tofile(raw)
        out = tmp_path / "o.jxl"
        subprocess.run([enc, str(raw), str(w), str(h), str(nc), "8", str(effort), str(out)], check=True)
        dec = jxlo.Decoded(out.read_bytes(), dumps=False).rgb8
        exp = np.repeat(img, 3, -1) if nc == 1 else img
        assert np.array_equal(dec, exp), (w, h, nc, effort)


def test_reference_decode_test_1x1_stream(built):
    """The only codestream embedded in the reference's tests (lib/jxl/decode_test.cc:2512-2517): a 1x1 image that
    libjxl itself encoded as a VarDCT frame with an alpha channel (Modular, default Squeeze transform). The oracle decodes
    it completely: signature, SizeHeader, ImageMetadata with an extra channel, FrameHeader, single-section TOC, DC global
    (quantiser, block context map, colour correlation, global MA tree), DC group, AC global (dequant tables, coefficient
    orders, ANS histograms), the AC group and the Squeeze-coded alpha, with the ANS final-state and section-size checks of
    every stream satisfied. The reference only asserts decoder statuses for it, so the pixel itself is not a golden value."""
    import jxlo
    data = open(os.path.join(GOLDEN, "ref_decode_test_1x1.jxl"), "rb").read()
    assert len(data) == 68
    o = jxlo.Decoded(data)
    assert o.out_size == (1, 1) and o.info["modular"] == 0 and o.info["channels"] == 4
    assert o.rgb8.shape == (1, 1, 4)
    assert o.rgb8[0, 0, 3] == 255  # opaque
    o.close()


@pytest.mark.parametrize("seed", [1, 2])
def test_dc_consistency_all_strategies(built, seed):
    """ac_strategy_test.cc:96-222 property: with no AC, the mean of every 8x8 block of the inverse transform equals the
    DC sample of that block (LowestFrequenciesFromDC consistency), for all 27 strategies."""
    import libjxl_amd as J
    import jxlo
    used = 0
    # 8..64-class strategies mixed in one stream, then each 128/256-class strategy with DCT8 as filler
    for mask in [0x1FFFFF, 0x1C0001] + [(1 << s) | 1 for s in range(21, 27)]:
        data = J.encode_random(640, 520, seed=seed, zero_ac=1, skip_dc_smoothing=1, strategy_mask=mask)
        d = jxlo.Decoded(data)
        used |= d.info["used_acs"]
        yb, xb = d.info["ysize_blocks"], d.info["xsize_blocks"]
        x = d.planes("xyb_idct")
        dc = d.buffer("dc").reshape(3, yb, xb)
        means = x.reshape(3, yb, 8, xb, 8).mean(axis=(2, 4))
        assert np.abs(means - dc).max() < 1e-6
    assert used == (1 << 27) - 1


@pytest.mark.parametrize("strategy", [0, 4, 5, 6, 7, 8, 9, 10, 11, 18, 19, 20, 21, 22, 23])
def test_dct_family_roundtrip(built, strategy):
    """ac_strategy_test.cc:29-93 / dct_test.cc:315-367 spirit: forward transform (the encoder's own float matrix DCT)
    followed by the oracle's inverse reproduces the image up to quantisation at a very small distance."""
    import libjxl_amd as J
    import jxlo
    n = 256 if strategy < 21 else 512
    y, x = np.mgrid[0:n, 0:n]
    img = np.stack([128 + 80 * np.sin(x / 31.0) * np.cos(y / 47.0), 128 + 60 * np.cos((x + y) / 53.0), 90 + 0.2 * x], -1)
    img = np.clip(img, 0, 255).astype(np.uint8)
    data = J.encode_rgb8(img, distance=0.05, strategy_mode=2, strategy_mask=(1 << strategy), gab=0, epf_iters=0, seed=strategy)
    d = jxlo.Decoded(data)
    assert d.info["used_acs"] & (1 << strategy)
    err = np.abs(d.rgb8.astype(int) - img.astype(int))
    assert err.max() <= 3, err.max()


def test_image_roundtrip_quality_and_stats(built):
    """jxl_test.cc RoundtripSmallD1-style sanity at d1.0 with the default loop filter (Gaborish + EPF1)."""
    import libjxl_amd as J
    import jxlo
    img = J.synth_image(512, 384, seed=5)
    data = J.encode_rgb8(img)
    d = jxlo.Decoded(data)
    assert d.info["epf_iters"] == 1 and d.info["gab"] == 1
    rmse = np.sqrt(((d.rgb8.astype(float) - img) ** 2).mean())
    assert rmse < 6.0
    assert 0.3 < len(data) * 8 / (512 * 384) < 6.0


def test_random_streams_decode(built):
    import libjxl_amd as J
    import jxlo
    for seed, epf in [(1, 0), (2, 1), (3, 2), (4, 3)]:
        d = jxlo.Decoded(J.encode_random(333, 270, seed=seed, epf_iters=epf))
        assert d.rgb8.shape == (270, 333, 3)
        assert np.isfinite(d.planes("rgbf")).all()


def test_oracle_rejects_corrupt_streams(built):
    import libjxl_amd as J
    import jxlo
    data = bytearray(J.encode_rgb8(J.synth_image(300, 200)))
    with pytest.raises(RuntimeError):
        jxlo.Decoded(bytes(data[: len(data) // 2]))
    bad = bytearray(data)
    for i in range(len(bad) - 400, len(bad) - 300):
        bad[i] ^= 0x5A
    with pytest.raises(RuntimeError):
        jxlo.Decoded(bytes(bad))


# ---- closed-form known answers for the inverse transforms (lib/jxl/dct_for_test.h:23-94, tolerance dct_test.cc:191-216)
DCT_STRATEGIES = {0: (8, 8), 4: (16, 16), 5: (32, 32), 6: (16, 8), 7: (8, 16), 8: (32, 8), 9: (8, 32), 10: (32, 16), 11: (16, 32),
                  18: (64, 64), 19: (64, 32), 20: (32, 64), 21: (128, 128), 22: (128, 64), 23: (64, 128), 24: (256, 256),
                  25: (256, 128), 26: (128, 256)}  # AcStrategy raw value -> (rows, columns) of pixels


def basis_stream(J, strategy, stride=7):
    """A stream whose every varblock is `strategy` with ONE non-zero Y coefficient (the test encoder's strategy_mode 3)
    and the list of (block y0, block x0, natural position) it holds, in the encoder's block numbering."""
    R, C = DCT_STRATEGIES[strategy]
    n_pos = R * C - (R // 8) * (C // 8)
    count = min(n_pos, 256 if R * C > 1024 else n_pos)
    cols = max(1, min(256 // C, int(np.ceil(np.sqrt(count * R / C)))))
    per_group_rows = 256 // R
    rows = -(-count // cols)
    rows = -(-rows // per_group_rows) * per_group_rows if rows > per_group_rows else rows
    xs, ys = cols * C, rows * R
    data = J.encode_random(xs, ys, strategy_mode=3, strategy_mask=1 << strategy, seed=stride, gab=0, epf_iters=0, skip_dc_smoothing=1)
    # block numbering: groups in raster order, blocks in raster order of their top-left corner inside each group
    blocks = []
    for gy in range(0, ys, 256):
        for gx in range(0, xs, 256):
            for y0 in range(gy, min(gy + 256, ys), R):
                for x0 in range(gx, min(gx + 256, xs), C):
                    blocks.append((y0, x0))
    lrows, lcols, cstride = min(R, C) // 8, max(R, C) // 8, max(R, C)
    naturals = [k for k in range(R * C) if not (k // cstride < lrows and k % cstride < lcols)]
    return data, [(y0, x0, naturals[(j * (stride | 1)) % n_pos]) for j, (y0, x0) in enumerate(blocks)], (R, C)


def basis_function(R, C, k):
    """The float64 definition (dct_for_test.h:23-62): coefficient k of the R x C block -> pixels. Coefficients are stored
    with the short side as rows (coeff_order_fwd.h:27-43); square and tall blocks are stored transposed."""
    if R < C:
        ky, kx = divmod(k, C)
    else:
        kx, ky = divmod(k, R)
    y = (np.arange(R) + 0.5)[:, None]
    x = (np.arange(C) + 0.5)[None, :]
    return (np.sqrt(2.0) if ky else 1.0) * np.cos(y * ky * np.pi / R) * (np.sqrt(2.0) if kx else 1.0) * np.cos(x * kx * np.pi / C)


def check_basis_planes(plane_y, blocks, shape):
    """Every block must be ONE basis function (up to the dequantisation scale): max |block / scale - basis| within the
    reference's per-basis-vector bar 1e-7 * N (dct_test.cc:211), N = the longer side, times 2 for the two passes."""
    R, C = shape
    worst = 0.0
    for y0, x0, k in blocks:
        got = plane_y[y0:y0 + R, x0:x0 + C].astype(np.float64)
        want = basis_function(R, C, k)
        scale = (got * want).sum() / (want * want).sum()
        assert scale > 0, (y0, x0, k)
        worst = max(worst, np.abs(got / scale - want).max())
    return worst


@pytest.mark.parametrize("strategy", sorted(DCT_STRATEGIES))
def test_idct_basis_functions_oracle(built, strategy):
    import jxlo
    data, blocks, shape = basis_stream(built, strategy)
    o = jxlo.Decoded(data)
    worst = check_basis_planes(o.planes("xyb_idct")[1], blocks, shape)
    o.close()
    assert worst < 2e-7 * max(shape) + 1e-6, worst


# First vectors of kExpected in lib/jxl/xorshift128plus_test.cc:60-257 (Xorshift128Plus rng(12345), successive Fill()s).
XORSHIFT_12345 = [
    [0x6E901576D477CBB1, 0xE9E53789195DA2A2, 0xB681F6DDA5E0AE99, 0x8EFD18CE21FD6896, 0xA898A80DF75CF532, 0x50CEB2C9E2DE7E32,
     0x3CA7C2FEB25C0DD0, 0xA4D0866B80B4D836],
    [0x8CD6A1E6233D3A26, 0x3D4603ADE98B112D, 0xDC427AF674019E36, 0xE28B4D230705AC53, 0x7297E9BBA88783DD, 0x34D3D23CFCD9B41A,
     0x5A223615ADBE96B8, 0xE5EB529027CFBD01],
    [0xC1894CF00DFAC6A2, 0x18EDF8AE9085E404, 0x8E936625296B4CCD, 0x31971EF3A14A899B, 0xBE87535FCE0BF26A, 0x576F7A752BC6649F,
     0xA44CBADCE0C6B937, 0x3DBA819BB17A353A],
    [0x27CE38DFCC1C5EB6, 0x920BEB5606340256, 0x3986CBC40C9AFC2C, 0xE22BCB3EEB1E191E, 0x6E1FCDD3602A8FBA, 0x052CB044E5415A29,
     0x46266646EFB9ECD7, 0x8F44914618D29335],
]


def test_noise_generator_known_answers(built):
    """The noise generator (xorshift128plus-inl.h) against the reference's own golden values."""
    import ctypes
    import jxlo
    L = jxlo.lib()
    L.jxlo_xorshift_fill.argtypes = [ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64), ctypes.c_size_t]
    out = (ctypes.c_uint64 * (8 * len(XORSHIFT_12345)))()
    L.jxlo_xorshift_fill(12345, out, len(XORSHIFT_12345))
    assert [list(out[8 * i:8 * i + 8]) for i in range(len(XORSHIFT_12345))] == XORSHIFT_12345


def test_noise_stream_statistics(built):
    """A frame with the noise flag: the synthesis is zero-mean high-pass noise whose strength follows the LUT
    (stage_noise.cc:64-260): the image keeps its mean, gains variance, and a zero LUT leaves it untouched."""
    import jxlo
    J = built
    img = np.full((160, 288, 3), 128, np.uint8)
    base = jxlo.Decoded(J.encode_rgb8(img))
    noisy = jxlo.Decoded(J.encode_rgb8(img, noise=60))
    a, b = base.planes("xyb_filtered")[1][:, :288], noisy.planes("xyb_filtered")[1][:, :288]
    assert abs(float(b.mean() - a.mean())) < 2e-3
    assert float(b.std()) > float(a.std()) + 0.005
    # neighbouring 256-wide groups use different generators: no repetition across the group boundary
    d = (b - a)
    assert np.abs(d[:, 0:32] - d[:, 256:288]).max() > 1e-3
    base.close()
    noisy.close()


def test_colour_stage_closed_forms_oracle(built):
    """XYB -> linear sRGB -> sRGB of the oracle against the definition of XYB and the reference's own closed forms
    (tests/color_kat.py: opsin_image_test.cc:28-135): pins the inverse opsin matrix, the biases, the cube and the transfer
    function against numbers no decoder produced."""
    import ctypes
    import jxlo
    import color_kat
    L = jxlo.lib()
    L.jxlo_color_kat.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]

    def convert(xyb, linear):
        xyb = np.ascontiguousarray(xyb, np.float32)
        out = np.empty((xyb.shape[1], 3), np.float32)
        L.jxlo_color_kat(xyb.ctypes.data, xyb.shape[1], 1 if linear else 0, out.ctypes.data)
        return out
    color_kat.check(convert)


@pytest.mark.parametrize("kw", [dict(gab=1, epf_iters=1), dict(gab=0, epf_iters=3), dict(gab=1, epf_iters=2), dict(gab=1, epf_iters=0),
                                dict(gab=1, epf_iters=3, distance=3.0)])
def test_loop_filters_against_a_float64_third_reading(built, kw):
    """tests/filters_f64.py restates Gaborish and the three EPF stages in float64 NumPy straight from stage_gaborish.cc /
    stage_epf.cc / loop_filter.cc; the oracle's filtered planes must agree with it on decoded frames (ragged size, every
    filter combination). No reference-decoded pixels exist here, so a reading that neither the oracle's nor
    the kernels' author-shared code takes part in is what stands in for them."""
    import filters_f64 as F
    import jxlo
    J = built
    data = J.encode_rgb8(J.synth_image(203, 117, seed=7), **kw)
    o = jxlo.Decoded(data)
    i = o.info
    assert (i["gab"], i["epf_iters"]) == (kw["gab"], kw["epf_iters"])
    sig = o.buffer("inv_sigma")
    sig = (np.zeros(i["xsize_blocks"] * i["ysize_blocks"], np.float32) if sig is None else sig).reshape(i["ysize_blocks"], -1)[:, :i["xsize_blocks"]]
    want = F.loop_filters(o.planes("xyb_idct"), sig, i["xsize"], i["ysize"], i["gab"], i["epf_iters"])
    got = o.planes("xyb_filtered")[:, :i["ysize"], :i["xsize"]]
    o.close()
    assert np.abs(got - want).max() < 2e-6


@pytest.mark.parametrize("kw", [dict(), dict(distance=3.0, epf_iters=3), dict(distance=0.5, epf_iters=2)])
def test_dc_path_against_a_float64_third_reading(built, kw):
    """Adaptive DC smoothing (compressed_dc.cc:50-52,64-198) and the EPF's 1 / sigma per block (epf.cc:39-81) restated in
    float64 NumPy (tests/filters_f64.py) from the reference's formulas: the oracle's smoothed DC image and its 1 / sigma
    plane must agree with that reading on decoded frames (the same reading the device kernels are held to on the GPU box)."""
    import filters_f64 as F
    import jxlo
    J = built
    data = J.encode_rgb8(J.synth_image(403, 277, seed=5), **kw)
    o = jxlo.Decoded(data)
    i = o.info
    yb, xb = i["ysize_blocks"], i["xsize_blocks"]
    p = o.dc_params
    raw, got = o.buffer("dc_unsmoothed").reshape(3, yb, xb), o.buffer("dc").reshape(3, yb, xb)
    want = F.dc_smoothing(raw, p["dc_step"])
    assert np.abs(raw - got).max() > 1e-4, "the stream must exercise the smoothing"
    assert np.abs(got - want).max() < 2e-6
    sig = o.buffer("inv_sigma").reshape(yb, xb)
    want = F.inv_sigma_blocks(o.buffer("acs").reshape(yb, xb), o.buffer("quant").reshape(yb, xb), o.buffer("sharpness").reshape(yb, xb),
                              p["quant_scale"], p["epf_quant_mul"], p["epf_sharp_lut"])
    o.close()
    assert np.abs(sig / want - 1.0).max() < 2e-6


def test_coded_upsampling_weights_are_parsed_and_used(built):
    """CustomTransformData with coded upsampling weights (image_metadata.cc:87-214): both front-ends read the header (the
    host parser reaches the frame behind it), the oracle's output follows the coded matrix, not the default one."""
    import jxlo
    J = built
    img = J.synth_image(203, 149, seed=8)
    plain = jxlo.Decoded(J.encode_rgb8(img, upsampling=2), dumps=False)
    J.set_custom_upsampling(1, seed=5)
    try:
        data = J.encode_rgb8(img, upsampling=2)
    finally:
        J.set_custom_upsampling(0)
    o = jxlo.Decoded(data, dumps=False)
    f = J.Frame(data)
    assert (f.info["out_xsize"], f.info["out_ysize"]) == o.out_size == (203, 149)
    f.close()
    assert (np.abs(o.rgb8.astype(int) - plain.rgb8.astype(int)) > 2).mean() > 0.01
    o.close()
    plain.close()


@pytest.mark.parametrize("xs,ys,h,v", [(17, 9, 1, 1), (16, 8, 1, 1), (33, 20, 1, 0), (20, 33, 0, 1), (1, 1, 1, 1), (2, 3, 1, 1)])
def test_chroma_upsampling_against_a_numpy_reading(built, xs, ys, h, v):
    """render_pipeline/stage_chroma_upsampling.cc:29-111 read independently: each stage turns a sample into two, 3/4 of itself
    + 1/4 of the neighbour on that side (one multiply, one fused multiply-add), horizontally first; the stage's image is the
    channel's own ceil(size / 2) samples and it mirrors about that (low_memory_render_pipeline.cc:348-355, 668-683). Odd
    sizes, a single sample, both directions and each alone."""
    import ctypes
    import jxlo
    L = jxlo.lib()
    L.jxlo_chroma_upsample_kat.argtypes = [ctypes.c_void_p] + [ctypes.c_size_t] * 4 + [ctypes.c_int] * 2
    L.jxlo_chroma_upsample_kat.restype = None
    rng = np.random.default_rng(xs * 100 + ys)
    stride, rows = xs + 7, ys + 5
    ws, hs = ((xs + 1) // 2 if h else xs), ((ys + 1) // 2 if v else ys)
    plane = np.full((rows, stride), np.nan, np.float32)
    plane[:hs, :ws] = rng.normal(size=(hs, ws)).astype(np.float32)
    src = plane[:hs, :ws].astype(np.float64)

    def up(a, axis):  # one stage along `axis`, in float64 with the float32 roundings of Mul and MulAdd made explicit
        a = np.moveaxis(a, axis, 0)
        prev, nxt = np.concatenate([a[:1], a[:-1]]), np.concatenate([a[1:], a[-1:]])  # mirrored: the edge sample itself
        cur = (a.astype(np.float32) * np.float32(0.75)).astype(np.float64)
        out = np.empty((2 * a.shape[0],) + a.shape[1:], np.float64)
        out[0::2] = (np.float32(0.25) * prev + cur).astype(np.float32)  # (exact in float64, rounded once: a fused multiply-add)
        out[1::2] = (np.float32(0.25) * nxt + cur).astype(np.float32)
        return np.moveaxis(out, 0, axis)

    want = src
    if h:
        want = up(want, 1)
    if v:
        want = up(want, 0)
    L.jxlo_chroma_upsample_kat(plane.ctypes.data, stride, xs, ys, rows, h, v)
    assert np.array_equal(plane[:ys, :xs], want[:ys, :xs].astype(np.float32))


@pytest.mark.parametrize("cs", [4, 8, 12, 1, 0b100100, 0b011011])
def test_chroma_subsampled_round_trip_on_a_smooth_image(built, cs):
    """No reference-made subsampled stream exists here, so the subsampled path is pinned by what it must reproduce: a smooth
    image coded at d0.5 with its channels subsampled (the synthetic encoder box-averages them) comes back within 46 dB in
    every mode. A half-sample phase error in the upsampling, a channel's blocks attached to the wrong varblocks or DC taken
    from the wrong grid each cost 10 dB and more on this image."""
    import jxlo
    J = built
    h, w = 277, 333
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 37.0) * np.cos(y / 29.0), 128 + 90 * np.cos(x / 23.0 + y / 41.0), 128 + 80 * np.sin(y / 31.0)], -1).clip(0, 255).astype(np.uint8)
    o = jxlo.Decoded(J.encode_rgb8(img, distance=0.5, color_transform=2, chroma_subsampling=cs, strategy_mode=0), dumps=False)
    d = o.rgb8[..., :3].astype(np.float64) - img
    o.close()
    assert 10 * np.log10(255.0 ** 2 / (d * d).mean()) > 46.0


def test_patch_blending_against_the_reference_tests_known_answers(built):
    """lib/jxl/alpha_test.cc:29-86 (BlendingWithNonPremultiplied, BlendingWithPremultiplied, Mul) holds expected values for
    PerformAlphaBlending and PerformMulBlending: the patch stage's blend above (= that function with the frame below and the
    patch on top, blending.cc:127-136, 163-165) and multiply must reproduce them, and blend below is the same call with the
    layers swapped. The NumPy reading the GPU test's oracle is checked against (tests/test_patches.py) is held to them too."""
    import ctypes
    import jxlo
    import test_patches as TP
    L = jxlo.lib()
    fp = ctypes.POINTER(ctypes.c_float)
    L.jxlo_patch_blend_kat.argtypes = [fp, ctypes.c_float, fp, ctypes.c_float] + [ctypes.c_int] * 6 + [fp]
    L.jxlo_patch_blend_kat.restype = None

    def blend(bg, bga, fg, fga, mode, clamp, ec_mode=0, ec_clamp=0, premultiplied=False, has_alpha=True):
        out = (ctypes.c_float * 4)()
        L.jxlo_patch_blend_kat((ctypes.c_float * 3)(*bg), bga, (ctypes.c_float * 3)(*fg), fga, mode, clamp, ec_mode, ec_clamp,
                               1 if premultiplied else 0, 1 if has_alpha else 0, out)
        got = np.array(list(out), np.float32)
        f32 = np.float32
        col, a = TP._blend_np(np.array(bg, f32).reshape(3, 1, 1), np.full((1, 1), bga, f32), np.array(fg, f32).reshape(3, 1, 1),
                              np.full((1, 1), fga, f32), mode, clamp, ec_mode, ec_clamp, premultiplied)
        assert np.abs(col.ravel() - got[:3]).max() <= 1e-4 * max(1.0, float(np.abs(got[:3]).max())) and abs(float(a.ravel()[0]) - got[3]) < 1e-6
        return got

    bg, bga, fg, fga = (100.0, 110.0, 120.0), 180.0 / 255, (25.0, 21.0, 23.0), 15420.0 / 65535
    got = blend(bg, bga, fg, fga, 4, 0)  # alpha_test.cc:29-43
    assert np.abs(got[:3] - [77.2, 83.0, 90.6]).max() < 0.05 and abs(got[3] - 3174.0 / 4095) < 1e-5
    got = blend(bg, bga, fg, 2.0, 4, 1)   # :44-51: alpha 2 clamped to 1: the top layer alone
    assert np.abs(got[:3] - fg).max() < 0.05 and abs(got[3] - 1.0) < 1e-5
    got = blend(bg, bga, fg, fga, 4, 0, premultiplied=True)  # :53-67
    assert np.abs(got[:3] - [101.5, 105.1, 114.8]).max() < 0.05 and abs(got[3] - 3174.0 / 4095) < 1e-5
    got = blend(bg, bga, fg, 2.0, 4, 1, premultiplied=True)  # :68-75
    assert np.abs(got[:3] - fg).max() < 0.05 and abs(got[3] - 1.0) < 1e-5
    # blend below = the same function with the layers swapped: the frame on top of the patch
    got = blend(fg, fga, bg, bga, 5, 0)
    assert np.abs(got[:3] - [77.2, 83.0, 90.6]).max() < 0.05 and abs(got[3] - 3174.0 / 4095) < 1e-5
    # :77-86 Mul: 100 * 25, and with the top value clamped to 1
    got = blend((100.0,) * 3, 1.0, (25.0,) * 3, 1.0, 3, 0)
    assert np.abs(got[:3] - 2500.0).max() < 0.05
    got = blend((100.0,) * 3, 1.0, (25.0,) * 3, 1.0, 3, 1)
    assert np.abs(got[:3] - 100.0).max() < 0.05
