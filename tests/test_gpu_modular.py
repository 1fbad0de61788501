"""GPU tests (-m gpu) of Modular (lossless) decode, SURVEY.md §8 row f1 / BASELINE.json configs[3]. The expectations are
reference-pinned: the fjxl fixtures are output of the reference's own enc_fast_lossless.cc, and the image each was made
from is a deterministic function of its name -- so the HIP path must reproduce it BIT-EXACTLY, no oracle in between."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
MANIFEST = json.load(open(os.path.join(ROOT, "tests", "golden", "fjxl_manifest.json")))
FJXL = os.path.join(ROOT, "oracle", "_ref", "fjxl_enc")


@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_fjxl_fixture_bit_exact_on_gpu(built, name):
    import make_fjxl_golden as G
    J = built
    img = G.golden_image(name)
    data = open(os.path.join(ROOT, "tests", "golden", name + ".jxl"), "rb").read()
    out = J.decode_lossless(data, num_channels=img.shape[2])
    assert out.shape == img.shape
    assert np.array_equal(out, img), "%d samples differ" % int((out != img).sum())


def test_lossless_formats_and_channel_subsets(built):
    """u16 / f32 output of 8-bit lossless data: exact multiples (v * 257, v / 255); RGB output of an RGBA file drops alpha,
    RGBA output of an RGB file is opaque."""
    import make_fjxl_golden as G
    J = built
    name = "fjxl_520x260_rgba_e5"
    img = G.golden_image(name)
    data = open(os.path.join(ROOT, "tests", "golden", name + ".jxl"), "rb").read()
    assert np.array_equal(J.decode_lossless(data, 3), img[..., :3])
    assert np.array_equal(J.decode_lossless(data, 4, data_type=3), img.astype(np.uint16) * 257)
    f = J.decode_lossless(data, 4, data_type=0)
    assert np.abs(f - img.astype(np.float32) / 255.0).max() < 1e-6
    rgb_name = "fjxl_300x280_rgb_e2"
    out = J.decode_lossless(open(os.path.join(ROOT, "tests", "golden", rgb_name + ".jxl"), "rb").read(), 4)
    assert np.array_equal(out[..., :3], G.golden_image(rgb_name)) and (out[..., 3] == 255).all()


@pytest.mark.skipif(not os.path.exists(FJXL), reason="the reference encoder binary (oracle/_ref/fjxl_enc) is not built here")
@pytest.mark.parametrize("size,channels,effort", [((3840, 2160), 3, 2), ((1000, 700), 4, 1), ((640, 480), 1, 3), ((777, 513), 3, 0),
                                                  ((2048, 2048), 4, 5)])
def test_live_reference_encoder_output_on_gpu(built, tmp_path, size, channels, effort):
    """Fresh output of the reference's encoder (configs[3]: 3840x2160 lossless), decoded by the HIP path: bit-exact."""
    J = built
    w, h = size
    rgb = J.synth_image(w, h, seed=31 + effort)
    y, x = np.mgrid[0:h, 0:w]
    planes = [rgb[..., 1]] if channels == 1 else [rgb[..., 0], rgb[..., 1], rgb[..., 2]]
    if channels == 4:
        planes.append(((x * 3 + y * 5) % 256).astype(np.uint8))
    img = np.ascontiguousarray(np.stack(planes, -1))
    raw = os.path.join(str(tmp_path), "in.raw")
    out = os.path.join(str(tmp_path), "out.jxl")
    img.tofile(raw)
    subprocess.run([FJXL, raw, str(w), str(h), str(channels), "8", str(effort), out], check=True)
    got = J.decode_lossless(open(out, "rb").read(), channels)
    assert np.array_equal(got, img), "%d samples differ" % int((got != img).sum())


def test_lossless_through_the_decoder_api(built, tmp_path):
    """The JxlDecoder boundary hands Modular frames to the same path (plain-C replay program)."""
    import make_fjxl_golden as G
    import replay_util as R
    name = "fjxl_300x280_rgb_e2"
    data = open(os.path.join(ROOT, "tests", "golden", name + ".jxl"), "rb").read()
    rc, events, out, px = R.run(data, tmp_path, "u8", 3)
    assert rc == 0 and events[-2:] == ["FULL_IMAGE", "SUCCESS"], out
    assert np.array_equal(np.frombuffer(px, np.uint8).reshape(280, 300, 3), G.golden_image(name))
    name = "fjxl_520x260_rgba_e5"
    data = open(os.path.join(ROOT, "tests", "golden", name + ".jxl"), "rb").read()
    rc, events, out, px = R.run(R.container(data), tmp_path, "u8", 4, "chunk=5000")
    assert rc == 0, out
    assert np.array_equal(np.frombuffer(px, np.uint8).reshape(260, 520, 4), G.golden_image(name))


def _lossless_image(J, w, h, channels, seed):
    rgb = J.synth_image(w, h, seed=seed)
    y, x = np.mgrid[0:h, 0:w]
    alpha = ((x * 3 + y * 5) % 256).astype(np.uint8)
    if channels == 1:
        return np.ascontiguousarray(rgb[..., 1:2])
    if channels == 2:
        return np.ascontiguousarray(np.dstack([rgb[..., 1], alpha]))
    return np.ascontiguousarray(rgb if channels == 3 else np.dstack([rgb, alpha]))


@pytest.mark.parametrize("flags,size,channels", [
    (0, (300, 280), 3),                      # rANS, gradient predictor, one tree leaf per channel
    (16, (300, 280), 3),                     # + RCT (YCoCg)
    (16 | 4, (520, 300), 4),                 # + weighted predictor, split on its error property
    (16 | 8, (600, 400), 3),                 # + Squeeze: stream 0, DC groups and AC groups all carry channels
    (1 | 2 | 16, (300, 280), 3),             # prefix codes + LZ77 (special distances)
    (32 | 64 | 2, (280, 300), 2),            # all 14 predictors, previous-channel property, LZ77 over rANS
    (16 | 4 | 8 | 32 | 64, (700, 300), 4),   # everything at once
    (4 | 8, (64, 48), 1),                    # a single-section frame
    (16, (1, 1), 3),
])
def test_lossless_round_trip_every_feature(built, flags, size, channels):
    """Streams of the repository's lossless test encoder (rANS or prefix codes, LZ77 with special distances, MA trees over
    the channel / weighted-predictor-error / neighbourhood / previous-channel properties, all 14 predictors, RCT, Squeeze):
    lossless, so the HIP path must return the encoder's INPUT exactly -- an expectation no decoder produced."""
    J = built
    img = _lossless_image(J, size[0], size[1], channels, seed=7 + flags)
    data = J.encode_lossless(img, flags, seed=flags)
    out = J.decode_lossless(data, num_channels=channels)
    assert out.shape == img.shape
    assert np.array_equal(out, img), "%d samples differ" % int((out != img).sum())


@pytest.mark.parametrize("flags", [16 | 4 | 8, 1 | 2 | 16])
def test_corrupt_modular_streams_are_flagged_not_fatal(built, flags):
    """Damaged sample data (bytes flipped deep inside the group sections) must end as an error status of the affected
    streams - ANS final state, over-read or LZ77 overflow - never as a device fault: every table index the kernel forms
    is validated at upload or bounded by construction, every sample write stays inside its channel rectangle."""
    J = built
    img = _lossless_image(J, 600, 400, 3, seed=3)
    data = bytearray(J.encode_lossless(img, flags))
    rng = np.random.default_rng(5)
    for pos in rng.integers(len(data) // 2, len(data) - 8, 40):
        data[int(pos)] ^= 0x5A
    try:
        f = J.ModFrame(bytes(data))
    except J.JxlAmdError:
        return  # (the damage reached a header the host front-end checks)
    c = J.HipContext()
    try:
        c.upload_modular(f)
        c.run_modular()
        r, status, _ = c.modular_status()
        assert r != 0 and any(status), "40 flipped bytes went unnoticed"
        # the context stays usable
        g = J.ModFrame(J.encode_lossless(img, flags))
        c.upload_modular(g)
        c.run_modular()
        r, status, _ = c.modular_status()
        assert r == 0 and not any(status)
        assert np.array_equal(c.pixels(), img)
        g.close()
    finally:
        c.close()
        f.close()


@pytest.mark.parametrize("lanes", [4, 64])
def test_several_streams_per_wave(built, lanes, monkeypatch):
    """Small frames put one stream in a wave; sets of many frames put up to 64 (jxlhip_modular_run_batch chooses): the
    multi-lane form of the stream kernel (shared tree and symbol tables in LDS, divergent lanes) on every feature."""
    J = built
    monkeypatch.setenv("JXLHIP_MOD_LANES", str(lanes))
    for flags, size, channels in ((0, (300, 280), 3), (16 | 4 | 8 | 32 | 64, (700, 300), 4), (1 | 2 | 16, (300, 280), 3)):
        img = _lossless_image(J, size[0], size[1], channels, seed=11 + flags)
        out = J.decode_lossless(J.encode_lossless(img, flags, seed=flags), num_channels=channels)
        assert np.array_equal(out, img), (lanes, flags)


def test_4k_lossless_squeeze_ma_tree(built):
    """BASELINE.json configs[3]: 3840x2160 Modular lossless (Squeeze + MA tree), integer bit-exact; and the same frames in
    one batched launch."""
    J = built
    img = _lossless_image(J, 3840, 2160, 3, seed=177)
    flags = J.LOSSLESS_RCT | J.LOSSLESS_SQUEEZE | J.LOSSLESS_WP
    data = J.encode_lossless(img, flags)
    assert np.array_equal(J.decode_lossless(data, 3), img)
    frames = [J.ModFrame(data) for _ in range(3)]
    ctxs = [J.HipContext() for _ in range(3)]
    for c, f in zip(ctxs, frames):
        c.upload_modular(f)
    J.run_modular_batch(ctxs)
    for c in ctxs:
        r, status, _ = c.modular_status()
        assert r == 0 and not any(status)
        assert np.array_equal(c.pixels(), img)
        c.close()
    for f in frames:
        f.close()


def test_reference_wasm_cross_stream_on_gpu(built, tmp_path):
    """crossJxl of tools/wasm_demo/jxl_decoder_test.js:33-40: a libjxl-made (jxl_from_tree) 20x20 Modular frame with a
    palette and an MA tree, decoded by k_modular_streams; integer work, so exactly the independent decoder's samples.
    The reference's own test decodes it to 16-bit RGB (6 bytes per pixel, :124-126)."""
    import jxlo
    import replay_util as R
    J = built
    data = open(os.path.join(ROOT, "tests", "golden", "ref_wasm_cross.jxl"), "rb").read()
    want = jxlo.Decoded(data, dumps=False).rgb8
    got = J.decode_lossless(data, 3)
    assert got.shape == (20, 20, 3) and np.array_equal(got, want)
    assert len(np.unique(got.reshape(-1, 3), axis=0)) >= 2  # (a drawing, not a flat field)
    rc, events, out, px = R.run(data, tmp_path, "u16", 3)
    assert rc == 0, out
    assert np.array_equal(np.frombuffer(px, np.uint16).reshape(20, 20, 3), want.astype(np.uint16) * 257)


def test_reference_jni_wrapper_streams_on_gpu(built, tmp_path):
    """The two streams of the reference's Java wrapper test (DecoderTest.java:12-20) through the drop-in boundary on the
    GPU, in the pixel formats that test asks for (:47-67: RGBA_8888, RGB_888 and the two half-float forms; it asserts the
    buffer sizes): a 1024x1024 10-bit Modular image with a palette-free MA tree, and a 1x1 image with an alpha channel.
    Integer work: the samples equal the independent decoder's."""
    import jxlo
    import replay_util as R
    for name, dim in (("ref_jni_simple_1024.jxl", 1024), ("ref_jni_pixel_alpha_1x1.jxl", 1)):
        data = open(os.path.join(ROOT, "tests", "golden", name), "rb").read()
        o = jxlo.Decoded(data, dumps=False)
        want = o.rgb8.copy()
        o.close()
        for fmt, nc, bytes_per in (("u8", 4, 4), ("u8", 3, 3), ("f16", 4, 8), ("f16", 3, 6)):
            rc, events, out, px = R.run(data, tmp_path, fmt, nc)
            assert rc == 0 and events[-1] == "SUCCESS", out
            assert len(px) == dim * dim * bytes_per
            if fmt == "u8":
                got = np.frombuffer(px, np.uint8).reshape(dim, dim, nc)
                assert np.abs(got[..., :3].astype(int) - want[..., :3].astype(int)).max() <= 1  # (10-bit samples scaled to 8 bits in float)
                if nc == 4:
                    assert np.array_equal(got[..., 3], want[..., 3] if want.shape[2] == 4 else np.full((dim, dim), 255, np.uint8))
    assert len(np.unique(want)) >= 1


@pytest.mark.parametrize("case", ["plain", "alpha_squeeze_wp", "splines"])
def test_xyb_modular_frames_go_through_the_colour_stage(built, tmp_path, case):
    """An XYB Modular frame ("lossy Modular", dec_modular.cc:583-631): the stream kernel decodes the integers Y, X, B - Y,
    the output kernel scales them by the DC quantisation steps and runs the colour stage of every XYB frame (after the
    splines, where there are any)."""
    import jxlo
    import replay_util as R
    J = built
    img = J.synth_image(300, 200, seed=5)
    flags = J.MODULAR_XYB
    src = img
    if case == "alpha_squeeze_wp":
        flags |= J.LOSSLESS_SQUEEZE | J.LOSSLESS_WP
        src = np.dstack([img, img[..., 0]])
    if case == "splines":
        J.set_splines([dict(points=[(20, 30), (120, 90), (220, 40)], color=[[40] + [0] * 31, [300, 10] + [0] * 30, [0] * 32], sigma=[12] + [0] * 31)])
    try:
        data = J.encode_lossless(src, flags)
    finally:
        J.set_splines(None)
    o = jxlo.Decoded(data)
    want8, wantf = o.rgb8.copy(), o.planes("rgbf").transpose(1, 2, 0).copy()
    o.close()
    nc = src.shape[2]
    if case != "splines":  # (quantised to the DC steps: 45 dB; the test spline is a bright stroke over it)
        assert 10 * np.log10(255.0 ** 2 / np.mean((want8[..., :3].astype(float) - img) ** 2)) > 44
    rc, events, out, px = R.run(data, tmp_path, "u8", nc)
    assert rc == 0, out
    got = np.frombuffer(px, np.uint8).reshape(200, 300, nc)
    assert np.abs(got.astype(int) - want8.astype(int)).max() <= 1
    if nc == 4:
        assert np.array_equal(got[..., 3], img[..., 0])
    rc, events, out, px = R.run(data, tmp_path, "f32", 3)
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.float32).reshape(200, 300, 3) - wantf).max() < 5e-5


def np_sample_to_float(v, bits, exp_bits):
    """NumPy reading of the reference's integer -> float step for Modular colour channels, written from the text of
    dec_modular.cc:128-185 (int_to_float) and :633-690 (1 / (2^bits - 1), single precision below 23 bits, double from
    there); shares no code with the kernel (ModSampleToFloat) or the oracle."""
    v = np.asarray(v, np.int64)
    if exp_bits == 0:
        if bits < 23:
            return v.astype(np.float32) * np.float32(1.0 / ((1 << bits) - 1))
        return (v.astype(np.float64) * (1.0 / ((1 << bits) - 1))).astype(np.float32)
    f = (v & 0xFFFFFFFF).astype(np.uint64)
    if bits == 32:
        return f.astype(np.uint32).view(np.float32)
    bias = (1 << (exp_bits - 1)) - 1
    mant_bits = bits - exp_bits - 1
    mant_shift = 23 - mant_bits
    sign = ((f >> np.uint64(bits - 1)) & np.uint64(1)).astype(np.uint64) << np.uint64(31)
    f = f & np.uint64((1 << (bits - 1)) - 1)
    exp = (f >> np.uint64(mant_bits)).astype(np.int64)
    mant = (f & np.uint64((1 << mant_bits) - 1)).astype(np.int64) << mant_shift
    out = np.zeros(f.shape, np.uint64)
    special = exp == (1 << exp_bits) - 1
    sub = (exp == 0) & (mant != 0) & (exp_bits < 8)
    m, e = mant.copy(), exp.copy()
    for _ in range(24):  # "while ((mantissa & 0x800000) == 0) { mantissa <<= 1; exp--; }" then exp++
        go = sub & ((m & 0x800000) == 0)
        m = np.where(go, m << 1, m)
        e = np.where(go, e - 1, e)
    e = np.where(sub, e + 1, e)
    m = np.where(sub, m & 0x7FFFFF, m)
    normal = ((e - bias + 127).astype(np.uint64) << np.uint64(23)) | m.astype(np.uint64)
    out = np.where(special, np.uint64(0xFF << 23) | mant.astype(np.uint64), normal)
    out = np.where(f == 0, np.uint64(0), out) | sign
    return out.astype(np.uint32).view(np.float32)


def _deep_samples(kind, h, w, nc, seed):
    """(int32 samples, bits, exp_bits): smooth content plus noise so that every predictor and context sees real work."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    base = np.stack([0.5 + 0.45 * np.sin(x / (9.0 + c)) * np.cos(y / (13.0 - c)) for c in range(nc)], -1)
    base = np.clip(base + rng.normal(0, 0.01, base.shape), 0.0, 1.0)
    if kind == "f32":
        f = base.astype(np.float32)
        # zero, subnormals, values above one. (Patterns a neighbour's distance of 2^30 and more away -- infinity, the largest
        # values -- are beyond the test encoder: the top bit of its token values marks LZ77 lengths. binary32 samples are
        # copied as they are, so the narrower types below carry the special-value branches.)
        f[0, :9, 0] = [0.0, 1.0, 2.5, 1.5, 1.9990234, 0.75, 1e-30, 1.4e-45, 1e-40]
        return f.view(np.int32).copy(), 32, 8
    if kind == "f16":
        f = base.astype(np.float16)
        f[0, :8, 0] = [0.0, 1.0, 6e-8, 6.1e-5, 65504.0, np.inf, 5.96e-8, 0.333]  # binary16 subnormals normalise on the way
        return f.view(np.uint16).astype(np.int32), 16, 5
    if kind == "f24":  # 1 + 7 + 16: a custom width (image_metadata.cc allows any 2..8 exponent / 2..23 mantissa bits)
        u = base.astype(np.float32).view(np.uint32)
        s = (u >> 31) & 1
        e = ((u >> 23) & 0xFF).astype(np.int64) - 127 + 63
        m = (u >> 7) & 0xFFFF
        ok = (e > 0) & (e < 127)
        v = np.where(ok, (s.astype(np.int64) << 23) | (e << 16) | m, 0)
        v[0, :4, 0] = [0, 1, 0x7F, (127 << 16) | 5]  # two subnormals of the type and a NaN with payload
        return v.astype(np.int32), 24, 7
    bits = int(kind[1:])
    return np.round(base * ((1 << bits) - 1)).astype(np.int64).astype(np.int32), bits, 0


@pytest.mark.parametrize("kind,flags,size,nc", [
    ("f32", 0, (300, 70), 3),            # binary32 samples as they are (lossless float: what cjxl writes for PFM / EXR input)
    ("f32", 4 | 8, (520, 300), 3),       # + weighted predictor + Squeeze over the bit patterns, several groups
    ("f16", 4, (300, 280), 4),           # binary16 with an 8-bit alpha channel
    ("f16", 16 | 32, (280, 300), 1),     # grey
    ("f24", 0, (260, 40), 3),            # a custom float width
    ("u20", 16 | 4, (300, 280), 3),      # integers above 16 bits: single-precision scale
    ("u24", 16 | 8, (520, 260), 3),      # double-precision scale from 23 bits on
    ("u28", 0, (64, 48), 1),
])
def test_float_and_deep_integer_samples(built, kind, flags, size, nc):
    """Modular frames whose colour samples are floats (any width the format allows) or integers of more than 16 bits
    (dec_modular.cc:128-185,633-690; refused until round 4). Lossless: the float32 output must be, bit for bit, the NumPy
    reading of the reference's conversion applied to the ENCODER'S INPUT -- no decoder in between; the oracle's integer
    channels are checked against the same input on the way."""
    J = built
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import jxlo
    w, h = size
    v, bits, exp_bits = _deep_samples(kind, h, w, 3 if nc >= 3 else 1, seed=len(kind) + flags)
    if nc in (2, 4):
        y, x = np.mgrid[0:h, 0:w]
        v = np.dstack([v, ((x * 3 + y * 5) % 256).astype(np.int32)])
    data = J.encode_lossless_samples(v, bits, exp_bits, flags=flags, seed=flags)
    o = jxlo.Decoded(data, dumps=True)
    assert o.info["bits"] == bits
    assert np.array_equal(o.buffer("modular").reshape(nc, h, w), np.moveaxis(v, -1, 0)), "oracle integers differ from the input"
    o.close()
    got = J.decode_lossless(data, num_channels=nc, data_type=0)
    assert got.shape == (h, w, nc) and got.dtype == np.float32
    ncol = 3 if nc >= 3 else 1
    want = np_sample_to_float(v[..., :ncol], bits, exp_bits)
    assert np.array_equal(got[..., :ncol].view(np.uint32), want.view(np.uint32)), \
        "%d samples differ" % int((got[..., :ncol].view(np.uint32) != want.view(np.uint32)).sum())
    if nc in (2, 4):
        assert np.array_equal(got[..., ncol], v[..., ncol].astype(np.float32) * np.float32(1.0 / 255))
    if kind == "f16":  # the narrow type's own reading: NumPy's binary16 -> binary32 widening is exact
        assert np.array_equal(want.view(np.uint32), v[..., :ncol].astype(np.uint16).view(np.float16).astype(np.float32).view(np.uint32))
        half = J.decode_lossless(data, num_channels=nc, data_type=5)
        assert np.array_equal(half[..., :ncol].view(np.uint16), v[..., :ncol].astype(np.uint16))
    if kind == "u20":  # 16-bit output of deeper integers: rounded, as the reference's writer does (stage_write.cc)
        u16 = J.decode_lossless(data, num_channels=nc, data_type=3)
        assert np.abs(u16.astype(np.int64) - np.round(want.astype(np.float64) * 65535).astype(np.int64)).max() <= 1
