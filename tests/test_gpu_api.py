"""GPU tests (-m gpu) of the drop-in boundary and of the stream features libjxl's other efforts use: every output format
of JxlPixelFormat, callbacks, containers and chunked input through the plain-C replay of the reference's DecodeImageJXL
sequence; the reference's own VarDCT + alpha stream through the HIP kernels; prefix-coded and LZ77 AC streams."""
import os

import numpy as np
import pytest

import replay_util as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _oracle_float(jxlo, data):
    o = jxlo.Decoded(data)
    f = o.planes("rgbf").transpose(1, 2, 0).copy()  # HxWx3 float, before the 8-bit conversion
    rgb8 = o.rgb8.copy()
    o.close()
    return f, rgb8


def test_float_output_matches_oracle_float(built, tmp_path):
    """JXL_TYPE_FLOAT (what benchmark_xl and djxl --disable_output ask for: benchmark_codec_jxl.cc:313,
    djxl_main.cc:599-600) is written by the row-streaming filter kernel itself (stage_write.cc:334-370)."""
    import jxlo
    J = built
    data = J.encode_rgb8(J.synth_image(333, 222, seed=5))
    ref_f, _ = _oracle_float(jxlo, data)
    rc, events, out, px = R.run(data, tmp_path, "f32", 3)
    assert rc == 0 and events[-2:] == ["FULL_IMAGE", "SUCCESS"], out
    got = np.frombuffer(px, np.float32).reshape(222, 333, 3)
    assert np.abs(got - ref_f).max() < 3e-5  # float samples in [0, 1]: the bar of the float planes, through the sRGB curve
    # other filter configurations take the generic writer (k_color_out): same bar
    for kw in (dict(epf_iters=2), dict(gab=0, epf_iters=0), dict(upsampling=2)):
        data = J.encode_rgb8(J.synth_image(200, 150, seed=7), **kw)
        ref_f, _ = _oracle_float(jxlo, data)
        rc, events, out, px = R.run(data, tmp_path, "f32", 3)
        assert rc == 0, out
        got = np.frombuffer(px, np.float32).reshape(150, 200, 3)
        assert np.abs(got - ref_f).max() < 3e-5, kw


@pytest.mark.parametrize("fmt,channels,extra", [("u8", 4, ()), ("u8", 3, ("callback",)), ("u16", 3, ()), ("u16", 4, ("mt",)),
                                                ("f16", 3, ()), ("f32", 4, ("callback",)), ("u8", 3, ("chunk=1000",))])
def test_every_pixel_format_and_delivery(built, tmp_path, fmt, channels, extra):
    """Formats follow stage_write.cc: u8 = round(clamp(v * 255 + dither)), u16 = round(clamp(v * 65535)) (no dither),
    f16 / f32 = the float sample; alpha of an image without alpha is opaque; buffer, callback and multithreaded
    callback deliver the same pixels."""
    import jxlo
    J = built
    data = J.encode_rgb8(J.synth_image(257, 131, seed=11))
    ref_f, ref_8 = _oracle_float(jxlo, data)
    rc, events, out, px = R.run(R.container(data) if "chunk=1000" in extra else data, tmp_path, fmt, channels, *extra)
    assert rc == 0, out
    dt = {"u8": np.uint8, "u16": np.uint16, "f16": np.float16, "f32": np.float32}[fmt]
    got = np.frombuffer(px, dt).reshape(131, 257, channels)
    if fmt == "u8":
        assert np.abs(got[..., :3].astype(int) - ref_8.astype(int)).max() <= 1
    elif fmt == "u16":
        want = np.rint(np.clip(ref_f * 65535.0, 0, 65535))
        assert np.abs(got[..., :3].astype(np.int64) - want).max() <= 3  # 3e-5 of float difference
    else:
        assert np.abs(got[..., :3].astype(np.float32) - ref_f).max() < (2e-3 if fmt == "f16" else 5e-5)  # (planes within 2e-5, then the sRGB curve)
    if channels == 4:
        opaque = {"u8": 255, "u16": 65535, "f16": 1.0, "f32": 1.0}[fmt]
        assert (got[..., 3] == opaque).all()
    if "mt" in extra:
        assert "mt init=1 destroy=1" in out


def test_noise_frame_through_the_decoder_api(built, tmp_path):
    """A frame with noise synthesis (frame flag kNoise: dec_noise.cc, stage_noise.cc) through JxlDecoder: the generator is
    deterministic (seeded by the frame index and the group origins), so the pixels must equal the oracle's."""
    import jxlo
    J = built
    data = J.encode_rgb8(J.synth_image(300, 270, seed=9), noise=50)
    plain = J.encode_rgb8(J.synth_image(300, 270, seed=9))
    ref_f, ref_8 = _oracle_float(jxlo, data)
    _, plain_8 = _oracle_float(jxlo, plain)
    assert np.abs(ref_8.astype(int) - plain_8.astype(int)).mean() > 1.0  # (the stream really carries noise)
    rc, events, out, px = R.run(data, tmp_path, "u8", 3)
    assert rc == 0 and events[-2:] == ["FULL_IMAGE", "SUCCESS"], out
    got = np.frombuffer(px, np.uint8).reshape(270, 300, 3)
    assert np.abs(got.astype(int) - ref_8.astype(int)).max() <= 1
    rc, events, out, px = R.run(data, tmp_path, "f32", 3)
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.float32).reshape(270, 300, 3) - ref_f).max() < 5e-5


@pytest.mark.parametrize("cs", [4, 8, 12])
def test_chroma_subsampled_ycbcr_frames_through_the_decoder_api(built, tmp_path, cs):
    """What a recompressed 4:2:0 / 4:2:2 / 4:4:0 JPEG holds (YCbCr frame, subsampled chroma, RAW quantisation table, no loop
    filter) through JxlDecoder, at sizes that are not whole MCUs: pixels equal to the oracle's."""
    import jxlo
    J = built
    data = J.encode_rgb8(J.synth_image(333, 277, seed=9), color_transform=2, chroma_subsampling=cs, strategy_mode=0, raw_quant=1, gab=0,
                         epf_iters=0)
    ref_f, ref_8 = _oracle_float(jxlo, data)
    rc, events, out, px = R.run(data, tmp_path, "u8", 3)
    assert rc == 0 and events[-2:] == ["FULL_IMAGE", "SUCCESS"], out
    got = np.frombuffer(px, np.uint8).reshape(277, 333, 3)
    assert np.abs(got.astype(int) - ref_8.astype(int)).max() <= 1
    rc, events, out, px = R.run(data, tmp_path, "f32", 3)
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.float32).reshape(277, 333, 3) - ref_f).max() < 5e-5


def test_images_with_an_embedded_icc_profile(built, tmp_path):
    """Streams whose original colours are described by an ICC profile (want_icc): the profile - here the reference's own
    test vector, lib/jxl/icc_codec_test.cc:52-211 - comes back through JxlDecoderGetICCProfileSize /
    JxlDecoderGetColorAsICCProfile(ORIGINAL) byte for byte, the structured profile is reported absent for it
    (decode.h:728-730), and the pixels are those of the same image without a profile: XYB frames are rendered to sRGB,
    Modular frames keep their samples (whose profile, target DATA, is then the embedded one)."""
    J = built
    g = os.path.join(ROOT, "tests", "golden")
    coded = open(os.path.join(g, "ref_icc_test_profile.enc"), "rb").read()
    want = open(os.path.join(g, "ref_icc_test_profile.icc"), "rb").read()
    h = 1469598103934665603
    for b in want:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    img = J.synth_image(300, 200, seed=6)
    plain, plain_l = J.encode_rgb8(img), J.encode_lossless(img, J.LOSSLESS_RCT)
    J.set_embedded_icc(coded)
    try:
        with_icc, with_icc_l = J.encode_rgb8(img), J.encode_lossless(img, J.LOSSLESS_RCT)
    finally:
        J.set_embedded_icc(None)
    _, _, _, px0 = R.run(plain, tmp_path, "u8", 3)
    rc, events, out, px = R.run(with_icc, tmp_path, "u8", 3)
    assert rc == 0 and events[-2:] == ["FULL_IMAGE", "SUCCESS"], out
    assert "original icc size=%d fnv1a=%016x encoded_profile=0" % (len(want), h) in out, out
    assert "COLOR_ENCODING tf=13 icc=0" in out  # the pixels: sRGB
    assert px == px0
    rc, events, out, px = R.run(with_icc_l, tmp_path, "u8", 3)
    assert rc == 0 and events[-2:] == ["FULL_IMAGE", "SUCCESS"], out
    assert "original icc size=%d fnv1a=%016x" % (len(want), h) in out
    assert "COLOR_ENCODING tf=-1 icc=%d" % len(want) in out, out  # the pixels are in the embedded profile's space
    assert np.array_equal(np.frombuffer(px, np.uint8).reshape(200, 300, 3), img)


def test_linear_output(built, tmp_path):
    import jxlo
    J = built
    data = J.encode_rgb8(J.synth_image(200, 100, seed=2))
    ref_f, _ = _oracle_float(jxlo, data)
    rc, events, out, px = R.run(data, tmp_path, "f32", 3, "linear")
    assert rc == 0 and "COLOR_ENCODING tf=8" in out, out
    got = np.frombuffer(px, np.float32).reshape(100, 200, 3)
    a = np.abs(got)  # (the transfer function is odd: out-of-gamut negatives keep their sign, stage_from_linear.cc:42-54)
    srgb = np.sign(got) * np.where(a <= 0.0031308, a * 12.92, 1.055 * np.power(np.maximum(a, 1e-12), 1 / 2.4) - 0.055)
    assert np.abs(srgb - ref_f).max() < 2e-3  # the same image, one transfer function apart
    assert np.abs(got - ref_f).max() > 0.05


def test_reference_vardct_alpha_stream_through_the_gpu(built, tmp_path):
    """The one libjxl-made VarDCT codestream in the reference tree (lib/jxl/decode_test.cc:2512-2517: 1x1, VarDCT colour,
    prefix-coded AC, a Squeeze-coded alpha channel) through the product: host front-end, HIP kernels, RGBA out. The
    reference asserts only statuses for it; here the GPU pixel must equal the independent decoder's."""
    import jxlo
    J = built
    data = open(os.path.join(ROOT, "tests", "golden", "ref_decode_test_1x1.jxl"), "rb").read()
    o = jxlo.Decoded(data)
    want = o.rgb8.copy()
    assert want.shape == (1, 1, 4)
    rc, events, out, px = R.run(data, tmp_path, "u8", 4)
    assert rc == 0 and events == ["BASIC_INFO", "COLOR_ENCODING", "FRAME", "NEED_IMAGE_OUT_BUFFER", "FULL_IMAGE", "SUCCESS"], out
    got = np.frombuffer(px, np.uint8).reshape(1, 1, 4)
    assert np.abs(got.astype(int) - want.astype(int)).max() <= 1
    assert got[0, 0, 3] == want[0, 0, 3]  # alpha is integer work: exact
    # stage by stage, like every other stream: coefficients bit-exact, planes within the float bar
    f = J.Frame(data)
    c = J.HipContext()
    c.upload(f)
    c.run_entropy()
    c.sync()
    r, flags = c.errors()
    assert r == 0
    assert np.array_equal(c.download("coeffs").astype(np.int32)[0, :, :64], o.planes("coeffs")[0, :, :64])
    c.run_transform()
    c.sync()
    assert np.abs(c.download("xyb_idct") - o.planes("xyb_idct")).max() < 2e-5
    c.close()
    f.close()
    o.close()


@pytest.mark.parametrize("mode", [1, 2, 3])
def test_prefix_coded_and_lz77_ac_streams(built, mode):
    """AC streams as libjxl's fastest efforts (prefix codes, dec_huffman.h:28-41) and slowest efforts (LZ77,
    dec_ans.h:288-353) write them: prefix codes alone on the lane kernel's prefix form, anything with LZ77 on
    k_entropy_generic; coefficients bit-exact, pixels within the usual bar."""
    import jxlo
    from test_gpu_parity import _compare
    J = built
    data = J.encode_rgb8(J.synth_image(520, 300, seed=3), ac_code_mode=mode)
    base = J.encode_rgb8(J.synth_image(520, 300, seed=3))
    rgb = _compare(J, jxlo, data)
    assert np.abs(rgb.astype(int) - J.decode_rgb8(base).astype(int)).max() <= 1  # same image as the rANS stream
    _compare(J, jxlo, J.encode_random(300, 260, seed=5 + mode, ac_code_mode=mode))
    _compare(J, jxlo, J.encode_rgb8(J.synth_image(300, 200, seed=4), ac_code_mode=mode, num_passes=2, num_histograms=3))


def test_prefix_streams_on_both_kernels_and_in_a_batch(built, monkeypatch):
    """Prefix-coded frames: the lane kernel's prefix form against the wave-per-section kernel (JXLHIP_NO_LANE_PREFIX=1),
    and a set of frames (one launch) against single decodes."""
    J = built
    imgs = [J.synth_image(600, 420, seed=20 + i) for i in range(3)]
    datas = [J.encode_rgb8(im, ac_code_mode=1, distance=1.0 + 0.5 * i) for i, im in enumerate(imgs)]
    single = [J.decode_rgb8(d) for d in datas]
    monkeypatch.setenv("JXLHIP_NO_LANE_PREFIX", "1")
    generic = [J.decode_rgb8(d) for d in datas]
    monkeypatch.delenv("JXLHIP_NO_LANE_PREFIX")
    for a, b in zip(single, generic):
        assert np.array_equal(a, b)
    frames = [J.Frame(d) for d in datas]
    ctxs = [J.HipContext() for _ in datas]
    for c, f in zip(ctxs, frames):
        c.upload(f)
    J.run_entropy_batch(ctxs)
    J.run_transform_batch(ctxs)
    J.run_filter_color_batch(ctxs)
    for c, want in zip(ctxs, single):
        assert np.array_equal(c.rgb8(), want)
        r, flags = c.errors()
        assert r == 0 and not any(flags)
    for c, f in zip(ctxs, frames):
        c.close()
        f.close()


def test_image_with_alpha_in_group_sections(built, tmp_path):
    """An alpha channel larger than a group continues behind the coefficients of every AC group section
    (dec_frame.cc:511-542): the entropy stage reports where each coefficient stream ended, the host decodes from there."""
    import jxlo
    J = built
    rgb = J.synth_image(600, 300, seed=8)
    alpha = ((np.mgrid[0:300, 0:600][1] * 255) // 599).astype(np.uint8)
    data = J.encode_rgba8(np.dstack([rgb, alpha]))
    o = jxlo.Decoded(data, dumps=False)
    want = o.rgb8.copy()
    assert want.shape == (300, 600, 4) and np.array_equal(want[..., 3], alpha)
    rc, events, out, px = R.run(data, tmp_path, "u8", 4)
    assert rc == 0, out
    got = np.frombuffer(px, np.uint8).reshape(300, 600, 4)
    assert np.array_equal(got[..., 3], alpha)
    assert np.abs(got[..., :3].astype(int) - want[..., :3].astype(int)).max() <= 1


@pytest.mark.parametrize("lossless", [False, True])
@pytest.mark.parametrize("nlayers", [1, 2])
def test_unpremultiplied_output_of_associated_alpha(built, tmp_path, lossless, nlayers):
    """JxlDecoderSetUnpremultiplyAlpha (decode.h; stage_write.cc:359-361,460-482): the colour samples of an image whose
    alpha is associated leave divided by max(alpha, 2^-26), after the transfer function, in the outputs that carry alpha.
    A single VarDCT / Modular frame (the generic pixel writer, the Modular writer) and a two-layer still (the canvas
    writer); the expectation is that division applied to the premultiplied f32 output of the same decoder. Three-channel
    output and images with unassociated alpha are left alone."""
    J = built
    yy, xx = np.mgrid[0:200, 0:320]
    alpha = np.clip((xx * 255) // 250, 0, 255).astype(np.uint8)  # 0 in the first column, 255 beyond column 250
    alpha[:8, :] = 0
    rgb = J.synth_image(320, 200, seed=31)
    pre = (rgb.astype(np.uint32) * alpha[..., None] // 255).astype(np.uint8)
    layers = [dict(img=np.dstack([pre, alpha]))]
    if nlayers == 2:
        patch = np.dstack([J.synth_image(64, 48, seed=32) // 2, np.full((48, 64), 128, np.uint8)])
        layers[0]["save_as"] = 1
        layers.append(dict(img=patch, x0=100, y0=60, mode=2, source=1))
    data = J.encode_layers(layers, lossless=lossless, premultiplied=True)
    rc, _, out, px = R.run(data, tmp_path, "f32", 4)
    assert rc == 0, out
    plain = np.frombuffer(px, np.float32).reshape(200, 320, 4)
    rc, _, out, px = R.run(data, tmp_path, "f32", 4, "unpremul")
    assert rc == 0, out
    got = np.frombuffer(px, np.float32).reshape(200, 320, 4)
    mul = np.float32(1.0) / np.maximum(np.float32(2.0 ** -26), plain[..., 3])
    want = plain[..., :3] * mul[..., None]
    assert np.array_equal(got[..., 3], plain[..., 3])
    assert np.allclose(got[..., :3], want, rtol=2e-7, atol=0)
    assert np.abs(got[..., :3] - plain[..., :3]).max() > 0.2  # (it did something)
    # 8-bit RGBA: the same values through the 8-bit conversion
    rc, _, out, px = R.run(data, tmp_path, "u8", 4, "unpremul")
    assert rc == 0, out
    got8 = np.frombuffer(px, np.uint8).reshape(200, 320, 4)
    assert np.abs(got8[..., :3].astype(int) - np.clip(np.rint(want * 255), 0, 255).astype(int)).max() <= 1
    # no alpha in the output: nothing to divide by
    rc, _, out, px3 = R.run(data, tmp_path, "f32", 3, "unpremul")
    assert rc == 0, out
    assert np.array_equal(np.frombuffer(px3, np.float32).reshape(200, 320, 3), plain[..., :3])
    # alpha that is not associated: the option does nothing (dec_frame.h:209)
    data_u = J.encode_layers(layers, lossless=lossless, premultiplied=False)
    rc, _, out, a = R.run(data_u, tmp_path, "f32", 4)
    rc2, _, out2, b = R.run(data_u, tmp_path, "f32", 4, "unpremul")
    assert rc == 0 and rc2 == 0 and a == b


@pytest.mark.parametrize("ups,ecu", [(1, 2), (1, 4), (2, 2), (2, 4), (2, 8), (4, 4), (8, 8)])
def test_extra_channel_with_an_upsampling_factor(built, tmp_path, ups, ecu):
    """An alpha channel coded at ceil(image / its own factor) (frame_header.cc:265-283, dec_modular.cc:262-271; what
    `cjxl --resampling` / `--ec_resampling` write), with the frame at the same or a smaller factor: the host decodes the
    smaller channel (its groups' rectangles shifted against the frame's), the device upsamples it with the image's kernels
    (dec_cache.cc:172-190, 203-212). Sizes that are not multiples of either factor; alpha larger than a group."""
    import jxlo
    J = built
    xs, ys = 1101, 613
    rgb = J.synth_image(xs, ys, seed=8)
    yy, xx = np.mgrid[0:ys, 0:xs]
    alpha = (128 + 100 * np.sin(xx / 37.0) * np.cos(yy / 29.0) + 20 * ((xx // 64 + yy // 64) & 1)).clip(0, 255).astype(np.uint8)
    data = J.encode_rgba8(np.dstack([rgb, alpha]), upsampling=ups, ec_upsampling=ecu)
    o = jxlo.Decoded(data, dumps=False)
    want8, wantf = o.rgb8.copy(), o.buffer("alphaf").reshape(ys, xs)
    o.close()
    assert want8.shape == (ys, xs, 4) and np.abs(want8[..., 3].astype(int) - alpha).mean() < 6
    rc, events, out, px = R.run(data, tmp_path, "u8", 4)
    assert rc == 0, out
    got = np.frombuffer(px, np.uint8).reshape(ys, xs, 4)
    d = np.abs(got.astype(int) - want8.astype(int))
    assert d.max() <= 1 and (d[..., 3] > 0).mean() < 1e-3
    rc, events, out, px = R.run(data, tmp_path, "f32", 4)
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.float32).reshape(ys, xs, 4)[..., 3] - wantf).max() < 1e-6
    # ... and as an extra channel buffer of its own (JxlDecoderSetExtraChannelBuffer: the integers of its bit depth, undithered)
    rc, events, out, px = R.run(data, tmp_path, "u8", 4, "ec")
    assert rc == 0, out
    ec = np.frombuffer(px[xs * ys * 4:], np.uint8).reshape(ys, xs).astype(int)
    d = np.abs(ec - np.rint(wantf.astype(np.float64) * 255.0))
    assert d.max() <= 1 and (d > 0).mean() < 1e-3


def _oriented(img, orientation):
    """The image a viewer shows for `orientation` (EXIF numbering): mirror y, mirror x, then transpose, as the reference's
    writer applies them (stage_write.cc:441-458, 664-699)."""
    bits = [0, 0, 1, 3, 2, 4, 6, 7, 5][orientation]
    if bits & 2:
        img = img[::-1]
    if bits & 1:
        img = img[:, ::-1]
    if bits & 4:
        img = img.transpose(1, 0, 2)
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("orientation", [3, 5, 6, 8])  # (mirror x | y, transpose, y + transpose, x + transpose: every bit, every pairing)
def test_orientation_is_undone_by_the_pixel_writer(built, tmp_path, orientation):
    """JxlDecoderSetKeepOrientation(false) (the default, and what DecodeImageJXL asks for: jxl.cc:213): basic info,
    buffer size and frame header give the oriented size (decode.cc:985-990, 2233-2240) and the pixels and extra channel
    planes arrive flipped / transposed. With keep_orientation the coded image comes back and the orientation is
    reported. The oracle renders the coded orientation; numpy applies the reference's flips to it."""
    import jxlo
    J = built
    xs, ys = 301, 143
    rgb = J.synth_image(xs, ys, seed=21)
    alpha = ((np.mgrid[0:ys, 0:xs][0] * 7 + np.mgrid[0:ys, 0:xs][1] * 3) & 255).astype(np.uint8)
    J.set_orientation(orientation)
    try:
        vardct = J.encode_rgba8(np.dstack([rgb, alpha]), epf_iters=1)
        lossless = J.encode_lossless(np.dstack([rgb, alpha]))
    finally:
        J.set_orientation(1)
    oxs, oys = (ys, xs) if orientation > 4 else (xs, ys)
    for name, data in (("vardct", vardct), ("lossless", lossless)):
        o = jxlo.Decoded(data, dumps=False)
        coded8 = o.rgb8.copy()
        o.close()
        assert coded8.shape == (ys, xs, 4) and np.array_equal(coded8[..., 3], alpha), name
        want = _oriented(coded8, orientation)
        rc, events, out, px = R.run(data, tmp_path, "u8", 4, "ec")
        assert rc == 0, out
        assert "event BASIC_INFO %ux%u " % (oxs, oys) in out and "orientation=1" in out, out
        assert "event FRAME %ux%u " % (oxs, oys) in out, out
        got = np.frombuffer(px[:oxs * oys * 4], np.uint8).reshape(oys, oxs, 4)
        assert np.array_equal(got[..., 3], want[..., 3]), name
        if name == "lossless":
            assert np.array_equal(got, want)
        else:
            assert np.abs(got[..., :3].astype(int) - want[..., :3].astype(int)).max() <= 1  # (dither cell of the new place)
        ec = np.frombuffer(px[oxs * oys * 4:], np.uint8).reshape(oys, oxs)
        assert np.array_equal(ec, want[..., 3]), name
        # float samples carry no dither: the oriented oracle floats, at the bar of the unoriented test
        if name == "vardct":
            o = jxlo.Decoded(data)
            ref_f = _oriented(o.planes("rgbf").transpose(1, 2, 0).copy(), orientation)
            o.close()
            rc, events, out, px = R.run(data, tmp_path, "f32", 3, "callback")
            assert rc == 0, out
            assert np.abs(np.frombuffer(px, np.float32).reshape(oys, oxs, 3) - ref_f).max() < 5e-5
        # keep_orientation: the coded pixels, the orientation left to the caller
        rc, events, out, px = R.run(data, tmp_path, "u8", 4, "keep")
        assert rc == 0 and "event BASIC_INFO %ux%u " % (xs, ys) in out and "orientation=%d" % orientation in out, out
        got = np.frombuffer(px, np.uint8).reshape(ys, xs, 4)
        assert np.abs(got.astype(int) - coded8.astype(int)).max() <= (0 if name == "lossless" else 1)


@pytest.mark.gpu
@pytest.mark.parametrize("passes", [1, 2, "420"])
def test_flush_image_draws_what_has_arrived(built, tmp_path, passes):
    """JxlDecoderFlushImage (decode.cc:2458-2475, FrameDecoder::Flush dec_frame.cc:735-795): with a part of the frame's
    bytes, every group is drawn from its leading passes whose sections are whole (dec_frame.cc:620-680), from the DC image
    alone when it has none. Expectation: the oracle told to use the same prefix of the codestream. The input comes in chunks; every time the
    decoder runs out inside the frame the replay program flushes and keeps the buffer."""
    import os
    import jxlo
    J = built
    if passes == "420":  # a chroma-subsampled YCbCr frame: the groups drawn from DC alone take each channel's DC from its own grid
        data = J.encode_rgb8(J.synth_image(1100, 800, seed=9), color_transform=2, chroma_subsampling=4, strategy_mode=0)
    else:
        data = J.encode_rgb8(J.synth_image(1100, 800, seed=9), num_passes=passes)  # 5 x 4 groups
    chunk = len(data) // 7
    rc, events, out, px = R.run(data, tmp_path, "u8", 3, "flush", "chunk=%d" % chunk)
    assert rc == 0 and events.count("FULL_IMAGE") == 1, out
    final = np.frombuffer(px, np.uint8).reshape(800, 1100, 3)
    o = jxlo.Decoded(data, dumps=False)
    assert np.abs(final.astype(int) - o.rgb8.astype(int)).max() <= 1
    o.close()
    flushed = [l for l in out.splitlines() if l.startswith("flushed ")]
    assert len(flushed) >= 3, out  # (the frame header, TOC and DC image fit the first chunks; several chunks of AC follow)
    partial_seen = 0
    for line in flushed:
        k, given = int(line.split()[1]), int(line.split("bytes_given=")[1])
        got = np.fromfile(os.path.join(str(tmp_path), "out.raw.flush%d" % k), np.uint8).reshape(800, 1100, 3)
        want = jxlo.Decoded(data, dumps=False, prefix=given)
        d = np.abs(got.astype(int) - want.rgb8.astype(int))
        want.close()
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, line
        if np.abs(got.astype(int) - final.astype(int)).max() > 8:
            partial_seen += 1  # (this flush really lacked groups)
    assert partial_seen >= 2
    # without a buffer, outside a frame, or with everything there, nothing is flushed (and nothing breaks)
    rc, events, out, px = R.run(data, tmp_path, "u8", 3, "flush")
    assert rc == 0 and "flushed" not in out


@pytest.mark.gpu
def test_frame_progression_event_at_the_dc_step(built, tmp_path):
    """JXL_DEC_FRAME_PROGRESSION (decode.h; decode.cc:1421-1428,1492-1500): subscribed, the decoder pauses ONCE per frame when
    the frame's DC image is decoded and sections are still missing, with an intended downsampling ratio of 8; a flush at that
    point draws exactly what has arrived (the oracle told to use the same prefix). Not subscribed, or with the whole frame
    there, the event never comes; JxlDecoderSetProgressiveDetail takes kDC / kLastPasses / kPasses only (decode.cc:2963-2973)."""
    import ctypes
    import os
    import jxlo
    J = built
    data = J.encode_rgb8(J.synth_image(1100, 800, seed=9), num_passes=2)
    chunk = len(data) // 7
    rc, events, out, px = R.run(data, tmp_path, "u8", 3, "progression", "chunk=%d" % chunk)
    assert rc == 0 and events.count("FULL_IMAGE") == 1 and events.count("FRAME_PROGRESSION") == 1, out
    assert events.index("FRAME") < events.index("FRAME_PROGRESSION") < events.index("FULL_IMAGE"), events
    line = [l for l in out.splitlines() if l.startswith("event FRAME_PROGRESSION")][0]
    assert "ratio=8" in line, line
    given = int(line.split("bytes_given=")[1])
    assert "flushed 0 bytes_given=%d" % given in out, out
    got = np.fromfile(os.path.join(str(tmp_path), "out.raw.flush0"), np.uint8).reshape(800, 1100, 3)
    want = jxlo.Decoded(data, dumps=False, prefix=given)
    d = np.abs(got.astype(int) - want.rgb8.astype(int))
    want.close()
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    final = np.frombuffer(px, np.uint8).reshape(800, 1100, 3)
    assert np.abs(got.astype(int) - final.astype(int)).max() > 8  # (the step really lacked groups)
    # the same input without the subscription, and the whole file at once with it
    rc, events, out, _ = R.run(data, tmp_path, "u8", 3, "chunk=%d" % chunk)
    assert rc == 0 and "FRAME_PROGRESSION" not in events
    rc, events, out, _ = R.run(data, tmp_path, "u8", 3, "progression")
    assert rc == 0 and "FRAME_PROGRESSION" not in events and events.count("FULL_IMAGE") == 1
    L = J.lib()
    L.JxlDecoderCreate.restype = ctypes.c_void_p
    L.JxlDecoderCreate.argtypes = [ctypes.c_void_p]
    L.JxlDecoderSetProgressiveDetail.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.JxlDecoderGetIntendedDownsamplingRatio.argtypes = [ctypes.c_void_p]
    L.JxlDecoderGetIntendedDownsamplingRatio.restype = ctypes.c_size_t
    L.JxlDecoderDestroy.argtypes = [ctypes.c_void_p]
    dec = L.JxlDecoderCreate(None)
    assert [L.JxlDecoderSetProgressiveDetail(dec, v) for v in (0, 1, 2, 3, 4, 5, 6)] == [1, 0, 0, 0, 1, 1, 1]  # (1 = JXL_DEC_ERROR)
    assert L.JxlDecoderGetIntendedDownsamplingRatio(dec) == 8
    L.JxlDecoderDestroy(dec)


@pytest.mark.gpu
@pytest.mark.parametrize("detail,passes", [(3, 2), (2, 2), (3, 3), (2, 3)])
def test_frame_progression_steps_by_passes(built, tmp_path, detail, passes):
    """JxlDecoderSetProgressiveDetail(kPasses): after the DC step the decoder pauses every time another pass is whole for
    every group (decode.cc:1502-1512, dec_frame.h:144-200); a flush at a step draws every group from the passes it has
    (dec_frame.cc:620-680,735-795) = the oracle told to use the same prefix, and the steps come closer to the final image.
    kLastPasses pauses at the last pass of a downsampling level only (frame_header.h:286-309): the writer's two-pass frames
    name none (only the DC step comes), its three-pass frames name 4x after pass 0 (one more step, and the intended
    downsampling ratio drops from 8 to 4 there)."""
    import os
    import jxlo
    J = built
    data = J.encode_rgb8(J.synth_image(1100, 800, seed=9), num_passes=passes)
    chunk = len(data) // 23
    rc, events, out, px = R.run(data, tmp_path, "u8", 3, "progression", "detail=%d" % detail, "chunk=%d" % chunk)
    assert rc == 0 and events.count("FULL_IMAGE") == 1, out
    final = np.frombuffer(px, np.uint8).reshape(800, 1100, 3).astype(int)
    steps = [l for l in out.splitlines() if l.startswith("event FRAME_PROGRESSION")]
    assert len(steps) == (passes if detail == 3 else (2 if passes == 3 else 1)), out  # the DC step + one per pass but the last
    errs = []
    for k, line in enumerate(steps):
        assert ("ratio=4" if passes == 3 and k >= 1 else "ratio=8") in line, line
        given = int(line.split("bytes_given=")[1])
        assert "flushed %d bytes_given=%d" % (k, given) in out, out
        got = np.fromfile(os.path.join(str(tmp_path), "out.raw.flush%d" % k), np.uint8).reshape(800, 1100, 3)
        want = jxlo.Decoded(data, dumps=False, prefix=given)
        d = np.abs(got.astype(int) - want.rgb8.astype(int))
        want.close()
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, line
        errs.append(float(np.abs(got.astype(int) - final).mean()))
    assert all(a > b for a, b in zip(errs, errs[1:])) and errs[-1] > 0, errs  # every step adds detail; none is the final image


@pytest.mark.gpu
@pytest.mark.parametrize("env,kw", [({}, dict(ac_code_mode=3)), ({}, dict(ac_code_mode=1)), ({}, dict(ac_code_mode=2)),
                                    (dict(JXLHIP_ENTROPY="0"), {}), (dict(JXLHIP_ENTROPY="1"), {})])
def test_partial_passes_on_the_other_entropy_kernels(built, tmp_path, monkeypatch, env, kw):
    """The same steps for prefix / LZ77 coded streams (the generic kernel, or the lane kernel's prefix form) and with the
    section-per-workgroup rANS kernels (JXLHIP_ENTROPY = 0 / 1): they step over sections that have not arrived (size 0) and
    leave such a group, or such a pass of it, out."""
    import os
    import jxlo
    J = built
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    data = J.encode_rgb8(J.synth_image(900, 600, seed=19), num_passes=2, **kw)
    chunk = len(data) // 17
    rc, events, out, px = R.run(data, tmp_path, "u8", 3, "progression", "detail=3", "chunk=%d" % chunk)
    assert rc == 0 and events.count("FULL_IMAGE") == 1, out
    final = np.frombuffer(px, np.uint8).reshape(600, 900, 3)
    o = jxlo.Decoded(data, dumps=False)
    assert np.abs(final.astype(int) - o.rgb8.astype(int)).max() <= 1
    o.close()
    steps = [l for l in out.splitlines() if l.startswith("event FRAME_PROGRESSION")]
    assert len(steps) == 2, out
    for k, line in enumerate(steps):
        given = int(line.split("bytes_given=")[1])
        assert "flushed %d bytes_given=%d" % (k, given) in out, out
        got = np.fromfile(os.path.join(str(tmp_path), "out.raw.flush%d" % k), np.uint8).reshape(600, 900, 3)
        want = jxlo.Decoded(data, dumps=False, prefix=given)
        d = np.abs(got.astype(int) - want.rgb8.astype(int))
        want.close()
        assert d.max() <= 1 and (d > 0).mean() < 2e-3, line


@pytest.mark.gpu
def test_preview_frame_through_the_decoder_api(built, tmp_path):
    """JXL_DEC_PREVIEW_IMAGE (decode.cc:1326-1343, 1448-1450, 1554-1559, 2522-2562): the preview has its own buffer request
    and event, no FRAME event, and is stepped over when nobody subscribed to it."""
    import os
    import jxlo
    J = built
    img, pv = J.synth_image(600, 400, seed=3), J.synth_image(75, 50, seed=4)
    data = J.encode_with_preview(img, pv)
    want_pv = jxlo.Decoded(data, dumps=False, preview=True).rgb8
    want = jxlo.Decoded(data, dumps=False).rgb8
    rc, events, out, px = R.run(data, tmp_path, "u8", 3)
    assert rc == 0, out
    assert [e for e in events if e not in ("BASIC_INFO", "COLOR_ENCODING")] == [
        "NEED_PREVIEW_OUT_BUFFER", "PREVIEW_IMAGE", "FRAME", "NEED_IMAGE_OUT_BUFFER", "FULL_IMAGE", "SUCCESS"], out
    assert "size=%d preview=75x50" % (75 * 50 * 3) in out
    got_pv = np.fromfile(os.path.join(str(tmp_path), "out.raw.preview"), np.uint8).reshape(50, 75, 3)
    assert np.abs(got_pv.astype(int) - want_pv.astype(int)).max() <= 1
    assert np.abs(np.frombuffer(px, np.uint8).reshape(400, 600, 3).astype(int) - want.astype(int)).max() <= 1
    rc, events, out, px = R.run(R.container(data), tmp_path, "u8", 3, "nopreview", "chunk=3000")
    assert rc == 0 and "PREVIEW_IMAGE" not in events and events.count("FULL_IMAGE") == 1, out
    assert np.abs(np.frombuffer(px, np.uint8).reshape(400, 600, 3).astype(int) - want.astype(int)).max() <= 1


def test_more_contexts_than_pool_streams_on_concurrent_host_threads(built):
    """Contexts share a pool of 4 HIP streams per device (JXLHIP_STREAMS), so a context's synchronisation also waits for
    whatever unrelated contexts have queued on the same stream, and destroyed contexts are recycled with their buffers
    (RecycleContext): 12 host threads, each with a context of its own, upload / decode / download different frames at once,
    three rounds each (so every thread also gets recycled contexts); every result must be bit-identical to the serial decode
    of the same stream (no test used more contexts than pool streams from concurrent threads)."""
    import concurrent.futures
    J = built
    streams = [J.encode_rgb8(J.synth_image(300 + 37 * i, 200 + 23 * i, seed=50 + i), distance=1.0 + 0.5 * (i % 3), noise=40 * (i % 2))
               for i in range(6)]
    serial = [J.decode_rgb8(s) for s in streams]

    def work(t):
        out = []
        for r in range(3):
            i = (t + r) % len(streams)
            f = J.Frame(streams[i], threads=1)
            c = J.HipContext()
            c.upload(f)
            c.run_all()
            out.append((i, c.rgb8()))
            c.close()
            f.close()
        return out

    with concurrent.futures.ThreadPoolExecutor(12) as ex:
        for res in ex.map(work, range(12)):
            for i, rgb in res:
                assert np.array_equal(rgb, serial[i]), "stream %d decoded differently on a shared stream" % i
