"""GPU parity tests (-m gpu): the HIP path, called through the C ABI, against the oracle on the same streams.
Bars: quantised coefficients bit-exact (integer work); float planes within 2e-5 absolute (values are O(1); the
reference's own fast-vs-slow pipeline tolerance is 2e-4, render_pipeline_test.cc:320-327); RGB8 within 1 level (the
final rounding may flip on a float difference of 1e-7; the reference itself differs across SIMD targets by more)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _used_mask(o):
    """Per group/channel: which coefficient slots belong to a varblock (the rest of the 65536-slot plane is unused)."""
    i = o.info
    acs = o.buffer("acs").reshape(i["ysize_blocks"], i["xsize_blocks"])
    xg = (i["xsize"] + 255) // 256
    used = np.zeros(i["num_groups"], np.int64)
    for g in range(i["num_groups"]):
        gy, gx = divmod(g, xg)
        used[g] = acs[gy * 32:(gy + 1) * 32, gx * 32:(gx + 1) * 32].size * 64
    return used


def _compare(J, jxlo, data, check_rgb=True, crop_idct=False):
    f = J.Frame(data, threads=2)
    o = jxlo.Decoded(data)
    c = J.HipContext()
    try:
        c.set_option("keep_filtered", 1)
        c.upload(f)
        c.run_entropy()
        c.sync()
        r, flags = c.errors()
        assert r == 0 and not any(flags)
        co = c.download("coeffs").astype(np.int32)
        ref = o.planes("coeffs")
        used = _used_mask(o)
        for g in range(o.info["num_groups"]):
            assert np.array_equal(co[g, :, :used[g]], ref[g, :, :used[g]]), "coefficients differ in group %d" % g
        c.run_transform()
        c.sync()
        x = c.download("xyb_idct")
        ys, xs = o.info["ysize"], o.info["xsize"]
        if crop_idct:  # (chroma-subsampled frames: the padding beyond the frame is not specified once the channels are upsampled)
            assert np.abs(x[:, :ys, :xs] - o.planes("xyb_idct")[:, :ys, :xs]).max() < 2e-5
        else:
            assert np.abs(x - o.planes("xyb_idct")).max() < 2e-5
        c.run_filter_color()
        c.sync()
        xf = c.download("xyb_filtered")[:, :ys, :xs]
        assert np.abs(xf - o.planes("xyb_filtered")[:, :, :xs]).max() < 2e-5
        rgb = c.rgb8()
        d = np.abs(rgb.astype(int) - o.rgb8.astype(int))
        assert d.max() <= 1
        assert (d > 0).mean() < 1e-3
        return rgb
    finally:
        c.close()
        f.close()
        o.close()


@pytest.mark.parametrize("size,kw", [
    ((64, 64), dict()),                                   # single TOC entry: AC data starts mid-byte
    ((256, 256), dict(strategy_mode=1)),                  # BASELINE.json configs[0] at its size: exactly one full 256x256 group, d1.0
    ((256, 256), dict(strategy_mode=2, distance=2.0)),    # ... with every transform size and the second EPF iteration
    ((8, 8), dict(strategy_mode=0)),                      # one block
    ((1, 1), dict(strategy_mode=0)),
    ((257, 255), dict()),                                 # ragged groups, xsize not a multiple of 8
    ((520, 300), dict()),
    ((520, 300), dict(distance=2.0)),                     # EPF1 + EPF2
    ((520, 300), dict(distance=4.5, gab=0)),              # EPF0 + EPF1 + EPF2, no Gaborish
    ((520, 300), dict(epf_iters=0, gab=0)),               # no filters
    ((777, 513), dict(strategy_mode=2, random_cmap=1)),   # random DCT-family tiling up to 256x256, random CfL
    ((300, 200), dict(skip_dc_smoothing=1)),
    ((600, 400), dict(noise=40)),                         # noise synthesis (frame flag kNoise), several 256x256 generators
    ((257, 255), dict(noise=200, distance=2.0)),          # ... ragged groups (partial last generator step per row), EPF2
    ((520, 300), dict(noise=80, custom_cmap=1)),          # ... with a coded base colour correlation
    ((1000, 700), dict(num_histograms=3)),                # several AC histogram sets (libjxl's streaming encoder)
    ((1300, 1100), dict(num_histograms=30, strategy_mode=2, distance=2.0)),  # one set per group
    ((600, 400), dict(custom_cmap=1, random_cmap=1)),     # coded colour correlation (factor, bases, DC factors), qm scales
    ((700, 520), dict(custom_bctx=1)),                    # coded block context map: quant-field and DC thresholds
    ((1000, 700), dict(custom_bctx=1, custom_orders=1, num_histograms=3, strategy_mode=2)),
    ((520, 300), dict(custom_bctx=1, num_passes=2)),
    ((700, 520), dict(custom_orders=1)),                  # coded (Lehmer) coefficient orders, different per channel
    ((777, 600), dict(custom_orders=1, strategy_mode=2, num_histograms=2, distance=2.0)),
    ((520, 300), dict(custom_orders=1, num_passes=2)),    # ... and different per pass
    ((700, 520), dict(num_passes=2)),                     # progressive: two passes, pass 0 shifted by one bit
    ((200, 100), dict(num_passes=2)),                     # ... with a single group
    ((700, 520), dict(num_passes=3)),                     # three passes (shifts 2, 1, 0) and a downsampling bracket in the frame header
    ((1000, 700), dict(num_passes=3, num_histograms=3, custom_orders=1, distance=2.0)),
    ((1000, 700), dict(num_passes=2, num_histograms=3, distance=2.0)),
    ((520, 300), dict(color_transform=2)),                # an image that is not XYB encoded: YCbCr frame (stage_ycbcr.cc), 4:4:4
    ((777, 513), dict(color_transform=2, distance=2.0, strategy_mode=2)),
    ((520, 300), dict(color_transform=1)),                # ... ColorTransform kNone: the channels are the sRGB samples
    ((600, 400), dict(color_transform=2, upsampling=2)),  # ... through the upsampling kernel's colour stage
    ((520, 300), dict(raw_quant=1, strategy_mode=0)),     # the 8x8 DCT's dequantisation table coded RAW (a Modular image in AC global)
    ((777, 513), dict(raw_quant=1)),                      # ... beside library tables for the other transforms
    ((600, 400), dict(raw_quant=1, color_transform=2, strategy_mode=0, epf_iters=0, gab=0)),  # what a recompressed 4:4:4 JPEG looks like
])
def test_image_streams(built, size, kw):
    import jxlo
    J = built
    _compare(J, jxlo, J.encode_rgb8(J.synth_image(size[0], size[1], seed=size[0]), **kw))


CS420, CS422, CS440 = 4, 8, 12  # channel modes of Cb, Y, Cr (frame_header.h:81-166): Y at 2x2 / 2x1 / 1x2 samples per MCU


@pytest.mark.parametrize("size,kw", [
    ((520, 300), dict(chroma_subsampling=CS420)),                          # what a recompressed 4:2:0 JPEG's frame looks like
    ((520, 300), dict(chroma_subsampling=CS422)),
    ((520, 300), dict(chroma_subsampling=CS440, distance=2.0)),
    ((257, 255), dict(chroma_subsampling=CS420)),                          # odd sizes: half-MCU edges, ragged groups
    ((17, 9), dict(chroma_subsampling=CS420)),
    ((8, 8), dict(chroma_subsampling=CS420)),                              # one MCU: 2 x 2 blocks for one block of image
    ((1, 1), dict(chroma_subsampling=CS422)),
    ((777, 513), dict(chroma_subsampling=CS420, distance=4.5, gab=0)),     # every EPF stage behind the upsampled channels
    ((600, 400), dict(chroma_subsampling=CS420, epf_iters=0, gab=0, raw_quant=1)),  # no filters, RAW table: the JPEG case
    ((600, 400), dict(chroma_subsampling=CS420, random_cmap=1)),           # chroma from luma still applies to AC (dec_group.cc:432-441)
    ((700, 520), dict(chroma_subsampling=CS420, custom_bctx=1)),           # DC thresholds read the subsampled DC (compressed_dc.cc:253-290)
    ((1000, 700), dict(chroma_subsampling=CS422, num_histograms=3, custom_orders=1)),
    ((520, 300), dict(chroma_subsampling=CS420, num_passes=2)),
    ((600, 400), dict(chroma_subsampling=CS420, upsampling=2)),            # ... and the frame upsampled behind them
    ((600, 400), dict(chroma_subsampling=CS420, noise=60)),
    ((520, 300), dict(chroma_subsampling=1)),                              # legal oddities: Cb at twice the resolution of Y and Cr
    ((520, 300), dict(chroma_subsampling=0b100100)),                       # Cb full, Y and Cr subsampled 2x1
    ((520, 300), dict(chroma_subsampling=0b011011)),                       # Cb 1x2, Y 2x1, Cr 2x2: every shift pair occurs
])
def test_chroma_subsampled_frames(built, size, kw):
    """YCbCr frames with subsampled channels (frame_header.h:81-166; dec_group.cc:568-578: a channel's blocks ride on the
    varblocks that lie on its grid; compressed_dc.cc:232-250: DC without chroma from luma; stage_chroma_upsampling.cc: the
    channels back at full resolution in front of the filters): coefficients bit for bit, planes and pixels at the usual bars."""
    import jxlo
    J = built
    _compare(J, jxlo, J.encode_rgb8(J.synth_image(size[0], size[1], seed=size[0]), color_transform=2, strategy_mode=0, **kw), crop_idct=True)


@pytest.mark.parametrize("seed,kw", [
    (1, dict(chroma_subsampling=CS420)),
    (2, dict(chroma_subsampling=CS422, custom_bctx=1, custom_orders=1)),
    (3, dict(chroma_subsampling=CS440, custom_cmap=1, custom_lf=1, epf_iters=3)),
    (4, dict(chroma_subsampling=CS420, ac_code_mode=1)),      # prefix codes: the lane kernel's other symbol reader
    (5, dict(chroma_subsampling=CS420, ac_code_mode=2)),      # LZ77: the one-lane-per-section kernel
    (6, dict(chroma_subsampling=CS420, ac_code_mode=3)),
    (7, dict(chroma_subsampling=CS420, max_clusters=128)),    # alias tables read in place from global memory
    (8, dict(chroma_subsampling=CS420, num_passes=2, num_histograms=2)),
    (9, dict(chroma_subsampling=0b011011)),                   # Y subsampled too: a varblock may carry no luma for the chroma-from-luma term
    (10, dict(chroma_subsampling=1, num_passes=2)),           # ... with the passes merged into the natural layout
    (11, dict(chroma_subsampling=0b100100, ac_code_mode=2)),  # ... through the one-lane-per-section kernel
])
def test_chroma_subsampled_random_streams(built, seed, kw):
    """Random streams over every transform a subsampled frame may use (the ten that cover one block: dec_modular.cc:534-538),
    random chroma-from-luma maps, sharpness and quantisation fields."""
    import jxlo
    J = built
    _compare(J, jxlo, J.encode_random(777, 600, seed=seed, color_transform=2, **kw), crop_idct=True)


@pytest.mark.parametrize("choice", ["1", "0"])
def test_chroma_subsampled_frames_through_the_fallback_entropy_kernels(built, choice):
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, libjxl_amd as J, jxlo\n"
        "for data in (J.encode_random(520, 300, seed=6, color_transform=2, chroma_subsampling=4),\n"
        "             J.encode_rgb8(J.synth_image(600, 400, seed=9), color_transform=2, chroma_subsampling=8, strategy_mode=0)):\n"
        "    o = jxlo.Decoded(data)\n"
        "    f = J.Frame(data); c = J.HipContext(); c.upload(f); c.run_entropy(); c.sync()\n"
        "    r, flags = c.errors(); assert r == 0\n"
        "    c.run_transform(); c.run_filter_color()\n"
        "    d = np.abs(c.rgb8().astype(int) - o.rgb8.astype(int)); assert d.max() <= 1, d.max()\n"
        "    xs, ys = o.info['xsize'], o.info['ysize']\n"
        "    x = c.download('xyb_idct'); assert np.abs(x[:, :ys, :xs] - o.planes('xyb_idct')[:, :ys, :xs]).max() < 2e-5\n"
        "    c.close(); f.close(); o.close()\n"
        "print('ok')\n") % (root, os.path.join(root, "oracle"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, JXLHIP_ENTROPY=choice))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("strategy", list(range(27)))
def test_every_strategy_random_stream(built, strategy):
    import jxlo
    J = built
    _compare(J, jxlo, J.encode_random(512, 512, seed=100 + strategy, strategy_mask=(1 << strategy) | 1))


@pytest.mark.parametrize("seed,epf", [(1, 0), (2, 1), (3, 2), (4, 3)])
def test_all_strategies_mixed(built, seed, epf):
    import jxlo
    J = built
    _compare(J, jxlo, J.encode_random(777, 600, seed=seed, epf_iters=epf, custom_orders=seed & 1, custom_bctx=(seed >> 1) & 1, custom_cmap=int(seed == 4)))


@pytest.mark.parametrize("epf", [0, 1, 2, 3])
def test_custom_loop_filter_and_dc_dequant_headers(built, epf):
    """Coded (non-default) Gaborish weights, EPF sharpness LUT / channel scales / sigma parameters and DC dequantisation
    steps (loop_filter.cc:20-100, quantizer.h): the GPU must take them from the stream like the oracle."""
    import jxlo
    J = built
    _compare(J, jxlo, J.encode_random(700, 520, seed=30 + epf, epf_iters=epf, custom_lf=1, custom_cmap=epf & 1))


def test_full_size_4k(built):
    """BASELINE.json config[1]: 3840x2160 d1.0 — direct comparison with the oracle (a few seconds of CPU)."""
    import jxlo
    J = built
    _compare(J, jxlo, J.encode_rgb8(J.synth_image(3840, 2160)))


def test_context_reuse_and_idempotence(built):
    """Re-running the stages on resident inputs gives identical bytes; a context can take a smaller frame after a
    larger one."""
    J = built
    a = J.encode_rgb8(J.synth_image(600, 400))
    b = J.encode_rgb8(J.synth_image(200, 100, seed=2))
    c = J.HipContext()
    fa, fb = J.Frame(a), J.Frame(b)
    c.upload(fa)
    c.run_all()
    r1 = c.rgb8()
    c.run_all()
    assert np.array_equal(r1, c.rgb8())
    c.upload(fb)
    c.run_all()
    r2 = c.rgb8()
    assert r2.shape == (100, 200, 3)
    assert np.array_equal(r2, J.decode_rgb8(b))
    c.close()


@pytest.mark.parametrize("lanes,wait_shift", [(1, 3), (4, 0), (16, 3), (64, 3), (64, 0), (64, 6)])
def test_lane_packing_of_sections(built, lanes, wait_shift, monkeypatch):
    """The lane-parallel entropy kernel must give the same coefficients however many sections share a wave and
    however block transitions are batched (JXLHIP_LANES / JXLHIP_WAIT_SHIFT only steer the work distribution)."""
    import jxlo
    J = built
    monkeypatch.setenv("JXLHIP_LANES", str(lanes))
    monkeypatch.setenv("JXLHIP_WAIT_SHIFT", str(wait_shift))
    for data in (J.encode_random(1100, 900, seed=lanes + 40), J.encode_rgb8(J.synth_image(2100, 1300, seed=lanes), distance=1.5)):
        f = J.Frame(data, threads=2)
        o = jxlo.Decoded(data)
        c = J.HipContext()
        try:
            c.upload(f)
            c.run_entropy()
            c.sync()
            r, flags = c.errors()
            assert r == 0 and not any(flags)
            co = c.download("coeffs").astype(np.int32)
            ref = o.planes("coeffs")
            used = _used_mask(o)
            for g in range(o.info["num_groups"]):
                assert np.array_equal(co[g, :, :used[g]], ref[g, :, :used[g]]), "coefficients differ in group %d" % g
        finally:
            c.close()
            f.close()
            o.close()


@pytest.mark.parametrize("choice", ["1", "0"])
def test_fallback_entropy_kernels_natural_layout(built, choice):
    """JXLHIP_ENTROPY=1 / 0 select the wave-per-section kernels (k_entropy_uni / k_entropy_ans) that multi-pass frames
    and frames with oversized tables fall back to; they write the natural, zero-filled coefficient layout, which the
    transform kernels must take as well. The choice is latched per process, hence the subprocess."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, libjxl_amd as J, jxlo\n"
        "for data in (J.encode_rgb8(J.synth_image(700, 520, seed=4), distance=2.0), J.encode_random(520, 300, seed=6),\n"
        "             J.encode_rgb8(J.synth_image(600, 400, seed=9), num_histograms=3)):\n"
        "    o = jxlo.Decoded(data)\n"
        "    f = J.Frame(data); c = J.HipContext(); c.upload(f); c.run_entropy(); c.sync()\n"
        "    r, flags = c.errors(); assert r == 0\n"
        "    co = c.download('coeffs').astype(np.int32); ref = o.planes('coeffs')\n"
        "    acs = o.buffer('acs'); n = o.info['num_groups']\n"
        "    c.run_transform(); c.run_filter_color()\n"
        "    d = np.abs(c.rgb8().astype(int) - o.rgb8.astype(int)); assert d.max() <= 1, d.max()\n"
        "    x = c.download('xyb_idct'); assert np.abs(x - o.planes('xyb_idct')).max() < 2e-5\n"
        "    c.close(); f.close(); o.close()\n"
        "print('ok')\n") % (root, os.path.join(root, "oracle"))
    env = dict(os.environ, JXLHIP_ENTROPY=choice)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("env", [{"JXLHIP_FILTER_ROWS1": "1"}, {"JXLHIP_FILTER_TILES": "1"}, {"JXLHIP_IDCT_MATRIX": "1"}])
def test_alternate_kernel_forms_agree_with_the_oracle(built, env):
    """The kernels the default path replaced stay selectable (one-column row filter, LDS-tile filter, matrix-form IDCT):
    each of them has to meet the same parity bars, so they cross-check the default forms (different tilings, different
    summation structure). The environment is read per launch; a subprocess keeps it out of the other tests."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "import numpy as np, libjxl_amd as J, jxlo\n"
        "for data in (J.encode_rgb8(J.synth_image(1111, 777, seed=4)), J.encode_random(520, 300, seed=6),\n"
        "             J.encode_rgb8(J.synth_image(64, 40, seed=9), distance=2.0)):\n"
        "    o = jxlo.Decoded(data)\n"
        "    f = J.Frame(data); c = J.HipContext(); c.set_option('keep_filtered', 1); c.upload(f); c.run_all(); c.sync()\n"
        "    r, flags = c.errors(); assert r == 0\n"
        "    x = c.download('xyb_idct'); assert np.abs(x - o.planes('xyb_idct')).max() < 2e-5\n"
        "    xs, ys = o.info['xsize'], o.info['ysize']\n"
        "    xf = c.download('xyb_filtered')[:, :ys, :xs]; assert np.abs(xf - o.planes('xyb_filtered')[:, :, :xs]).max() < 2e-5\n"
        "    d = np.abs(c.rgb8().astype(int) - o.rgb8.astype(int)); assert d.max() <= 1, d.max()\n"
        "    c.close(); f.close(); o.close()\n"
        "print('ok')\n") % (root, os.path.join(root, "oracle"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]


@pytest.mark.parametrize("factor", [2, 4, 8])
def test_upsampled_frames(built, factor):
    """Frames coded at 1/2, 1/4, 1/8 size with the upsampling flag (stage_upsampling.cc): the GPU produces the full-size
    image like the oracle, for ragged image sizes, and through the JxlDecoder-level helper as well."""
    import jxlo
    J = built
    # (the third stream: with noise, which an upsampled frame gets at the IMAGE's resolution, behind the upsampling --
    # dec_cache.cc:206-216; one generator per 256 x 256 square of the image, several of them and ragged ones here)
    for data in (J.encode_rgb8(J.synth_image(701, 523, seed=factor), upsampling=factor),
                 J.encode_random(300, 203, seed=20 + factor, upsampling=factor, epf_iters=2),
                 J.encode_rgb8(J.synth_image(701, 523, seed=40 + factor), upsampling=factor, noise=120)):
        o = jxlo.Decoded(data, dumps=False)
        f = J.Frame(data)
        assert (f.info["out_xsize"], f.info["out_ysize"]) == o.out_size
        assert f.info["xsize"] == (o.out_size[0] + factor - 1) // factor
        f.close()
        rgb = J.decode_rgb8(data)
        assert rgb.shape == o.rgb8.shape
        d = np.abs(rgb.astype(int) - o.rgb8.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 2e-3
        o.close()


@pytest.mark.parametrize("factor", [2, 4, 8])
def test_upsampling_with_coded_weights(built, factor):
    """Images whose header codes its own upsampling weights (CustomTransformData, image_metadata.cc:87-214; applied by
    stage_upsampling.cc:59-84 like the default ones): the GPU upsamples with the coded matrix like the oracle, and the
    result differs from what the default weights give (so the coded ones were really used)."""
    import jxlo
    J = built
    img = J.synth_image(411, 307, seed=30 + factor)
    plain = J.decode_rgb8(J.encode_rgb8(img, upsampling=factor))
    J.set_custom_upsampling(7, seed=factor)
    try:
        data = J.encode_rgb8(img, upsampling=factor)
    finally:
        J.set_custom_upsampling(0)
    o = jxlo.Decoded(data, dumps=False)
    rgb = J.decode_rgb8(data)
    assert rgb.shape == o.rgb8.shape == plain.shape
    d = np.abs(rgb.astype(int) - o.rgb8.astype(int))
    o.close()
    assert d.max() <= 1 and (d > 0).mean() < 2e-3
    assert (np.abs(rgb.astype(int) - plain.astype(int)) > 2).mean() > 0.01


def test_randomized_parity_sweep(built):
    """A short run of scripts/fuzz_parity.py: random sizes, distances, filter settings, strategy sets, histogram counts."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_parity.py"), "24", "3"], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_int32_coefficient_storage(built):
    """Streams whose histograms allow magnitudes beyond 15 bits make the decoder store int32 coefficients
    (dec_frame.cc:412-429); the entropy stage must stay bit-exact and the later stages must take the same layout."""
    import jxlo
    J = built
    data = J.encode_random(700, 520, seed=8, big_coeffs=1)
    f = J.Frame(data, threads=2)
    assert f.info["coef_bits"] == 32
    o = jxlo.Decoded(data)
    c = J.HipContext()
    try:
        c.set_option("keep_filtered", 1)
        c.upload(f)
        c.run_all()
        c.sync()
        r, flags = c.errors()
        assert r == 0 and not any(flags)
        co = c.download("coeffs")
        assert co.dtype == np.int32 and np.abs(co).max() > 40000
        ref = o.planes("coeffs")
        used = _used_mask(o)
        for g in range(o.info["num_groups"]):
            assert np.array_equal(co[g, :, :used[g]], ref[g, :, :used[g]]), "coefficients differ in group %d" % g
        # pixel values are huge here: compare the transform output relative to the block's own dynamic range
        x, xr = c.download("xyb_idct"), o.planes("xyb_idct")
        assert np.abs(x - xr).max() <= 1e-5 * max(1.0, float(np.abs(xr).max()))
    finally:
        c.close()
        f.close()
        o.close()


def test_shared_planes_between_frame_sets(built):
    """jxlhip_share_planes: the contexts of a second frame set keep their XYB planes in the first set's buffers. The
    two sets' transform -> filter sequences run from different streams; every launch that overwrites shared planes has to
    wait for the filter launch that last read them, so repeated alternating decodes stay identical to unshared ones."""
    J = built
    a_data = J.encode_rgb8(J.synth_image(900, 600))
    b_data = J.encode_rgb8(J.synth_image(880, 590, seed=7), distance=2.0)  # smaller frame, other filter settings
    ref_a, ref_b = J.decode_rgb8(a_data), J.decode_rgb8(b_data)
    fa, fb = J.Frame(a_data), J.Frame(b_data)
    A = [J.HipContext() for _ in range(3)]
    B = [J.HipContext() for _ in range(3)]
    try:
        for a, b in zip(A, B):
            b.share_planes(a)
        for c in A:
            c.upload(fa)
        for c in B:
            c.upload(fb)
        J.run_entropy_batch(A)
        J.run_entropy_batch(B)
        for _ in range(4):
            J.run_transform_batch(A)
            J.run_filter_color_batch(A)
            J.run_transform_batch(B)
            J.run_filter_color_batch(B)
        for c in A:
            assert np.array_equal(c.rgb8(), ref_a)
        for c in B:
            assert np.array_equal(c.rgb8(), ref_b)
        # a lender that is too small for the borrower's frame is refused at launch
        B[0].share_planes(None)
        A[0].share_planes(B[0])
        A[0].upload(fa)
        B[0].upload(fb)
        J.run_entropy_batch([A[0]])
        with pytest.raises(Exception):
            J.run_transform_batch([A[0]])
    finally:
        for c in A + B:
            c.close()


def test_async_filter_overlaps_next_entropy(built):
    """Option "filter_async": the filter launch of a frame set runs on a second stream while the next entropy launch of
    the same set is already queued; transforms, re-uploads and downloads still wait for it."""
    J = built
    datas = [J.encode_rgb8(J.synth_image(700, 500, seed=s), distance=d) for s, d in ((1, 1.0), (2, 2.0), (3, 1.0))]
    refs = [J.decode_rgb8(d) for d in datas]
    frames = [J.Frame(d) for d in datas]
    ctxs = [J.HipContext() for _ in range(3)]
    try:
        ctxs[0].set_option("filter_async", 1)
        for rot in range(3):  # every round decodes a different assignment of frames to contexts
            for i, c in enumerate(ctxs):
                c.upload(frames[(i + rot) % 3])
            for _ in range(3):
                J.run_entropy_batch(ctxs)
                J.run_transform_batch(ctxs)
                J.run_filter_color_batch(ctxs)
            J.run_entropy_batch(ctxs)  # queued behind nothing the filter needs
            for i, c in enumerate(ctxs):
                assert np.array_equal(c.rgb8(), refs[(i + rot) % 3])
    finally:
        for c in ctxs:
            c.close()


def test_batched_entropy_launch(built):
    """jxlhip_run_entropy_batch: frames of different geometry in one launch decode exactly as one by one, the batch
    description follows a re-upload, and the per-frame stages after it see the batch's coefficients."""
    import jxlo
    J = built
    streams = [J.encode_rgb8(J.synth_image(600, 400)), J.encode_rgb8(J.synth_image(1300, 520, seed=5), distance=2.0),
               J.encode_random(512, 300, seed=9), J.encode_rgb8(J.synth_image(64, 64, seed=3)),
               J.encode_rgb8(J.synth_image(900, 420, seed=8), num_passes=2, num_histograms=2)]  # progressive: lanes + pass merge
    ctxs = [J.HipContext() for _ in streams]
    frames = [J.Frame(s) for s in streams]
    try:
        for rounds in range(2):
            for c, f in zip(ctxs, frames):
                c.upload(f)
            J.run_entropy_batch(ctxs)
            if rounds == 0:  # per-frame downstream launches
                for c in ctxs:
                    c.run_transform()
                    c.run_filter_color()
            else:            # one launch per kernel for the whole set (mixed geometry and filter settings)
                J.run_transform_batch(ctxs)
                J.run_filter_color_batch(ctxs)
            for c, s in zip(ctxs, streams):
                c.sync()
                r, flags = c.errors()
                assert r == 0 and not any(flags)
                o = jxlo.Decoded(s)
                co = c.download("coeffs").astype(np.int32)
                ref = o.planes("coeffs")
                used = _used_mask(o)
                for g in range(o.info["num_groups"]):
                    assert np.array_equal(co[g, :, :used[g]], ref[g, :, :used[g]])
                d = np.abs(c.rgb8().astype(int) - o.rgb8.astype(int))
                assert d.max() <= 1
                o.close()
            # second round: every context takes another frame -> the cached batch description must be rebuilt
            frames = frames[1:] + frames[:1]
            streams = streams[1:] + streams[:1]
    finally:
        for c in ctxs:
            c.close()
        for f in frames:
            f.close()


@pytest.mark.parametrize("kind", ["image", "random_epf3", "chroma_420"])
def test_band_decode_matches_whole_frame(built, kind):
    """One frame split into bands of group rows (the multi-GPU split of a large frame, here on one device): every band
    context produces exactly the rows of the whole-frame decode, with no data exchanged between the contexts."""
    from libjxl_amd import sharding
    J = built
    if kind == "image":
        data = J.encode_rgb8(J.synth_image(1000, 1300, seed=11), distance=2.0)  # gab + EPF1 + EPF2: 4 rows of halo
    elif kind == "chroma_420":  # the vertical chroma upsampling reads one more (subsampled) row either side
        data = J.encode_rgb8(J.synth_image(700, 1100, seed=13), distance=2.0, color_transform=2, chroma_subsampling=4, strategy_mode=0)
    else:
        data = J.encode_random(700, 1100, seed=12, epf_iters=3)                 # all strategies, 7 rows of halo
    f = J.Frame(data, threads=2)
    ys = f.info["ysize"]
    rows = (ys + 255) // 256
    whole = J.decode_rgb8(data)
    for world in (2, 3, rows):
        stitched = np.zeros_like(whole)
        for rank in range(world):
            b0, b1 = sharding.band_of(rows, rank, world)
            if b0 == b1:
                continue
            c = J.HipContext()
            try:
                c.upload(f, band=(b0, b1))
                c.run_all()
                r, flags = c.errors()
                assert r == 0
                y0, y1 = b0 * 256, min(b1 * 256, ys)
                stitched[y0:y1] = c.rgb8_rows(y0, y1)
            finally:
                c.close()
        assert np.array_equal(stitched, whole), "band split over %d ranks differs from the whole-frame decode" % world
    f.close()


@pytest.mark.parametrize("kw", [dict(), dict(num_passes=2), dict(num_histograms=3, strategy_mode=2), dict(max_clusters=200, distance=0.5)])
def test_lane_kernel_with_alias_tables_in_global_memory(built, kw, monkeypatch):
    """Sets of frames whose alias tables would not leave every frame resident (libjxl-sized tables: 128+ clusters) run
    the lane kernel with the tables read in place (JXLHIP_GALIAS=1 forces that form): same coefficients, bit for bit."""
    import jxlo
    monkeypatch.setenv("JXLHIP_GALIAS", "1")
    _compare(built, jxlo, built.encode_rgb8(built.synth_image(600, 420, seed=13), **kw))


@pytest.mark.parametrize("kw,env", [(dict(max_clusters=128), {}), (dict(max_clusters=128, distance=0.5), dict(JXLHIP_LANES="64")),
                                    (dict(max_clusters=128, distance=0.3, num_passes=2), {}), (dict(max_clusters=100, num_histograms=3, strategy_mode=2), {}),
                                    (dict(max_clusters=128, distance=0.5, strategy_mode=0), dict(JXLHIP_LANES_CPP="1")),
                                    (dict(max_clusters=128, distance=4.0), dict(JXLHIP_LANES_CPP="1")), (dict(), {})])
def test_lane_kernel_with_six_byte_alias_tables(built, kw, env, monkeypatch, capfd):
    """Alias tables of up to 128 clusters x 2^6 slots (libjxl-sized) stay in LDS in a six-byte form when that keeps a launch
    resident (two frames per CU instead of one); JXLHIP_A6=2 picks the form whenever a frame is eligible (log_alpha 5 and 6
    here). Same coefficients, bit for bit, from the hand-written trip and from the C++ trip over that layout."""
    import jxlo
    monkeypatch.setenv("JXLHIP_A6", "2")
    monkeypatch.setenv("JXLHIP_PACK_DEBUG", "1")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    data = built.encode_rgb8(built.synth_image(1600, 1200, seed=21), **kw)
    f = built.Frame(data)
    assert f.info["log_alpha"] in (5, 6) and f.info["num_clusters"] <= 128
    f.close()
    _compare(built, jxlo, data)
    assert "tables lds6" in capfd.readouterr().err  # (the launch plan's own words: the form under test did run)


def test_six_byte_alias_tables_in_a_batched_launch(built, monkeypatch, capfd):
    """One launch over frames of different size whose tables differ in cluster count and log_alpha (5 and 6, as the benchmark's
    libjxl-sized streams do): every frame's workgroup stages its own tables in the six-byte form; coefficients and pixels as
    one by one."""
    import jxlo
    J = built
    monkeypatch.setenv("JXLHIP_A6", "2")
    monkeypatch.setenv("JXLHIP_PACK_DEBUG", "1")
    streams = [J.encode_rgb8(J.synth_image(1600, 1200, seed=21), max_clusters=128),                # log_alpha 5, 78 clusters
               J.encode_rgb8(J.synth_image(1200, 900, seed=22), max_clusters=128, distance=0.5),    # log_alpha 6
               J.encode_rgb8(J.synth_image(900, 700, seed=23), max_clusters=100, distance=0.3),
               J.encode_rgb8(J.synth_image(64, 64, seed=3), max_clusters=128)]
    frames = [J.Frame(s) for s in streams]
    assert {f.info["log_alpha"] for f in frames} >= {5, 6}
    ctxs = [J.HipContext() for _ in streams]
    try:
        for c, f in zip(ctxs, frames):
            c.upload(f)
        J.run_entropy_batch(ctxs)
        J.run_transform_batch(ctxs)
        J.run_filter_color_batch(ctxs)
        for c, s in zip(ctxs, streams):
            c.sync()
            r, flags = c.errors()
            assert r == 0 and not any(flags)
            o = jxlo.Decoded(s)
            co = c.download("coeffs").astype(np.int32)
            ref = o.planes("coeffs")
            used = _used_mask(o)
            for g in range(o.info["num_groups"]):
                assert np.array_equal(co[g, :, :used[g]], ref[g, :, :used[g]])
            assert np.abs(c.rgb8().astype(int) - o.rgb8.astype(int)).max() <= 1
            o.close()
        assert "tables lds6" in capfd.readouterr().err
    finally:
        for c in ctxs:
            c.close()
        for f in frames:
            f.close()


def test_colour_stage_closed_forms_gpu(built):
    """The colour kernel itself (k_color_out through jxlhip_debug_color) against the definition of XYB and the reference's
    closed-form colour tests (tests/color_kat.py: opsin_image_test.cc:28-135): roundtrip of the 13 colours of
    OpsinRoundtrip, every grey of VerifyGray and a 9x9x9 lattice through float64 forward XYB, linear and sRGB-encoded."""
    import color_kat
    J = built
    f = J.Frame(J.encode_rgb8(J.synth_image(64, 64, seed=1)))
    c = J.HipContext()
    try:
        c.upload(f)
        color_kat.check(lambda xyb, linear: c.debug_color(xyb, linear))
    finally:
        c.close()
        f.close()


def test_no_kernel_writes_outside_its_buffers(built, monkeypatch):
    """Debug build aid (JXLHIP_GUARD=1): every device buffer sits between two 4 KiB guard bands that no kernel may touch.
    A tour of the kernels: ragged sizes (also heights that are multiples of 64), every filter depth, upsampling, two
    passes, prefix / LZ77 codes, noise, int32 coefficients, and Modular frames with every feature."""
    J = built
    monkeypatch.setenv("JXLHIP_GUARD", "1")
    cases = [((257, 255), dict()), ((512, 512), dict(distance=2.0)), ((520, 300), dict(distance=4.5, gab=0)),
             ((300, 200), dict(upsampling=2)), ((520, 300), dict(num_passes=2)), ((300, 280), dict(ac_code_mode=3)),
             ((600, 400), dict(noise=40)), ((777, 513), dict(strategy_mode=2, random_cmap=1)), ((64, 64), dict(big_coeffs=1, strategy_mode=2))]
    for size, kw in cases:
        if kw.get("big_coeffs"):
            data = J.encode_random(size[0], size[1], **kw)
        else:
            data = J.encode_rgb8(J.synth_image(size[0], size[1], seed=5), **kw)
        f = J.Frame(data, threads=2)
        c = J.HipContext()
        try:
            c.upload(f)
            c.run_all()
            c.sync()
            assert c.check_guards() == 0, (size, kw, c.check_guards())
        finally:
            c.close()
            f.close()
    img = J.synth_image(700, 300, seed=4)
    for flags in (0, 16 | 4 | 8, 1 | 2 | 16):
        f = J.ModFrame(J.encode_lossless(img, flags))
        c = J.HipContext()
        try:
            c.upload_modular(f)
            c.run_modular()
            c.sync()
            assert c.check_guards() == 0, (flags, c.check_guards())
        finally:
            c.close()
            f.close()
    # the forward path (sharpening row kernel, transform tiles, device tokenisation) on ragged sizes, and the six-byte table
    # form of the lane kernel
    for size in ((301, 143), (57, 65), (640, 333), (8, 8)):
        c = J.HipContext()
        try:
            im = J.synth_image(size[0], size[1], seed=6)
            a = J.encode_rgb8_gpu(im, c, device_tokens=True, distance=1.0, cfl_fit=1)
            assert a == J.encode_rgb8_gpu(im, c, distance=1.0, cfl_fit=1)
            c.sync()
            assert c.check_guards() == 0, (size, c.check_guards())
        finally:
            c.close()
    monkeypatch.setenv("JXLHIP_A6", "2")
    f = J.Frame(J.encode_rgb8(J.synth_image(1600, 1200, seed=21), max_clusters=128, distance=0.5), threads=2)
    c = J.HipContext()
    try:
        c.upload(f)
        c.run_all()
        c.sync()
        assert c.check_guards() == 0, c.check_guards()
    finally:
        c.close()
        f.close()


def test_no_result_depends_on_bytes_outside_the_buffers(built, monkeypatch):
    """Stray READS: with JXLHIP_GUARD=1 a fresh device allocation is filled, guard bands and
    body, with JXLHIP_GUARD_BYTE. The same tour of the kernels is decoded under three patterns (0xA5, 0x00, 0xFF = NaN as a
    float): every output (inverse-transform planes, pixels; Modular samples) must be identical, so no kernel's result
    depends on what lies outside its buffers (up to 4 KiB either side) or on memory nothing has written. (The coefficient
    buffer itself is not compared: the tail of a partial group's area is never written, nor read.)"""
    import hashlib
    J = built
    monkeypatch.setenv("JXLHIP_GUARD", "1")
    cases = [((257, 255), dict()), ((520, 300), dict(distance=4.5, gab=0, epf_iters=3)), ((300, 200), dict(upsampling=4)),
             ((520, 300), dict(num_passes=2)), ((300, 280), dict(ac_code_mode=3)), ((600, 400), dict(noise=40)),
             ((777, 513), dict(strategy_mode=2, random_cmap=1))]
    streams = [J.encode_rgb8(J.synth_image(size[0], size[1], seed=5), **kw) for size, kw in cases]
    lossless = [J.encode_lossless(J.synth_image(700, 300, seed=4), flags) for flags in (0, 16 | 4 | 8, 1 | 2 | 16)]

    def tour():
        out = []
        for data in streams:
            f = J.Frame(data, threads=2)
            c = J.HipContext()
            try:
                c.upload(f)
                c.run_all()
                c.sync()
                assert c.check_guards() == 0
                h = hashlib.sha256()
                for a in (c.download("xyb_idct"), c.rgb8()):
                    h.update(np.ascontiguousarray(a).tobytes())
                out.append(h.hexdigest())
            finally:
                c.close()
                f.close()
        for data in lossless:
            out.append(hashlib.sha256(J.decode_lossless(data).tobytes()).hexdigest())
        return out

    results = []
    for byte in ("0xA5", "0x00", "0xFF"):
        monkeypatch.setenv("JXLHIP_GUARD_BYTE", byte)
        results.append(tour())
    assert results[0] == results[1] == results[2], [i for i in range(len(results[0])) if len({r[i] for r in results}) > 1]


def test_corrupt_sections_are_flagged_not_fatal(built):
    J = built
    data = bytearray(J.encode_rgb8(J.synth_image(520, 300)))
    f = J.Frame(bytes(data))
    n = len(data)
    for i in range(n - 3000, n - 2900):  # inside the AC group sections (they are last in the file)
        data[i] ^= 0xA5
    f2 = J.Frame(bytes(data))
    c = J.HipContext()
    c.upload(f2)
    c.run_all()
    r, flags = c.errors()
    assert r != 0 and any(flags)
    c.upload(f)
    c.run_all()
    r, flags = c.errors()
    assert r == 0
    c.close()


def test_jxl_decoder_api_full_image(built):
    """The drop-in boundary: same call sequence as lib/extras/dec/jxl.cc:140-669 with a thread runner."""
    import jxlo
    J = built
    L = J.lib()
    L.JxlDecoderCreate.restype = ctypes.c_void_p
    L.JxlDecoderCreate.argtypes = [ctypes.c_void_p]
    for n in ("JxlDecoderDestroy", "JxlDecoderProcessInput", "JxlDecoderCloseInput"):
        getattr(L, n).argtypes = [ctypes.c_void_p]
    L.JxlDecoderSubscribeEvents.argtypes = [ctypes.c_void_p, ctypes.c_int]
    L.JxlDecoderSetInput.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_size_t]
    L.JxlDecoderSetParallelRunner.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]

    class Fmt(ctypes.Structure):
        _fields_ = [("num_channels", ctypes.c_uint32), ("data_type", ctypes.c_int), ("endianness", ctypes.c_int), ("align", ctypes.c_size_t)]

    L.JxlDecoderImageOutBufferSize.argtypes = [ctypes.c_void_p, ctypes.POINTER(Fmt), ctypes.POINTER(ctypes.c_size_t)]
    L.JxlDecoderSetImageOutBuffer.argtypes = [ctypes.c_void_p, ctypes.POINTER(Fmt), ctypes.c_void_p, ctypes.c_size_t]
    data = J.encode_rgb8(J.synth_image(333, 222))
    ref = jxlo.Decoded(data, dumps=False).rgb8
    for nc, align in [(3, 0), (4, 0), (3, 64)]:
        dec = L.JxlDecoderCreate(None)
        pool = L.JxlThreadParallelRunnerCreate(None, 2)
        assert L.JxlDecoderSetParallelRunner(dec, ctypes.cast(L.JxlThreadParallelRunner, ctypes.c_void_p), pool) == 0
        assert L.JxlDecoderSubscribeEvents(dec, 0x40 | 0x1000) == 0
        L.JxlDecoderSetInput(dec, data, len(data))
        L.JxlDecoderCloseInput(dec)
        assert L.JxlDecoderProcessInput(dec) == 0x40
        assert L.JxlDecoderProcessInput(dec) == 5  # JXL_DEC_NEED_IMAGE_OUT_BUFFER
        fmt = Fmt(nc, 2, 0, align)
        size = ctypes.c_size_t()
        assert L.JxlDecoderImageOutBufferSize(dec, ctypes.byref(fmt), ctypes.byref(size)) == 0
        stride = 333 * nc if not align else (333 * nc + align - 1) // align * align
        assert size.value == stride * 221 + 333 * nc
        buf = np.zeros(size.value, np.uint8)
        assert L.JxlDecoderSetImageOutBuffer(dec, ctypes.byref(fmt), buf.ctypes.data, size.value) == 0
        assert L.JxlDecoderProcessInput(dec) == 0x1000  # JXL_DEC_FULL_IMAGE
        assert L.JxlDecoderProcessInput(dec) == 0
        rows = np.stack([buf[y * stride: y * stride + 333 * nc].reshape(333, nc) for y in range(222)])
        assert np.abs(rows[..., :3].astype(int) - ref.astype(int)).max() <= 1
        if nc == 4:
            assert (rows[..., 3] == 255).all()
        L.JxlDecoderDestroy(dec)
        L.JxlThreadParallelRunnerDestroy(pool)
    # an upsampled frame: the caller sees the image size, not the coded frame's
    data = J.encode_rgb8(J.synth_image(333, 222), upsampling=2)
    ref = jxlo.Decoded(data, dumps=False).rgb8
    dec = L.JxlDecoderCreate(None)
    assert L.JxlDecoderSubscribeEvents(dec, 0x1000) == 0
    L.JxlDecoderSetInput(dec, data, len(data))
    L.JxlDecoderCloseInput(dec)
    assert L.JxlDecoderProcessInput(dec) == 5
    fmt = Fmt(3, 2, 0, 0)
    size = ctypes.c_size_t()
    assert L.JxlDecoderImageOutBufferSize(dec, ctypes.byref(fmt), ctypes.byref(size)) == 0
    assert size.value == 333 * 222 * 3
    buf = np.zeros(size.value, np.uint8)
    assert L.JxlDecoderSetImageOutBuffer(dec, ctypes.byref(fmt), buf.ctypes.data, size.value) == 0
    assert L.JxlDecoderProcessInput(dec) == 0x1000
    assert np.abs(buf.reshape(222, 333, 3).astype(int) - ref.astype(int)).max() <= 1
    L.JxlDecoderDestroy(dec)


# ---- closed-form known answers through the HIP kernels (the same streams and bar as tests/test_oracle.py)
@pytest.mark.parametrize("strategy", [0, 4, 5, 6, 7, 8, 9, 10, 11, 18, 19, 20, 21, 22, 23, 24, 25, 26])
def test_idct_basis_functions_gpu(built, strategy):
    """Every coefficient position of every DCT-family strategy, one per varblock: the block the transform kernels produce
    must be that position's float64 basis function (lib/jxl/dct_for_test.h:23-94) within the reference's per-basis-vector
    bar 1e-7 * N (dct_test.cc:191-216). Covers k_idct_fast (8..32), k_dct (64 class) and k_dct_big (128 / 256 class),
    the scan-order coefficient layout and the coefficient orders, against a closed form instead of the oracle."""
    from test_oracle import basis_stream, check_basis_planes
    J = built
    data, blocks, shape = basis_stream(J, strategy)
    f = J.Frame(data)
    c = J.HipContext()
    try:
        c.upload(f)
        c.run_entropy()
        c.run_transform()
        c.sync()
        r, flags = c.errors()
        assert r == 0
        worst = check_basis_planes(c.download("xyb_idct")[1], blocks, shape)
    finally:
        c.close()
        f.close()
    assert worst < 2e-7 * max(shape) + 1e-6, worst


def test_16k_frame_whole_and_in_8_bands(built):
    """BASELINE.json configs[2]: a 16384x16384 d1.0 frame. Whole-frame decode against the oracle (coefficients bit-exact,
    RGB8 within one level on < 0.1 % of the samples), then the 8-band split the multi-GPU mode uses (band r = rows of
    groups sharding.band_of(64, r, 8), decoded with one group row of overlap either side, no exchange): the stitched bands
    must be bit-identical to the whole-frame decode."""
    import jxlo
    from libjxl_amd import sharding
    J = built
    N = 16384
    data = J.encode_rgb8(J.synth_image(N, N, seed=21), distance=1.0)
    f = J.Frame(data, threads=16)
    assert f.info["num_groups"] == 4096 and f.info["num_dc_groups"] == 64
    c = J.HipContext()
    try:
        c.upload(f)
        c.run_entropy()
        c.sync()
        r, flags = c.errors()
        assert r == 0 and not any(flags)
        co = c.download("coeffs")
        c.run_transform()
        c.run_filter_color()
        c.sync()
        whole = c.rgb8()
        o = jxlo.Decoded(data)
        ref = o.planes("coeffs")
        used = _used_mask(o)
        for g in range(0, 4096, 37):  # every 37th group: ~110 groups spread over the frame
            assert np.array_equal(co[g, :, :used[g]].astype(np.int32), ref[g, :, :used[g]]), "coefficients differ in group %d" % g
        del co, ref
        want = o.rgb8
        bad = 0
        for y0 in range(0, N, 2048):  # row blocks: keeps the int conversions small
            d = np.abs(whole[y0:y0 + 2048].astype(np.int16) - want[y0:y0 + 2048].astype(np.int16))
            assert d.max() <= 1
            bad += int((d > 0).sum())
        assert bad < 1e-3 * N * N * 3
        o.close()
        del want
        for rank in range(8):
            b0, b1 = sharding.band_of(64, rank, 8)
            assert b1 - b0 == 8
            c.upload(f, band=(b0, b1))
            c.run_all()
            rows = c.rgb8_rows(b0 * 256, b1 * 256)
            assert np.array_equal(rows, whole[b0 * 256:b1 * 256]), "band %d differs from the whole-frame decode" % rank
    finally:
        c.close()
        f.close()


@pytest.mark.parametrize("kw", [dict(gab=1, epf_iters=1), dict(gab=0, epf_iters=3), dict(gab=1, epf_iters=2), dict(gab=1, epf_iters=0),
                                dict(gab=1, epf_iters=3, distance=3.0)])
def test_filter_kernels_against_a_float64_third_reading(built, kw):
    """The HIP filter kernels (k_filter_rows2 for Gaborish + EPF1, k_filter_fused for the other combinations) against
    tests/filters_f64.py, the float64 NumPy restatement of stage_gaborish.cc / stage_epf.cc that neither the oracle nor the
    kernels share code with: the kernels' own inverse-transform output and their own 1 / sigma go in, their filtered
    planes must match to 2e-5 (the bar of the oracle comparison; observed ~1e-6)."""
    import filters_f64 as F
    import jxlo
    J = built
    data = J.encode_rgb8(J.synth_image(331, 245, seed=11), **kw)
    f = J.Frame(data, threads=2)
    o = jxlo.Decoded(data)
    i = o.info
    sig = o.buffer("inv_sigma")
    sig = (np.zeros(i["xsize_blocks"] * i["ysize_blocks"], np.float32) if sig is None else sig).reshape(i["ysize_blocks"], -1)[:, :i["xsize_blocks"]]
    o.close()
    c = J.HipContext()
    try:
        c.set_option("keep_filtered", 1)
        c.upload(f)
        c.run_all()
        c.sync()
        idct = c.download("xyb_idct")
        got = c.download("xyb_filtered")[:, :i["ysize"], :i["xsize"]]
    finally:
        c.close()
        f.close()
    want = F.loop_filters(idct, sig, i["xsize"], i["ysize"], i["gab"], i["epf_iters"])
    assert np.abs(got - want).max() < 2e-5


@pytest.mark.parametrize("kw", [dict(), dict(epf_iters=3), dict(gab=0, epf_iters=2), dict(epf_iters=2)])
def test_bands_with_halo_exchange_match_the_whole_frame(built, kw):
    """The multi-GPU band split without redundant decoding (option "band_halo", jxlhip_halo_*): three bands of
    a 700x1500 frame (6 rows of groups), each in a context of its own that entropy-decodes and transforms ONLY its own
    rows of groups; after the transform stage the boundary rows (Gaborish 1 + EPF up to 6) move between the contexts as
    dense device blocks through sharding.exchange_halos (here device-to-device inside one process; between GPUs the same
    blocks go through RCCL send / recv), then each band is filtered. The stitched bands are bit-identical to the
    whole-frame decode, for every filter depth."""
    import ctypes
    from libjxl_amd import sharding
    J = built
    hip = ctypes.CDLL("libamdhip64.so")
    blocks = []

    def device_block(nbytes):  # plain device memory for a halo block (what an RCCL send / recv would move between GPUs)
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(nbytes)) == 0
        blocks.append(p)
        return p

    data = J.encode_rgb8(J.synth_image(700, 1500, seed=33), **kw)
    f = J.Frame(data, threads=4)
    c = J.HipContext()
    c.upload(f)
    c.run_all()
    whole = c.rgb8()
    c.close()
    world = 3
    bands = [sharding.band_of(6, r, world) for r in range(world)]
    ctxs = []
    for r in range(world):
        b = J.HipContext()
        b.set_option("band_halo", 1)
        b.upload(f, band=bands[r])
        b.run_entropy()
        b.run_transform()
        ctxs.append(b)
    rows = ctxs[0].halo_rows()
    assert rows == 1 * (kw.get("gab", 1) != 0) + {0: 0, 1: 2, 2: 3, 3: 6}[kw.get("epf_iters", 1)]
    n = ctxs[0].halo_floats()
    mailbox = {}

    def run_rank(r):
        # (one process plays every rank in turn: `send` leaves the block in a mailbox, `recv` of the peer picks it up; the
        # two passes below stand for the two sides of each blocking send / recv pair)
        def pack(side):
            t = device_block(n * 4)
            ctxs[r].halo_pack(side, t, n * 4)
            return t

        def unpack(side, block):
            ctxs[r].halo_unpack(side, block, n * 4)

        return pack, unpack

    for r in range(world):  # every rank posts what its neighbours need ...
        pack, _ = run_rank(r)
        if r > 0:
            mailbox[(r, r - 1)] = pack(0)
        if r + 1 < world:
            mailbox[(r, r + 1)] = pack(1)
    for r in range(world):  # ... and takes what they posted, in the order exchange_halos prescribes
        pack, unpack = run_rank(r)
        sharding.exchange_halos(r, world, lambda side: None, unpack, lambda block, peer: None, lambda peer, r=r: mailbox[(peer, r)])
    for r in range(world):
        ctxs[r].run_filter_color()
        ctxs[r].sync()
        b0, b1 = bands[r]
        got = ctxs[r].rgb8_rows(b0 * 256, min(b1 * 256, 1500))
        assert np.array_equal(got, whole[b0 * 256:min(b1 * 256, 1500)]), "band %d differs from the whole-frame decode" % r
        ctxs[r].close()
    f.close()
    for p in blocks:
        hip.hipFree(p)


@pytest.mark.parametrize("env", [dict(AMD_SERIALIZE_KERNEL="3"), dict(GPU_MAX_HW_QUEUES="1"), dict(HIP_LAUNCH_BLOCKING="1")])
def test_the_pipelined_bench_ends_when_the_runtime_runs_kernels_one_at_a_time(built, env):
    """The entropy gate is a device-side wait without a bound (jxl_hip_api.hip EntropyGate): under a runtime that runs
    one kernel at a time a launch gated on ANOTHER launch's workgroups being resident may wait for a kernel that is not
    allowed to start. The library therefore drops the gate in such processes (RuntimeMaySerialise) whatever the caller
    asked for: the default pipelined schedule of bench.py, which asks for the gate, must end under each of the
    serialising switches. A child process under a timeout, small frames."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--size", "768x512", "--batch", "12", "--distinct", "2", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--e2e-frames", "0"]
    r = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["value"] > 0 and line["steps"] == 3


@pytest.mark.parametrize("filter_async", [0, 1])
def test_band_halos_through_an_asynchronous_transport(built, filter_async):
    """The set forms jxlhip_halo_pack_batch / _unpack_batch with a transport that is what RCCL is to this library: work on
    ANOTHER stream that the host does not wait for (a receive that has merely been enqueued). Two bands
    of a 600x1100 frame, three frames per band set, batched launches. The "network" is a second torch stream that first
    sleeps (the host runs far ahead of it), then copies the packed blocks into receive buffers that start out as NaN; the
    unpack and the filter launch are enqueued while that stream is still asleep. Any missing dependency (unpack before the
    copy has landed, filter before the unpack) leaves NaN or stale rows at the band edges and the bands stop matching the
    whole-frame decode. Both filter placements: the set's own stream and its second stream (option "filter_async")."""
    import torch
    from libjxl_amd import sharding
    J = built
    H, W = 1100, 600
    frames = [J.Frame(J.encode_rgb8(J.synth_image(W, H, seed=70 + i), distance=1.0 + i), threads=4) for i in range(3)]
    whole = []
    for f in frames:
        c = J.HipContext()
        c.upload(f)
        c.run_all()
        whole.append(c.rgb8())
        c.close()
    world = 2
    bands = [sharding.band_of((H + 255) // 256, r, world) for r in range(world)]
    sets = []
    for r in range(world):
        cs = [J.HipContext() for _ in frames]
        if filter_async:
            cs[0].set_option("filter_async", 1)
        for c, f in zip(cs, frames):
            c.set_option("band_halo", 1)
            c.upload(f, band=bands[r])
        sets.append(cs)
    n = max(c.halo_floats() for cs in sets for c in cs)
    net = torch.cuda.Stream()
    for cs in sets:
        J.run_entropy_batch(cs)
        J.run_transform_batch(cs)
    # (the blocks exist, filled with NaN, before anything is enqueued: the fills are not part of what is being ordered)
    blocks = [torch.full((len(frames) * n,), float("nan"), dtype=torch.float32, device="cuda") for _ in range(4)]
    torch.cuda.synchronize()
    with torch.cuda.stream(net):
        for it, (r, peer, side) in enumerate(((0, 1, 1), (1, 0, 0))):  # band 0's last rows to band 1's top; band 1's first rows to band 0's bottom
            sent, got = blocks[2 * it], blocks[2 * it + 1]
            J.halo_pack_batch(sets[r], side, sent.data_ptr(), n * 4, net.cuda_stream)
            torch.cuda._sleep(200_000_000)  # ~0.1 s of "network latency" on the transport's stream
            got.copy_(sent, non_blocking=True)
            J.halo_unpack_batch(sets[peer], 1 - side, got.data_ptr(), n * 4, net.cuda_stream)
    for cs in sets:  # enqueued at once: nothing above has waited for the device
        J.run_filter_color_batch(cs)
    for r, cs in enumerate(sets):
        b0, b1 = bands[r]
        for i, c in enumerate(cs):
            c.sync()
            got = c.rgb8_rows(b0 * 256, min(b1 * 256, H))
            assert np.array_equal(got, whole[i][b0 * 256:min(b1 * 256, H)]), "band %d of frame %d differs from the whole-frame decode" % (r, i)
    torch.cuda.synchronize()
    for cs in sets:
        for c in cs:
            c.close()
    for f in frames:
        f.close()


@pytest.mark.parametrize("kw", [dict(), dict(distance=3.0, epf_iters=3), dict(distance=0.5, epf_iters=2), dict(size=(1000, 700)),
                                dict(skip_dc_smoothing=1, custom_cmap=1), dict(custom_lf=1, custom_cmap=1, size=(2300, 300))])
def test_dc_path_kernels_against_the_oracle_and_a_float64_reading(built, kw):
    """Row a12 on the device (csrc/hip/jxl_hip_dc.h): the host front-end hands over the coded DC integers and the
    sharpness field; k_dc_dequant (DequantDC, compressed_dc.cc:201-296; also with non-default DC steps and chroma-from-luma
    DC factors, and across several DC groups), k_dc_smooth (AdaptiveDCSmoothing, :64-198) and k_epf_sigma (ComputeSigma,
    epf.cc:39-81) run inside the upload. What the transform and filter stages then read is compared with the oracle's
    planes (2e-6, the bar VERDICT round 2 item 4 sets) and with the float64 reading of tests/filters_f64.py."""
    import filters_f64 as F
    import jxlo
    J = built
    kw = dict(kw)
    w, h = kw.pop("size", (403, 277))
    data = J.encode_rgb8(J.synth_image(w, h, seed=5), **kw)
    o = jxlo.Decoded(data)
    i = o.info
    yb, xb = i["ysize_blocks"], i["xsize_blocks"]
    p = o.dc_params
    raw, dc_o, sig_o = o.buffer("dc_unsmoothed"), o.buffer("dc").reshape(3, yb, xb), o.buffer("inv_sigma").reshape(yb, xb)
    raw = None if raw is None else raw.reshape(3, yb, xb)  # (kSkipAdaptiveDCSmoothing: nothing was smoothed)
    acs, quant, sharp = o.buffer("acs").reshape(yb, xb), o.buffer("quant").reshape(yb, xb), o.buffer("sharpness").reshape(yb, xb)
    o.close()
    f = J.Frame(data, threads=2)
    c = J.HipContext()
    try:
        c.upload(f)
        dc_g = c.download("dc").reshape(3, yb, xb)
        sig_g = c.download("inv_sigma").reshape(yb, xb)
    finally:
        c.close()
        f.close()
    assert np.abs(dc_g - dc_o).max() < 2e-6 and np.abs(sig_g / sig_o - 1.0).max() < 2e-6
    if raw is None:
        assert np.array_equal(dc_g, dc_o)  # DequantDC alone: the same operations in the same order
    else:
        assert np.abs(dc_g - F.dc_smoothing(raw, p["dc_step"])).max() < 2e-6
    assert np.abs(sig_g / F.inv_sigma_blocks(acs, quant, sharp, p["quant_scale"], p["epf_quant_mul"], p["epf_sharp_lut"]) - 1.0).max() < 2e-6
