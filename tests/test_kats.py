"""Known-answer tests taken from the reference's own test suite, run against BOTH the product's host front-end
(libjxl_amd/csrc/host/jxh_*.h) and the oracle (oracle/jxlo_*.h) -- each compiled into its own copy of tests/c/host_kats.cc,
so that neither is only ever compared with its twin:
  * alias-table invariants (lib/jxl/ans_common_test.cc:26-44), hybrid-uint worked examples and round trip
    (lib/jxl/dec_ans.h:47-67, entropy_coder_test.cc:20-59), Lehmer codes (lehmer_code_test.cc);
  * the bit reader's byte-order known answers (bit_reader_test.cc:28-255), the U32 / U64 / F16 field coders with the bit
    counts the reference's test states (fields_test.cc:57-209), and the dequantisation-table builder on the reference's
    DCTUniform vector (quant_weights_test.cc:185-271: every entry of every table is 4);
  * the fjxl fixtures (output of the reference's enc_fast_lossless.cc): bit reader, field coders, headers, TOC, prefix
    codes, LZ77, hybrid uint, context maps, MA-tree Modular decode, RCT and Palette of each front-end must reproduce the
    encoder's input image exactly.
"""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
_BINS = {}


def _binary(which):
    if which not in _BINS:
        out = "/tmp/libjxl_amd_kats_%s_%d" % (which, os.getpid())
        cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-function", os.path.join(ROOT, "tests", "c", "host_kats.cc"), "-o", out]
        if which == "oracle":
            cmd.insert(1, "-DKAT_ORACLE")
        subprocess.run(cmd, check=True)
        _BINS[which] = out
    return _BINS[which]


@pytest.mark.parametrize("which", ["product", "oracle"])
@pytest.mark.parametrize("kat", ["alias", "hybrid", "lehmer", "fastmath", "bits", "fields", "quant"])
def test_reference_closed_form_kats(which, kat):
    r = subprocess.run([_binary(which), kat], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr


def test_dct_scale_tables_match_the_reference_source():
    """dct_scales.h:42-353 as the reference lists them (ref_constant_floats.json): the IDCT kernels' butterfly multipliers
    (WcTable<N> literals of jxl_hip_kernels.h = WcMultipliers<N>) and the scales of the LLF-from-DC step, which the oracle
    (jxlo_vardct.h ResampleScale) and the HIP layer (jxl_hip_api.hip: c_resample) both compute from the formula in the
    reference's comment: that formula must reproduce the reference's DCTResampleScales<n, 8n> tables."""
    import math
    import re
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constant_floats.json")))
    text = open(os.path.join(ROOT, "libjxl_amd", "csrc", "hip", "jxl_hip_kernels.h")).read()
    seen = 0
    for n, body in re.findall(r"struct WcTable<(\d+)>\s*\{[^{]*\{(.*?)\};", text, re.S):
        vals = [float(v) for v in re.findall(r"[-+]?\d+\.\d+e[-+]\d+", body)]
        assert len(vals) == int(n) // 2
        if n in ref["wc_multipliers"]:
            want = ref["wc_multipliers"][n]
            assert len(want) == len(vals) and all(abs(a - b) <= 2e-9 * abs(b) for a, b in zip(vals, want)), n
            seen += 1
        else:  # WcTable<2>: 1 / (2 cos(pi / 4)), the same formula (the reference has no 2-point table)
            assert abs(vals[0] - 1.0 / (2.0 * math.cos(math.pi / 4))) < 1e-9
    assert seen == 5  # N = 4, 8, 16, 32, 64
    for n in (1, 2, 4, 8, 16, 32):
        N = 8.0 * n
        ours = [1.0 / (math.cos(i / (2 * N) * math.pi) * math.cos(i / N * math.pi) * math.cos(i / (N / 2) * math.pi)) for i in range(n)]
        want = ref["dct_resample_scales"]["%d_%d" % (n, 8 * n)]
        assert len(want) == n and all(abs(a - b) <= 1e-12 * abs(b) + 1e-15 for a, b in zip(ours, want)), n
    for f in ("oracle/jxlo_vardct.h", "libjxl_amd/csrc/hip/jxl_hip_api.hip"):  # both spell exactly that formula
        src = open(os.path.join(ROOT, f)).read()
        assert "std::cos(i / (2 * N) * M_PI) * std::cos(i / N * M_PI) * std::cos(i / (N / 2) * M_PI)" in src, f


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_float_constants_match_the_reference_source(which):
    """The floating-point constants the format fixes, as the reference's source lists them
    (tests/golden/ref_constant_floats.json, extracted by tests/golden/make_float_tables_golden.py: default upsampling
    weights, the dither table, the AFV basis, inverse opsin matrix and bias, quant biases, DC quantisation steps, Gaborish and
    EPF defaults, the 17 default dequantisation-table definitions), against what each front-end carries: its default-constructed headers and the .inc tables next to it (the
    product's are what the kernels are built with). A transcription error in a copy, or one shared by all copies, fails
    here; self-consistency tests cannot see it."""
    import numpy as np
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_constant_floats.json")))
    r = subprocess.run([_binary(which), "defaults"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    got = {}
    for line in r.stdout.splitlines():
        name, *vals = line.split()
        got[name] = np.array([float(v) for v in vals])

    def same(name, want, tol=2e-7):
        want = np.asarray(want, np.float64)
        assert got[name].shape == want.shape, name
        assert np.all(np.abs(got[name] - np.float32(want)) <= tol * np.maximum(1.0, np.abs(want))), name

    for name in ("upsampling_weights2", "upsampling_weights4", "upsampling_weights8", "dither32", "afv_basis", "inverse_opsin",
                 "quant_bias", "epf_sharp_lut", "epf_channel_scale"):
        same(name, ref[name])
    same("opsin_bias", [-ref["opsin_bias"][0]] * 3)  # (the decoder's default is kNegOpsinAbsorbanceBiasRGB: opsin_params.h:58-60)
    same("gab_weights", [ref["gab_weight1"], ref["gab_weight2"]] * 3)
    same("epf_scalars", [ref["epf_quant_mul"], ref["epf_pass0_sigma_scale"], ref["epf_pass2_sigma_scale"], ref["epf_border_sad_mul"]])
    same("dc_quant", [1.0 / v for v in ref["inv_dc_quant"]])
    for k, want in enumerate(ref["quant_library"]):  # the 17 default dequantisation-table definitions (quant_weights.cc:533-1106)
        same("quant_library_%d" % k, want, tol=1e-6)
    # the two EPF constants are literals in the code (epf.h:19-22)
    import re
    files = (["oracle/jxlo_render.h"] if which == "oracle" else
             ["libjxl_amd/csrc/hip/jxl_hip_dc.h", "libjxl_amd/csrc/hip/jxl_hip_filter_fused.h"])
    text = "".join(open(os.path.join(ROOT, f)).read() for f in files)
    # splines' channel weights (splines.cc:248) and the noise stage's constants (stage_noise.cc:147-193): literals as well
    sp = open(os.path.join(ROOT, "oracle/jxlo_splines.h" if which == "oracle" else "libjxl_amd/csrc/host/jxh_splines.h")).read()
    cw = [float(v) for v in re.findall(r"[\d.]+", re.search(r"kChannelWeight\[4\]\s*=\s*\{([^}]*)\}", sp).group(1).replace("f", ""))]
    assert cw == ref["spline_channel_weight"]
    nz = open(os.path.join(ROOT, "oracle/jxlo_render.h" if which == "oracle" else "libjxl_amd/csrc/hip/jxl_hip_filter_fused.h")).read()
    for const in (ref["noise_rg_corr"], ref["noise_rgn_corr"], ref["noise_norm_const"]):
        assert ("%sf" % repr(const)) in nz, const
    sig = {float(v) for v in re.findall(r"-1\.17157\d+", text)}
    mins = {float(v) for v in re.findall(r"-3\.90524\d+", text)}
    assert sig and mins and all(abs(v - ref["inv_sigma_num"]) < 1e-12 for v in sig) and all(abs(v - ref["min_sigma"]) < 1e-12 for v in mins)


def test_embedded_icc_profile_reference_vector():
    """lib/jxl/icc_codec_test.cc:52-211 (kEncodedTestProfile -> kTestProfile), extracted by tests/golden/make_icc_golden.py:
    the product's ICC decoder (41-context entropy decode + the inverse of the profile predictor) against the reference's
    own coded profile, plus 400 damaged copies."""
    g = os.path.join(ROOT, "tests", "golden")
    r = subprocess.run([_binary("product"), "icc", os.path.join(g, "ref_icc_test_profile.enc"), os.path.join(g, "ref_icc_test_profile.icc")],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr


MANIFEST = json.load(open(os.path.join(ROOT, "tests", "golden", "fjxl_manifest.json")))


@pytest.mark.parametrize("which", ["product", "oracle"])
@pytest.mark.parametrize("name", sorted(MANIFEST))
def test_front_end_decodes_reference_encoder_output(which, name, tmp_path):
    import make_fjxl_golden as G
    img = G.golden_image(name)
    assert hashlib.sha256(img.tobytes()).hexdigest() == MANIFEST[name]["pixels_sha256"]
    raw = os.path.join(str(tmp_path), "in.raw")
    img.tofile(raw)
    h, w, c = img.shape
    r = subprocess.run([_binary(which), "fjxl", os.path.join(ROOT, "tests", "golden", name + ".jxl"), raw, str(w), str(h), str(c)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr


def test_host_front_end_under_sanitizers_on_damaged_streams(built, tmp_path):
    """tests/c/host_fuzz.cc: the product's host parsers (image header + ICC, VarDCT frame plan with its host Modular
    decode of the DC groups, Modular frame plan, several frames) built with AddressSanitizer + UBSan, over streams of
    every feature the writer has, each damaged 150 times (seeded). Damage must end in jxh::Error; any sanitizer report
    or crash fails. (GPU sanitizers are not available on the pool: the device side has guard bands instead.)"""
    import numpy as np
    J = built
    out = os.path.join(str(tmp_path), "host_fuzz")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wall",
                    "-Wno-unused-function", os.path.join(ROOT, "tests", "c", "host_fuzz.cc"), "-o", out], check=True)
    img = J.synth_image(300, 200, seed=5)
    frames = [J.synth_image(200, 150, seed=s) for s in (1, 2)]
    J.set_embedded_icc(open(os.path.join(ROOT, "tests", "golden", "ref_icc_test_profile.enc"), "rb").read())
    try:
        with_icc = J.encode_rgb8(img)
    finally:
        J.set_embedded_icc(None)
    J.set_orientation(6)
    try:
        oriented = J.encode_lossless(img)
    finally:
        J.set_orientation(1)
    J.set_splines([dict(points=[(20, 30), (120, 90), (220, 40)], color=[[40] + [0] * 31, [300, 10] + [0] * 30, [0] * 32], sigma=[12] + [0] * 31),
                   dict(points=[(50, 180), (150, 120)], color=[[0] * 32, [-200] + [0] * 31, [100] + [0] * 31], sigma=[6, 2] + [0] * 30)])
    try:
        with_splines = [J.encode_rgb8(img), J.encode_lossless(img)]
    finally:
        J.set_splines(None)
    with_splines.append(open(os.path.join(ROOT, "tests", "golden", "ref_wasm_splines.jxl"), "rb").read())
    with_splines.append(J.encode_lossless(img, J.MODULAR_XYB | J.LOSSLESS_SQUEEZE))
    with_splines.append(J.encode_patched(img, J.synth_image(64, 48, seed=9), [dict(x0=4, y0=6, xsize=20, ysize=16, positions=[(10, 10, 2, 0), (250, 170, 1, 0)]),
                                                                            dict(x0=30, y0=0, xsize=30, ysize=40, positions=[(60, 120, 3, 1)])]))
    patch = J.synth_image(80, 60, seed=2)
    ramp = ((np.mgrid[0:60, 0:80][1] * 255) // 79).astype(np.uint8)
    with_splines.append(J.encode_layers([dict(img=np.dstack([img, img[..., 0]]), save_as=1, duration=2),
                                         dict(img=np.dstack([patch, ramp]), x0=-10, y0=100, mode=2, alpha_mode=2, source=1, save_as=1),
                                         dict(img=np.dstack([patch, ramp]), x0=250, y0=-20, mode=4, alpha_mode=3, source=1, clamp=1, duration=1)],
                                        tps=(10, 1), lossless=True))
    streams = with_splines + [J.encode_rgb8(img), J.encode_rgba8(np.dstack([img, img[..., 0]])), J.encode_random(264, 200, seed=3),
               J.encode_rgb8(img, ac_code_mode=3, num_passes=2, custom_orders=1, custom_bctx=1, noise=50), J.encode_rgb8(img, upsampling=2),
               J.encode_lossless(img, J.LOSSLESS_RCT | J.LOSSLESS_SQUEEZE | J.LOSSLESS_WP),
               J.encode_lossless(np.dstack([img, img[..., 1]]), J.LOSSLESS_RCT), J.encode_animation(frames, [1, 2]),
               J.encode_animation(frames, [1, 2], lossless=True), with_icc, oriented,
               J.encode_with_dc_frame(img), J.encode_with_dc_frame(img, dc_vardct=True),
               # round 4: images that are not XYB encoded, RAW tables, subsampled chroma, upsampled extra channels, alpha patches
               J.encode_rgb8(img, color_transform=2, chroma_subsampling=4, strategy_mode=0, raw_quant=1, custom_bctx=1),
               J.encode_random(264, 200, seed=5, color_transform=2, chroma_subsampling=0b011011),
               J.encode_rgb8(img, color_transform=1), J.encode_rgba8(np.dstack([img, img[..., 0]]), upsampling=2, ec_upsampling=4),
               J.encode_patched(np.dstack([img, img[..., 1]]), np.dstack([J.synth_image(64, 48, seed=9), np.full((48, 64), 99, np.uint8)]),
                                [dict(x0=4, y0=6, xsize=20, ysize=16, positions=[(10, 10, 4, 1, 5, 0), (250, 170, 7, 0, 3, 1)])], atlas_vardct=True),
               J.encode_patched(img, J.synth_image(64, 48, seed=9), [dict(x0=4, y0=6, xsize=20, ysize=16, positions=[(10, 10, 2, 0), (250, 170, 5, 0)])],
                                lossless=True, lossless_flags=J.LOSSLESS_RCT)]
    J.set_custom_upsampling(7, seed=3)
    try:
        streams.append(J.encode_rgb8(img, upsampling=4))
    finally:
        J.set_custom_upsampling(0)
    files = []
    for i, s in enumerate(streams):
        files.append(os.path.join(str(tmp_path), "s%d.jxl" % i))
        open(files[-1], "wb").write(s)
    r = subprocess.run([out, "150"] + files, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "refused" in r.stdout


def test_decoder_api_under_sanitizers_on_damaged_files(built, tmp_path):
    """tests/c/api_fuzz.cc: the decode.h state machine (container walk, chunked input, box buffers, header events, ICC
    getters, animation frame walk, SkipFrames) built with AddressSanitizer + UBSan around libjxl_amd/csrc/api/jxl_api.cc,
    over bare and boxed files, damaged and fed in random pieces. Every run must end (success, error or end of input)
    within a bounded number of calls; a sanitizer report, a crash or a decoder that keeps asking fails."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import replay_util as R
    J = built
    out = os.path.join(str(tmp_path), "api_fuzz")
    subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-Wall",
                    "-Wno-unused-function", "-Wno-subobject-linkage", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "api_fuzz.cc"), "-o", out, "-lpthread"], check=True)
    img = J.synth_image(300, 200, seed=5)
    frames = [J.synth_image(200, 150, seed=s) for s in (1, 2, 3)]
    J.set_embedded_icc(open(os.path.join(ROOT, "tests", "golden", "ref_icc_test_profile.enc"), "rb").read())
    try:
        with_icc = J.encode_rgb8(img)
    finally:
        J.set_embedded_icc(None)
    anim = J.encode_animation(frames, [1, 2, 3])
    import numpy as np
    patch = J.synth_image(80, 60, seed=2)
    ramp = ((np.mgrid[0:60, 0:80][1] * 255) // 79).astype(np.uint8)
    layered = J.encode_layers([dict(img=np.dstack([img, img[..., 0]]), save_as=1),
                               dict(img=np.dstack([patch, ramp]), x0=-10, y0=100, mode=2, alpha_mode=2, source=1)], lossless=True)
    files = []
    for i, s in enumerate([J.encode_rgb8(img), R.container(J.encode_rgb8(img)), R.container(with_icc, pieces=3), anim,
                           R.container(anim, pieces=4), R.container(J.encode_lossless(img), pieces=1), layered,
                           J.encode_with_dc_frame(img)]):
        files.append(os.path.join(str(tmp_path), "f%d.jxl" % i))
        open(files[-1], "wb").write(s)
    r = subprocess.run([out, "60"] + files, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    assert "reached JXL_DEC_SUCCESS" in r.stdout
