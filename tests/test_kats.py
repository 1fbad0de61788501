"""Known-answer tests taken from the reference's own test suite, run against BOTH the product's host front-end
(libjxl_amd/csrc/host/jxh_*.h) and the oracle (oracle/jxlo_*.h) -- each compiled into its own copy of tests/c/host_kats.cc,
so that neither is only ever compared with its twin:
  * alias-table invariants (lib/jxl/ans_common_test.cc:26-44), hybrid-uint worked examples and round trip
    (lib/jxl/dec_ans.h:47-67, entropy_coder_test.cc:20-59), Lehmer codes (lehmer_code_test.cc);
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
@pytest.mark.parametrize("kat", ["alias", "hybrid", "lehmer"])
def test_reference_closed_form_kats(which, kat):
    r = subprocess.run([_binary(which), kat], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr


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
