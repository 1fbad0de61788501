#!/usr/bin/env python3
"""Extracts the floating-point constants the format fixes from the reference's SOURCE into
tests/golden/ref_constant_floats.json, so that a test can hold every copy in this repository (host front-end, oracle, the
.inc tables the kernels are built with) to the numbers the reference lists, not to each other:
  default upsampling weights kWeights2 / 4 / 8                lib/jxl/image_metadata.cc:98-214
  blue-noise dither table kDither (32 rows of 32 + 16 repeat) lib/jxl/render_pipeline/stage_write.cc:59-257
  AFV basis k4x4AFVBasis                                      lib/jxl/dec_transforms-inl.h:96-385
  default inverse opsin matrix, opsin bias                    lib/jxl/cms/opsin_params.h:36-62
  kDefaultQuantBias                                           lib/jxl/quantizer.h:52-57
  kInvDCQuant (DC quantisation steps = 1 / these)             lib/jxl/quant_weights.h:289-299
  loop-filter defaults (Gaborish weights, EPF parameters)     lib/jxl/loop_filter.cc:28-87
  kInvSigmaNum, kMinSigma                                     lib/jxl/epf.h:19-22
  default dequantisation-table parameters (the "library")     lib/jxl/quant_weights.cc:533-1106
  WcMultipliers<N>, DCTResampleScales<FROM, TO>               lib/jxl/dct_scales.h:42-353
usage (where /root/reference exists): python tests/golden/make_float_tables_golden.py"""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/lib/jxl"
NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def text(path):
    return re.sub(r"//.*", "", open(os.path.join(REF, path)).read())


def floats(body):
    return [float(t) for t in re.findall(NUM, re.sub(r"(?<=[\d.])f\b", "", body))]


def array(src, name):
    m = re.search(r"%s(?:\s*\[[^\]]*\])*\s*(?:=\s*)?\{(.*?)\};" % re.escape(name), src, re.S)
    return floats(m.group(1))


def f16_defaults(src, field):
    """The default the visitor gives `field`: visitor->F16(<expr>, &field)."""
    m = re.search(r"F16\(\s*([^,]+?),\s*&%s\s*\)" % re.escape(field), src, re.S)
    expr = re.sub(r"(?<=[\d.])f\b", "", m.group(1)).strip()
    return float(eval(expr, {"__builtins__": {}}))  # (a literal or a product of two literals)


def main():
    out = {}
    md = text("image_metadata.cc")
    out["upsampling_weights2"] = array(md, "kWeights2")
    out["upsampling_weights4"] = array(md, "kWeights4")
    out["upsampling_weights8"] = array(md, "kWeights8")
    d = array(text("render_pipeline/stage_write.cc"), "kDither")
    assert len(d) == 48 * 32
    out["dither32"] = [d[r * 48 + c] for r in range(32) for c in range(32)]
    assert all(d[r * 48 + 32 + c] == d[r * 48 + c] for r in range(32) for c in range(16))  # the padding repeats the row
    out["afv_basis"] = array(text("dec_transforms-inl.h"), "k4x4AFVBasis")
    op = text("cms/opsin_params.h")
    out["inverse_opsin"] = array(op, "kDefaultInverseOpsinAbsorbanceMatrix")
    out["opsin_bias"] = floats(re.search(r"kOpsinAbsorbanceBias0\s*=\s*([^;]+);", op).group(1))
    qb = re.search(r"kDefaultQuantBias\[4\]\s*=\s*\{(.*?)\};", text("quantizer.h"), re.S).group(1)
    out["quant_bias"] = [float(eval(re.sub(r"(?<=[\d.])f\b", "", e).strip(), {"__builtins__": {}})) for e in qb.split(",") if e.strip()]
    out["inv_dc_quant"] = array(text("quant_weights.h"), "kInvDCQuant")
    lf = text("loop_filter.cc")
    out["gab_weight1"] = f16_defaults(lf, "gab_x_weight1")
    out["gab_weight2"] = f16_defaults(lf, "gab_x_weight2")
    for ch in "yb":
        assert f16_defaults(lf, "gab_%s_weight1" % ch) == out["gab_weight1"] and f16_defaults(lf, "gab_%s_weight2" % ch) == out["gab_weight2"]
    out["epf_channel_scale"] = [f16_defaults(lf, "epf_channel_scale[%d]" % i) for i in range(3)]
    out["epf_quant_mul"] = f16_defaults(lf, "epf_quant_mul")
    out["epf_pass0_sigma_scale"] = f16_defaults(lf, "epf_pass0_sigma_scale")
    out["epf_pass2_sigma_scale"] = f16_defaults(lf, "epf_pass2_sigma_scale")
    out["epf_border_sad_mul"] = f16_defaults(lf, "epf_border_sad_mul")
    assert re.search(r"float\(i\)\s*/\s*float\(kEpfSharpEntries - 1\)", lf)  # the sharpness LUT defaults to i / 7
    out["epf_sharp_lut"] = [i / 7.0 for i in range(8)]
    ep = text("epf.h")
    out["inv_sigma_num"] = floats(re.search(r"kInvSigmaNum\s*=\s*([^;]+);", ep).group(1))[0]
    out["min_sigma"] = floats(re.search(r"kMinSigma\s*=\s*([^;]+);", ep).group(1))[0]
    # the default dequantisation-table parameters ("library", quant_weights.cc:533-1106): per table kind, in the order of the
    # enum, every V(...) of its definition in source order (DCT kinds: the three channels' distance bands; IDENTITY / DCT2X2:
    # the per-channel weights; DCT4X4 / DCT4X8: bands then multipliers; AFV: the 27 weights, its bands being DCT4X8's and
    # DCT4X4's)
    qw = text("quant_weights.cc")
    lib = qw[qw.index("struct DequantMatricesLibraryDef"):qw.index("DequantMatricesLibraryDef::DCT()", qw.index("DCT128X256()"))]
    parts = re.split(r"static (?:constexpr )?QuantEncodingInternal (\w+)\(\)", lib)
    names, bodies = parts[1::2], parts[2::2]
    assert names == ["DCT", "IDENTITY", "DCT2X2", "DCT4X4", "DCT16X16", "DCT32X32", "DCT8X16", "DCT8X32", "DCT16X32", "DCT4X8", "AFV0",
                     "DCT64X64", "DCT32X64", "DCT128X128", "DCT64X128", "DCT256X256", "DCT128X256"], names
    out["quant_library"] = [[float(eval(re.sub(r"(?<=[\d.])f\b", "", e).strip(), {"__builtins__": {}})) for e in re.findall(r"\bV\(([^)]*)\)", b)]
                            for b in bodies]
    # IDCT butterfly multipliers and the resample scales of the LLF-from-DC step (dct_scales.h:42-353)
    ds = text("dct_scales.h")
    out["wc_multipliers"] = {n: floats(b) for n, b in re.findall(r"struct WcMultipliers<(\d+)>\s*\{[^{]*\{(.*?)\};", ds, re.S)}
    out["dct_resample_scales"] = {"%s_%s" % (a, b): floats(body)
                                  for a, b, body in re.findall(r"struct DCTResampleScales<(\d+),\s*(\d+)>\s*\{[^{]*\{(.*?)\};", ds, re.S)}
    # splines: kChannelWeight (splines.cc:248); noise: the correlation and normalisation constants (stage_noise.cc:147-193)
    out["spline_channel_weight"] = array(text("splines.cc"), "kChannelWeight")
    sn = text("render_pipeline/stage_noise.cc")
    out["noise_rg_corr"] = floats(re.search(r"kRGCorr\s*=\s*Set\(d,\s*([^)]+)\)", sn).group(1))[0]
    out["noise_rgn_corr"] = floats(re.search(r"kRGNCorr\s*=\s*Set\(d,\s*([^)]+)\)", sn).group(1))[0]
    out["noise_norm_const"] = floats(re.search(r"norm_const\s*=\s*Set\(d,\s*([^)]+)\)", sn).group(1))[0]
    sizes = {k: (len(v) if isinstance(v, list) else 1) for k, v in out.items()}
    assert (sizes["upsampling_weights2"], sizes["upsampling_weights4"], sizes["upsampling_weights8"], sizes["dither32"], sizes["afv_basis"],
            sizes["inverse_opsin"], sizes["quant_bias"], sizes["inv_dc_quant"]) == (15, 55, 210, 1024, 256, 9, 4, 3), sizes
    json.dump(out, open(os.path.join(HERE, "ref_constant_floats.json"), "w"), indent=0)
    print(sizes)


if __name__ == "__main__":
    main()
