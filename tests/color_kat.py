"""Closed-form expectations for the colour stage, taken from the reference's own tests (lib/jxl/opsin_image_test.cc:28-135,
lib/jxl/cms/transfer_functions_test.cc) and from the definition of XYB (lib/jxl/cms/opsin_params.h:19-45 absorbance matrix
and bias; lib/jxl/enc_xyb.cc:50-110: mixed = M * rgb + bias, cube root minus the bias's, X = (L - M) / 2, Y = (L + M) / 2,
B = S), in float64. No decoder produced any of these numbers."""
import numpy as np

M = np.array([[0.30, 1.0 - 0.078 - 0.30, 0.078],
              [0.23, 1.0 - 0.078 - 0.23, 0.078],
              [0.24342268924547819, 0.20476744424496821, 1.0 - 0.24342268924547819 - 0.20476744424496821]])
BIAS = 0.0037930732552754493


def linear_srgb_to_xyb(rgb):
    """rgb: [n, 3] linear sRGB in [0, 1] (intensity target 255) -> [3, n] XYB."""
    mixed = np.maximum(rgb.astype(np.float64) @ M.T + BIAS, 0.0)
    lms = np.cbrt(mixed) - np.cbrt(BIAS)
    return np.stack([0.5 * (lms[:, 0] - lms[:, 1]), 0.5 * (lms[:, 0] + lms[:, 1]), lms[:, 2]])


def srgb_encode(v):
    """IEC 61966-2-1, extended to negative values by odd symmetry like stage_from_linear.cc:42-54."""
    a = np.abs(v.astype(np.float64))
    return np.sign(v) * np.where(a <= 0.0031308, a * 12.92, 1.055 * np.power(np.maximum(a, 1e-30), 1 / 2.4) - 0.055)


def roundtrip_colors():
    """The colours of OpsinImageTest.OpsinRoundtrip, every grey of VerifyGray, and a lattice of the RGB cube."""
    cs = [(0, 0, 0), (1 / 255, 1 / 255, 1 / 255), (128 / 255, 128 / 255, 128 / 255), (1, 1, 1), (0, 0, 1 / 255), (0, 0, 128 / 255), (0, 0, 1),
          (0, 1 / 255, 0), (0, 128 / 255, 0), (0, 1, 0), (1 / 255, 0, 0), (128 / 255, 0, 0), (1, 0, 0)]
    cs += [(i / 255, i / 255, i / 255) for i in range(1, 255)]
    g = np.linspace(0, 1, 9)
    cs += [(r, gg, b) for r in g for gg in g for b in g]
    return np.array(cs, np.float64)


def check(convert):
    """convert(xyb [3, n] float32, linear) -> rgb [n, 3]: the decoder's colour stage under test."""
    # OpsinImageTest.VerifyOpsinAbsorbanceInverseMatrix, from the outside: the inverse of the stage is the absorbance matrix
    rgb = roundtrip_colors()
    xyb = linear_srgb_to_xyb(rgb)
    # VerifyZero / VerifyGray hold for the forward definition used here
    assert abs(xyb[:, 0]).max() < 1e-9
    greys = slice(13, 13 + 254)
    assert np.abs(xyb[0, greys]).max() < 1e-6 and np.abs(xyb[2, greys] / xyb[1, greys] - 1.0).max() < 3e-5
    lin = convert(xyb.astype(np.float32), True)
    # OpsinRoundtrip: the reference's bar is 1e-3; float32 arithmetic on [0, 1] does 2e-5
    assert np.abs(lin - rgb).max() < 2e-5, np.abs(lin - rgb).max()
    enc = convert(xyb.astype(np.float32), False)
    assert np.abs(enc - srgb_encode(rgb)).max() < 6e-5, np.abs(enc - srgb_encode(rgb)).max()
    return float(np.abs(lin - rgb).max()), float(np.abs(enc - srgb_encode(rgb)).max())
