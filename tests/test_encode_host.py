"""CPU tests of the encode seam: the forward hook of the stream writer, without a GPU."""
import ctypes

import numpy as np
import pytest


@pytest.mark.parametrize("kw", [dict(), dict(strategy_mode=0, distance=2.0), dict(gab=0, distance=0.5)])
def test_forward_hook_with_the_cpu_form_writes_the_same_stream(built, kw):
    """jxlenc_encode_rgb8_forward hands the pixel-domain half to a function with jxlhip_enc_forward's signature; given
    the CPU form of that function (jxlenc_forward_cpu) the stream is byte-identical to jxlenc_encode_rgb8's: the
    descriptor (quantiser parameters, dequantisation tables) and the model hand-over lose nothing."""
    J = built
    E = J._enc_lib()
    img = J.synth_image(301, 143, seed=3)
    pp = ctypes.POINTER(ctypes.POINTER(ctypes.c_uint8))
    E.jxlenc_encode_rgb8_forward.argtypes = [ctypes.c_char_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.POINTER(J.EncParams), ctypes.c_void_p,
                                             ctypes.c_void_p, pp, ctypes.POINTER(ctypes.c_size_t), ctypes.POINTER(ctypes.c_double)]
    p = J._params(**kw)
    out, n = ctypes.POINTER(ctypes.c_uint8)(), ctypes.c_size_t()
    r = E.jxlenc_encode_rgb8_forward(img.tobytes(), 301, 143, ctypes.byref(p), ctypes.cast(E.jxlenc_forward_cpu, ctypes.c_void_p), None,
                                     ctypes.byref(out), ctypes.byref(n), None)
    assert J._finish(E, r, out, n, "jxlenc_encode_rgb8_forward") == J.encode_rgb8(img, **kw)


def test_forward_model_shapes_and_refusals(built):
    J = built
    img = J.synth_image(300, 270, seed=4)
    m = J.enc_forward_model(img)  # CPU form
    assert m["acs"].shape == (34, 38) and m["coeffs"].shape == (4, 3, 65536)
    first = (m["acs"] & 1) == 1
    assert (m["qf"][first] >= 1).all() and (m["qf"][~first] == 0).all()
    # every block is covered by exactly one transform: the covered areas of the first blocks add up to the frame
    cov_x = np.array([1, 1, 1, 1, 2, 4, 1, 2, 1, 4, 2, 4, 1, 1, 1, 1, 1, 1, 8, 4, 8, 16, 8, 16, 32, 16, 32])
    cov_y = np.array([1, 1, 1, 1, 2, 4, 2, 1, 4, 1, 4, 2, 1, 1, 1, 1, 1, 1, 8, 8, 4, 16, 16, 8, 32, 32, 16])
    st = m["acs"][first] >> 1
    assert (cov_x[st] * cov_y[st]).sum() == 34 * 38
    with pytest.raises(J.JxlAmdError):
        J.enc_forward_model(img, None, strategy_mode=2)  # the random tiling mode is not a forward-path mode
