"""DC frames (what `cjxl --progressive_dc` emits): a kDCFrame (frame_header.h:319,439; 1/8 of the image per
level) is rendered like any frame and kept before its colour transform (dec_cache.cc:221-224); a later frame with
kUseDcFrame (frame_header.h:348) takes that image as its DC image instead of decoding one (passes_state.cc:62-77,
dec_frame.cc:319-326), without adaptive smoothing (:347-356). CPU part: the oracle and the host parse; GPU part: the planes
stay on the device (DC slot of the canvas -> JxlHipFrameDesc::dc_device) behind the decoder API, against the oracle."""
import numpy as np
import pytest

import replay_util as R


def _psnr(a, b):
    return 10 * np.log10(255.0 ** 2 / np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))


@pytest.mark.parametrize("kw", [dict(), dict(dc_vardct=True)])
def test_oracle_and_host_take_the_dc_image_from_a_dc_frame(built, kw):
    import jxlo
    J = built
    img = J.synth_image(333, 250, seed=12)
    data = J.encode_with_dc_frame(img, **kw)
    plain = J.encode_rgb8(img)
    o = jxlo.Decoded(data)
    assert o.out_size == (333, 250)
    # the picture is the image (its DC now comes from the 1/8-size frame: close to, not equal to, the plain stream's)
    assert _psnr(o.rgb8, img) > 24.0
    dc = o.buffer("dc").reshape(3, 32, 42)
    o.close()
    # the DC image IS the DC frame's output before the colour transform: the first frame decoded alone, as coded XYB
    first = (J.Frame if kw else J.ModFrame)(data)
    assert (first.info["xsize"], first.info["ysize"]) == (42, 32) and not first.is_last
    second = J.Frame(data, frame_pos=first.end, frame_index=1)
    assert (second.info["xsize"], second.info["ysize"]) == (333, 250) and second.is_last and second.end == len(data)
    first.close()
    second.close()
    assert len(data) < len(plain) + 4000 and np.isfinite(dc).all() and np.abs(dc[1]).max() > 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(dc_vardct=True), dict(epf_iters=2, size=(1000, 700))])
def test_dc_frames_through_the_gpu(built, tmp_path, kw):
    import jxlo
    J = built
    kw = dict(kw)
    w, h = kw.pop("size", (333, 250))
    img = J.synth_image(w, h, seed=12)
    data = J.encode_with_dc_frame(img, **kw)
    o = jxlo.Decoded(data)
    want8, wantf = o.rgb8.copy(), o.planes("rgbf").transpose(1, 2, 0).copy()
    o.close()
    rc, events, out, px = R.run(data, tmp_path, "f32", 3)
    assert rc == 0 and [e for e in events if e in ("FRAME", "FULL_IMAGE")] == ["FRAME", "FULL_IMAGE"], out
    got = np.frombuffer(px, np.float32).reshape(h, w, 3)
    assert np.abs(got - wantf).max() < 1e-4
    rc, events, out, px = R.run(R.container(data), tmp_path, "u8", 3, "chunk=4000")
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.uint8).reshape(h, w, 3).astype(int) - want8.astype(int)).max() <= 1


@pytest.mark.gpu
def test_a_frame_that_names_a_missing_dc_frame_is_refused(built, tmp_path):
    """passes_state.cc:70-75: kUseDcFrame without a decoded DC frame of that level is an error, not a guess."""
    J = built
    img = J.synth_image(200, 120, seed=3)
    data = J.encode_with_dc_frame(img)
    first = J.ModFrame(data)
    E = J._enc_lib()
    header = E.jxlenc_last_header_bytes()
    alone = data[:header] + data[first.end:]  # image header + the main frame only
    first.close()
    rc, events, out, px = R.run(alone, tmp_path, "u8", 3)
    assert rc != 0 and "FULL_IMAGE" not in events, out
