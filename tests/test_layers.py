"""Frames composed on a canvas (SURVEY.md §8 f2 "blending", f4 "multi-frame coalescing"): crops, the five blend modes,
reference slots, invisible layers, animations whose frames are deltas over the previous canvas (lib/jxl/blending.cc,
alpha.cc, render_pipeline/stage_blending.cc, dec_cache.cc:268-290). CPU part: the oracle against the closed form of
alpha-over and the host's placement fields; GPU part: the canvas kernels (jxlhip_canvas_*) behind the decoder API
against the oracle."""
import numpy as np
import pytest

import replay_util as R

W, H = 200, 150


def _parts(J):
    base = J.synth_image(W, H, seed=1)
    opaque = np.full((H, W), 255, np.uint8)
    patch = J.synth_image(80, 60, seed=2)
    ramp = ((np.mgrid[0:60, 0:80][1] * 255) // 79).astype(np.uint8)
    return base, opaque, patch, ramp


def test_oracle_alpha_over_closed_form(built):
    """A patch with an alpha ramp blended (kBlend, alpha.cc:44-58: non-premultiplied) over an opaque base kept in slot 1:
    inside the patch rectangle out = fg * a + bg * (1 - a) (new alpha = 1), outside the base shows."""
    import jxlo
    J = built
    base, opaque, patch, ramp = _parts(J)
    data = J.encode_layers([dict(img=np.dstack([base, opaque]), save_as=1),
                            dict(img=np.dstack([patch, ramp]), x0=50, y0=40, mode=2, alpha_mode=2, source=1)], lossless=True)
    o = jxlo.Decoded(data, dumps=False)
    got = o.rgb8.astype(int)
    o.close()
    a = ramp.astype(np.float64)[..., None] / 255
    want = base.astype(np.float64).copy()
    want[40:100, 50:130] = patch * a + want[40:100, 50:130] * (1 - a)
    assert got.shape == (H, W, 4) and (got[..., 3] == 255).all()
    assert np.abs(got[..., :3] - np.rint(want)).max() <= 1
    assert np.array_equal(got[:40, :, :3], base[:40])  # untouched outside the rectangle


def test_host_reports_the_placement_of_every_frame(built, tmp_path):
    J = built
    base, opaque, patch, ramp = _parts(J)
    data = J.encode_layers([dict(img=np.dstack([base, opaque]), save_as=1, duration=2),
                            dict(img=np.dstack([patch, ramp]), x0=-10, y0=100, mode=2, alpha_mode=2, source=1, save_as=1, duration=0),
                            dict(img=np.dstack([patch, ramp]), x0=150, y0=-20, mode=1, alpha_mode=3, source=1, clamp=1, duration=1)],
                           tps=(10, 1), lossless=True)
    L = J.lib()
    import ctypes

    class Placement(ctypes.Structure):
        _fields_ = [("x0", ctypes.c_int32), ("y0", ctypes.c_int32)] + [(n, ctypes.c_uint32) for n in (
            "xsize", "ysize", "custom_size", "frame_type", "mode", "alpha_mode", "source", "alpha_source", "clamp", "alpha_clamp", "duration",
            "is_last", "save_as_reference", "save_before_color_transform", "dc_level", "use_dc_frame")]  # include/jxl_amd.h

    # (this mirror is written to by the C side: the sized form never writes more than the mirror holds, and the library
    # reports its own struct size, which must be the header's)
    L.jxlamd_sizeof_frame_placement.restype = ctypes.c_size_t
    assert L.jxlamd_sizeof_frame_placement() == ctypes.sizeof(Placement)
    L.jxlamd_modframe_placement_sized.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    L.jxlamd_modframe_placement_sized.restype = ctypes.c_size_t
    pos, seen = 0, []
    for k in range(3):
        f = J.ModFrame(data, frame_pos=pos, frame_index=k)
        p = Placement()
        assert L.jxlamd_modframe_placement_sized(f._h, ctypes.byref(p), ctypes.sizeof(p)) == ctypes.sizeof(p)
        short = (ctypes.c_uint32 * 6)(*([0xAAAAAAAA] * 6))  # an older caller's 16-byte struct + guard words: nothing past 16 bytes is written
        assert L.jxlamd_modframe_placement_sized(f._h, short, 16) == ctypes.sizeof(p)
        assert (short[2], short[3], short[4], short[5]) == (p.xsize, p.ysize, 0xAAAAAAAA, 0xAAAAAAAA)
        seen.append((p.x0, p.y0, p.xsize, p.ysize, p.mode, p.alpha_mode, p.source, p.duration, p.is_last, p.save_as_reference))
        pos = f.end
        f.close()
    assert seen == [(0, 0, W, H, 0, 0, 0, 2, 0, 1), (-10, 100, 80, 60, 2, 2, 1, 0, 0, 1), (150, -20, 80, 60, 1, 3, 1, 1, 1, 0)]


def _oracle_frames(data, n):
    import jxlo
    out = []
    for k in range(n):
        o = jxlo.Decoded(data, frame=k)
        f = o.planes("rgbf").transpose(1, 2, 0).copy()
        out.append((o.rgb8.copy(), f))
        o.close()
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("lossless,case", [(True, "blend"), (False, "blend"), (True, "add_mul"), (False, "outside"), (True, "premultiplied")])
def test_layered_stills_through_the_gpu(built, tmp_path, lossless, case):
    """Only the last frame of a layered still is shown; the layers before it are decoded, blended and kept in their
    slots without any event (decode.cc:1346-1408)."""
    J = built
    base, opaque, patch, ramp = _parts(J)
    first = dict(img=np.dstack([base, opaque]), save_as=1)
    if case in ("blend", "premultiplied"):
        layers = [first, dict(img=np.dstack([patch, ramp]), x0=50, y0=40, mode=2, alpha_mode=2, source=1)]
    elif case == "add_mul":
        layers = [first, dict(img=np.dstack([patch // 4, ramp]), x0=10, y0=20, mode=1, alpha_mode=0, source=1, save_as=2),
                  dict(img=np.dstack([patch, ramp]), x0=100, y0=70, mode=4, alpha_mode=4, source=2, clamp=1, save_as=3),
                  dict(img=np.dstack([patch, ramp]), x0=60, y0=5, mode=3, alpha_mode=3, source=3, clamp=1)]
    else:  # rectangles that stick out of the canvas on every side, and one entirely outside
        layers = [first, dict(img=np.dstack([patch, ramp]), x0=-30, y0=-20, mode=2, alpha_mode=2, source=1, save_as=1),
                  dict(img=np.dstack([patch, ramp]), x0=W - 40, y0=H - 25, mode=2, alpha_mode=2, source=1, save_as=1),
                  dict(img=np.dstack([patch, ramp]), x0=W + 5, y0=10, mode=0, alpha_mode=0, source=1)]
    data = J.encode_layers(layers, lossless=lossless, premultiplied=case == "premultiplied")  # (alpha.cc:22-31 for associated alpha)
    (want8, wantf), = _oracle_frames(data, 1)
    rc, events, out, px = R.run(data, tmp_path, "u8", 4)
    assert rc == 0 and [e for e in events if e in ("FRAME", "FULL_IMAGE")] == ["FRAME", "FULL_IMAGE"], out
    got = np.frombuffer(px, np.uint8).reshape(H, W, 4)
    assert np.abs(got.astype(int) - want8.astype(int)).max() <= (0 if lossless else 1), case
    rc, events, out, px = R.run(data, tmp_path, "f32", 3, "callback")
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.float32).reshape(H, W, 3) - wantf).max() < (1e-6 if lossless else 1e-4)


@pytest.mark.gpu
def test_animation_of_deltas_over_the_previous_canvas(built, tmp_path):
    """What a converted GIF looks like: every frame a crop blended over the canvas the frame before left in slot 1; an
    invisible layer in between; JxlDecoderSkipFrames still composes what it skips."""
    J = built
    base, opaque, patch, ramp = _parts(J)
    layers = [dict(img=np.dstack([base, opaque]), save_as=1, duration=2),
              dict(img=np.dstack([patch, ramp]), x0=20, y0=30, mode=2, alpha_mode=2, source=1, save_as=1, duration=3),
              dict(img=np.dstack([patch[::-1], ramp]), x0=90, y0=80, mode=2, alpha_mode=2, source=1, save_as=1, duration=0),
              dict(img=np.dstack([patch, np.full_like(ramp, 255)]), x0=150, y0=-20, mode=0, alpha_mode=0, source=1, duration=1)]
    data = J.encode_layers(layers, tps=(10, 1), lossless=True)
    want = _oracle_frames(data, 3)
    rc, events, out, px = R.run(data, tmp_path, "u8", 4, "frames", "ec")
    assert rc == 0 and events.count("FRAME") == 3 and events.count("FULL_IMAGE") == 3, out
    got = np.frombuffer(px[:3 * W * H * 4], np.uint8).reshape(3, H, W, 4)
    for k in range(3):
        assert np.array_equal(got[k], want[k][0]), k
    rc, events, out, px = R.run(R.container(data), tmp_path, "u8", 4, "frames", "skip=2", "chunk=5000")
    assert rc == 0 and events.count("FULL_IMAGE") == 1, out
    assert np.array_equal(np.frombuffer(px, np.uint8).reshape(H, W, 4), want[2][0])


@pytest.mark.gpu
def test_reference_alpha_blending_vectors_on_the_kernel(built):
    """lib/jxl/alpha_test.cc:25-86 (BlendingWithNonPremultiplied, BlendingWithPremultiplied, Mul): the reference's own
    known answers for PerformAlphaBlending / PerformMulBlending, through k_canvas_blend itself (jxlhip_debug_blend)."""
    import ctypes
    J = built
    L = J.lib()

    class Blend(ctypes.Structure):
        _fields_ = [("x0", ctypes.c_int32), ("y0", ctypes.c_int32)] + [(n, ctypes.c_uint32) for n in (
            "mode", "alpha_mode", "source", "alpha_source", "clamp", "alpha_clamp")] + [("save_slot", ctypes.c_int32)]

    L.jxlhip_debug_blend.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(Blend), ctypes.c_uint32,
                                     ctypes.c_uint32, ctypes.c_void_p]

    def blend(bg, fg, mode, alpha_mode, premultiplied, clamp):
        bgp = np.ascontiguousarray(np.array(bg, np.float32).reshape(4, 1))
        fgp = np.ascontiguousarray(np.array(fg, np.float32).reshape(1, 4))
        out = np.zeros((4, 1), np.float32)
        b = Blend(0, 0, mode, alpha_mode, 0, 0, clamp, clamp, -1)
        assert L.jxlhip_debug_blend(0, bgp.ctypes.data, fgp.ctypes.data, 1, ctypes.byref(b), 1, premultiplied, out.ctypes.data) == 0
        return out[:, 0]

    bg = [100, 110, 120, 180.0 / 255]
    fg, fg2 = [25, 21, 23, 15420.0 / 65535], [25, 21, 23, 2.0]
    o = blend(bg, fg, 2, 2, 0, 0)
    assert np.abs(o[:3] - [77.2, 83.0, 90.6]).max() < 0.05 and abs(o[3] - 3174.0 / 4095) < 1e-5
    o = blend(bg, fg2, 2, 2, 0, 1)
    assert np.abs(o[:3] - fg2[:3]).max() < 0.05 and abs(o[3] - 1.0) < 1e-5
    o = blend(bg, fg, 2, 2, 1, 0)
    assert np.abs(o[:3] - [101.5, 105.1, 114.8]).max() < 0.05 and abs(o[3] - 3174.0 / 4095) < 1e-5
    o = blend(bg, fg2, 2, 2, 1, 1)
    assert np.abs(o[:3] - fg2[:3]).max() < 0.05 and abs(o[3] - 1.0) < 1e-5
    o = blend([100, 100, 100, 1], [25, 25, 25, 1], 4, 0, 0, 0)
    assert np.abs(o[:3] - 2500).max() < 0.05
    o = blend([100, 100, 100, 1], [25, 25, 25, 1], 4, 0, 0, 1)
    assert np.abs(o[:3] - 100).max() < 0.05


@pytest.mark.gpu
@pytest.mark.parametrize("lossless", [True, False])
def test_layers_delivered_one_by_one_without_coalescing(built, tmp_path, lossless):
    """JxlDecoderSetCoalescing(false) (decode.cc:973-979, 1350-1354, 2725-2768): every regular frame is a still of its own
    size, unblended, and its frame header carries the crop, the blend mode, the source slot and the slot it is saved to;
    the buffer sizes follow the frame (decode.cc:980-1001). Lossless layers come back as the arrays that were coded."""
    J = built
    base, opaque, patch, ramp = _parts(J)
    layers = [dict(img=np.dstack([base, opaque]), save_as=1),
              dict(img=np.dstack([patch, ramp]), x0=-10, y0=100, mode=2, alpha_mode=2, source=1, save_as=2),
              dict(img=np.dstack([patch[::-1].copy(), ramp]), x0=150, y0=20, mode=3, alpha_mode=3, source=2, clamp=1)]
    data = J.encode_layers(layers, lossless=lossless)  # (the writer gives every layer an explicit size and origin)
    rc, events, out, px = R.run(data, tmp_path, "u8", 4, "layers")
    assert rc == 0 and [e for e in events if e in ("FRAME", "FULL_IMAGE")] == ["FRAME", "FULL_IMAGE"] * 3, out
    heads = [l for l in out.splitlines() if l.startswith("layer ")]
    assert heads == ["layer crop=1 x0=0 y0=0 blend=0 source=0 alpha=0 clamp=0 save_as=1",
                     "layer crop=1 x0=-10 y0=100 blend=2 source=1 alpha=0 clamp=0 save_as=2",
                     "layer crop=1 x0=150 y0=20 blend=3 source=2 alpha=0 clamp=1 save_as=0"], out
    sizes = [l.split()[2] for l in out.splitlines() if l.startswith("event FRAME")]
    assert sizes == ["%dx%d" % (W, H), "80x60", "80x60"], out
    pos = 0
    for k, layer in enumerate(layers):
        h, w = layer["img"].shape[:2]
        got = np.frombuffer(px[pos:pos + h * w * 4], np.uint8).reshape(h, w, 4)
        pos += h * w * 4
        if lossless:
            assert np.array_equal(got, layer["img"]), k
        else:  # VarDCT colour at d1.0, lossless alpha
            assert np.array_equal(got[..., 3], layer["img"][..., 3]), k
            assert np.abs(got[..., :3].astype(int) - layer["img"][..., :3].astype(int)).mean() < 4.0, k
    assert pos == len(px)
    # the same stream coalesced is one still of the canvas size
    rc, events, out, px = R.run(data, tmp_path, "u8", 4)
    assert rc == 0 and events.count("FULL_IMAGE") == 1 and len(px) == W * H * 4
