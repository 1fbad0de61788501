"""Patches (lib/jxl/dec_patch_dictionary.cc, render_pipeline/stage_patches.cc): a reference-only frame
kept before its colour transform (what libjxl codes its patch atlas as: an XYB Modular frame), and a frame whose
dictionary draws rectangles of it over the decoded XYB planes, before the splines. CPU part: the oracle's placement
against the atlas decoded on its own, the host's parse; GPU part: k_patches_add behind the decoder API against the oracle."""
import numpy as np
import pytest

import replay_util as R


def _case(J):
    img = J.synth_image(300, 200, seed=5)
    atlas = J.synth_image(64, 48, seed=9)
    patches = [dict(x0=4, y0=6, xsize=20, ysize=16, positions=[(10, 10, 2, 0), (100, 50, 2, 0), (250, 170, 1, 0), (255, 175, 2, 0)]),
               dict(x0=30, y0=0, xsize=30, ysize=40, positions=[(60, 120, 3, 1), (200, 20, 2, 0), (270, 160, 0, 0)])]
    return img, atlas, patches


def test_oracle_places_replaced_patches_exactly(built):
    """PatchBlendMode kReplace copies the reference's XYB samples: inside such a rectangle (where nothing else is drawn) the
    picture is the atlas decoded as a still of its own, pixel for pixel; outside every rectangle it is the frame without
    patches."""
    import jxlo
    J = built
    img, atlas, _ = _case(J)
    patches = [dict(x0=4, y0=6, xsize=20, ysize=16, positions=[(250, 170, 1, 0), (16, 8, 1, 0)])]
    data = J.encode_patched(img, atlas, patches, lossless=True)
    got = jxlo.Decoded(data, dumps=False).rgb8
    plain = jxlo.Decoded(J.encode_lossless(img, J.MODULAR_XYB), dumps=False).rgb8
    alone = jxlo.Decoded(J.encode_lossless(atlas, J.MODULAR_XYB), dumps=False).rgb8
    # (the 8-bit conversion dithers by position: compare within 1 level)
    for (x, y) in ((250, 170), (16, 8)):
        assert np.abs(got[y:y + 16, x:x + 20].astype(int) - alone[6:22, 4:24].astype(int)).max() <= 1
    mask = np.ones(got.shape[:2], bool)
    mask[170:186, 250:270] = False
    mask[8:24, 16:36] = False
    assert np.array_equal(got[mask], plain[mask])


def test_host_parses_reference_frames_and_dictionaries(built):
    J = built
    img, atlas, patches = _case(J)
    for kw in (dict(), dict(atlas_vardct=True)):
        data = J.encode_patched(img, atlas, patches, **kw)
        first = (J.Frame if kw else J.ModFrame)(data)
        assert (first.info["xsize"], first.info["ysize"]) == (64, 48) and not first.is_last
        second = J.Frame(data, frame_pos=first.end, frame_index=1)
        assert (second.info["xsize"], second.info["ysize"]) == (300, 200) and second.is_last and second.end == len(data)
        first.close()
        second.close()


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(atlas_vardct=True), dict(epf_iters=2, noise=40)])
def test_patches_through_the_gpu(built, tmp_path, kw):
    """Reference frame (XYB Modular like libjxl's, or VarDCT) -> XYB slot on the device canvas; VarDCT frame with add /
    replace / multiply / none patches, overlapping ones in dictionary order; with EPF2 + noise behind them."""
    import jxlo
    J = built
    img, atlas, patches = _case(J)
    data = J.encode_patched(img, atlas, patches, **kw)
    o = jxlo.Decoded(data)
    want8, wantf = o.rgb8.copy(), o.planes("rgbf").transpose(1, 2, 0).copy()
    o.close()
    rc, events, out, px = R.run(data, tmp_path, "f32", 3)
    assert rc == 0 and [e for e in events if e in ("FRAME", "FULL_IMAGE")] == ["FRAME", "FULL_IMAGE"], out
    got = np.frombuffer(px, np.float32).reshape(200, 300, 3)
    assert np.abs(got - wantf).max() < 1e-4
    rc, events, out, px = R.run(R.container(data), tmp_path, "u8", 3, "chunk=4000")
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.uint8).reshape(200, 300, 3).astype(int) - want8.astype(int)).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("ups", [2, 4])
def test_patches_on_an_upsampled_frame(built, tmp_path, ups):
    """dec_cache.cc:193-212: the patches of a frame coded at 1 / 2 or 1 / 4 of the image's size are placed at the FRAME's
    resolution (positions in frame pixels) before the upsampling stage enlarges them with everything else."""
    import jxlo
    J = built
    img, atlas, _ = _case(J)
    fx, fy = (300 + ups - 1) // ups, (200 + ups - 1) // ups
    patches = [dict(x0=4, y0=6, xsize=20, ysize=16, positions=[(3, 2, 2, 0), (fx - 22, fy - 18, 1, 0)]),
               dict(x0=30, y0=0, xsize=30, ysize=40, positions=[(fx // 2 - 15, 4, 3, 1), (1, fy - 41, 2, 0)])]
    data = J.encode_patched(img, atlas, patches, upsampling=ups)
    o = jxlo.Decoded(data)
    want8, wantf = o.rgb8.copy(), o.planes("rgbf").transpose(1, 2, 0).copy()
    o.close()
    rc, events, out, px = R.run(data, tmp_path, "f32", 3)
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.float32).reshape(200, 300, 3) - wantf).max() < 1e-4
    rc, events, out, px = R.run(data, tmp_path, "u8", 3)
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.uint8).reshape(200, 300, 3).astype(int) - want8.astype(int)).max() <= 1


def test_patch_rectangles_are_checked_against_the_reference_frames(built):
    """jxlamd_frame_set_patch_sources (dec_patch_dictionary.cc:63-83): a patch that names an empty slot, or reaches outside
    the reference frame it names, is refused before anything is uploaded (the pointers are only carried, never read here)."""
    import ctypes
    J = built
    L = J.lib()
    img, atlas, patches = _case(J)
    data = J.encode_patched(img, atlas, patches)  # slot 1, rectangles up to x = 60, y = 40 of the 64 x 48 atlas
    first = J.ModFrame(data)
    f = J.Frame(data, frame_pos=first.end, frame_index=1)
    first.close()
    fp = ctypes.POINTER(ctypes.c_float)
    L.jxlamd_frame_set_patch_sources.argtypes = [ctypes.c_void_p, ctypes.POINTER(fp), ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]
    L.jxlamd_last_error.restype = ctypes.c_char_p
    dummy = (ctypes.c_float * 4)()
    some = ctypes.cast(dummy, fp)

    def check(slots, w, h):
        planes = (fp * 4)(*[some if s else fp() for s in slots])
        return L.jxlamd_frame_set_patch_sources(f._h, planes, (ctypes.c_uint32 * 4)(*w), (ctypes.c_uint32 * 4)(*h))

    assert check([0, 1, 0, 0], [0, 64, 0, 0], [0, 48, 0, 0]) == 0
    assert check([1, 0, 1, 1], [64, 0, 64, 64], [48, 0, 48, 48]) != 0 and b"missing" in L.jxlamd_last_error()
    assert check([0, 1, 0, 0], [0, 59, 0, 0], [0, 48, 0, 0]) != 0 and b"outside" in L.jxlamd_last_error()
    assert check([0, 1, 0, 0], [0, 64, 0, 0], [0, 39, 0, 0]) != 0 and b"outside" in L.jxlamd_last_error()
    f.close()


def _alpha_case(J):
    """RGBA frame and RGBA atlas; positions over every colour mode and every alpha-channel mode, some overlapping."""
    yy, xx = np.mgrid[0:200, 0:300]
    img = np.dstack([J.synth_image(300, 200, seed=5), (40 + 150 * (0.5 + 0.5 * np.sin(xx / 23.0) * np.cos(yy / 17.0))).astype(np.uint8)])
    ay, ax = np.mgrid[0:48, 0:64]
    atlas = np.dstack([J.synth_image(64, 48, seed=9), ((ax * 4 + ay * 3) & 255).astype(np.uint8)])
    pos = []
    k = 0
    for mode in range(8):
        for ec_mode in range(8):
            # (x, y, colour mode, clamp, alpha mode, alpha clamp): a 6 x 11 grid of 20 x 16 rectangles, a few pixels of overlap
            pos.append((4 + (k % 13) * 22, 3 + (k // 13) * 17, mode, (k >> 1) & 1, ec_mode, k & 1))
            k += 1
    return img, atlas, [dict(x0=4, y0=6, xsize=24, ysize=18, positions=pos)]


def _blend_np(bg, bga, fg, fga, mode, clamp, ec_mode, ec_clamp, premultiplied):
    """blending.cc:40-190 + alpha.cc:17-101 for an image whose one extra channel is alpha, read independently of the oracle's
    restatement: float32 arithmetic, array-wise. bg / fg: [3][h][w]; returns the blended colour planes and alpha."""
    f = np.float32
    c01 = lambda v: np.clip(v, f(0), f(1))
    one = f(1)
    a = {0: bga, 1: fga, 2: bga + fga, 3: bga * (c01(fga) if ec_clamp else fga),
         4: one - (one - (c01(fga) if ec_clamp else fga)) * (one - bga), 5: one - (one - (c01(bga) if ec_clamp else bga)) * (one - fga),
         6: bga, 7: fga}[ec_mode]
    if mode == 0:
        out = bg
    elif mode == 1:
        out = fg
    elif mode == 2:
        out = bg + fg
    elif mode == 3:
        out = bg * (c01(fg) if clamp else fg)
    elif mode in (4, 5):
        bot, bota, top, topa = (bg, bga, fg, fga) if mode == 4 else (fg, fga, bg, bga)
        ta = c01(topa) if clamp else topa
        new_a = one - (one - ta) * (one - bota)
        if premultiplied:
            out = top + bot * (one - ta)
        else:
            with np.errstate(divide="ignore"):
                r = np.where(new_a > 0, one / new_a, f(0)).astype(f)
            out = (top * ta + bot * bota * (one - ta)) * r
        a = new_a
    elif mode == 6:
        out = bg + fg * (c01(fga) if clamp else fga)
    else:
        out = fg + bg * (c01(bga) if clamp else bga)
    return out.astype(f), a.astype(f)


@pytest.mark.parametrize("premultiplied", [False, True])
def test_oracle_blends_patches_through_alpha_like_the_reference_text(built, premultiplied):
    """Every PatchBlendMode on the colour channels x every one on the alpha channel (dec_patch_dictionary.h:32-58): the
    oracle's patched XYB planes and alpha against a NumPy reading of blending.cc / alpha.cc applied to the frame decoded
    without patches and the atlas decoded on its own."""
    import jxlo
    J = built
    img, atlas, patches = _alpha_case(J)
    data = J.encode_patched(img, atlas, patches, atlas_vardct=True, premultiplied=premultiplied)
    o = jxlo.Decoded(data)
    got, got_a = o.planes("xyb_filtered").copy(), o.buffer("alphaf").reshape(200, 300).copy()
    o.close()
    o = jxlo.Decoded(J.encode_rgba8(img))
    bg, bga = o.planes("xyb_filtered").copy(), o.buffer("alphaf").reshape(200, 300).copy()
    o.close()
    o = jxlo.Decoded(J.encode_rgba8(atlas))
    fg, fga = o.planes("xyb_filtered").copy(), o.buffer("alphaf").reshape(48, 64).copy()
    o.close()
    assert np.array_equal((bga * 255).round().astype(np.uint8), img[..., 3])
    p = patches[0]
    for (x, y, mode, clamp, ec_mode, ec_clamp) in p["positions"]:
        sy, sx = slice(p["y0"], p["y0"] + p["ysize"]), slice(p["x0"], p["x0"] + p["xsize"])
        dy, dx = slice(y, y + p["ysize"]), slice(x, x + p["xsize"])
        out, a = _blend_np(bg[:, dy, dx], bga[dy, dx], fg[:, sy, sx], fga[sy, sx], mode, clamp, ec_mode, ec_clamp, premultiplied)
        bg[:, dy, dx] = out
        bga[dy, dx] = a
    xs = 300
    assert np.abs(got[:, :, :xs] - bg[:, :, :xs]).max() < 1e-6
    assert np.abs(got_a - bga).max() < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize("premultiplied,atlas_vardct", [(False, True), (True, True), (False, False), (True, False)])
def test_patches_that_blend_through_alpha_on_the_gpu(built, tmp_path, premultiplied, atlas_vardct):
    """Every PatchBlendMode on the colour channels x every one on the alpha channel through JxlDecoder: the reference frame's
    alpha travels with its XYB planes in the canvas slot, k_patches_add blends colour and alpha, and the pixel writer, the
    extra-channel buffer and the canvas all see the blended alpha."""
    import jxlo
    J = built
    img, atlas, patches = _alpha_case(J)
    # (the reference frame as a VarDCT frame, or as libjxl codes its patch frames: an XYB Modular frame, alpha beside it)
    data = J.encode_patched(img, atlas, patches, atlas_vardct=atlas_vardct, premultiplied=premultiplied)
    o = jxlo.Decoded(data)
    want8, wantf, wanta = o.rgb8.copy(), o.planes("rgbf").transpose(1, 2, 0).copy(), o.buffer("alphaf").reshape(200, 300).copy()
    o.close()
    assert np.abs(wanta - img[..., 3] / np.float32(255)).max() > 0.2  # (the patches really changed the alpha channel)
    rc, events, out, px = R.run(data, tmp_path, "f32", 4)
    assert rc == 0 and [e for e in events if e in ("FRAME", "FULL_IMAGE")] == ["FRAME", "FULL_IMAGE"], out
    got = np.frombuffer(px, np.float32).reshape(200, 300, 4)
    assert np.abs(got[..., 3] - wanta).max() < 1e-6
    # (colour: a blend divides by the new alpha, which may be tiny: relative to the sample)
    assert (np.abs(got[..., :3] - wantf) <= 1e-4 * np.maximum(1.0, np.abs(wantf))).all()
    rc, events, out, px = R.run(data, tmp_path, "u8", 4, "ec")
    assert rc == 0, out
    got8 = np.frombuffer(px[:300 * 200 * 4], np.uint8).reshape(200, 300, 4)
    assert np.abs(got8.astype(int) - want8.astype(int)).max() <= 1
    ec = np.frombuffer(px[300 * 200 * 4:], np.uint8).reshape(200, 300).astype(int)
    assert np.abs(ec - np.rint(np.clip(wanta.astype(np.float64), 0, 1) * 255.0)).max() <= 1


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["xyb", "rgb"])
def test_patches_on_a_modular_frame(built, tmp_path, kind):
    """A Modular frame with a patch dictionary (what libjxl's lossless mode writes for screen content: the frame flag is read
    in DC global like a VarDCT frame's, dec_frame.cc:271-285): the colour samples go to float planes, the patches are drawn
    over them from the reference slot, then the pixel writer reads them back. XYB-coded ("lossy Modular") and plain RGB
    (RCT) frames; add, replace, multiply and the alpha-less forms of the blend modes (blending.cc:154-168), overlapping."""
    import jxlo
    J = built
    img, atlas, _ = _case(J)
    patches = [dict(x0=4, y0=6, xsize=20, ysize=16, positions=[(10, 10, 2, 0), (100, 50, 1, 0), (250, 170, 4, 0), (255, 175, 6, 1)]),
               dict(x0=30, y0=0, xsize=30, ysize=40, positions=[(60, 120, 3, 1), (200, 20, 7, 0), (270, 160, 0, 0), (5, 150, 5, 0)])]
    data = J.encode_patched(img, atlas, patches, lossless=True, lossless_flags=None if kind == "xyb" else J.LOSSLESS_RCT)
    o = jxlo.Decoded(data)
    want8, wantf = o.rgb8.copy(), o.planes("rgbf").transpose(1, 2, 0).copy()
    o.close()
    plain = jxlo.Decoded(J.encode_lossless(img, J.MODULAR_XYB if kind == "xyb" else J.LOSSLESS_RCT), dumps=False).rgb8
    assert (want8 != plain).any(axis=2).sum() > 1500  # (the patches are there)
    rc, events, out, px = R.run(data, tmp_path, "f32", 3)
    assert rc == 0 and [e for e in events if e in ("FRAME", "FULL_IMAGE")] == ["FRAME", "FULL_IMAGE"], out
    assert np.abs(np.frombuffer(px, np.float32).reshape(200, 300, 3) - wantf).max() < 1e-4
    rc, events, out, px = R.run(data, tmp_path, "u8", 3)
    assert rc == 0, out
    assert np.abs(np.frombuffer(px, np.uint8).reshape(200, 300, 3).astype(int) - want8.astype(int)).max() <= 1
