"""Animations (multi-frame codestreams): a sequence of frames that each replace the whole canvas
(decode.cc:1346-1350 is_last_of_still; full size, BlendMode kReplace, a duration); crops, blending and invisible layers:
tests/test_layers.py. CPU part: oracle, host parse and the header-only event sequence; GPU part: pixels of every
frame through the replay of the reference's call sequence."""
import ctypes
import struct

import numpy as np
import pytest

import replay_util as R


def _frames(J, n=3, size=(200, 150)):
    return [J.synth_image(size[0], size[1], seed=40 + i) for i in range(n)]


def test_oracle_decodes_every_frame_of_an_animation(built):
    import jxlo
    J = built
    fr = _frames(J)
    data = J.encode_animation(fr, [3, 5, 7], tps=(25, 1), num_loops=2)
    for k, img in enumerate(fr):
        o = jxlo.Decoded(data, dumps=False, frame=k)
        still = jxlo.Decoded(J.encode_rgb8(img), dumps=False)  # the same frame coded as a still: same payload, same pixels
        assert np.array_equal(o.rgb8, still.rgb8)
        assert o.animation == dict(have_animation=1, tps_numerator=25, tps_denominator=1, num_loops=2, duration=[3, 5, 7][k],
                                   is_last=int(k == 2), timecode=0)
        o.close()
        still.close()
    with pytest.raises(RuntimeError, match="no such frame"):
        jxlo.Decoded(data, dumps=False, frame=3)
    lossless = J.encode_animation(fr, [1, 1, 1], lossless=True)
    for k, img in enumerate(fr):
        o = jxlo.Decoded(lossless, dumps=False, frame=k)
        assert np.array_equal(o.rgb8, img)
        o.close()


def test_host_parses_frame_after_frame(built):
    J = built
    fr = _frames(J)
    for lossless in (False, True):
        data = J.encode_animation(fr, [2, 4, 6], lossless=lossless)
        pos = 0
        for k in range(3):
            f = (J.ModFrame if lossless else J.Frame)(data, frame_pos=pos, frame_index=k)
            assert (f.duration, f.is_last) == ([2, 4, 6][k], k == 2)
            assert f.end > pos
            pos = f.end
            f.close()
        assert pos == len(data)
    # a zero-duration frame is an invisible layer: it parses like any frame (composition: tests/test_layers.py)
    layered = J.encode_animation(fr[:2], [0, 1])
    f = J.Frame(layered)
    assert (f.duration, f.is_last) == (0, False)
    f.close()


def test_decoder_api_walks_the_frames_without_pixels(built):
    """FRAME events only (decode.cc:1431-1437: the frames' bytes are skipped), in one piece and in small pieces."""
    J = built
    L = J.lib()
    vp = ctypes.c_void_p
    L.JxlDecoderCreate.restype = vp
    L.JxlDecoderCreate.argtypes = [vp]
    for n in ("JxlDecoderDestroy", "JxlDecoderProcessInput", "JxlDecoderCloseInput"):
        getattr(L, n).argtypes = [vp]
    L.JxlDecoderSubscribeEvents.argtypes = [vp, ctypes.c_int]
    L.JxlDecoderSetInput.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    L.JxlDecoderReleaseInput.argtypes = [vp]
    L.JxlDecoderReleaseInput.restype = ctypes.c_size_t
    L.JxlDecoderGetBasicInfo.argtypes = [vp, vp]
    L.JxlDecoderGetFrameHeader.argtypes = [vp, vp]
    L.JxlDecoderSkipFrames.argtypes = [vp, ctypes.c_size_t]
    L.JxlDecoderSkipFrames.restype = None
    data = J.encode_animation(_frames(J, 4), [3, 1, 300, 2], tps=(1000, 1001), num_loops=5)

    def walk(chunk, skip=0):
        dec = L.JxlDecoderCreate(None)
        assert L.JxlDecoderSubscribeEvents(dec, 0x40 | 0x400) == 0
        if skip:
            L.JxlDecoderSkipFrames(dec, skip)
        seen, pos, given = [], 0, min(chunk, len(data))
        buf = data[:given]
        L.JxlDecoderSetInput(dec, buf, len(buf))
        if given == len(data):
            L.JxlDecoderCloseInput(dec)
        while True:
            st = L.JxlDecoderProcessInput(dec)
            if st == 0x40:
                info = (ctypes.c_uint8 * 512)()
                assert L.JxlDecoderGetBasicInfo(dec, info) == 0
                raw = bytes(info)
                # JxlBasicInfo (codestream_header.h:90-231): have_animation is the 12th 32-bit field, the animation header
                # follows orientation, the four channel counts, alpha_premultiplied and the preview size
                fields = struct.unpack_from("<12I", raw, 0)
                assert fields[1:3] == (200, 150) and fields[11] == 1, fields  # xsize, ysize, have_animation
                tps = struct.unpack_from("<4I", raw, 80)
                assert tps == (1000, 1001, 5, 0), tps
            elif st == 0x400:
                fh = (ctypes.c_uint8 * 256)()
                assert L.JxlDecoderGetFrameHeader(dec, fh) == 0
                duration, timecode, name_length, is_last = struct.unpack_from("<4I", bytes(fh), 0)
                seen.append((duration, is_last))
            elif st == 2:  # JXL_DEC_NEED_MORE_INPUT
                left = L.JxlDecoderReleaseInput(dec)
                pos += len(buf) - left
                assert pos + left < len(data), "asks for input past the end"
                given = min(len(data) - pos, left + chunk)
                buf = data[pos:pos + given]
                L.JxlDecoderSetInput(dec, buf, len(buf))
                if pos + given == len(data):
                    L.JxlDecoderCloseInput(dec)
            else:
                assert st == 0, st
                break
        L.JxlDecoderDestroy(dec)
        return seen

    assert walk(len(data)) == [(3, 0), (1, 0), (300, 0), (2, 1)]
    assert walk(1500) == [(3, 0), (1, 0), (300, 0), (2, 1)]
    assert walk(len(data), skip=2) == [(300, 0), (2, 1)]


@pytest.mark.gpu
@pytest.mark.parametrize("lossless", [False, True, "ycbcr420"])
def test_every_frame_of_an_animation_through_the_gpu(built, tmp_path, lossless):
    """The replay of DecodeImageJXL's loop (jxl.cc:495-640: one NEED_IMAGE_OUT_BUFFER + FULL_IMAGE per frame) gets every
    frame's pixels; each equals the oracle's decode of that frame (VarDCT: +-1, lossless: exact); with alpha, in a
    container delivered in pieces; JxlDecoderSkipFrames drops leading frames."""
    import jxlo
    J = built
    fr = _frames(J, 3, size=(300, 200))
    alpha = ((np.mgrid[0:200, 0:300][1] * 255) // 299).astype(np.uint8)
    frames = [np.dstack([f, np.roll(alpha, 17 * i, axis=1)]) for i, f in enumerate(fr)]
    if lossless == "ycbcr420":  # frames of an image that is not XYB encoded, chroma subsampled, composed on the canvas like any other
        lossless = False
        data = J.encode_animation(frames, [4, 2, 9], color_transform=2, chroma_subsampling=4, strategy_mode=0)
    else:
        data = J.encode_animation(frames, [4, 2, 9], lossless=lossless)
    want = []
    for k in range(3):
        o = jxlo.Decoded(data, dumps=False, frame=k)
        want.append(o.rgb8.copy())
        o.close()
    tol = 0 if lossless else 1
    # (whole; in a container in pieces; the input handed back and in again after every frame)
    for stream, extra in ((data, ()), (R.container(data), ("chunk=3000",)), (data, ("swap",))):
        rc, events, out, px = R.run(stream, tmp_path, "u8", 4, "frames", *extra)
        assert rc == 0, out
        assert [e for e in events if e in ("FRAME", "NEED_IMAGE_OUT_BUFFER", "FULL_IMAGE")] == ["FRAME", "NEED_IMAGE_OUT_BUFFER", "FULL_IMAGE"] * 3, out
        assert out.count("duration=4") == 1 and out.count("duration=2") == 1 and out.count("duration=9") == 1
        got = np.frombuffer(px, np.uint8).reshape(3, 200, 300, 4)
        for k in range(3):
            assert np.array_equal(got[k][..., 3], frames[k][..., 3])
            assert np.abs(got[k].astype(int) - want[k].astype(int)).max() <= tol, k
    rc, events, out, px = R.run(data, tmp_path, "u8", 4, "frames", "skip=2")
    assert rc == 0 and events.count("FULL_IMAGE") == 1, out
    got = np.frombuffer(px, np.uint8).reshape(200, 300, 4)
    assert np.abs(got.astype(int) - want[2].astype(int)).max() <= tol


@pytest.mark.gpu
def test_noise_of_later_frames_uses_the_frame_index(built, tmp_path):
    """dec_frame.cc:160-168, stage_noise.cc: the noise generator is seeded with the number of visible frames before the
    frame, so the same image as frame 0 and as frame 1 gets different noise; the oracle applies the same rule."""
    import jxlo
    J = built
    img = J.synth_image(256, 256, seed=9)
    data = J.encode_animation([img, img], [1, 1], noise=300)
    rc, events, out, px = R.run(data, tmp_path, "f32", 3, "frames")
    assert rc == 0, out
    got = np.frombuffer(px, np.float32).reshape(2, 256, 256, 3)
    assert np.abs(got[0] - got[1]).max() > 1e-3
    for k in range(2):
        o = jxlo.Decoded(data, frame=k)
        ref = o.planes("rgbf").transpose(1, 2, 0).copy()
        o.close()
        assert np.abs(got[k] - ref).max() < 1e-4, k


def test_frame_names_come_out_of_the_frame_header(built):
    """frame_header.cc:431: JxlFrameHeader.name_length and JxlDecoderGetFrameName (decode.cc:2778-2792: the buffer holds
    the name and its terminating zero)."""
    J = built
    L = J.lib()
    vp = ctypes.c_void_p
    L.JxlDecoderCreate.restype = vp
    L.JxlDecoderCreate.argtypes = [vp]
    for n in ("JxlDecoderDestroy", "JxlDecoderProcessInput", "JxlDecoderCloseInput"):
        getattr(L, n).argtypes = [vp]
    L.JxlDecoderSubscribeEvents.argtypes = [vp, ctypes.c_int]
    L.JxlDecoderSetInput.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    L.JxlDecoderGetFrameHeader.argtypes = [vp, vp]
    L.JxlDecoderGetFrameName.argtypes = [vp, ctypes.c_char_p, ctypes.c_size_t]
    name = "layer é with a longer name than sixteen bytes"
    J.set_frame_name(name)
    try:
        streams = [J.encode_rgb8(J.synth_image(64, 48, seed=1)), J.encode_lossless(J.synth_image(64, 48, seed=1))]
    finally:
        J.set_frame_name("")
    for data in streams:
        dec = L.JxlDecoderCreate(None)
        L.JxlDecoderSubscribeEvents(dec, 0x400)
        L.JxlDecoderSetInput(dec, data, len(data))
        L.JxlDecoderCloseInput(dec)
        assert L.JxlDecoderProcessInput(dec) == 0x400
        fh = (ctypes.c_uint8 * 256)()
        assert L.JxlDecoderGetFrameHeader(dec, fh) == 0
        n = struct.unpack_from("<4I", bytes(fh), 0)[2]
        assert n == len(name.encode())
        buf = ctypes.create_string_buffer(n + 1)
        assert L.JxlDecoderGetFrameName(dec, buf, n + 1) == 0 and buf.value.decode() == name
        assert L.JxlDecoderGetFrameName(dec, buf, n) == 1  # too small
        L.JxlDecoderDestroy(dec)


@pytest.mark.gpu
def test_skip_current_frame_only_drops_that_frame(built, tmp_path):
    """JxlDecoderSkipCurrentFrame (decode.cc:904-915) steps over ONE frame: the frames behind it still come with their
    FRAME / FULL_IMAGE events and pixels; a skipped frame that later frames are blended with still reaches its reference
    slot, so the canvas the caller gets afterwards is the one it would have got without the skip."""
    import jxlo
    J = built
    fr = _frames(J, 3, size=(300, 200))
    data = J.encode_animation(fr, [4, 2, 9])
    want = []
    for k in range(3):
        o = jxlo.Decoded(data, dumps=False, frame=k)
        want.append(o.rgb8.copy())
        o.close()
    rc, events, out, px = R.run(data, tmp_path, "u8", 3, "frames", "skipcur=0")
    assert rc == 0 and "skipped current frame" in out, out
    assert [e for e in events if e in ("FRAME", "NEED_IMAGE_OUT_BUFFER", "FULL_IMAGE")] == ["FRAME"] + ["FRAME", "NEED_IMAGE_OUT_BUFFER", "FULL_IMAGE"] * 2, out
    got = np.frombuffer(px, np.uint8).reshape(2, 200, 300, 3)
    for k in range(2):
        assert np.abs(got[k].astype(int) - want[k + 1].astype(int)).max() <= 1, k
    # skipping the LAST frame ends the decode cleanly
    rc, events, out, px = R.run(data, tmp_path, "u8", 3, "frames", "skipcur=2")
    assert rc == 0 and events.count("FULL_IMAGE") == 2, out
    # a delta animation: frame 1 is a patch blended over frame 0 (kept in slot 1); skipping frame 0 must not lose it
    base = J.synth_image(200, 150, seed=1)
    opaque = np.full((150, 200), 255, np.uint8)
    patch = J.synth_image(80, 60, seed=2)
    ramp = ((np.mgrid[0:60, 0:80][1] * 255) // 79).astype(np.uint8)
    layered = J.encode_layers([dict(img=np.dstack([base, opaque]), save_as=1, duration=2),
                               dict(img=np.dstack([patch, ramp]), x0=50, y0=40, mode=2, alpha_mode=2, source=1, duration=3)],
                              tps=(10, 1), lossless=True)
    o = jxlo.Decoded(layered, dumps=False, frame=1)
    ref = o.rgb8.copy()
    o.close()
    rc, events, out, px = R.run(layered, tmp_path, "u8", 4, "frames", "skipcur=0")
    assert rc == 0 and events.count("FULL_IMAGE") == 1, out
    got = np.frombuffer(px, np.uint8).reshape(150, 200, 4)
    assert np.abs(got.astype(int) - ref.astype(int)).max() <= 1
