"""Development helper for the GPU box: stage-by-stage comparison of the HIP path with the oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import libjxl_amd as J
import jxlo


def compare(name, data):
    f = J.Frame(data)
    o = jxlo.Decoded(data)
    c = J.HipContext()
    c.upload(f)
    c.run_entropy(); c.sync()
    r, flags = c.errors()
    co = c.download("coeffs").astype(np.int32)
    ref = o.planes("coeffs")
    ncoef = int((co != ref).sum())
    c.run_transform(); c.sync()
    x = c.download("xyb_idct"); xr = o.planes("xyb_idct")
    e1 = np.abs(x - xr).max(axis=(1, 2))
    c.run_filter_color(); c.sync()
    xf = c.download("xyb_filtered")[:, :o.info["ysize"], :]; xfr = o.planes("xyb_filtered")
    xs = o.info["xsize"]
    e2 = np.abs(xf[:, :, :xs] - xfr[:, :, :xs]).max(axis=(1, 2))
    rgb = c.rgb8(); d = np.abs(rgb.astype(int) - o.rgb8.astype(int))
    print("%-28s err=%d flags=%s coef_mismatch=%d idct_maxerr=%s filt_maxerr=%s rgb8 maxdiff=%d ndiff=%d  ms=(%.3f %.3f %.3f)" % (
        name, r, [f_ for f_ in flags if f_][:4], ncoef, np.array2string(e1, precision=6), np.array2string(e2, precision=6), d.max(), int((d > 0).sum()),
        c.stage_ms(0), c.stage_ms(1), c.stage_ms(2)))
    sys.stdout.flush()
    bad = ncoef != 0 or r != 0 or d.max() > 1
    if bad and e1.max() > 1e-3:
        # locate worst block
        ch = int(e1.argmax()); yy, xx = np.unravel_index(np.abs(x[ch] - xr[ch]).argmax(), x[ch].shape)
        acs = o.buffer("acs").reshape(o.info["ysize_blocks"], o.info["xsize_blocks"])
        print("   worst idct at ch %d (%d,%d) block strategy %d first=%d" % (ch, yy, xx, acs[yy // 8, xx // 8] >> 1, acs[yy // 8, xx // 8] & 1))
    c.close(); f.close(); o.close()
    return not bad


if __name__ == "__main__":
    ok = True
    img = J.synth_image(520, 300, seed=3)
    ok &= compare("image d1 mode1", J.encode_rgb8(img))
    ok &= compare("image d1 dct8", J.encode_rgb8(img, strategy_mode=0))
    ok &= compare("image 64x64 single group", J.encode_rgb8(J.synth_image(64, 64), strategy_mode=1))
    ok &= compare("image d2 epf2", J.encode_rgb8(img, distance=2.0))
    ok &= compare("image d4.5 epf3 nogab", J.encode_rgb8(img, distance=4.5, gab=0))
    for s in range(27):
        ok &= compare("random strategy %d" % s, J.encode_random(512, 512, seed=s, strategy_mask=(1 << s) | 1))
    ok &= compare("random all", J.encode_random(777, 600, seed=99))
    big = J.synth_image(3840, 2160)
    t = time.time(); data = J.encode_rgb8(big); print("4K encode %.1fs %d bytes" % (time.time() - t, len(data)))
    ok &= compare("4K d1", data)
    print("ALL OK" if ok else "FAILURES")
