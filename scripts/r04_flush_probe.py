"""Measurement / debugging aid: JxlDecoderFlushImage over chunked input for several stream kinds; prints what was refused and why."""
import os, sys, numpy as np, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import libjxl_amd as J, replay_util as R, jxlo
for kw in (dict(ac_code_mode=3), dict(ac_code_mode=1), dict(ac_code_mode=2), dict()):
    for passes in (1, 2):
        data = J.encode_rgb8(J.synth_image(900, 600, seed=19), num_passes=passes, **kw)
        tmp = tempfile.mkdtemp()
        rc, events, out, px = R.run(data, tmp, "u8", 3, "flush", "chunk=%d" % (len(data) // 9))
        res = []
        for line in [l for l in out.splitlines() if l.startswith("flushed ")]:
            k, given = int(line.split()[1]), int(line.split("bytes_given=")[1])
            got = np.fromfile(os.path.join(tmp, "out.raw.flush%d" % k), np.uint8).reshape(600, 900, 3)
            want = jxlo.Decoded(data, dumps=False, prefix=given)
            res.append(int(np.abs(got.astype(int) - want.rgb8.astype(int)).max()))
            want.close()
        print(kw, "passes", passes, "rc", rc, "flush max diffs", res, [l for l in out.splitlines() if "refused" in l][:2], flush=True)
