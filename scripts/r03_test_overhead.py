#!/usr/bin/env python3
"""Where a small parity case spends its wall time (suite hygiene): encode, oracle decode, context creation, GPU stages."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, libjxl_amd as J, jxlo
def T(label, f):
    t = time.time(); r = f(); print("%-28s %.3f s" % (label, time.time() - t)); return r
for rep in range(3):
    print("--- rep", rep)
    img = T("synth", lambda: J.synth_image(600, 400, seed=rep))
    data = T("encode", lambda: J.encode_rgb8(img))
    o = T("oracle Decoded(dumps)", lambda: jxlo.Decoded(data))
    c = T("HipContext()", lambda: J.HipContext())
    f = T("Frame()", lambda: J.Frame(data))
    T("upload", lambda: c.upload(f))
    T("entropy+sync", lambda: (c.run_entropy(), c.sync()))
    T("download coeffs", lambda: c.download("coeffs"))
    T("oracle planes coeffs", lambda: o.planes("coeffs"))
    T("transform+sync", lambda: (c.run_transform(), c.sync()))
    T("filter+sync", lambda: (c.run_filter_color(), c.sync()))
    T("rgb8", lambda: c.rgb8())
    T("close", lambda: (c.close(), f.close(), o.close()))
