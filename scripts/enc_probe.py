"""Forward (encoder) path probe: 3840x2160 synthetic image, GPU forward path vs the CPU writer's model, timings."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import libjxl_amd as J

xs, ys = (int(v) for v in (sys.argv[1:3] if len(sys.argv) > 2 else (3840, 2160)))
img = J.synth_image(xs, ys, seed=7)
ctx = J.HipContext()
for i in range(3):
    t = {}
    t0 = time.time()
    data = J.encode_rgb8_gpu(img, ctx, timings=t, distance=1.0, cfl_fit=1)
    print("gpu encode: total %.3f s, forward call %.3f s, kernels %.3f ms (%.0f MP/s), assemble %.3f s, %d bytes" % (
        time.time() - t0, t["forward_s"], t["kernels_ms"], xs * ys / t["kernels_ms"] / 1e3, t["assemble_s"], len(data)), flush=True)
if "--compare" in sys.argv:
    t0 = time.time()
    ref = J.encode_rgb8(img, distance=1.0, cfl_fit=1)
    print("cpu encode: %.3f s, %d bytes" % (time.time() - t0, len(ref)))
    g, c = J.enc_forward_model(img, ctx, distance=1.0, cfl_fit=1), J.enc_forward_model(img, None, distance=1.0, cfl_fit=1)
    for k in g:
        print(k, "differs on %.5f%%" % (100.0 * (g[k] != c[k]).mean()), "max", int(np.abs(g[k].astype(np.int64) - c[k]).max()))
ctx.close()
