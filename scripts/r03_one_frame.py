#!/usr/bin/env python3
"""The lane kernel on ONE frame at a time (one 64-lane wave when JXLHIP_LANES=64): the chain of each benchmark frame alone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, libjxl_amd as J
for i in range(8):
    f = J.Frame(bench.make_stream(3840, 2160, 1.0, 177 + i), threads=8)
    c = J.HipContext(0)
    c.upload(f)
    ms = []
    for _ in range(3):
        J.run_entropy_batch([c])
        c.sync()
        ms.append(c.stage_ms(0))
    print("frame %d alone: %.2f ms" % (i, min(ms)), flush=True)
    c.close()
