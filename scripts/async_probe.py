#!/usr/bin/env python3
"""Measurement aid (GPU box): stage times of a 256-frame set after synchronous / asynchronous table uploads."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
data = bench.make_stream(3840, 2160, 1.0)
f = J.Frame(data, 8)
ctxs = [J.HipContext(0) for _ in range(256)]
t0 = time.perf_counter()
for c in ctxs:
    c.upload(f)
    if mode == "syncdev":
        c.sync()
print(mode, "uploads: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
for rep in range(4):
    J.run_entropy_batch(ctxs)
    J.run_transform_batch(ctxs)
    J.run_filter_color_batch(ctxs)
    ctxs[0].sync()
    if rep == 0 and mode == "syncall":
        for c in ctxs:
            c.sync()
    print(mode, rep, ["%.1f" % ctxs[0].stage_ms(k) for k in range(3)])
