#!/usr/bin/env python3
"""Isolated timings of the batched transform and filter+colour launches over a resident frame set (no other kernel
running): min and median of 12 launches. usage: stage_times.py [FRAMES]   (GPU box only)"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
data = bench.make_stream(3840, 2160, 1.0)
frame = J.Frame(data, threads=8)
ctxs = [J.HipContext(0) for _ in range(n)]
for c in ctxs:
    c.upload(frame)
J.run_entropy_batch(ctxs)
ctxs[0].sync()
px = 3840 * 2160
for name, fn, which, alg in (("transform", J.run_transform_batch, 1, 18.4 * px), ("filter+colour", J.run_filter_color_batch, 2, 15.06 * px)):
    ms = []
    for _ in range(12):
        fn(ctxs)
        ctxs[0].sync()
        ms.append(ctxs[0].stage_ms(which) / n)
    print("%s: min %.4f median %.4f ms/frame  (%.0f GB/s algorithmic at min)" % (name, min(ms), statistics.median(ms), alg / min(ms) / 1e6), flush=True)
for c in ctxs:
    c.close()
