#!/usr/bin/env python3
"""Isolated stage timings of one resident 4K frame (no other stream active): transform and filter+colour (GPU box only)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402

data = bench.make_stream(3840, 2160, 1.0)
frame = J.Frame(data, threads=8)
c = J.HipContext(0)
c.upload(frame)
c.run_entropy()
c.sync()
px = 3840 * 2160
for name, fn, which, alg in (("transform", c.run_transform, 1, 18.4 * px), ("filter+colour", c.run_filter_color, 2, 15.06 * px)):
    ms = []
    for _ in range(6):
        fn()
        c.sync()
        ms.append(c.stage_ms(which))
    best = min(ms[1:])
    print("%s: %.3f ms  (%.1f GB/s algorithmic)" % (name, best, alg / best / 1e6), flush=True)
c.close()
