#!/usr/bin/env python3
"""Isolated timings of the batched transform and filter+colour launches over a resident frame set of the 8 distinct benchmark
frames (no other kernel running): min and median of 8 launches, under the environment settings of each "K=V,K=V" argument.
usage: r03_stage_times.py FRAMES [CFG ...]   (GPU box only)"""
import os, statistics, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, libjxl_amd as J
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
frames = [J.Frame(bench.make_stream(3840, 2160, 1.0, 177 + i), threads=8) for i in range(8)]
ctxs = [J.HipContext(0) for _ in range(n)]
for i, c in enumerate(ctxs):
    c.upload(frames[i % 8])
J.run_entropy_batch(ctxs)
ctxs[0].sync()
px = 3840 * 2160
for cfg in (sys.argv[2:] or ["base"]):
    keys = []
    for kv in cfg.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            os.environ[k] = v
            keys.append(k)
    for name, fn, which, alg in (("transform", J.run_transform_batch, 1, 18.4 * px), ("filter+colour", J.run_filter_color_batch, 2, 15.06 * px)):
        ms = []
        for _ in range(8):
            fn(ctxs)
            ctxs[0].sync()
            ms.append(ctxs[0].stage_ms(which) / n)
        print("   ", " ".join("%.4f" % m for m in ms))
        print("[%s] %s: min %.4f median %.4f ms/frame  (%.0f GB/s algorithmic at median, %.3f of 8 TB/s)" % (cfg, name, min(ms), statistics.median(ms), alg / statistics.median(ms) / 1e6, alg / statistics.median(ms) / 1e6 / 8000), flush=True)
    for k in keys:
        del os.environ[k]
for c in ctxs:
    c.close()
