#!/usr/bin/env python3
"""End-to-end single-frame timing through the JxlDecoder-level pieces: host parse (headers, DC groups, tables), H2D
upload, the three GPU stages, D2H download. usage: e2e_time.py [WxH]   (GPU box only)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402

size = sys.argv[1] if len(sys.argv) > 1 else "3840x2160"
xs, ys = [int(v) for v in size.split("x")]
data = bench.make_stream(xs, ys, 1.0)
c = J.HipContext(0)
best = None
for rep in range(4):
    t0 = time.perf_counter()
    f = J.Frame(data, threads=16)
    t1 = time.perf_counter()
    c.upload(f)
    c.sync()
    t2 = time.perf_counter()
    c.run_all()
    c.sync()
    t3 = time.perf_counter()
    rgb = c.rgb8()
    t4 = time.perf_counter()
    f.close()
    cur = (t4 - t0, t1 - t0, t2 - t1, t3 - t2, t4 - t3)
    if best is None or cur[0] < best[0]:
        best = cur
px = xs * ys
print("%s, %d bytes: total %.1f ms = parse %.1f + upload %.1f + gpu %.1f + download %.1f  -> %.1f MP/s end to end" % (
    size, len(data), best[0] * 1e3, best[1] * 1e3, best[2] * 1e3, best[3] * 1e3, best[4] * 1e3, px * 1e-6 / best[0]))
c.close()
