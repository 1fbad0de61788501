#!/usr/bin/env python3
"""Measurement aid (GPU box): host time of frame uploads, alone and from several threads (JXLHIP_UPLOAD_PROF=1 adds the
library's own per-phase times on stderr)."""
import concurrent.futures
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402

data = bench.make_stream(3840, 2160, 1.0)
f = J.Frame(data, 8)
ctxs = [J.HipContext(0) for _ in range(8)]
for c in ctxs:
    c.upload(f)
for n in (1, 2, 4, 8):
    with concurrent.futures.ThreadPoolExecutor(n) as pool:
        t0 = time.perf_counter()
        list(pool.map(lambda i: [ctxs[i].upload(f) for _ in range(10)], range(n)))
        dt = time.perf_counter() - t0
    print("%d uploader threads: %.2f ms per upload per thread, %.2f ms per frame overall" % (n, dt / 10 * 1e3, dt / (10 * n) * 1e3))
