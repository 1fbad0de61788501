#!/usr/bin/env python3
"""Measurement aid (GPU box): where an end-to-end frame's time goes on the host side: parse at several thread counts,
parses in parallel, upload, download into pageable and pinned memory."""
import concurrent.futures
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402

data = bench.make_stream(3840, 2160, 1.0)
for th in (1, 4, 16):
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        f = J.Frame(data, th)
        t.append(time.perf_counter() - t0)
        f.close()
    print("parse, %2d threads: %.2f ms" % (th, min(t) * 1e3))
for workers, th in ((8, 4), (16, 4), (32, 2), (32, 1), (64, 1)):
    with concurrent.futures.ThreadPoolExecutor(workers) as pool:
        t0 = time.perf_counter()
        fr = list(pool.map(lambda _: J.Frame(data, th), range(workers * 4)))
        dt = time.perf_counter() - t0
    print("%2d parses in flight x %d threads: %.2f ms per frame" % (workers, th, dt / len(fr) * 1e3))
    for f in fr:
        f.close()
c = J.HipContext(0)
f = J.Frame(data, 8)
t = []
for _ in range(5):
    t0 = time.perf_counter()
    c.upload(f)
    t.append(time.perf_counter() - t0)
print("upload: %.2f ms" % (min(t) * 1e3))
c.run_all()
c.sync()
a = np.empty((2160, 3840, 3), np.uint8)
p = torch.empty((2160, 3840, 3), dtype=torch.uint8, pin_memory=True).numpy()
for name, buf in (("pageable", a), ("pinned", p)):
    t = []
    for _ in range(5):
        t0 = time.perf_counter()
        J._check(J.lib().jxlhip_download_rgb8(c._h, buf.ctypes.data, 3840 * 3), "download")
        t.append(time.perf_counter() - t0)
    print("download into %s memory: %.2f ms" % (name, min(t) * 1e3))
