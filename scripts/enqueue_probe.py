#!/usr/bin/env python3
"""Measurement aid (GPU box): host time of the three batch calls on freshly uploaded sets of 256 frames, one set at a time
and two sets from two threads (the end-to-end pipeline's runner threads)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402

data = bench.make_stream(3840, 2160, 1.0)
f = J.Frame(data, 8)
N = 256
sets = [[J.HipContext(0) for _ in range(N)] for _ in range(2)]


def cycle(cs, tag, out):
    for c in cs:
        c.upload(f)
    t = [time.perf_counter()]
    J.run_entropy_batch(cs); t.append(time.perf_counter())
    J.run_transform_batch(cs); t.append(time.perf_counter())
    J.run_filter_color_batch(cs); t.append(time.perf_counter())
    cs[0].sync(); t.append(time.perf_counter())
    out.append("%s host ms: entropy %.1f transform %.1f filter %.1f sync %.1f | kernel ms %s" % (
        tag, (t[1] - t[0]) * 1e3, (t[2] - t[1]) * 1e3, (t[3] - t[2]) * 1e3, (t[4] - t[3]) * 1e3,
        ["%.1f" % cs[0].stage_ms(k) for k in range(3)]))


out = []
for rep in range(3):
    cycle(sets[0], "alone %d" % rep, out)
for rep in range(2):
    ths = [threading.Thread(target=cycle, args=(sets[k], "pair %d.%d" % (rep, k), out)) for k in range(2)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
print("\n".join(out))
