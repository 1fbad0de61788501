#!/usr/bin/env python3
"""Entropy-stage time of a progressive (two-pass) 4K frame set: lane kernel + pass merge against the wave-per-section
fallback kernels (JXLHIP_ENTROPY=1, set by the caller). usage: progressive_time.py [FRAMES]   (GPU box only)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libjxl_amd as J  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
data = J.encode_rgb8(J.synth_image(3840, 2160, seed=177), num_passes=2)
frame = J.Frame(data, threads=8)
ctxs = [J.HipContext(0) for _ in range(n)]
for c in ctxs:
    c.upload(frame)
ms = []
for _ in range(3):
    for c in ctxs:
        c.sync()
    t0 = time.perf_counter()
    J.run_entropy_batch(ctxs)
    for c in ctxs:
        c.sync()
    ms.append((time.perf_counter() - t0) * 1e3)  # wall clock: the fallback runs one launch per frame
J.run_transform_batch(ctxs)
J.run_filter_color_batch(ctxs)
ctxs[0].sync()
ref = J.decode_rgb8(data)
same = all((c.rgb8() == ref).all() for c in ctxs[:2])
print("progressive 4K x %d, JXLHIP_ENTROPY=%s: entropy %.2f ms/set (%.3f ms/frame), pixels equal to the single-frame decode: %s" % (
    n, os.environ.get("JXLHIP_ENTROPY", "2"), min(ms), min(ms) / n, same), flush=True)
for c in ctxs:
    c.close()
