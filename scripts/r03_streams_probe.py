#!/usr/bin/env python3
"""Does the mere existence of many HIP streams (one per idle context) change how long an 8-frame entropy launch takes?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, libjxl_amd as J
frames = [J.Frame(bench.make_stream(3840, 2160, 1.0, 177 + i), threads=8) for i in range(8)]
ctxs = [J.HipContext(0) for _ in range(8)]
for i, c in enumerate(ctxs):
    c.upload(frames[i])
def run(tag):
    ms = []
    for _ in range(4):
        J.run_entropy_batch(ctxs)
        ctxs[0].sync()
        ms.append(ctxs[0].stage_ms(0))
    print("%s: entropy %.2f ms/launch" % (tag, min(ms[1:])), flush=True)
run("8 contexts")
idle = [J.HipContext(0) for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 56)]
run("8 busy + %d idle contexts" % len(idle))
for c in idle[:8]:
    c.upload(frames[0])  # (their streams have been used once)
run("... after 8 of the idle ones uploaded a frame")
for c in idle:
    c.upload(frames[0])
run("... after all idle ones uploaded a frame")
