#!/usr/bin/env python3
"""Lane kernel alone on N frames cycling through D distinct benchmark frames (seeds 177..): does a launch of copies of the
same frames behave like a launch of different frames? usage: r03_distinct_probe.py N D [D ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, libjxl_amd as J
n = int(sys.argv[1])
dmax = max(int(a) for a in sys.argv[2:])
frames = [J.Frame(bench.make_stream(3840, 2160, 1.0, 177 + i), threads=8) for i in range(dmax)]
ctxs = [J.HipContext(0) for _ in range(n)]
for a in sys.argv[2:]:
    d = int(a)
    for i, c in enumerate(ctxs):
        c.upload(frames[i % d])
    ms = []
    for _ in range(4):
        J.run_entropy_batch(ctxs)
        ctxs[0].sync()
        ms.append(ctxs[0].stage_ms(0))
    alone = []
    print("frames %d distinct %d: entropy %.2f ms/launch" % (n, d, min(ms[1:])), flush=True)
for c in ctxs:
    c.close()
