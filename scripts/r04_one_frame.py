#!/usr/bin/env python3
"""The entropy stage of ONE 4K frame under several lane layouts (GPU box): what the single-image latency is made of.
usage: r04_one_frame.py ["K=V,K=V" ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, libjxl_amd as J
f = J.Frame(bench.make_stream(3840, 2160, 1.0, 177), threads=8)
for cfg in (sys.argv[1:] or ["base"]):
    keys = []
    for kv in cfg.split(","):
        if "=" in kv:
            k, v = kv.split("=")
            os.environ[k] = v
            keys.append(k)
    c = J.HipContext(0)
    c.upload(f)
    ms = []
    for _ in range(4):
        J.run_entropy_batch([c])
        c.sync()
        ms.append(c.stage_ms(0))
    c.run_transform(); c.run_filter_color(); c.sync()
    print("[%s] entropy %.2f ms (transform %.2f, filter %.2f)" % (cfg, min(ms[1:]), c.stage_ms(1), c.stage_ms(2)), flush=True)
    c.close()
    for k in keys:
        del os.environ[k]
