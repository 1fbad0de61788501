#!/usr/bin/env python3
"""Round-3 probe of the lane-parallel entropy kernel alone (GPU box only): BATCH frames cycling through 8 distinct benchmark
streams, one launch at a time, under the environment settings given as "K=V,K=V" strings (one configuration each).
usage: r03_entropy_probe.py BATCH "CFG" ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402


def main():
    batch = int(sys.argv[1])
    datas = [bench.make_stream(3840, 2160, 1.0, 177 + i) for i in range(8)]
    frames = [J.Frame(d, threads=8) for d in datas]
    ctxs = [J.HipContext(0) for _ in range(batch)]
    for cfg in sys.argv[2:]:
        keys = []
        for kv in cfg.split(","):
            if "=" in kv:
                k, v = kv.split("=")
                os.environ[k] = v
                keys.append(k)
        for i, c in enumerate(ctxs):
            c.upload(frames[i % 8])
        ms = []
        for it in range(4):
            J.run_entropy_batch(ctxs)
            ctxs[0].sync()
            ms.append(ctxs[0].stage_ms(0))
        r, flags = ctxs[0].errors()
        print("batch %d [%s]: entropy %.2f ms/launch (%.4f ms/frame) err=%d" % (batch, cfg, min(ms[1:]), min(ms[1:]) / batch, r), flush=True)
        for k in keys:
            del os.environ[k]
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
