#!/usr/bin/env python3
"""Sweeps the work-distribution knobs of the lane-parallel entropy kernel on the benchmark stream (GPU box only).
usage: entropy_sweep.py BATCH "LANES:WAIT_SHIFT[:TARGET_WAVES[:SPREAD[:WAVES_PER_WG]]]" ...   (LANES 0 = automatic; SPREAD = percent of the minimum lane count)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402


def main():
    batch = int(sys.argv[1])
    data = bench.make_stream(3840, 2160, 1.0)
    frame = J.Frame(data, threads=8)
    ctxs = [J.HipContext(0) for _ in range(batch)]
    for cfg in sys.argv[2:]:
        parts = cfg.split(":")
        os.environ["JXLHIP_LANES"] = parts[0]
        os.environ["JXLHIP_WAIT_SHIFT"] = parts[1]
        if len(parts) > 2:
            os.environ["JXLHIP_TARGET_WAVES"] = parts[2]
        if len(parts) > 3:
            os.environ["JXLHIP_SPREAD"] = parts[3]
        if len(parts) > 4:
            os.environ["JXLHIP_WPG"] = parts[4]
        for c in ctxs:
            c.upload(frame)  # new generation -> the batch description is rebuilt with the new knobs
        ms = []
        for it in range(4):
            J.run_entropy_batch(ctxs)
            ctxs[0].sync()
            ms.append(ctxs[0].stage_ms(0))
        r, flags = ctxs[0].errors()
        print("batch %d lanes %s wait_shift %s %s: entropy %.2f ms/launch (%.3f ms/frame) err=%d" % (
            batch, parts[0], parts[1], ":".join(parts[2:]), min(ms[1:]), min(ms[1:]) / batch, r), flush=True)
    for c in ctxs:
        c.close()


if __name__ == "__main__":
    main()
