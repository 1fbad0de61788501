#!/usr/bin/env python3
"""How much a 640-frame entropy launch loses to what runs beside it: alone, beside a loop of transform launches of
another frame set, beside a loop of filter+colour launches (GPU box only). usage: r03_interference.py [FRAMES [SIDE]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, libjxl_amd as J
n = int(sys.argv[1]) if len(sys.argv) > 1 else 640
m = int(sys.argv[2]) if len(sys.argv) > 2 else 64
frames = [J.Frame(bench.make_stream(3840, 2160, 1.0, 177 + i), threads=8) for i in range(8)]
A = [J.HipContext(0) for _ in range(n)]
B = [J.HipContext(0) for _ in range(m)]
for i, c in enumerate(A + B):
    c.upload(frames[i % 8])
B[0].set_option("filter_async", 1)
J.run_entropy_batch(A); J.run_entropy_batch(B); J.run_transform_batch(B)
A[0].sync(); B[0].sync()
first = len(sys.argv) > 3 and sys.argv[3] == "side-first"  # the side work is already running when the entropy launch arrives
for name, side in (("alone", None), ("beside transform launches", J.run_transform_batch), ("beside filter+colour launches", J.run_filter_color_batch),
                   ("alone again", None)):
    for rep in range(2):
        loops = 0
        if side is not None and first:
            for _ in range(6):
                side(B)
                loops += 1
        J.run_entropy_batch(A)
        if side is not None:
            per = 640 // m
            for _ in range(3 * per):  # (about 3 x the entropy launch's length of side work, queued at once)
                side(B)
                loops += 1
        A[0].sync(); B[0].sync()
        print("%-32s entropy launch %.1f ms  (%d side launches of %d frames, last one %.2f ms)" % (
            name, A[0].stage_ms(0), loops, m, B[0].stage_ms(1 if side is J.run_transform_batch else 2) if side else 0.0))
