#!/usr/bin/env python3
"""One-off randomized parity sweep (GPU box): random sizes / distances / filter settings / strategy sets / histogram
counts, GPU vs oracle: coefficients bit-exact, RGB8 within one level. usage: fuzz_parity.py [N] [SEED]"""
import os
import random
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import jxlo  # noqa: E402
import libjxl_amd as J  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for it in range(n):
    xs, ys = rnd.randint(1, 1400), rnd.randint(1, 1100)
    kw = dict(seed=rnd.randint(1, 10 ** 6), epf_iters=rnd.choice([-1, 0, 1, 2, 3]), gab=rnd.choice([-1, 0, 1]),
              num_histograms=rnd.choice([0, 1, 2, 7]), max_clusters=rnd.choice([0, 1, 4, 100]),
              upsampling=rnd.choice([0, 0, 0, 2, 4, 8]), num_passes=rnd.choice([1, 1, 1, 2]),
              custom_orders=rnd.choice([0, 1]), custom_bctx=rnd.choice([0, 1]), custom_cmap=rnd.choice([0, 0, 1]), custom_lf=rnd.choice([0, 0, 1]))
    if rnd.random() < 0.5:
        kind = "random"
        data = J.encode_random(xs, ys, strategy_mask=rnd.choice([0, 0, rnd.getrandbits(27) | 1]), **kw)
    else:
        kind = "image"
        seed = kw.pop("seed")
        data = J.encode_rgb8(J.synth_image(xs, ys, seed=seed), distance=rnd.choice([0.3, 1.0, 2.0, 4.5, 8.0]),
                             strategy_mode=rnd.choice([0, 1, 2]), random_cmap=rnd.choice([0, 1]), **kw)
    o = jxlo.Decoded(data, dumps=False)
    try:
        rgb = J.decode_rgb8(data, threads=4)
        d = np.abs(rgb.astype(int) - o.rgb8.astype(int))
        ok = d.max() <= 1 and (d > 0).mean() < 2e-3
    except Exception as e:  # noqa: BLE001
        ok = False
        print("EXC", e)
    o.close()
    print("%3d %-6s %4dx%-4d %s -> %s" % (it, kind, xs, ys, kw, "ok" if ok else "MISMATCH"), flush=True)
    bad += not ok
print("mismatches:", bad)
sys.exit(1 if bad else 0)
