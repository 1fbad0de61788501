#!/usr/bin/env python3
"""Per-channel / per-column-parity count of 8-bit differences between the GPU decode and the oracle (debug aid)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, libjxl_amd as J, jxlo
for (w, h) in ((64, 64), (700, 520)):
    data = J.encode_rgb8(J.synth_image(w, h, seed=w))
    out = J.decode_rgb8(data)
    ref = jxlo.Decoded(data, dumps=False).rgb8
    d = out.astype(int) - ref.astype(int)
    print(w, h, "per channel", [(d[:, :, c] != 0).mean() for c in range(3)])
    print("  per x parity", [(d[:, e::2, :] != 0).mean() for e in range(2)], "per y parity", [(d[e::2] != 0).mean() for e in range(2)])
    print("  x&31 hist ch0", [int((d[:, x::32, 0] != 0).sum()) for x in range(32)])
