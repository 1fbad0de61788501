#!/usr/bin/env python3
"""Single-image wall time through the JxlDecoder API (djxl's definition) of a 3840x2160 VarDCT frame WITH an alpha channel
(RGBA8 out): the alpha plane is a Modular extra channel whose groups sit behind the AC groups' coefficients and decode on
the host (on the caller's runner since round 3). usage (GPU box): r03_rgba_single.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402

img = J.synth_image(3840, 2160, seed=177)
alpha = (np.mgrid[0:2160, 0:3840][1] * 255 // 3839).astype(np.uint8)
alpha ^= (img[..., 1] >> 3)  # (not a flat ramp: some entropy)
data = J.encode_rgba8(np.dstack([img, alpha]))
print(json.dumps({"rgba": bench.single_image_api(J, data, 3840, 2160, reps=6, channels=4), "bytes": len(data)}))
print(json.dumps({"rgb": bench.single_image_api(J, bench.make_stream(3840, 2160, 1.0), 3840, 2160, reps=6)}))
