#!/usr/bin/env python3
"""Measurement + full-size parity on REFERENCE-encoder output (GPU box; not collected by pytest): 3840x2160 images
encoded by the reference's own lossless encoder (oracle/_ref/fjxl_enc = lib/jxl/enc_fast_lossless.cc built by
`make -C oracle ref`), decoded as ONE set by the HIP Modular path; every frame must equal its input bit for bit.
usage: python tests/measure_fjxl_4k.py [frames=96] [effort=2] [distinct=4]"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import libjxl_amd as J  # noqa: E402

FJXL = os.path.join(ROOT, "oracle", "_ref", "fjxl_enc")


def main():
    frames = int(sys.argv[1]) if len(sys.argv) > 1 else 96
    effort = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    distinct = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    if not os.path.exists(FJXL):
        raise SystemExit("oracle/_ref/fjxl_enc is missing (make -C oracle ref, where /root/reference exists)")
    w, h = 3840, 2160
    imgs, datas = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(distinct):
            img = J.synth_image(w, h, seed=61 + i)
            raw, out = os.path.join(tmp, "in.raw"), os.path.join(tmp, "out.jxl")
            img.tofile(raw)
            subprocess.run([FJXL, raw, str(w), str(h), "3", "8", str(effort), out], check=True)
            imgs.append(img)
            datas.append(open(out, "rb").read())
    mods = [J.ModFrame(d) for d in datas]
    ctxs = [J.HipContext(0) for _ in range(frames)]
    for i, c in enumerate(ctxs):
        c.upload_modular(mods[i % distinct])
    J.run_modular_batch(ctxs)  # warm-up
    ctxs[0].sync()
    times = []
    for _ in range(3):
        t0 = time.perf_counter()
        J.run_modular_batch(ctxs)
        ctxs[0].sync()
        times.append(time.perf_counter() - t0)
    for i in list(range(0, frames, max(1, frames // 8))) + [frames - 1]:
        r, status, _ = ctxs[i].modular_status()
        assert r == 0 and not any(status), (i, status)
        assert np.array_equal(ctxs[i].pixels(), imgs[i % distinct]), "frame %d differs from the encoder's input" % i
    best = min(times)
    print("fjxl effort %d, %d x 3840x2160 frames per launch (%d distinct, %.2f bpp, %d streams per frame): %.1f ms = %.0f MP/s, bit-exact"
          % (effort, frames, distinct, sum(len(d) for d in datas) * 8.0 / (distinct * w * h), mods[0].info["num_streams"], best * 1e3,
             frames * w * h * 1e-6 / best))


if __name__ == "__main__":
    main()
