"""Generates tests/golden/fjxl_*.jxl with the REFERENCE's own standalone lossless encoder.

The encoder binary is oracle/_ref/fjxl_enc, compiled in place from /root/reference/lib/jxl/enc_fast_lossless.cc by
`make -C oracle ref` (nothing of the reference is copied into this repository: the committed files are encoder OUTPUT,
i.e. data).  The expected pixels are not stored: every fixture's input image is a deterministic function of its name
(see `golden_image`), which the tests re-evaluate and compare bit-exactly with what the oracle decodes.

Run from the repository root in the authoring container:  python tests/golden/make_fjxl_golden.py
"""
import hashlib
import json
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
ENC = os.path.join(ROOT, "oracle", "_ref", "fjxl_enc")

# name: (width, height, channels, kind, effort)
CASES = {
    "fjxl_1x1_rgb_e2": (1, 1, 3, "noise", 2),
    "fjxl_7x5_rgb_e0": (7, 5, 3, "ramp", 0),
    "fjxl_37x29_rgba_e2": (37, 29, 4, "smooth", 2),
    "fjxl_64x64_gray_e5": (64, 64, 1, "smooth", 5),
    "fjxl_64x64_graya_e2": (64, 64, 2, "noise", 2),
    "fjxl_256x256_rgb_e1": (256, 256, 3, "smooth", 1),
    "fjxl_300x280_rgb_e2": (300, 280, 3, "smooth", 2),      # 4 groups: TOC, multi-group Modular, RCT
    "fjxl_300x280_rgb_e2_noise": (300, 280, 3, "noise", 2),  # incompressible: long prefix codes, raw bits
    "fjxl_520x260_rgba_e5": (520, 260, 4, "ramp", 5),        # LZ77 run-length path
    "fjxl_100x100_rgb_flat_e2": (100, 100, 3, "flat", 2),    # palette + RLE
}


def golden_image(name):
    w, h, nc, kind, _ = CASES[name]
    seed = int(hashlib.sha256(name.encode()).hexdigest()[:8], 16)
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    if kind == "ramp":
        a = np.stack([(x * 3 + y * (c + 1)) % 256 for c in range(nc)], -1)
    elif kind == "noise":
        a = rng.integers(0, 256, (h, w, nc))
    elif kind == "flat":
        a = np.zeros((h, w, nc), np.int64) + np.array([200, 40, 90, 255][:nc])
        a[h // 3: h // 2, w // 4: w // 2] = np.array([10, 250, 30, 128][:nc])
    else:
        a = np.stack([128 + 60 * np.sin(x / 17.0 + c) + 50 * np.cos(y / 23.0) for c in range(nc)], -1)
        a = a + rng.integers(-3, 4, (h, w, nc))
        a[h // 4: h // 2, w // 4: w // 2] = 37
    return np.clip(a, 0, 255).astype(np.uint8)


def main():
    assert os.path.exists(ENC), "build the reference encoder first: make -C oracle ref"
    manifest = {}
    for name, (w, h, nc, kind, effort) in CASES.items():
        img = golden_image(name)
        raw = os.path.join("/tmp", name + ".raw")
        img.tofile(raw)
        out = os.path.join(HERE, name + ".jxl")
        subprocess.run([ENC, raw, str(w), str(h), str(nc), "8", str(effort), out], check=True)
        manifest[name] = {"width": w, "height": h, "channels": nc, "kind": kind, "effort": effort,
                          "jxl_bytes": os.path.getsize(out), "pixels_sha256": hashlib.sha256(img.tobytes()).hexdigest()}
        print(name, manifest[name]["jxl_bytes"])
    json.dump(manifest, open(os.path.join(HERE, "fjxl_manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
