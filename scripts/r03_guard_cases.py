#!/usr/bin/env python3
"""Which kernel tour case misbehaves under JXLHIP_GUARD=1 with poisoned allocations: every case in a process of its own
(a device fault aborts the process), exit code and the end of stderr per case. usage (GPU box): r03_guard_cases.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [((257, 255), dict()), ((512, 512), dict(distance=2.0)), ((520, 300), dict(distance=4.5, gab=0)),
         ((300, 200), dict(upsampling=2)), ((520, 300), dict(num_passes=2)), ((300, 280), dict(ac_code_mode=3)),
         ((600, 400), dict(noise=40)), ((777, 513), dict(strategy_mode=2, random_cmap=1)), ((64, 64), dict(big_coeffs=1, strategy_mode=2)),
         ((520, 300), dict(distance=4.5, gab=0, epf_iters=3)), ((300, 200), dict(upsampling=4))]
CHILD = r"""
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import libjxl_amd as J
size, kw = %r
data = J.encode_random(size[0], size[1], **kw) if kw.get("big_coeffs") else J.encode_rgb8(J.synth_image(size[0], size[1], seed=5), **kw)
f = J.Frame(data, threads=2)
c = J.HipContext()
c.upload(f)
c.run_all()
c.sync()
h = hashlib.sha256()
for a in (c.download("xyb_idct"), c.rgb8()):
    h.update(np.ascontiguousarray(a).tobytes())
print("guards", c.check_guards(), h.hexdigest()[:16])
"""
for case in CASES:
    line = [repr(case)]
    for byte in ("0xA5", "0x00", "0xFF"):
        env = dict(os.environ, JXLHIP_GUARD="1", JXLHIP_GUARD_BYTE=byte)
        r = subprocess.run([sys.executable, "-c", CHILD % (ROOT, case)], capture_output=True, text=True, timeout=300, env=env)
        line.append("%s rc=%d %s %s" % (byte, r.returncode, r.stdout.strip()[-40:], r.stderr.strip()[-200:].replace("\n", " | ") if r.returncode else ""))
    print("  ".join(line), flush=True)
