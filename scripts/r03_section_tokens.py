#!/usr/bin/env python3
"""Token counts per AC section of the 8 benchmark frames (lane kernel's debug aid: bit 1 reports them in the flag words)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench, libjxl_amd as J
os.environ["JXLHIP_LANES_DEBUG"] = "2"
for i in range(8):
    d = bench.make_stream(3840, 2160, 1.0, 177 + i)
    f = J.Frame(d, threads=8)
    c = J.HipContext(0)
    c.upload(f)
    J.run_entropy_batch([c])
    c.sync()
    r, flags = c.errors()
    t = np.array(flags, dtype=np.int64)
    print("frame %d: sections %d tokens total %d mean %.0f max %d  sorted top5 %s  bytes %d" % (i, len(t), t.sum(), t.mean(), t.max(), sorted(t)[-5:], len(d)), flush=True)
    c.close()
