#!/usr/bin/env python3
"""Which per-context resource slows a long kernel down once there are many contexts: plain HIP streams? events? pinned memory?"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench, libjxl_amd as J
frames = [J.Frame(bench.make_stream(3840, 2160, 1.0, 177 + i), threads=8) for i in range(8)]
ctxs = [J.HipContext(0) for _ in range(8)]
for i, c in enumerate(ctxs):
    c.upload(frames[i])
def run(tag):
    ms = []
    for _ in range(3):
        J.run_entropy_batch(ctxs)
        ctxs[0].sync()
        ms.append(ctxs[0].stage_ms(0))
    print("%s: entropy %.2f ms/launch" % (tag, min(ms[1:])), flush=True)
hip = ctypes.CDLL("libamdhip64.so")
run("8 contexts")
ev = [ctypes.c_void_p() for _ in range(2000)]
for e in ev:
    hip.hipEventCreate(ctypes.byref(e))
run("+ 2000 events")
pinned = [ctypes.c_void_p() for _ in range(56)]
for p in pinned:
    hip.hipHostMalloc(ctypes.byref(p), ctypes.c_size_t(8 << 20), 0)
run("+ 56 pinned buffers of 8 MB")
for n in (8, 16, 24, 40, 56):
    st = [ctypes.c_void_p() for _ in range(8)]
    for s in st:
        hip.hipStreamCreateWithFlags(ctypes.byref(s), 1)  # hipStreamNonBlocking
    run("+ %d plain streams" % n)
