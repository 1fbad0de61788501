"""Measurement aid: N decodes of one 4K frame through the JxlDecoder C API (the span bench.py's e2e.single_image times),
meant to run under `rocprofv3 --kernel-trace --memory-copy-trace --hip-trace` so that one repetition's timeline (host
parse, uploads, the three stages, the download) can be read from the traces."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402

data = bench.make_stream(3840, 2160, 1.0, 177)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    t0 = time.perf_counter()
    r = bench.single_image_api(J, data, 3840, 2160, reps=3)
    print(i, r, time.perf_counter() - t0, flush=True)
