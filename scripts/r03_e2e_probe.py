#!/usr/bin/env python3
"""End-to-end leg of bench.py alone (compressed bytes in host memory -> RGB8 in pinned host memory), for probing the host
thread split: JXLAMD_E2E_MOVERS / JXLAMD_E2E_PARSERS override the pool sizes. usage: r03_e2e_probe.py [frames]  (GPU box)"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import libjxl_amd as J  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
datas = [bench.make_stream(3840, 2160, 1.0, seed=177 + i) for i in range(8)]
r = bench.end_to_end(J, datas, n, 0, 3840, 2160)
r.pop("span", None)
print(json.dumps(r))
