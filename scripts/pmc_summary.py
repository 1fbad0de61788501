#!/usr/bin/env python3
"""Condenses rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs into a per-kernel JSON summary.
usage: pmc_summary.py FETCH.csv WRITE.csv OUT.json FRAMES_PER_LAUNCH   (counter values are KiB per dispatch, see MI355X_MICROARCH.md HBM)"""
import collections
import csv
import json
import os
import sys


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write = load(sys.argv[1]), load(sys.argv[2])
    try:
        import subprocess
        commit = subprocess.check_output(["git", "-C", os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "rev-parse", "--short", "HEAD"],
                                         stderr=subprocess.DEVNULL).decode().strip()
    except Exception:  # noqa: BLE001 (the GPU box has no .git: the caller stamps the commit afterwards)
        commit = os.environ.get("JXL_PROFILE_COMMIT", "unknown")
    out = {"_frames_per_launch": int(sys.argv[4]), "_commit": commit, "_unit": "KiB per dispatch (rocprofv3 FETCH_SIZE / WRITE_SIZE, separate passes)",
           "_gfx950_correction": "MI355X_MICROARCH.md (HBM): FETCH_SIZE tallies 128-byte requests at 64 bytes on gfx950, i.e. reports half "
                                 "the bytes of wide (16 B/lane) reads: fetch_kib_corrected = 2 x FETCH_SIZE; WRITE_SIZE is exact for "
                                 "16 B/lane stores. Narrower access widths are uncalibrated in the guide: the corrected figure is an "
                                 "upper estimate for them. hbm_kib_per_dispatch = fetch_kib_corrected + write"}
    for k in sorted(set(fetch) | set(write)):
        if "jxlhip" not in k:
            continue
        out[k] = {"dispatches": fetch.get(k, write.get(k))[0],
                  "fetch_kib_per_dispatch": round(fetch[k][1] / fetch[k][0], 1) if k in fetch else None,
                  "write_kib_per_dispatch": round(write[k][1] / write[k][0], 1) if k in write else None}
        f, w = out[k]["fetch_kib_per_dispatch"], out[k]["write_kib_per_dispatch"]
        out[k]["fetch_kib_corrected"] = round(2 * f, 1) if f is not None else None
        out[k]["hbm_kib_per_dispatch"] = round(2 * f + w, 1) if f is not None and w is not None else None
    json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
