#!/usr/bin/env python3
"""Turns the two counter passes of tools/profile_traffic.sh into the per-launch HBM traffic of the dominant kernel
(the 3x3 reflect 128->128 resblock convolution forward: grid 4096 workgroups at batch 32, 256x256).

    python tools/summarize_traffic.py <kernel-name-substring> <out.json> [grid_size]

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE are in
KB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced streaming read (x2); WRITE_SIZE is exact."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime)


def mean_counter(path, counter, needle, grid):
    vals = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter and needle in r["Kernel_Name"] and (grid is None or int(r["Grid_Size"]) == grid):
            vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


def main():
    needle, out = sys.argv[1], sys.argv[2]
    grid = int(sys.argv[3]) if len(sys.argv) > 3 else 4096 * 512   # 4096 tiles x 512 threads (wave-specialised kernel)
    f, nf = mean_counter(newest(ROOT + "/gpurun_out/pmc_fetch/**/*counter_collection.csv"), "FETCH_SIZE", needle, grid)
    w, nw = mean_counter(newest(ROOT + "/gpurun_out/pmc_write/**/*counter_collection.csv"), "WRITE_SIZE", needle, grid)
    N, H, C, K = 32, 128, 128, 3
    algo = 2 * N * H * H * C * 4 + K * K * C * C * 4  # read x once, write y once, weights once (fp32 bytes)
    js = {"kernel": needle + " (grid %d threads = resblock 3x3 reflect 128->128 forward, N=32, 128x128)" % grid,
          "launches": nf, "FETCH_SIZE_KB_mean": f, "WRITE_SIZE_KB_mean": w,
          "correction": "gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE x2 for wide coalesced streaming reads; WRITE_SIZE exact; "
                        "both in KB; memory-side (fabric) requests, Infinity-Cache hits included",
          "hbm_bytes_per_launch": 2 * f * 1024 + w * 1024, "algorithmic_bytes_per_launch": algo,
          "command": "tools/profile_traffic.sh && python tools/summarize_traffic.py '%s' <out>" % needle}
    json.dump(js, open(out, "w"), indent=1)
    print(json.dumps(js))


if __name__ == "__main__":
    main()
