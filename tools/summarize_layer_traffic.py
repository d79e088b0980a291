#!/usr/bin/env python3
"""Per-layer HBM traffic from the counter passes of tools/profile_layer_traffic.sh.

    python tools/summarize_layer_traffic.py <tag> <out.json>

The microbenchmark brackets every (layer, pass) segment with pad_vector_kernel launches (--mark); dispatches between the
2nd and 3rd marker-free gap of a segment are its timed launches.  Corrections per /opt/skills/guides/MI355X_MICROARCH.md
(HBM / rocprofv3): FETCH_SIZE and WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half the bytes of a wide coalesced
streaming read (x2); WRITE_SIZE is exact."""
import csv
import re
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def segments(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    segs, cur, inside = [], [], False
    for r in rows:
        if "pad_vector_kernel" in r["Kernel_Name"] and int(r["Grid_Size"]) == 1792:
            if inside:
                segs.append(cur)
            cur, inside = [], not inside
            continue
        if inside:
            cur.append(r)
    return segs


def main():
    tag, out = sys.argv[1], sys.argv[2]
    meta = json.load(open(os.path.join(ROOT, "gpurun_out", "layers_%s.json" % tag)))
    res = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        f = max(glob.glob(os.path.join(ROOT, "gpurun_out", "pmc_layers_%s_%s" % (tag, c), "**", "*counter_collection.csv"),
                          recursive=True), key=os.path.getmtime)
        res[c] = segments(f, c)
    n = len(meta["segments"])
    assert len(res["FETCH_SIZE"]) == n and len(res["WRITE_SIZE"]) == n, (n, len(res["FETCH_SIZE"]), len(res["WRITE_SIZE"]))
    rows = []
    for i, seg in enumerate(meta["segments"]):
        iters = 3
        fetch = sum(float(r["Counter_Value"]) for r in res["FETCH_SIZE"][i]) / iters * 1024.0
        write = sum(float(r["Counter_Value"]) for r in res["WRITE_SIZE"][i]) / iters * 1024.0
        def kname(k):   # (names rocprofv3 left mangled: the function name)
            m = re.match(r"_Z(\d+)", k)
            return k[m.end():m.end() + int(m.group(1))] if m else k.split("(")[0][:80]
        kernels = sorted(set(kname(r["Kernel_Name"]) for r in res["FETCH_SIZE"][i]))
        hbm = 2.0 * fetch + write
        rows.append(dict(layer=seg["layer"], what=seg["what"], ms=round(seg["ms"], 4),
                         algorithmic_bytes=seg["algorithmic_bytes"], stored_bytes=seg["stored_bytes"],
                         hbm_bytes_per_launch=hbm, fetch_bytes_x2=2.0 * fetch, write_bytes=write,
                         traffic_over_algorithmic=round(hbm / seg["algorithmic_bytes"], 3),
                         achieved_GBps_algorithmic=round(seg["algorithmic_bytes"] / seg["ms"] / 1e6, 1),
                         frac_of_8TBps=round(seg["algorithmic_bytes"] / seg["ms"] / 1e6 / 8000.0, 4),
                         TFLOPs=round(seg["flops"] / seg["ms"] / 1e9, 1), kernels=kernels))
    js = dict(precision=meta["precision"], batch="as in tools/microbench_conv.py SHAPES (N=32, 256x256 generators)",
              correction="gfx950 (MI355X_MICROARCH.md, HBM): FETCH_SIZE x2 for wide coalesced streaming reads; WRITE_SIZE exact; both "
                         "in KB; memory-side requests, Infinity-Cache hits included.  ms = HIP events over 10 un-profiled launches",
              command="tools/profile_layer_traffic.sh <layers> %s" % tag, layers=rows)
    json.dump(js, open(out, "w"), indent=1)
    for r in rows:
        print("%-20s %-5s %7.3f ms  algo %7.1f MB  hbm %7.1f MB (%.2fx)  %6.1f GB/s algorithmic = %.3f of 8 TB/s  %s"
              % (r["layer"], r["what"], r["ms"], r["algorithmic_bytes"] / 1e6, r["hbm_bytes_per_launch"] / 1e6,
                 r["traffic_over_algorithmic"], r["achieved_GBps_algorithmic"], r["frac_of_8TBps"], ",".join(r["kernels"])[:100]))


if __name__ == "__main__":
    main()
