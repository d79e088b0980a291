#!/usr/bin/env python3
"""Per-(kernel, grid size) averages from a rocprofv3 kernel trace: one template instance serves several layers (e.g.
igemm_conv_x3_ws<false,false,true> = resblock data gradients on the 130x130 padded grid AND the D_B 4x4 convolutions), so
the per-kernel table of kernel_stats_md.py mixes shapes.   python tools/kernel_shapes_md.py <kernel_trace.csv> <out.md> [steps=3]"""
import collections
import csv
import re
import sys


def main():
    src, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(src)):
        name = r["Kernel_Name"]
        mz = re.match(r"_Z(\d+)", name)   # left mangled by rocprofv3: keep the function name
        if mz:
            name = name[mz.end():mz.end() + int(mz.group(1))]
        name = re.sub(r"\(.*", "", name)[:80]
        wg = int(r["Workgroup_Size_X"])
        agg[(name, int(r["Grid_Size_X"]) // wg, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
    total = sum(sum(v) for v in agg.values())
    with open(out, "w") as f:
        f.write("# per (kernel, workgroups) averages of the bench step — %s\n\n" % src)
        f.write("| ms/step | calls/step | avg us | min us | max us | workgroups x threads | kernel |\n|---|---|---|---|---|---|---|\n")
        for (name, nwg, wg), v in rows:
            if sum(v) / total < 0.002:
                continue
            f.write("| %.2f | %d | %.1f | %.1f | %.1f | %d x %d | `%s` |\n"
                    % (sum(v) / 1e3 / steps, round(len(v) / steps), sum(v) / len(v), min(v), max(v), nwg, wg, name))


if __name__ == "__main__":
    main()
