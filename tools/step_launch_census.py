#!/usr/bin/env python3
"""rocprofv3 kernel_trace.csv of `bench.py --steps 2 --warmup 1` (tools/profile_stats.sh) -> launches and small-kernel time of
ONE steady-state training step: the launches between the last two optimiser steps of the trace (adam_multi_kernel runs four
times per step; model construction, weight initialisation and the first step's lazy packing are left out, unlike the
whole-trace averages of kernel_stats_md.py).
    python tools/step_launch_census.py <kernel_trace.csv> <out.md>"""
import collections
import csv
import re
import sys


def short(name):
    m = re.match(r"_Z(\d+)", name)
    if m:
        name = name[m.end():m.end() + int(m.group(1))]
    return re.sub(r"\(.*", "", name)[:90]


def main():
    src, out = sys.argv[1], sys.argv[2]
    rows = sorted(csv.DictReader(open(src)), key=lambda r: int(r["Start_Timestamp"]))
    adam = [i for i, r in enumerate(rows) if "adam_multi_kernel" in r["Kernel_Name"]]
    assert len(adam) >= 8 and len(adam) % 4 == 0, "expected four adam_multi_kernel launches per step, found %d" % len(adam)
    seg = rows[adam[-5] + 1:adam[-1] + 1]          # behind the previous step's last Adam launch .. this step's last one
    cnt, tim = collections.Counter(), collections.Counter()
    for r in seg:
        n = short(r["Kernel_Name"])
        cnt[n] += 1
        tim[n] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    total = sum(tim.values())
    small = [n for n in cnt if tim[n] / cnt[n] < 20.0]
    with open(out, "w") as f:
        f.write("# one steady-state training step of %s (tools/step_launch_census.py)\n\n" % src)
        f.write("launches: **%d**; kernel time %.2f ms; kernels averaging < 20 us: **%d launches, %.2f ms**\n\n"
                % (len(seg), total / 1e3, sum(cnt[n] for n in small), sum(tim[n] for n in small) / 1e3))
        f.write("| launches | total us | avg us | kernel (averaging < 20 us) |\n|---|---|---|---|\n")
        for n in sorted(small, key=lambda n: -tim[n]):
            f.write("| %d | %.1f | %.1f | `%s` |\n" % (cnt[n], tim[n], tim[n] / cnt[n], n))
    print(open(out).read().split("\n\n")[1])


if __name__ == "__main__":
    main()
