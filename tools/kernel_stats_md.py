#!/usr/bin/env python3
"""rocprofv3 kernel_stats.csv (tools/profile_stats.sh: 1 warm-up + 2 timed bench steps) -> the per-step table kept in
profiles/.    python tools/kernel_stats_md.py <kernel_stats.csv> <out.md> [steps_in_trace=3] [precision]"""
import csv
import re
import sys


def short(name):
    m = re.match(r"_Z(\d+)", name)            # a name rocprofv3 left mangled (its demangler stops at __bf16 parameters of plain functions)
    if m:
        name = name[m.end():m.end() + int(m.group(1))]
    name = re.sub(r"\(.*", "", name)          # drop the argument list
    return name[:95]


def main():
    src, out = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    prec = sys.argv[4] if len(sys.argv) > 4 else "bf16x3"
    rows = list(csv.DictReader(open(src)))
    # (bench.py's sustained-MFMA-rate probe runs behind the timed region: three 50 ms launches that are not part of a step)
    probe = sum(float(r["TotalDurationNs"]) for r in rows if "mfma_rate_probe_kernel" in r["Name"])
    rows = [r for r in rows if "mfma_rate_probe_kernel" not in r["Name"]]
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline "
                "--precision %s   (tools/profile_stats.sh, tools/kernel_stats_md.py)\n\n" % prec)
        f.write("%d steps (1 warm-up + %d timed) in the trace: total %.1f ms = %.1f ms per step\n\n" % (steps, steps - 1, total / 1e6, total / 1e6 / steps))
        if probe:
            f.write("(not counted: %.1f ms of `mfma_rate_probe_kernel`, bench.py's live sustained-rate probe behind the timed region)\n\n" % (probe / 1e6))
        nl = sum(int(r["Calls"]) for r in rows) / float(steps)
        small = [r for r in rows if float(r["AverageNs"]) < 20e3]
        f.write("launches per step: %.0f; kernels averaging < 20 us: %.0f launches, %.2f ms per step\n\n"
                % (nl, sum(int(r["Calls"]) for r in small) / float(steps), sum(float(r["TotalDurationNs"]) for r in small) / 1e6 / steps))
        f.write("| ms/step | % | calls/step | avg us | kernel |\n|---|---|---|---|---|\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            t = float(r["TotalDurationNs"])
            if t / total < 0.001:
                continue
            f.write("| %.2f | %.1f | %d | %.1f | `%s` |\n" % (t / 1e6 / steps, 100 * t / total, round(int(r["Calls"]) / steps), float(r["AverageNs"]) / 1e3, short(r["Name"])))


if __name__ == "__main__":
    main()
