#!/usr/bin/env python3
"""GPU idle time inside the bench step, from a rocprofv3 --kernel-trace csv: the union of the kernel intervals against
the span from the first to the last kernel of the timed steps.  Tells whether the host keeps the queue fed.

    python tools/trace_gaps.py <kernel_trace.csv> [skip_fraction]   (skip the leading fraction of the trace = warm-up)"""
import csv
import sys


def main():
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
    rows.sort()
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    t0, t1 = rows[0][0], rows[-1][1]
    cut = t0 + (t1 - t0) * skip
    rows = [r for r in rows if r[0] >= cut]
    span = rows[-1][1] - rows[0][0]
    busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
    gaps = []
    for s, e, k in rows[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, k))
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print("kernels %d  span %.2f ms  busy %.2f ms  idle %.2f ms (%.1f %%)" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6,
                                                                          100.0 * (span - busy) / span))
    gaps.sort(reverse=True)
    print("gaps > 20 us: %d (%.2f ms); > 5 us: %d (%.2f ms)" % (sum(g > 20000 for g, _ in gaps), sum(g for g, _ in gaps if g > 20000) / 1e6,
                                                             sum(g > 5000 for g, _ in gaps), sum(g for g, _ in gaps if g > 5000) / 1e6))
    for g, k in gaps[:12]:
        print("  %8.1f us before %s" % (g / 1e3, k[:90]))


if __name__ == "__main__":
    main()
