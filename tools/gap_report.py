"""GPU idle time between kernels from a `rocprofv3 --kernel-trace` CSV (…_kernel_trace.csv).

usage: python tools/gap_report.py <kernel_trace.csv> [skip_fraction]

Takes the kernels of the trailing (1 - skip_fraction) of the trace (default 0.5: steady-state iterations), and prints
the wall span, the union of kernel busy time, the idle share, a histogram of the gaps and the largest gaps with the
kernels on either side - the first thing to look at when an iteration takes longer than the sum of its kernels."""
import csv
import sys


def main():
    path = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    rows = rows[int(len(rows) * skip):]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    busy, cur_end, gaps = 0, rows[0][0], []
    prev = None
    for s, e, n in rows:
        if s > cur_end:
            gaps.append((s - cur_end, prev, n))
            busy += e - s
            cur_end = e
        else:
            if e > cur_end:
                busy += e - cur_end
                cur_end = e
        prev = n
    span = t1 - t0
    print("kernels %d  span %.2f ms  busy %.2f ms  idle %.2f ms (%.1f %%)" % (len(rows), span / 1e6, busy / 1e6, (span - busy) / 1e6,
                                                                             100.0 * (span - busy) / span))
    edges = [2e3, 5e3, 1e4, 2e4, 5e4, 1e5, 1e6, 1e12]
    lo = 0
    for hi in edges:
        sel = [g[0] for g in gaps if lo <= g[0] < hi]
        print("  gaps %8.0f - %-8.0f ns: %6d  total %8.2f ms" % (lo, hi, len(sel), sum(sel) / 1e6))
        lo = hi
    short = lambda n: n.replace("(anonymous namespace)::", "")[:60]
    print("largest gaps:")
    for g, a, b in sorted(gaps, reverse=True)[:25]:
        print("  %9.1f us  after %-60s before %s" % (g / 1e3, short(a or ""), short(b)))
    # which kernel precedes most of the idle time
    by = {}
    for g, a, b in gaps:
        k = short(a or "")
        by[k] = by.get(k, 0) + g
    print("idle time by preceding kernel:")
    for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:15]:
        print("  %8.2f ms  %s" % (v / 1e6, k))


if __name__ == "__main__":
    main()
