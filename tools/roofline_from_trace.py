"""Per-shape view of the roofline kernel in a rocprofv3 kernel trace: the template instance conv_mfma_kernel<8,3,2,4> serves
several layer shapes, so the `--stats` average mixes them; this groups its launches by duration cluster and prints count /
mean per cluster, plus the paired conv_fixup_kernel launches that follow them.
usage: python tools/roofline_from_trace.py <kernel_trace.csv> [kernel substring]"""
import csv
import sys

path = sys.argv[1]
key = sys.argv[2] if len(sys.argv) > 2 else "conv_mfma_kernel<8, 3, 2, 4>"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
main, fix = [], []
for i, (s, e, n) in enumerate(rows):
    if key in n:
        main.append((e - s) / 1e3)
        if i + 1 < len(rows) and "conv_fixup_kernel" in rows[i + 1][2]:
            fix.append((e - s) / 1e3 + (rows[i + 1][1] - rows[i + 1][0]) / 1e3)
        else:
            fix.append((e - s) / 1e3)
print("kernel:", key, " launches:", len(main))
# (direct kernel: 1.5-2.5 ms = the B = 2 finest-level launches, the roofline shape; above: B = 4, the merged generator pass.
#  two-axis Winograd kernel, key "conv_wino2d_kernel<0": 0.9-1.3 ms = B = 2 at 13 x 144 x 256, ~2.1 ms = B = 4; below 0.9: the
#  level-8 volume inside a stage-9 iteration)
edges = [0, 100, 300, 600, 900, 1300, 1500, 2500, 1e9]
for lo, hi in zip(edges[:-1], edges[1:]):
    sel = [(m, f) for m, f in zip(main, fix) if lo <= m < hi]
    if sel:
        print("  %6.0f - %-8.0f us: %4d launches  mean %8.1f us  (+ fix-up: %8.1f us)" % (
            lo, hi if hi < 1e9 else float("inf"), len(sel), sum(m for m, _ in sel) / len(sel), sum(f for _, f in sel) / len(sel)))
