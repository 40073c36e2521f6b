"""Which kernels run right before / after the launches of a given kernel (second half of a rocprofv3 kernel trace):
finds where stray copy / fill / add kernels come from.
usage: python tools/trace_neighbours.py <kernel_trace.csv> <kernel name substring>"""
import collections
import csv
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "")))
rows.sort()
rows = rows[len(rows) // 2:]
c = collections.Counter()
for i, (s, n) in enumerate(rows):
    if sys.argv[2] in n:
        c[(rows[i - 1][1][:60] if i else "", rows[i + 1][1][:60] if i + 1 < len(rows) else "")] += 1
for k, v in c.most_common(15):
    print(v, k)
