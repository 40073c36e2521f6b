"""Per-kernel averages of one rocprofv3 --pmc counter_collection CSV (any counters) plus the MFMA-busy fraction when
SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE are present (busy summed over the 1024 SIMDs, GRBM over the 8 XCDs).
usage: python tools/pmc_busy.py <counter_collection.csv>"""
import collections
import csv
import re
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0]
    if name.startswith("void at::"):
        continue
    v = acc[(name, r["Grid_Size"])][r["Counter_Name"]]
    v[0] += float(r["Counter_Value"])
    v[1] += 1
names = sorted({c for d in acc.values() for c in d})
print("kernel,grid," + ",".join(names) + ",mfma_busy_fraction")
for k, d in sorted(acc.items()):
    vals = {c: d[c][0] / d[c][1] for c in d}
    frac = ""
    if "SQ_VALU_MFMA_BUSY_CYCLES" in vals and vals.get("GRBM_GUI_ACTIVE"):
        frac = "%.3f" % (vals["SQ_VALU_MFMA_BUSY_CYCLES"] / (vals["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0))
    print("%s,%s,%s,%s" % (k[0].replace(",", ";"), k[1], ",".join("%.4g" % vals.get(c, 0.0) for c in names), frac))
