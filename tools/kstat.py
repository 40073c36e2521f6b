"""print (calls, avg us, total ms) of the kernels whose name contains any of the given substrings, from a rocprofv3 *_kernel_stats.csv
usage: python tools/kstat.py <kernel_stats.csv> <substr> [...]"""
import csv
import sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in sys.argv[2:]):
        print("%-80s calls %6s  avg %9.1f us  total %9.2f ms" % (r["Name"].replace("(anonymous namespace)::", "")[:80], r["Calls"],
                                                                float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
