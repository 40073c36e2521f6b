"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (separate passes) into per-kernel averages.
usage: python tools/pmc_summary.py <fetch_csv> <write_csv> > profiles/xxx.csv
gfx950 correction (MI355X_MICROARCH.md, HBM section; calibrated here on channel_sum_partial_kernel, which reads each
input byte exactly once): FETCH_SIZE counts 128-byte requests as 64 -> fetch_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE is exact."""
import csv, sys, collections, re


def load(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        name = name.split("(")[0][-60:] if name.startswith("void at::") else name.split("(")[0]
        key = (name, r["Grid_Size"])
        acc[key][0] += float(r["Counter_Value"])
        acc[key][1] += 1
    return {k: v[0] / v[1] for k, v in acc.items()}


json_out = None
if "--roofline-json" in sys.argv:
    i = sys.argv.index("--roofline-json")
    json_out = sys.argv[i + 1]          # then: <csv path to cite> <B> <C> <T> <H> <W>
    cite, shape = sys.argv[i + 2], [int(v) for v in sys.argv[i + 3:i + 8]]
    del sys.argv[i:i + 8]
f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
print("kernel,grid,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_corrected")
for k in sorted(set(f) | set(w)):
    fk, wk = f.get(k, 0.0), w.get(k, 0.0)
    print("%s,%s,%.4g,%.4g,%.4g" % (k[0].replace(",", ";"), k[1], fk, wk, 2 * fk * 1024 + wk * 1024))

if json_out:
    # the roofline kernel of bench.py: the 64->64 3x3x3 conv instance with the largest fetch (the full-resolution launch of
    # tools/perf_conv.py at that shape) + the largest stream-K fix-up launch that follows it
    import json
    rows = {k: 2 * f.get(k, 0.0) * 1024 + w.get(k, 0.0) * 1024 for k in set(f) | set(w)}
    # (the Winograd kernel when the library runs the shape on it - conv_wino_kernel<3, 2, 0> - else the direct one)
    # (the kernel the library runs the shape on: two-axis Winograd conv_wino2d_kernel<0> - whole tiles, no fix-up launch -,
    # else one-axis conv_wino_kernel<3, 2, 0, *>, else the direct one)
    w2 = [k for k in rows if "conv_wino2d_kernel<0" in k[0]]
    wino = [k for k in rows if "conv_wino_kernel<" in k[0] and ", 0, " in k[0]]
    if w2:
        main = max(w2, key=lambda k: rows[k])
        ent = {"shape": shape, "bytes": round(rows[main]), "source": cite, "rows": {"%s grid %s" % main: round(rows[main])}}
    else:
        main = max(wino or [k for k in rows if "conv_mfma_kernel<8" in k[0] and k[0].rstrip(">").endswith(" 0")], key=lambda k: rows[k])
        fix = max((k for k in rows if ("conv_wino_fixup_kernel" if wino else "conv_fixup_kernel") in k[0]), key=lambda k: rows[k])
        ent = {"shape": shape, "bytes": round(rows[main] + rows[fix]), "source": cite,
               "rows": {"%s grid %s" % main: round(rows[main]), "%s grid %s" % fix: round(rows[fix])}}
    json.dump({"note": "HBM bytes per launch of bench.py's roofline kernel from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes "
                       "(gfx950: FETCH_SIZE x 2; tools/pmc_summary.py); bench.py reports `traffic` from here", "entries": [ent]},
              open(json_out, "w"), indent=1)
