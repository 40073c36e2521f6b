"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE counter_collection CSVs (separate passes) into per-kernel averages.
usage: python tools/pmc_summary.py <fetch_csv> <write_csv> [--roofline-json <json> <family> <csv to cite> B C T H W] > profiles/xxx.csv
(--roofline-json merges the per-launch HBM bytes of bench.py's roofline family - conv_fwd or weight_gradient - at that shape into
profiles/roofline_traffic.json, replacing an older entry of the same family and shape)
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
    json_out, family, cite = sys.argv[i + 1], sys.argv[i + 2], sys.argv[i + 3]
    shape = [int(v) for v in sys.argv[i + 4:i + 9]]
    del sys.argv[i:i + 9]
f = load(sys.argv[1], "FETCH_SIZE")
w = load(sys.argv[2], "WRITE_SIZE")
print("kernel,grid,FETCH_SIZE_KB,WRITE_SIZE_KB,hbm_bytes_corrected")
for k in sorted(set(f) | set(w)):
    fk, wk = f.get(k, 0.0), w.get(k, 0.0)
    print("%s,%s,%.4g,%.4g,%.4g" % (k[0].replace(",", ";"), k[1], fk, wk, 2 * fk * 1024 + wk * 1024))

if json_out:
    import json, os
    rows = {k: 2 * f.get(k, 0.0) * 1024 + w.get(k, 0.0) * 1024 for k in set(f) | set(w)}

    def biggest(pred):
        ks = [k for k in rows if pred(k[0])]
        return max(ks, key=lambda k: rows[k]) if ks else None

    if family == "conv_fwd":
        # the kernel the library runs the shape on (largest launch of the profiled tool = the full-resolution one): two-axis
        # Winograd (plain instance; whole tiles or its own stream-K tail) - else one-axis + fix-up - else direct + fix-up
        parts = [biggest(lambda n: "conv_wino2r_kernel<0" in n or "conv_wino2d_kernel<0" in n)]
        if parts[0] is None:
            wino = biggest(lambda n: "conv_wino_kernel<" in n and ", 0, " in n)
            parts = [wino, biggest(lambda n: "conv_wino_fixup_kernel" in n)] if wino else \
                    [biggest(lambda n: "conv_mfma_kernel<8" in n and n.rstrip(">").endswith(" 0")), biggest(lambda n: "conv_fixup_kernel" in n)]
        else:
            parts.append(biggest(lambda n: "conv_wino2d_fixup_kernel" in n))
    elif family == "weight_gradient":
        main = biggest(lambda n: "conv_wgradw2_kernel" in n) or biggest(lambda n: "conv_wgradw_kernel" in n) or \
            biggest(lambda n: "conv_wgrad3_kernel" in n) or biggest(lambda n: "conv_wgrad_kernel" in n)
        red = main[0].split("<")[0].replace("void ", "").replace("_kernel", "_reduce_kernel")
        parts = [main, biggest(lambda n: red in n)]
    else:
        raise SystemExit("unknown family " + family)
    parts = [k for k in parts if k is not None]
    ent = {"family": family, "shape": shape, "bytes": round(sum(rows[k] for k in parts)), "source": cite,
           "rows": {"%s grid %s" % k: round(rows[k]) for k in parts}}
    rec = {"entries": []}
    if os.path.exists(json_out):
        rec = json.load(open(json_out))
    rec["note"] = ("HBM bytes per launch of bench.py's roofline kernel families from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                   "passes (gfx950: FETCH_SIZE x 2; tools/pmc_summary.py); bench.py reports `traffic` from here")
    rec["entries"] = [e for e in rec["entries"] if not (e.get("family", "conv_fwd") == family and e["shape"] == shape)] + [ent]
    json.dump(rec, open(json_out, "w"), indent=1)
