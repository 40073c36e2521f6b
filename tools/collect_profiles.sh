#!/bin/bash
# copy what tools/profile_r03.sh left under gpurun_out/final_r03 into profiles/ (tracked) under r03_final_* names
S=${1:-gpurun_out/final_r03}; P=profiles
for c in video image video8 baseline; do grep '^{' $S/bench_$c.log > $P/r03_final_bench_line_$c.json; done
mv $P/r03_final_bench_line_video.json $P/r03_final_bench_line.json
cp $S/bench_kernel_stats.csv $P/r03_final_bench_kernel_stats.csv
cat $S/roofline_trace_conv.txt $S/roofline_trace_wgrad.txt > $P/r03_final_roofline_kernel_trace.txt
for st in 9 8 7 5; do cp $S/stages/stage${st}_kernel_stats.csv $P/r03_final_stage${st}_kernel_stats.csv; done
cp $S/pmc_traffic_conv_stage9.csv $P/r03_final_pmc_traffic_conv_stage9.csv
cp $S/pmc_traffic_wgrad_stage9.csv $P/r03_final_pmc_traffic_wgrad_stage9.csv
cp $S/pmc_busy_conv_stage9.csv $P/r03_final_pmc_busy_conv_stage9.csv
cp $S/pmc_busy_wgrad_stage9.csv $P/r03_final_pmc_busy_wgrad_stage9.csv
cp $S/perf_wino2r.txt $P/r03_perf_wino2r.txt
cp $S/mfma_fillers.txt $P/r03_mfma_fillers.txt
cp $S/stages.txt $P/r03_final_stage_breakdown.txt
python3 - <<PY
import json, csv
rec = json.load(open("$S/roofline_traffic.json"))
# the weight-gradient family = main kernel + its reduce kernel (rows of the csv)
rows = {r["kernel"]: float(r["hbm_bytes_corrected"]) for r in csv.DictReader(open("$S/pmc_traffic_wgrad_stage9.csv"))}
for e in rec["entries"]:
    if e.get("family") == "weight_gradient":
        main = [k for k in rows if "conv_wgradw2_kernel" in k][0]
        e["rows"] = {main + " grid 65024": round(rows[main]), "conv_wgradw2_reduce_kernel": round(rows["conv_wgradw2_reduce_kernel"])}
        e["bytes"] = round(rows[main] + rows["conv_wgradw2_reduce_kernel"])
json.dump(rec, open("$P/roofline_traffic.json", "w"), indent=1)
PY
ls $P | grep r03
