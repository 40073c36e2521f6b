#!/bin/bash
# landscape of the two-axis weight-gradient kernel (development): tiles, tile order, ablations, HBM traffic.
# usage (GPU box): bash tools/ab_wgrad2.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/abw2}; mkdir -p $O; out=$O/ab.txt; : > $out
run() { python3 tools/perf_wgrad_wino.py 20 8 9 2>/dev/null | grep stage | sed 's/.*two-axis/two-axis/' ; }
echo "== planner's pick, order 0" >> $out; run >> $out
echo "== order 1" >> $out; HPVG_WG2_ORDER=1 run >> $out
for f in 6,16 4,24 4,16 8,12 2,40 2,32 4,20 6,8 4,32; do
  echo "== HPVG_WG2_FORCE=$f" >> $out
  HPVG_WG2_FORCE=$f run >> $out
done
for v in w2nodma w2nomma; do
  if [ -f hp-vae-gan_amd/build/libhpvg_$v.so ]; then
    echo "== ablation $v (timing only)" >> $out
    HPVG_LIB=$GRAFT_REPO_ROOT/hp-vae-gan_amd/build/libhpvg_$v.so run >> $out
  fi
done
cat $out
for ord in 0 1; do
  HPVG_WG2_ORDER=$ord rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$ord -o f -- python3 tools/perf_wgrad_wino.py 3 9 > /dev/null 2>&1
  HPVG_WG2_ORDER=$ord rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w$ord -o w -- python3 tools/perf_wgrad_wino.py 3 9 > /dev/null 2>&1
  python3 tools/pmc_summary.py $(find $O/f$ord -name "*counter_collection.csv") $(find $O/w$ord -name "*counter_collection.csv") > $O/traffic_order$ord.csv 2>$O/pmc_err$ord.txt
  echo "== traffic order $ord"; grep -i "wgrad" $O/traffic_order$ord.csv
done
find $O -name "*counter_collection.csv" -delete
