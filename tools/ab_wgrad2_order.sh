#!/bin/bash
# tile walk of the two-axis weight gradient (development): time at stages 8 / 9, B = 2 / 4, and HBM traffic at stage 9
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/abw2o}; mkdir -p $O
for ord in 0 1; do for B in 2 4; do
  echo "== order $ord B=$B"; HPVG_WG2_ORDER=$ord HPVG_PERF_B=$B python3 tools/perf_wgrad_wino.py 30 8 9 2>/dev/null | grep stage | sed 's/direct.*two-axis/two-axis/' | cut -c1-110
done; done
for ord in 0 1; do
  HPVG_WG2_ORDER=$ord rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f$ord -o f -- python3 tools/perf_wgrad_wino.py 3 9 > /dev/null 2>&1
  HPVG_WG2_ORDER=$ord rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w$ord -o w -- python3 tools/perf_wgrad_wino.py 3 9 > /dev/null 2>&1
  python3 tools/pmc_summary.py $(find $O/f$ord -name "*counter_collection.csv") $(find $O/w$ord -name "*counter_collection.csv") > $O/traffic_order$ord.csv 2>$O/pmc_err$ord.txt
  echo "== traffic order $ord"; grep -i "wgradw2" $O/traffic_order$ord.csv
done
find $O -name "*counter_collection.csv" -delete
