#!/bin/bash
# where the two-axis weight gradient wins (development): W % 4 == 0 shapes of the configs' pyramids, B = 2 and 4, 3-D and 2-D
cd $GRAFT_REPO_ROOT
export HPVG_PERF_SHAPES="4,27,48;5,40,72;4,18,32;7,114,204;13,144,256;5,55,100;7,60,108;13,72,128"
for B in 2 4; do
  echo "== 3-D B=$B"; HPVG_PERF_B=$B python3 tools/perf_wgrad_wino.py 20 100 101 102 103 104 105 106 107 2>/dev/null | grep stage | sed 's/direct.*| wino \([0-9.]*\) ms.*two-axis/one-axis \1 ms | two-axis/' | cut -c1-140
done
export HPVG_PERF_SHAPES="1,192,256;1,153,204;1,36,48;1,96,128;1,75,100;1,144,256;1,27,48"
for B in 2 4; do
  echo "== 2-D B=$B"; HPVG_PERF_DIMS=2 HPVG_PERF_B=$B python3 tools/perf_wgrad_wino.py 20 100 101 102 103 104 105 106 2>/dev/null | grep stage | sed 's/direct.*| wino \([0-9.]*\) ms.*two-axis/one-axis \1 ms | two-axis/' | cut -c1-140
done
