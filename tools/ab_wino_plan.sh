#!/bin/bash
# Landscape of the Winograd conv's tile plans: every band count forced through HPVG_PLAN_NTW, stages 3-9 (one process per
# band count: the knob is read once).  usage (on the GPU box): tools/ab_wino_plan.sh <outfile> [max_ntw]
out=$1; max=${2:-8}
: > $out
echo "== planner's own pick" >> $out
python tools/perf_wino.py 20 3 4 5 6 7 8 9 2>/dev/null | grep stage >> $out
for n in $(seq 1 $max); do
  echo "== HPVG_PLAN_NTW=$n" >> $out
  HPVG_PLAN_NTW=$n python tools/perf_wino.py 20 3 4 5 6 7 8 9 2>/dev/null | grep stage >> $out
done
