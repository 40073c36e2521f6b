#!/bin/bash
# landscape of the 16-byte weight-gradient form's tiles (development): usage tools/ab_wgrad16.sh <outfile>
out=$1; : > $out
echo "== planner's pick" >> $out
python tools/perf_wgrad_wino.py 20 8 9 2>/dev/null | grep stage | cut -c1-150 >> $out
for f in 6,16 4,24 3,32 2,44 2,48 1,64 1,68 2,36 3,28 1,52 1,88 1,104; do
  echo "== HPVG_WG16_FORCE=$f" >> $out
  HPVG_WG16_FORCE=$f python tools/perf_wgrad_wino.py 20 8 9 2>/dev/null | grep stage | cut -c1-150 >> $out
done
