#!/bin/bash
# kernel stats of one eager iteration set per stage (development): usage bash tools/profile_stages.sh <outdir> <stages...>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; shift; mkdir -p $O
for st in "$@"; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$st -o s$st -- python3 bench.py --stages $st --steps 3 --warmup 1 --no-cpu-baseline --graph-stages none > $O/s$st.log 2>&1
  f=$(find $O/s$st -name "*kernel_stats.csv" | head -1)
  cp $f $O/stage${st}_kernel_stats.csv
  rm -rf $O/s$st
  echo "== stage $st"; python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/stage${st}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("total %.2f ms over 4 iterations" % (tot/1e6))
for r in sorted(rows,key=lambda r:-float(r["TotalDurationNs"]))[:28]:
    print("%5.1f%%  calls %5s avg %8.1f us  %s" % (100*float(r["TotalDurationNs"])/tot, r["Calls"], float(r["AverageNs"])/1e3, r["Name"].replace("(anonymous namespace)::","")[:90]))
PY
done
