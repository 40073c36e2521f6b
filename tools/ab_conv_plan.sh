#!/bin/bash
# Development: the measured landscape of the conv tile planner (bands x NB) against its own choice.
# usage: tools/ab_conv_plan.sh <outdir> <libhpvg.so> <stage> ...      (HPVG_PERF_B = batch size, default 2)
out=$1; lib=$2; shift 2; mkdir -p $out
f=$out/ab_conv_plan_b${HPVG_PERF_B:-2}.txt
for st in "$@"; do
  echo "== stage $st planner"; HPVG_LIB=$lib python tools/perf_conv.py $st 10 2>/dev/null | grep "conv_fwd 64->64  \|conv_bwd_data"
  for ntw in 1 2 3 4 5 6 8; do for nb in 1 2 4; do
    echo "== stage $st ntw $ntw NB $nb"; HPVG_LIB=$lib HPVG_PLAN_NTW=$ntw HPVG_PLAN_NB=$nb python tools/perf_conv.py $st 10 2>/dev/null | grep "conv_fwd 64->64  "
  done; done
done > $f 2>&1
