#!/bin/bash
# Development A/B: conv kernel instances with <= 4 accumulator tiles per wave compiled for FOUR co-resident workgroups per
# CU (-DHPVG_CONV_WGS4) against the default two, at several pyramid stages.  usage: tools/ab_conv_wgs.sh <outdir>
out=$1; mkdir -p $out
c=hp-vae-gan_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DHPVG_CONV_WGS4 -shared -I include -o /tmp/libhpvg_wgs4.so $c/conv_mfma.hip $c/conv_wgrad.hip $c/elementwise.hip $c/frames.hip $c/graph.hip $c/variants.hip || exit 1
for st in 9 7 5 3; do
  echo "== stage $st: default plan, 2 WG/CU"; python tools/perf_conv.py $st 10 2>/dev/null | grep "conv_fwd 64->64  \|conv_bwd_data"
  echo "== stage $st: NB=2 MB=2, 2 WG/CU"; HPVG_PLAN_NB=2 HPVG_PLAN_MB=2 python tools/perf_conv.py $st 10 2>/dev/null | grep "conv_fwd 64->64  \|conv_bwd_data"
  echo "== stage $st: NB=2 MB=2, 4 WG/CU"; HPVG_LIB=/tmp/libhpvg_wgs4.so HPVG_PLAN_NB=2 HPVG_PLAN_MB=2 python tools/perf_conv.py $st 10 2>/dev/null | grep "conv_fwd 64->64  \|conv_bwd_data"
  echo "== stage $st: NB=4 MB=1, 4 WG/CU"; HPVG_LIB=/tmp/libhpvg_wgs4.so HPVG_PLAN_NB=4 HPVG_PLAN_MB=1 python tools/perf_conv.py $st 10 2>/dev/null | grep "conv_fwd 64->64  \|conv_bwd_data"
  echo "== stage $st: planner's choice, 4 WG/CU build"; HPVG_LIB=/tmp/libhpvg_wgs4.so python tools/perf_conv.py $st 10 2>/dev/null | grep "conv_fwd 64->64  \|conv_bwd_data"
done > $out/ab_conv_wgs.txt 2>&1
cat $out/ab_conv_wgs.txt
