cd $GRAFT_REPO_ROOT
for v in "" r_nostage r_noa r_nobar r_noepi r_all; do
  if [ -z "$v" ]; then echo "== base"; python3 tools/perf_wino.py 20 9 2>/dev/null | grep two-axis
  else echo "== $v"; HPVG_LIB=$GRAFT_REPO_ROOT/hp-vae-gan_amd/build/libhpvg_$v.so python3 tools/perf_wino.py 20 9 2>/dev/null | grep two-axis; fi
done
