cd $GRAFT_REPO_ROOT
for v in "" w2nodma w2noxf w2nolds w2nosetup w2skel; do
  if [ -z "$v" ]; then echo "== base"; python3 tools/perf_wgrad_wino.py 20 9 2>/dev/null | grep stage | sed 's/.*two-axis/two-axis/'
  else echo "== $v"; HPVG_LIB=$GRAFT_REPO_ROOT/hp-vae-gan_amd/build/libhpvg_$v.so python3 tools/perf_wgrad_wino.py 20 9 2>/dev/null | grep stage | sed 's/.*two-axis/two-axis/'; fi
done
