#!/bin/bash
# A/B of two trees inside ONE gpurun call (box-to-box spread is larger than most single changes): the old tree lives in
# .ab_old/ (git archive <rev> | tar -x -C .ab_old; built there), the new one is the repo.  usage: tools/ab_bench.sh <outdir> <bench args...>
out=$1; shift
mkdir -p $out
for r in 1 2; do
  (cd .ab_old && python bench.py "$@" --no-cpu-baseline) > $out/old_$r.json 2>/dev/null
  python bench.py "$@" --no-cpu-baseline > $out/new_$r.json 2>/dev/null
done
python - $out <<'PY'
import json, sys, glob
out = sys.argv[1]
def load(p):
    l = [x for x in open(p) if x.startswith("{")]
    return json.loads(l[-1]) if l else None
for side in ("old", "new"):
    for r in (1, 2):
        d = load("%s/%s_%d.json" % (out, side, r))
        if d: print(side, r, "value %.4f" % d["value"], {k: round(v, 3) for k, v in d["per_stage_it_s"].items()}, "roof ms", d["roofline"]["avg_ms"] if d["roofline"] else None)
PY
