"""Soak run: N iterations of a few stages on the synthetic bench inputs, graph-replayed or eager; reports the first
iteration at which a loss or the clip norm stops being finite (none expected) and the loss trajectory.
usage: python tools/soak.py [graph|eager] [iters] [stages...]"""
import math
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

bench.CONFIG = os.environ.get("HPVG_SOAK_CONFIG", "video")   # "image": the 2-D path (BASELINE configs[1])
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 300
stages = [int(a) for a in sys.argv[3:]] or [2, 4, 6]
built, shapes = bench.build_gpu_stages(torch.device("cuda", 0), stages)
for s, tr, _step, real, rz in built:
    tr.step(real, rz)
    tr.step(real, rz)
    if mode == "graph":
        tr.enable_graph(real, rz)
    hist, bad = [], None
    for i in range(iters):
        out = tr.step(real, rz)
        if i % 10 == 9:
            torch.cuda.synchronize()
            sc = {k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1}
            sc["clip_norm"] = float(out["clip_info"][1])
            if bad is None and not all(math.isfinite(v) for v in sc.values()):
                bad = (i, sc)
            if i % 50 == 49:
                hist.append({k: round(v, 4) for k, v in sc.items()})
    print("stage", s, mode, "iters", tr.iteration, "first non-finite:", bad)
    for h in hist:
        print("   ", h)
