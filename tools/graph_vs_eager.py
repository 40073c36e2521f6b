"""Development: iterations/s of one stage eager against hipGraph replay (the bench keeps its top stage eager so that the
roofline kernel's launches can be bracketed by events; this prices that choice).
usage: python tools/graph_vs_eager.py [stage] [iters]      (HPVG_SOAK_CONFIG=image for the 2-D path)"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

bench.CONFIG = os.environ.get("HPVG_SOAK_CONFIG", "video")
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 9
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
built, _ = bench.build_gpu_stages(torch.device("cuda", 0), [stage])
s, tr, _step, real, rz = built[0]


def rate(tag):
    tr.step(real, rz)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tr.step(real, rz)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    print("%s stage %d %-6s %8.3f ms/iteration  %8.3f it/s" % (bench.CONFIG, s, tag, dt * 1e3, 1.0 / dt))


tr.step(real, rz)
rate("eager")
tr.enable_graph(real, rz)
rate("graph")
