"""Development tool: cProfile of the host side of one pyramid stage's train iterations."""
import cProfile, pstats, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
bench.CONFIG = sys.argv[1] if len(sys.argv) > 1 else "video"
stage = int(sys.argv[2]) if len(sys.argv) > 2 else 3
built, shapes = bench.build_gpu_stages(torch.device("cuda", 0), [stage])
s, tr, _step, real, rz = built[0]
for _ in range(3):
    tr.step(real, rz)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(5):
    tr.step(real, rz)
torch.cuda.synchronize()
print("stage", stage, "ms/iter", (time.perf_counter() - t0) / 5 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.step(real, rz)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
