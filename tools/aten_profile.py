"""Development tool: which torch (ATen) device kernels a train iteration still launches besides libhpvg's.
usage: python tools/aten_profile.py [video|image] [stage]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity
bench.CONFIG = sys.argv[1] if len(sys.argv) > 1 else "video"
stage = int(sys.argv[2]) if len(sys.argv) > 2 else 3
built, shapes = bench.build_gpu_stages(torch.device("cuda", 0), [stage])
s, tr, _step, real, rz = built[0]
for _ in range(3):
    tr.step(real, rz)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU]) as prof:
    tr.step(real, rz)
    torch.cuda.synchronize()
rows = [(e.key, e.count, e.self_cpu_time_total) for e in prof.key_averages() if e.key.startswith("aten::")]
rows.sort(key=lambda r: -r[1])
print("stage", stage, "aten ops in one iteration (name, calls, self cpu us):")
for r in rows[:30]:
    print("  %-40s %5d %8.0f" % r)
