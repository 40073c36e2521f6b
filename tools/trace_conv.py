"""Per-workgroup timeline of one conv launch (development tool).  Builds a -DHPVG_TRACE copy of the library into
gpurun_out/, launches the conv at a pyramid stage and prints where workgroups ran and for how long.
usage: python tools/trace_conv.py [stage]   (env HPVG_CONV_SK=0: one workgroup per tile; HPVG_PLAN_NB/MB: restrict the planner)"""
import sys, os, ctypes, subprocess, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join("/tmp", "libhpvg_trace%s.so" % os.environ.get("HPVG_TRACE_DEFS", "").replace(" ", "").replace("-D", "_"))
os.makedirs(os.path.dirname(out), exist_ok=True)
csrc = os.path.join(ROOT, "hp-vae-gan_amd", "csrc")
if not os.path.exists(out):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-DHPVG_TRACE", "-shared"] + os.environ.get("HPVG_TRACE_DEFS", "").split() + [
                           "-I", os.path.join(ROOT, "include"), "-o", out] + sorted(__import__("glob").glob(os.path.join(csrc, "*.hip"))))
os.environ["HPVG_LIB"] = out
import torch
import hp_vae_gan_amd
from hp_vae_gan_amd import lib as hplib, ops
lib = hplib.load()
SHAPES = {0: (4, 18, 33), 3: (5, 36, 65), 5: (5, 57, 102), 6: (7, 72, 129), 7: (7, 91, 162), 8: (7, 114, 204), 9: (13, 144, 256)}
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 6
T, H, W = SHAPES[stage]
x = torch.randn(2, 64, T, H, W, device="cuda")
w = torch.randn(64, 64, 3, 3, 3, device="cuda") * 0.05
for _ in range(3):
    ops.conv_fwd_raw(x, w, None)
torch.cuda.synchronize()
N = 4096
buf = (ctypes.c_ulonglong * (4 * N))()
lib.hpvg_debug_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert lib.hpvg_debug_trace_read(buf, N) == 0
rows = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3], i) for i in range(N) if buf[4 * i + 1]]
t0 = min(r[0] for r in rows)
t1 = max(r[1] for r in rows)
print("stage", stage, "workgroups", len(rows), "span %.1f us" % ((t1 - t0) / 100.0))
durs = sorted((r[1] - r[0]) / 100.0 for r in rows)
print("duration us: min %.1f  p50 %.1f  p90 %.1f  max %.1f" % (durs[0], durs[len(durs) // 2], durs[int(len(durs) * 0.9)], durs[-1]))
starts = sorted((r[0] - t0) / 100.0 for r in rows)
clk = sorted(((r[3] >> 8) / max(1, (r[1] - r[0]))) * 100.0 / 1e3 for r in rows)
print("shader clock GHz (cycle counter / wall): min %.3f p50 %.3f max %.3f" % (clk[0], clk[len(clk) // 2], clk[-1]))
print("start us: p50 %.1f  p90 %.1f  max %.1f" % (starts[len(starts) // 2], starts[int(len(starts) * 0.9)], starts[-1]))
place = collections.Counter()
members = collections.defaultdict(list)
for r in rows:
    hw = r[2]
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 0x7
    xcc = r[3] & 0xF
    place[(xcc, se, sh, cu)] += 1
    members[(xcc, se, sh, cu)].append((r[4], hw & 0xF, (hw >> 4) & 3, round((r[1] - t0) / 100.0, 1)))
hist = collections.Counter(place.values())
print("distinct CUs used:", len(place), " workgroups-per-CU histogram:", dict(sorted(hist.items())))
# concurrency profile: how many WGs are alive at 10 sample points
for f in (0.1, 0.3, 0.5, 0.7, 0.9, 0.97):
    t = t0 + f * (t1 - t0)
    print("  alive at %3d%% of span: %d" % (int(f * 100), sum(1 for r in rows if r[0] <= t < r[1])))
print("co-resident workgroups (block id, wave slot, simd, end us) on the first CUs:")
for k in sorted(members)[:6]:
    print("  ", k, members[k])
pbuf = (ctypes.c_ulonglong * (4 * N))()
lib.hpvg_debug_phase_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
if lib.hpvg_debug_phase_read(pbuf, N) == 0 and any(pbuf[i] for i in range(64)):
    import statistics
    for slot in (0, 1):
        ids = [r[4] for r in rows if (r[2] & 1) == slot]
        ph = [[pbuf[4 * i + k] / 2050.0 for i in ids] for k in range(4)]
        print("slot %d (n=%d): median us in [DMA issue, DMA wait+barrier, MFMA loop, chunk-top barrier + epilogue + setup] = %s"
              % (slot, len(ids), ["%.0f" % statistics.median(p) for p in ph]))
