"""Two-axis Winograd conv (forced, mode 5) against the one-axis kernel (mode 6) on given shapes, old (HPVG_WINO2R=0) or new kernel:
usage: python tools/perf_wino2.py reps "B,T,H,W;B,T,H,W;..."  (development)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hp_vae_gan_amd
from hp_vae_gan_amd import ops, lib as hplib
lib = hplib.load()
reps = int(sys.argv[1])
def bench(fn):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
for spec in sys.argv[2].split(";"):
    B, T, H, W = (int(v) for v in spec.split(","))
    x = torch.randn(B, 64, T, H, W, device="cuda"); w = torch.randn(64, 64, 3, 3, 3, device="cuda") * 0.05
    r = {}
    for m in (6, 5):
        lib.hpvg_conv_wino_config(m, -1)
        r[m] = bench(lambda: ops.conv_fwd_raw(x, w, None))
    plan = (ctypes_plan := None)
    ntl = B * T * ((((H + 1) // 2) * (W // 2) + 63) // 64)
    print("B=%d %dx%dx%d  tiles %d (%.2f rounds)  one-axis %.4f  two-axis %.4f  x%.3f  %.1f TFLOP/s alg" % (B, T, H, W, ntl, ntl / 256.0, r[6], r[5], r[6] / r[5], 2.0*B*64*64*27*T*H*W / r[5] / 1e9), flush=True)
lib.hpvg_conv_wino_config(1, -1)
