"""Micro-benchmark of the two MFMA conv kernels at a given pyramid-stage shape (development tool, not a test).
usage: python tools/perf_conv.py [stage] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hp_vae_gan_amd
from hp_vae_gan_amd import ops

SHAPES = {0: (4, 18, 33), 1: (4, 23, 41), 2: (4, 29, 52), 3: (5, 36, 65), 4: (5, 45, 81), 5: (5, 57, 102), 6: (7, 72, 129), 7: (7, 91, 162), 8: (7, 114, 204), 9: (13, 144, 256)}
stage = int(sys.argv[1]) if len(sys.argv) > 1 else 9
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
T, H, W = SHAPES[stage]
B = int(os.environ.get("HPVG_PERF_B", "2"))
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(B, 64, T, H, W, device=dev)
dy = torch.randn(B, 64, T, H, W, device=dev)
w = torch.randn(64, 64, 3, 3, 3, device=dev) * 0.05
b = torch.randn(64, device=dev)
flops = 2.0 * B * 64 * 64 * 27 * T * H * W


def bench(name, fn, fl=flops):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-28s stage %d  %8.3f ms  %7.2f TFLOP/s (%.1f%% of 157.3)" % (name, stage, ms, fl / ms / 1e9, 100 * fl / ms / 1e9 / 157.3))


bench("conv_fwd 64->64", lambda: ops.conv_fwd_raw(x, w, b))
sc = torch.rand(64, device=dev) + 0.5
sh = torch.randn(64, device=dev) * 0.1
bench("conv_fwd 64->64 fused BN+lrelu in", lambda: ops.conv_fwd_raw(x, w, b, in_affine=(sc, sh), in_lrelu=True))
bench("conv_bwd_data 64->64", lambda: ops.conv_fwd_raw(dy, w, None, flip=True))
bench("conv_bwd_weight 64->64", lambda: ops.conv_bwd_weight_raw(dy, x, w.shape))
w3 = torch.randn(3, 64, 3, 3, 3, device=dev) * 0.05
bench("conv_fwd 64->3 (tail)", lambda: ops.conv_fwd_raw(x, w3, None), fl=flops * 3 / 64)
wh = torch.randn(64, 3, 3, 3, 3, device=dev) * 0.05
x3 = torch.randn(B, 3, T, H, W, device=dev)
bench("conv_fwd 3->64 (head)", lambda: ops.conv_fwd_raw(x3, wh, None), fl=flops * 3 / 64)
bench("conv_bwd_weight 3->64", lambda: ops.conv_bwd_weight_raw(dy, x3, wh.shape), fl=flops * 3 / 64)
dy3 = torch.randn(B, 3, T, H, W, device=dev)
bench("conv_bwd_weight 64->3", lambda: ops.conv_bwd_weight_raw(dy3, x, w3.shape), fl=flops * 3 / 64)
bench("channel_sum", lambda: ops.channel_sum_raw(dy), fl=0)
