"""Weight gradient 64 -> 64 per pyramid stage: the direct kernels (conv_wgrad_kernel, or conv_wgrad3_kernel where the
library picks it) against the Winograd kernel (conv_wgradw_kernel), same process, switched through
hpvg_conv_bwd_weight_wino_config (development tool).  usage: python tools/perf_wgrad_wino.py [reps] [stages...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hp_vae_gan_amd
from hp_vae_gan_amd import ops, lib as hplib

SHAPES = {0: (4, 18, 33), 1: (4, 23, 41), 2: (4, 29, 52), 3: (5, 36, 65), 4: (5, 45, 81), 5: (5, 57, 102), 6: (7, 72, 129), 7: (7, 91, 162), 8: (7, 114, 204), 9: (13, 144, 256)}
# HPVG_PERF_SHAPES="T,H,W;T,H,W;..." adds shapes as stages 100, 101, ...
for k, spec in enumerate(filter(None, os.environ.get("HPVG_PERF_SHAPES", "").split(";"))):
    SHAPES[100 + k] = tuple(int(v) for v in spec.split(","))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
stages = [int(a) for a in sys.argv[2:]] or list(range(10))
B = int(os.environ.get("HPVG_PERF_B", "2"))
dims = int(os.environ.get("HPVG_PERF_DIMS", "3"))
lib = hplib.load()
dev = "cuda"


def bench(fn):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for st in stages:
    T, H, W = SHAPES[st]
    sp = (T, H, W) if dims == 3 else (H, W)
    torch.manual_seed(0)
    x = torch.randn(B, 64, *sp, device=dev)
    dy = torch.randn(B, 64, *sp, device=dev)
    wshape = (64, 64) + (3,) * dims
    flops = 2.0 * B * 64 * 64 * (27 if dims == 3 else 9) * x[0, 0].numel()
    row, out = {}, {}
    for mode in (0, 2, 5):
        lib.hpvg_conv_bwd_weight_wino_config(mode)
        row[mode] = bench(lambda: ops.conv_bwd_weight_raw(dy, x, wshape))
        out[mode] = ops.conv_bwd_weight_raw(dy, x, wshape)
    err = float((out[2] - out[0]).abs().max() / out[0].abs().max())
    err5 = float((out[5] - out[0]).abs().max() / out[0].abs().max())
    kind5 = lib.hpvg_conv_bwd_weight_kernel_kind(B, 64, 64, T if dims == 3 else 1, H, W, 3 if dims == 3 else 1)
    print("stage %d %s B=%d  direct %.4f ms | wino %.4f ms | x%.3f | wino %.1f TFLOP/s (algorithmic) | rel diff %.2e | two-axis(kind %d) %.4f ms x%.3f over one-axis, %.1f TFLOP/s, rel diff %.2e"
          % (st, sp, B, row[0], row[2], row[0] / row[2], flops / row[2] / 1e9, err, kind5, row[5], row[2] / row[5], flops / row[5] / 1e9, err5), flush=True)
lib.hpvg_conv_bwd_weight_wino_config(1)
