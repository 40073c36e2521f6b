"""Winograd (conv_wino_kernel) against the direct MFMA kernel at every pyramid-stage shape, same process, mode switched
through hpvg_conv_wino_config (development tool, not a test).  usage: python tools/perf_wino.py [reps] [stages...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import hp_vae_gan_amd
from hp_vae_gan_amd import ops, lib as hplib

SHAPES = {0: (4, 18, 33), 1: (4, 23, 41), 2: (4, 29, 52), 3: (5, 36, 65), 4: (5, 45, 81), 5: (5, 57, 102), 6: (7, 72, 129), 7: (7, 91, 162), 8: (7, 114, 204), 9: (13, 144, 256)}
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
    w = torch.randn(64, 64, *([3] * dims), device=dev) * 0.05
    b = torch.randn(64, device=dev)
    flops = 2.0 * B * 64 * 64 * (27 if dims == 3 else 9) * x[0, 0].numel()
    row = {}
    for mode in (0, 2) if dims == 2 else (0, 6, 5):   # direct | one-axis Winograd | two-axis where it can run (else one-axis)
        lib.hpvg_conv_wino_config(mode, -1)
        row[mode] = (bench(lambda: ops.conv_fwd_raw(x, w, b)), bench(lambda: ops.conv_fwd_raw(x, w, None, flip=True)))
    if dims == 3:
        lib.hpvg_conv_wino_config(0, -1); yd = ops.conv_fwd_raw(x, w, b)
        lib.hpvg_conv_wino_config(5, -1); y5 = ops.conv_fwd_raw(x, w, b)
        row[2] = row[6]
        print("stage %d  two-axis fwd %.4f bwd %.4f ms  x%.3f over one-axis  %.1f TFLOP/s (algorithmic)  rel diff vs direct %.2e"
              % (st, row[5][0], row[5][1], row[6][0] / row[5][0], flops / row[5][0] / 1e9, float((y5 - yd).abs().max() / yd.abs().max())), flush=True)
    y0 = None
    lib.hpvg_conv_wino_config(0, -1); y0 = ops.conv_fwd_raw(x, w, b)
    lib.hpvg_conv_wino_config(2, -1); y2 = ops.conv_fwd_raw(x, w, b)
    err = float((y2 - y0).abs().max() / y0.abs().max())
    print("stage %d %s B=%d  direct fwd %.4f bwd %.4f ms | wino fwd %.4f bwd %.4f ms | x%.3f x%.3f | wino %.1f TFLOP/s (algorithmic) | rel diff %.2e"
          % (st, sp, B, row[0][0], row[0][1], row[2][0], row[2][1], row[0][0] / row[2][0], row[0][1] / row[2][1],
             flops / row[2][0] / 1e9, err), flush=True)
lib.hpvg_conv_wino_config(1, -1)
