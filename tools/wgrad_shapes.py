import sys, os, collections
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, bench
from hp_vae_gan_amd import ops, lib as hplib
lib = hplib.load()
cnt = collections.Counter()
o1, o2 = ops.conv_bwd_weight_raw, ops.conv_bwd_weight_bias_raw
def w1(dy, x, ws, into=None):
    B, Co, T, H, W = ops.geom(dy); cnt[("w", B, x.shape[1], Co, T, H, W, lib.hpvg_conv_bwd_weight_kernel_kind(B, x.shape[1], Co, T, H, W, 3))] += 1
    return o1(dy, x, ws, into)
def w2(dy, x, ws, a, b):
    B, Co, T, H, W = ops.geom(dy); cnt[("wb", B, x.shape[1], Co, T, H, W, lib.hpvg_conv_bwd_weight_kernel_kind(B, x.shape[1], Co, T, H, W, 3))] += 1
    return o2(dy, x, ws, a, b)
ops.conv_bwd_weight_raw, ops.conv_bwd_weight_bias_raw = w1, w2
bench.CONFIG = "video"
built, shapes = bench.build_gpu_stages(torch.device("cuda"), [int(sys.argv[1])])
s, tr, step, real, rz = built[0]
step(); cnt.clear(); step(); torch.cuda.synchronize()
for k, v in sorted(cnt.items(), key=lambda kv: -kv[1]): print(v, k)
