"""Child process of tests/test_hip_ops.py::test_wgrad_all_taps_kernel_on_small_shapes: started with HPVG_WGRAD3=2, which sends
every 3x3x3 weight gradient through conv_wgrad3_kernel (by default only launches with >= 64 tiles per workgroup take it), and
compares with the oracle's (torch CPU autograd of the restated conv) on odd shapes: ragged tiles, T = 1 and 2, channel counts
that are not multiples of 32 / 64, several 64-blocks, accumulation into an existing gradient."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

import hp_vae_gan_amd  # noqa: E402,F401
from hp_vae_gan_amd import ops  # noqa: E402
from oracle import hpvg_oracle as O  # noqa: E402

assert os.environ.get("HPVG_WGRAD3") == "2"
torch.manual_seed(0)
worst = 0.0
for (B, Ci, Co, T, H, W) in [(2, 64, 64, 5, 19, 23), (1, 8, 8, 1, 7, 9), (2, 5, 70, 2, 9, 11), (1, 33, 16, 3, 12, 40), (3, 128, 96, 4, 10, 13),
                             (2, 64, 64, 7, 33, 65), (1, 24, 24, 13, 8, 8)]:
    x = torch.randn(B, Ci, T, H, W)
    dy = torch.randn(B, Co, T, H, W)
    w = torch.zeros(Co, Ci, 3, 3, 3, requires_grad=True)
    O.conv(x, w, None).backward(dy)
    want = w.grad
    got = ops.conv_bwd_weight_raw(dy.cuda(), x.cuda(), w.shape).cpu()
    err = float((got - want).abs().max()) / float(want.abs().max())
    worst = max(worst, err)
    assert err < 2e-5, ("wgrad3", (B, Ci, Co, T, H, W), err)
    # accumulate into an existing gradient, and bitwise reproducibility of a second launch
    base = torch.randn_like(want)
    into = base.cuda().clone()
    ops.conv_bwd_weight_raw(dy.cuda(), x.cuda(), w.shape, into=into)
    err = float((into.cpu() - (base + want)).abs().max()) / float(want.abs().max())
    assert err < 2e-5, ("wgrad3 accumulate", (B, Ci, Co, T, H, W), err)
    again = ops.conv_bwd_weight_raw(dy.cuda(), x.cuda(), w.shape).cpu()
    assert torch.equal(again, got), ("wgrad3 not reproducible", (B, Ci, Co, T, H, W))
print("wgrad3 ok, worst relative error %.2e" % worst)
