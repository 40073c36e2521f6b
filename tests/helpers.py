"""Shared helpers for the parity tests (fixture loading, tolerances, oracle state handling)."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# north_star tolerance: outputs within 1e-3 relative (fp32) of the reference's PyTorch-CPU path
RTOL = 1e-3


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


def opt_from(d, **kw):
    o = types.SimpleNamespace(**d)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def rel_err(a, b):
    """max |a-b| / max(|b|) - the relative error measure used for every tensor comparison."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / denom


def assert_close(a, b, rtol=RTOL, what="", atol=0.0):
    assert tuple(a.shape) == tuple(b.shape), "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    a64 = a.detach().double().cpu()
    b64 = b.detach().double().cpu()
    err = float((a64 - b64).abs().max()) if a64.numel() else 0.0
    scale = float(b64.abs().max()) if b64.numel() else 0.0
    assert err <= rtol * scale + atol, "%s: max abs err %.3e vs scale %.3e (rel %.3e > %.1e)" % (
        what, err, scale, err / max(scale, 1e-30), rtol)


def bn_bias_atol(k, grads, base=1e-7):
    """Absolute slack for the bias of a conv that feeds a BatchNorm: its true gradient is exactly 0 (BN removes the
    mean), both sides hold rounding noise there, so it is judged on the scale of the layer's weight gradient."""
    if k.endswith(".conv.bias") and (k[:-len("conv.bias")] + "norm.weight") in grads:
        w = grads[k[:-len("bias")] + "weight"]
        return 1e-4 * float(w.abs().max())
    return base


def oracle_state(sd, requires_grad=True):
    """state_dict (fixture) -> oracle parameter dict (clones; parameters get requires_grad)."""
    from oracle import hpvg_oracle as O
    P = {}
    for k, v in sd.items():
        t = v.clone()
        if requires_grad and O.is_param(k):
            t.requires_grad_(True)
        P[k] = t
    return P


# ------------------------------------------------------------------------------------------------ HIP-path drivers
def hip_opt(fx_opt, dims, scale_idx, device="cuda"):
    """`opt` blackboard for the product path from a golden fixture's option dict."""
    import hp_vae_gan_amd as hp  # noqa: F401
    from hp_vae_gan_amd import utils as hu
    opt = opt_from(fx_opt, device=device, dims=dims, scale_idx=scale_idx, Noise_Amps=[],
                   generator="GeneratorHPVAEGAN", discriminator="WDiscriminator3D" if dims == 3 else "WDiscriminator2D")
    hu.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    return opt


class NoiseFeed:
    """noise_source for GeneratorHPVAEGAN: hands out recorded N(0,1) tensors in order (and checks their shapes)."""

    def __init__(self, tensors, device):
        self.it = iter(tensors)
        self.device = device

    def __call__(self, ref):
        t = next(self.it)
        assert tuple(t.shape) == tuple(ref.shape), "noise shape %s, expected %s" % (tuple(t.shape), tuple(ref.shape))
        return t.to(self.device)


def run_hip_stage(fx, device="cuda"):
    """Drive the product path (hp_vae_gan_amd.train.StageTrainer) over a golden stage fixture; yields per-iteration
    (rec, out, netG, netD, trainer)."""
    import hp_vae_gan_amd as hp  # noqa: F401
    from hp_vae_gan_amd import train as hp_train
    from hp_vae_gan_amd.modules import networks_2d, networks_3d
    dims, s = fx["dims"], fx["scale_idx"]
    opt = hip_opt(fx["opt"], dims, s, device)
    nets = networks_3d if dims == 3 else networks_2d
    netG = nets.GeneratorHPVAEGAN(opt)
    for _ in range(s):
        netG.init_next_stage()
    netG.load_state_dict(fx["G_init"])
    netG.to(device)
    netD = None
    if fx["D_init"] is not None:
        netD = getattr(nets, opt.discriminator)(opt)
        netD.load_state_dict(fx["D_init"])
        netD.to(device)
    opt.Noise_Amps = list(fx["noise_amps_init"])
    opt.record_grads = True
    trainer = hp_train.StageTrainer(opt, netG, netD)
    real, real_zero = fx["real"].to(device), fx["real_zero"].to(device)
    for rec in fx["iters"]:
        netG.noise_source = NoiseFeed(rec["noises"], device)
        alpha = rec["alpha"] if rec["alpha"] is not None else None
        out = trainer.step(real, real_zero, noise_init=rec["noise_init"].to(device), alpha=alpha)
        yield rec, out, netG, netD, trainer


def arena_grads(arena, module):
    """name -> gradient tensor (views of the recorded flat gradient)."""
    return {n: p.grad for n, p in module.named_parameters()}


def flat_to_named(flat, arena, module):
    out = {}
    for n, p in module.named_parameters():
        o, k = arena.range[id(p)]
        out[n] = flat[o:o + k].view(p.shape)
    return out
