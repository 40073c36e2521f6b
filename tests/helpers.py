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
# reference-generated whole steps at the BASELINE channel widths (tests/golden/make_golden.py wide_fixture): the 256-wide
# pyramid of configs[2] / configs[1], and EVEN-width pyramids (e48 / e78 / e72: W = 48, 78, 72) that put the two-axis Winograd
# conv, the 16-byte Winograd weight gradient and the stage_tail instance inside a whole step
WIDE_FIXTURES = ["wide3d_vae_s0.pt", "wide3d_gan_s3.pt", "wide2d_vae_s1.pt", "wide2d_gan_s4.pt",
                 "wide3d_e48_vae_s0.pt", "wide3d_e78_gan_s2.pt", "wide3d_e72_gan_s2.pt"]


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


def run_hip_stage(fx, device="cuda", sync=True):
    """Drive the product path (hp_vae_gan_amd.train.StageTrainer) over a golden stage fixture; yields per-iteration
    (rec, out, netG, netD, trainer).

    sync: Adam's first step is ~ lr * sign(g), so two correct fp32 implementations land 2 lr apart on every weight whose
    gradient is smaller than their rounding difference, and everything computed AFTER an optimizer step then differs by what
    those few weights do to the network (1e-3 ... 1e-2 on a small critic: chaotic, not a kernel property).  The parity tests
    therefore judge each optimizer step by its UPDATE (helpers.compare_update) and then continue from the REFERENCE's
    post-step state: right after the critic's step (trainer.after_d_step) its recorded state replaces ours - out["D_hip_after"]
    keeps ours for the update check - and each further iteration starts from the previous iteration's recorded networks.  The
    generator step of a GAN stage (critic term, every generator gradient, clip norm) is thereby compared from an identical
    critic at the plain 1e-3."""
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
    prev = None
    for rec in fx["iters"]:
        if sync and prev is not None and "G_after" in prev:
            netG.load_state_dict(prev["G_after"])
            if netD is not None:
                netD.load_state_dict(prev["D_after"])
        hip_d = {}
        if sync and netD is not None and rec.get("D_after") is not None:
            def hook(tr, rec=rec, hip_d=hip_d):
                hip_d.update({k: v.detach().clone() for k, v in tr.netD.state_dict().items()})
                # PARAMETERS only: D_after was recorded at the end of the iteration, when the spectral-norm u / v had seen
                # one more forward (the generator step's) than they have at this point
                pnames = set(n for n, _ in tr.netD.named_parameters())
                tr.netD.load_state_dict({k: v for k, v in rec["D_after"].items() if k in pnames}, strict=False)
            trainer.after_d_step = hook
        netG.noise_source = NoiseFeed(rec["noises"], device)
        alpha = rec["alpha"] if rec["alpha"] is not None else None
        out = trainer.step(real, real_zero, noise_init=rec["noise_init"].to(device), alpha=alpha)
        out = dict(out)
        out["D_hip_after"] = hip_d if hip_d else (netD.state_dict() if netD is not None else None)
        prev = rec
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


# ------------------------------------------------------------------------------------------------ wide (BASELINE-width) steps
def wide_inputs(fx, keysG, keysD):
    """Everything a wide fixture's step consumes, regenerated from tests/detfill.py: (opt, G state, D state, real, real_zero,
    noise_init, [noise tensors in the reference's draw order after noise_init], alpha).  keysG / keysD: state_dicts (any
    values) with the reference's key layout and shapes."""
    import detfill
    from oracle import hpvg_oracle as O
    opt = opt_from(fx["opt"])
    O.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    dims, s = fx["dims"], fx["scale_idx"]
    G = detfill.fill_state(keysG, "G")
    D = detfill.fill_state(keysD, "D") if keysD is not None else None
    real = detfill.det_uniform([opt.batch_size, 3, *O.level_shape(s, opt, dims)], "real")
    real_zero = detfill.det_uniform([opt.batch_size, 3, *O.level_shape(0, opt, dims)], "real_zero") if s > 0 else real
    shapes = fx["expected"]["noise_shapes"]
    draws = [detfill.det_normal(tuple(sh), "n%d" % i) for i, sh in enumerate(shapes)]
    alpha = torch.full((1, 1), float(fx["expected"]["alpha"]))
    return opt, G, D, real, real_zero, draws[0], draws[1:], alpha


def _bn_fed_bias(name, names):
    return name.endswith(".conv.bias") and (name[:-len("conv.bias")] + "norm.weight") in names


def wide_compare(fx, got, lr_of_G, lr_d, what):
    """Compare a step's results with a wide fixture.  Tolerance per quantity: max(north_star 1e-3 relative, 2 x the spread
    the reference itself shows between its oneDNN / 1-thread and native-ATen / 8-thread evaluations of the same step).
    Parameter UPDATES (not parameters) are compared: frozen parameters must not move at all, trained ones must match the
    reference's Adam step except on a small fraction (sign flips of ~0 gradients), bounded by the reference's own."""
    import detfill
    exp, spr = fx["expected"], fx["spread"]
    for k in ("total_loss", "rec_vae_loss", "kl_loss", "errD_real", "errD_fake", "gradient_penalty", "rec_loss", "errG", "total_norm"):
        if k in exp:
            tol = max(RTOL * abs(exp[k]), 2 * spr[k], 1e-7)
            assert abs(float(got[k]) - exp[k]) <= tol, "%s.%s: %r vs %r (tol %.2e)" % (what, k, float(got[k]), exp[k], tol)
    assert [float(a) for a in got["noise_amps"]] == pytest_approx([float(a) for a in exp["noise_amps"]], 1e-4), what + ".noise_amps"

    def tensor(name, t, e, s, count, atol=0.0):
        assert list(t.shape) == e["shape"], "%s.%s: shape %s vs %s" % (what, name, list(t.shape), e["shape"])
        m = detfill.summarize(t.detach().cpu(), count)
        tol = max(RTOL * e["absmax"], 2 * s["sample"], atol)
        err = float((m["sample"].double() - e["sample"].double()).abs().max()) if m["sample"].numel() else 0.0
        assert err <= tol, "%s.%s: sample err %.3e > tol %.3e (absmax %.3e)" % (what, name, err, tol, e["absmax"])
        ntol = max(RTOL * e["norm"], 2 * s["norm"], atol * max(1.0, float(t.numel()) ** 0.5))
        assert abs(m["norm"] - e["norm"]) <= ntol, "%s.%s: norm %.6e vs %.6e (tol %.2e)" % (what, name, m["norm"], e["norm"], ntol)

    for k in ("generated", "generated_vae", "mu", "logvar", "fake"):
        if k in exp:
            tensor(k, got[k], exp[k], spr[k], 4096)
    for key, count in (("gradsG", 256), ("gradsD", 256)):
        if key not in exp:
            continue
        names = set(exp[key])
        # the per-tensor spread is a two-sample estimate; the largest RELATIVE spread over the tensors of the same family
        # (one network's gradients in one backward pass, BatchNorm-fed conv biases aside) steadies it
        fam = max([spr[key][n]["sample"] / e["absmax"] for n, e in exp[key].items()
                   if e is not None and e["absmax"] > 0 and not _bn_fed_bias(n, names)] + [0.0])
        for n, e in exp[key].items():
            if e is None:
                g = got[key].get(n)
                assert g is None or float(g.abs().max()) == 0.0, "%s.%s.%s should have no gradient" % (what, key, n)
                continue
            atol = max(1e-7, fam * e["absmax"])
            if _bn_fed_bias(n, names):   # exactly-zero true gradient: rounding noise, judged on the layer's weight gradient
                atol = 1e-4 * exp[key][n[:-len("bias")] + "weight"]["absmax"]
            tensor(key + "." + n, got[key][n], e, spr[key][n], count, atol=atol)
    for key, lr_of in (("G_delta", lr_of_G), ("D_delta", lambda n: lr_d)):
        if key not in exp:
            continue
        names = set(exp[key])
        for n, e in exp[key].items():
            d = got[key][n].detach().cpu().double().reshape(-1)
            lr = lr_of(n)
            if lr is None:   # outside every optimizer group: must not move
                assert e["absmax"] == 0.0 and float(d.abs().max()) == 0.0, "%s.%s.%s: a frozen parameter moved" % (what, key, n)
                continue
            assert float(d.abs().max()) <= lr * (1 + 1e-3) + 2.4e-7, "%s.%s.%s: |update| %.3e exceeds the Adam bound lr = %.3e" % (what, key, n, float(d.abs().max()), lr)
            if _bn_fed_bias(n, names):
                continue         # gradient is rounding noise: size and sign of the update are arbitrary on both sides
            if e["absmax"] > 0:  # the reference moved it: so must we (a skipped optimizer step / wrong arena range fails here)
                assert float(d.abs().max()) >= 0.5 * e["absmax"], "%s.%s.%s: update %.3e, reference %.3e" % (what, key, n, float(d.abs().max()), e["absmax"])
            idx = detfill.sample_idx(d.numel(), 256)
            diff = (d[idx] - e["sample"].double()).abs()
            frac = float((diff > lr / 10).double().mean())
            allowed = max(0.02, 2 * (spr[key][n]["frac_lr10"] or 0.0))
            assert frac <= allowed, "%s.%s.%s: %.1f %% of the sampled updates differ by more than lr/10 (allowed %.1f %%)" % (
                what, key, n, 100 * frac, 100 * allowed)
            # away from sign flips the first Adam step is -lr*g/(|g| + 1e-8) ~ -lr*sign(g): a group learning rate that is off
            # by a few per cent (a wrong lr_scale power) shows up here
            frac2 = float((diff > lr / 100).double().mean())
            assert frac2 <= allowed + 0.03, "%s.%s.%s: %.1f %% of the sampled updates differ by more than lr/100" % (what, key, n, 100 * frac2)
    for key in ("G_buffers", "D_buffers"):
        if key not in exp:
            continue
        for n, v in exp[key].items():
            if n.endswith("num_batches_tracked"):
                assert int(got[key][n]) == int(v), "%s.%s.%s" % (what, key, n)
            else:
                assert_close(got[key][n].float().cpu(), v.float(), RTOL, "%s.%s.%s" % (what, key, n), atol=max(1e-6, 2 * spr[key].get(n) if isinstance(spr.get(key), dict) and spr[key].get(n) is not None else 1e-6))


def pytest_approx(v, rel):
    import pytest
    return pytest.approx(v, rel=rel)


# ------------------------------------------------------------------------------------------------ recorded (small) step fixtures
def _amax(t):
    return float(t.detach().double().abs().max()) if t.numel() else 0.0


def compare_update(what, before, after_ref, after_got, lr, spread_frac=0.0, bn_fed=False, first_step=True):
    """A parameter's UPDATE (after - before) against the reference's.  lr None: the parameter is outside every optimizer
    group and must not move (exact).  Otherwise Adam's step: bounded by lr (first step) and equal to the reference's except
    where a ~0 gradient flipped its sign (fraction of elements off by more than lr/10, bounded by 2 % or twice the
    reference's own fraction between its evaluations)."""
    d_ref = (after_ref.double() - before.double()).reshape(-1)
    d_got = (after_got.detach().cpu().double() - before.double()).reshape(-1)
    if lr is None:
        assert _amax(d_ref) == 0.0 and _amax(d_got) == 0.0, "%s: a parameter outside the optimizer groups moved (%.3e)" % (what, _amax(d_got))
        return
    ulp = 1.2e-7 * _amax(before)   # the update is read off fp32 parameters: one rounding of the parameter itself
    bound = (lr * (1 + 1e-3) if first_step else 3 * lr) + ulp
    assert _amax(d_got) <= bound, "%s: |update| %.3e exceeds %.3e" % (what, _amax(d_got), bound)
    if bn_fed:
        return   # exactly-zero true gradient: rounding noise decides size and sign of the update on both sides
    if _amax(d_ref) > 0:
        assert _amax(d_got) >= 0.5 * _amax(d_ref), "%s: update %.3e, reference %.3e (skipped step / wrong range?)" % (what, _amax(d_got), _amax(d_ref))
    diff = (d_got - d_ref).abs()
    frac = float((diff > lr / 10).double().mean())
    allowed = max(0.02, 2 * spread_frac)
    assert frac <= allowed, "%s: %.2f %% of the updates differ from the reference's by more than lr/10 (allowed %.2f %%)" % (what, 100 * frac, 100 * allowed)
    frac2 = float((diff > lr / 100).double().mean())
    assert frac2 <= allowed + 0.03, "%s: %.2f %% of the updates differ by more than lr/100" % (what, 100 * frac2)


def compare_step(what, rec, spread, got, rtol=RTOL):
    """Losses, outputs, gradients and the clip norm of one recorded iteration.  `spread` (may be None for old fixtures):
    the reference's own per-quantity spread; tolerance = max(rtol * scale, 2 * spread[, family floor])."""
    spread = spread or {}
    for k in ("total_loss", "rec_vae_loss", "kl_loss", "errD_real", "errD_fake", "gradient_penalty", "rec_loss", "errG", "total_norm"):
        if k in rec and k in got:
            assert_close(torch.as_tensor(got[k]).reshape(()), rec[k].reshape(()), rtol, "%s.%s" % (what, k), atol=2 * spread.get(k, 0.0))
    for k in ("generated", "generated_vae", "mu", "logvar", "fake"):
        if k in rec and k in got and rec[k] is not None:
            assert_close(got[k], rec[k], rtol, "%s.%s" % (what, k), atol=2 * spread.get(k, 0.0))
    for key in ("gradsG", "gradsD"):
        if key not in rec or key not in got:
            continue
        names = set(n for n, g in rec[key].items() if g is not None)
        sp = spread.get(key, {})
        fam = max([sp.get(n, 0.0) / _amax(rec[key][n]) for n in names if _amax(rec[key][n]) > 0 and not _bn_fed_bias(n, names)] + [0.0])
        for n, g in rec[key].items():
            mine = got[key].get(n)
            if g is None:
                assert mine is None or _amax(mine) == 0.0, "%s.%s.%s should have no gradient" % (what, key, n)
                continue
            atol = max(1e-7, 2 * sp.get(n, 0.0), fam * _amax(g))
            if _bn_fed_bias(n, names):
                atol = max(bn_bias_atol(n, rec[key], 1e-7), 0.0)
            assert_close(mine, g, rtol, "%s.%s.%s" % (what, key, n), atol=atol)
