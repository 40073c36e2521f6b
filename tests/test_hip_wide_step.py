"""GPU parity of whole train iterations at the BASELINE widths (nfc 64, latent 128, the 256-wide pyramid of configs[2] -
the thing bench.py times): hp_vae_gan_amd.train.StageTrainer.step against the reference-generated summaries of
tests/golden/wide3d_*.pt.  Weights, clip and noise are closed-form (tests/detfill.py); the fixture holds losses, clip norm,
per-parameter gradient norms + strided samples, parameter UPDATES and buffers, and the reference's own spread."""
import pytest
import torch

from helpers import WIDE_FIXTURES, NoiseFeed, flat_to_named, hip_opt, load_golden, wide_compare, wide_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fname", WIDE_FIXTURES)
def test_wide_train_step_matches_reference(fname):
    _run_wide_step(fname)


# hpvg_conv_wino_config modes: 1 = by size (the default), 5 = the two-axis Winograd kernel wherever it can run (even W),
# 3 / 4 = the one-axis kernel with the rows-as-in-memory / halo'd-band staging form, 0 = the direct kernel;
# hpvg_conv_bwd_weight_wino_config modes: 1 = default, 5 = the two-axis Winograd kernel, 2 = the one-axis kernel
# on every wide layer, 4 = its 16-byte form on four waves, 3 = its 4-byte form only, 0 = the direct weight-gradient kernels
CONV_MODES = [1, 5, 3, 4, 0]
WGRAD_MODES = [1, 5, 2, 4, 3, 0]


@pytest.mark.parametrize("fname", ["wide3d_e48_vae_s0.pt", "wide3d_e78_gan_s2.pt", "wide3d_e72_gan_s2.pt", "wide3d_gan_s3.pt"])
def test_wide_train_step_under_every_forced_kernel(fname):
    """The same reference-generated step with each conv kernel family x each weight-gradient family forced at run time: the
    mask-bit hand-off (Conv.apply_bits), the fused bias gradient into the grad slots, the geometry-dependent weight packs and
    the BatchNorm / spectral-norm wiring are the whole step's, the kernels are the ones that own bench stages 8-9
    (W = 48 / 72: two-axis conv + 16-byte eight-wave weight gradient; W = 78: the stage_tail instance, H * W = 2 mod 4).
    Same tolerances as the default-mode test: a failure here is a finding, not a tolerance question."""
    from hp_vae_gan_amd import lib as hplib, ops
    lib = hplib.load()
    ran = []
    try:
        for cm in CONV_MODES:
            for wm in WGRAD_MODES:
                assert lib.hpvg_conv_wino_config(cm, -1) == cm
                assert lib.hpvg_conv_bwd_weight_wino_config(wm) == wm
                ops.weights_changed()
                _run_wide_step(fname, "%s conv mode %d wgrad mode %d" % (fname, cm, wm))
                ran.append((cm, wm))
    finally:
        lib.hpvg_conv_wino_config(1, -1)
        lib.hpvg_conv_bwd_weight_wino_config(1)
        ops.weights_changed()
    assert len(ran) == len(CONV_MODES) * len(WGRAD_MODES)


def test_forced_modes_select_the_kernels_they_name():
    """what the forced modes of the test above mean at the fixtures' top-level shapes (host queries, no launches)"""
    from hp_vae_gan_amd import lib as hplib
    lib = hplib.load()
    try:
        for (T, H, W) in [(4, 27, 48), (4, 43, 78), (5, 40, 72)]:
            for B in (2, 4):
                geo = (B, 64, 64, T, H, W, 3)
                lib.hpvg_conv_wino_config(5, -1)
                assert lib.hpvg_conv_fwd_kernel_kind(*geo) == 2
                for m in (3, 4):
                    lib.hpvg_conv_wino_config(m, -1)
                    assert lib.hpvg_conv_fwd_kernel_kind(*geo) == 1
                lib.hpvg_conv_wino_config(0, -1)
                assert lib.hpvg_conv_fwd_kernel_kind(*geo) == 0
                for m in (2, 3, 4, 6):
                    lib.hpvg_conv_bwd_weight_wino_config(m)
                    assert lib.hpvg_conv_bwd_weight_kernel_kind(*geo) == 2
                lib.hpvg_conv_bwd_weight_wino_config(5)
                assert lib.hpvg_conv_bwd_weight_kernel_kind(*geo) == 3      # (every even width: 48, 72 and 78 = 2 mod 4)
                lib.hpvg_conv_bwd_weight_wino_config(0)
                assert lib.hpvg_conv_bwd_weight_kernel_kind(*geo) in (0, 1)
    finally:
        lib.hpvg_conv_wino_config(1, -1)
        lib.hpvg_conv_bwd_weight_wino_config(1)


def _run_wide_step(fname, what=None):
    import hp_vae_gan_amd as hp  # noqa: F401
    from hp_vae_gan_amd import train as hp_train
    from hp_vae_gan_amd.modules import networks_2d, networks_3d
    dev = "cuda"
    fx = load_golden(fname)
    dims, s = fx["dims"], fx["scale_idx"]
    opt = hip_opt(fx["opt"], dims, s, dev)
    nets = networks_3d if dims == 3 else networks_2d
    netG = nets.GeneratorHPVAEGAN(opt)
    for _ in range(s):
        netG.init_next_stage()
    gan = opt.vae_levels < s + 1
    netD = getattr(nets, opt.discriminator)(opt) if gan else None
    _, G, D, real, real_zero, noise_init, noises, alpha = wide_inputs(fx, netG.state_dict(), netD.state_dict() if gan else None)
    netG.load_state_dict(G)
    netG.to(dev)
    if gan:
        netD.load_state_dict(D)
        netD.to(dev)
    opt.Noise_Amps = [1] + [0.05 + 0.01 * k for k in range(1, s)]
    opt.record_grads = True
    trainer = hp_train.StageTrainer(opt, netG, netD)
    G0 = {k: v.detach().clone() for k, v in netG.state_dict().items()}
    D0 = {k: v.detach().clone() for k, v in netD.state_dict().items()} if gan else None
    netG.noise_source = NoiseFeed(noises, dev)
    out = trainer.step(real.to(dev), real_zero.to(dev), noise_init=noise_init.to(dev), alpha=alpha if gan else None)
    got = {k: v for k, v in out.items() if k not in ("gradG_flat", "gradD_flat", "clip_info")}
    got["total_norm"] = out["clip_info"][1]
    got["noise_amps"] = opt.Noise_Amps
    got["gradsG"] = flat_to_named(out["gradG_flat"], trainer.arenaG, netG)
    G1 = netG.state_dict()
    pnames = set(n for n, _ in netG.named_parameters())
    got["G_delta"] = {k: G1[k].float() - G0[k].float() for k in pnames}
    got["G_buffers"] = {k: v for k, v in G1.items() if k not in pnames}
    if gan:
        got["gradsD"] = flat_to_named(out["gradD_flat"], trainer.arenaD, netD)
        D1 = netD.state_dict()
        dnames = set(n for n, _ in netD.named_parameters())
        got["D_delta"] = {k: D1[k].float() - D0[k].float() for k in dnames}
        got["D_buffers"] = {k: v for k, v in D1.items() if k not in dnames}
    lr_by_id = {}
    for params, lr in hp_train.generator_param_groups(opt, netG):
        for p in params:
            lr_by_id[id(p)] = lr
    lr_by_name = {n: lr_by_id.get(id(p)) for n, p in netG.named_parameters()}
    wide_compare(fx, got, lambda n: lr_by_name[n], opt.lr_d, what or fname)
