"""GPU parity of whole train iterations at the BASELINE widths (nfc 64, latent 128, the 256-wide pyramid of configs[2] -
the thing bench.py times): hp_vae_gan_amd.train.StageTrainer.step against the reference-generated summaries of
tests/golden/wide3d_*.pt.  Weights, clip and noise are closed-form (tests/detfill.py); the fixture holds losses, clip norm,
per-parameter gradient norms + strided samples, parameter UPDATES and buffers, and the reference's own spread."""
import pytest
import torch

from helpers import NoiseFeed, flat_to_named, hip_opt, load_golden, wide_compare, wide_inputs

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fname", ["wide3d_vae_s0.pt", "wide3d_gan_s3.pt", "wide2d_vae_s1.pt", "wide2d_gan_s4.pt"])
def test_wide_train_step_matches_reference(fname):
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
    wide_compare(fx, got, lambda n: lr_by_name[n], opt.lr_d, fname)
