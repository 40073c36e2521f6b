"""Whole train steps at the BASELINE widths (nfc 64, latent 128, the 256-wide pyramid of configs[2]): the CPU oracle
against the reference-generated summaries of tests/golden/wide3d_*.pt (closed-form inputs from tests/detfill.py)."""
import pytest
import torch

from helpers import WIDE_FIXTURES, load_golden, oracle_state, wide_compare, wide_inputs
from oracle import hpvg_oracle as O


def _containers(opt, dims, s, gan):
    """state_dicts with the reference's key layout (the product modules are parameter containers here; no compute)."""
    from hp_vae_gan_amd.modules import networks_2d, networks_3d
    nets = networks_3d if dims == 3 else networks_2d
    netG = nets.GeneratorHPVAEGAN(opt)
    for _ in range(s):
        netG.init_next_stage()
    netD = (nets.WDiscriminator3D if dims == 3 else nets.WDiscriminator2D)(opt) if gan else None
    return netG.state_dict(), (netD.state_dict() if gan else None)


@pytest.mark.parametrize("fname", WIDE_FIXTURES)
def test_oracle_wide_step(fname):
    from helpers import opt_from
    fx = load_golden(fname)
    dims, s = fx["dims"], fx["scale_idx"]
    opt0 = opt_from(fx["opt"], dims=dims)
    gan = opt0.vae_levels < s + 1
    kG, kD = _containers(opt0, dims, s, gan)
    opt, G, D, real, real_zero, noise_init, noises, alpha = wide_inputs(fx, kG, kD)
    PG = oracle_state(G)
    PD = oracle_state(D) if gan else None
    G0 = {k: v.detach().clone() for k, v in PG.items()}
    D0 = {k: v.detach().clone() for k, v in PD.items()} if gan else None
    amps = [1] + [0.05 + 0.01 * k for k in range(1, s)]
    it = iter(noises)
    O.noise_amp_for_stage(PG, opt, dims, s, real, real_zero, amps, it)
    out = O.train_step(PG, PD, opt, dims, s, real, real_zero, noise_init, it, alpha.reshape(()) if gan else None, amps, {}, {})
    got = dict(out)
    got["noise_amps"] = amps
    got["G_delta"] = {k: PG[k].detach() - G0[k] for k in G0 if O.is_param(k)}
    got["G_buffers"] = {k: PG[k].detach() for k in G0 if not O.is_param(k)}
    if gan:
        got["D_delta"] = {k: PD[k].detach() - D0[k] for k in D0 if O.is_param(k)}
        got["D_buffers"] = {k: PD[k].detach() for k in D0 if not O.is_param(k)}
    groups = O.g_param_groups(PG, opt, s)

    def lr_of(name):
        for prefix, lr in groups:
            if name.startswith(prefix):
                return lr
        return None
    wide_compare(fx, got, lr_of, opt.lr_d, fname)
