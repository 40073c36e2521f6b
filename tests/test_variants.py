"""The reference's variant models - Encode3DVAE_nb / Encode2DVAE_nb, Encode3DVAE1x1, GeneratorVAE_nb, reparameterize_bern,
kl_bern_criterion (networks_3d.py:38-45,110-160,409-485; networks_2d.py:115-165,272-348; losses.py:12-14) - against vectors
recorded from the reference's own classes (tests/golden/variants.pt): the oracle on CPU, the HIP modules on the GPU."""
import pytest
import torch

from helpers import RTOL, NoiseFeed, assert_close, bn_bias_atol, load_golden, opt_from, oracle_state
from oracle import hpvg_oracle as O

CASES = ["gen3d_rec", "gen3d_rand", "gen2d_rec", "gen2d_rand"]


def _loss(x, vae_out, mu, logvar, bern, real, video, mse, kl, klb):
    return 10.0 * (mse(x, real) + mse(vae_out, video)) + kl(mu, logvar) + klb(bern)


@pytest.mark.parametrize("case", CASES)
def test_oracle_generator_vae_nb(case):
    fx = load_golden("variants.pt")[case]
    opt = opt_from(fx["opt"])
    O.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    P = oracle_state(fx["G_init"])
    x, vae_out, (mu, logvar, bern) = O.generator_vae_nb_forward(P, opt, fx["dims"], fx["video"], fx["amps"], iter(fx["draws"]), fx["mode"])
    for k, v in (("x", x), ("vae_out", vae_out), ("mu", mu), ("logvar", logvar), ("bern", bern)):
        assert_close(v, fx[k], 2e-5, "%s.%s" % (case, k))
    assert_close(O.kl_bern_criterion(bern), fx["kl_bern"], 2e-5, case + ".kl_bern")
    loss = _loss(x, vae_out, mu, logvar, bern, fx["real"], fx["video"], O.mse, O.kl_criterion, O.kl_bern_criterion)
    assert_close(loss, fx["loss"], 2e-5, case + ".loss")
    names = [k for k in P if O.is_param(k)]
    grads = torch.autograd.grad(loss, [P[k] for k in names], allow_unused=True)
    for k, g in zip(names, grads):
        want = fx["grads"][k]
        if want is None:
            assert g is None or float(g.abs().max()) == 0.0, k
        else:
            assert_close(g, want, 2e-4, "%s.grad.%s" % (case, k), atol=bn_bias_atol(k, fx["grads"], 1e-7))
    for k, v in fx["G_after"].items():                      # BatchNorm running statistics / spectral-norm u, v moved by the forward
        if not O.is_param(k) and not k.endswith("num_batches_tracked"):
            assert_close(P[k].detach(), v, 2e-5, "%s.buffer.%s" % (case, k), atol=1e-7)


@pytest.mark.parametrize("case", ["enc1x1_3d", "enc1x1_2d"])
def test_oracle_encoder_1x1(case):
    fx = load_golden("variants.pt")[case]
    P = oracle_state(fx["E_init"])
    x = fx["x"].clone().requires_grad_(True)
    mu, logvar = O.encoder_1x1_forward(x, P)
    assert_close(mu, fx["mu"], 2e-5, case + ".mu")
    assert_close(logvar, fx["logvar"], 2e-5, case + ".logvar")
    names = [k for k in P if O.is_param(k)]
    grads = torch.autograd.grad([mu, logvar], [x] + [P[k] for k in names], [fx["gmu"], fx["glogvar"]])
    assert_close(grads[0], fx["dx"], 2e-4, case + ".dx")
    for k, g in zip(names, grads[1:]):
        assert_close(g, fx["dparams"][k], 2e-4, "%s.d%s" % (case, k), atol=1e-7)


@pytest.mark.gpu
@pytest.mark.parametrize("case", CASES)
def test_hip_generator_vae_nb(case):
    import hp_vae_gan_amd  # noqa: F401
    from hp_vae_gan_amd.modules import networks_2d, networks_3d
    from hp_vae_gan_amd.modules.losses import kl_bern_criterion, kl_criterion, mse_loss
    from helpers import hip_opt
    fx = load_golden("variants.pt")[case]
    dev = "cuda"
    dims = fx["dims"]
    opt = hip_opt(fx["opt"], dims, 2, dev)
    nets = networks_3d if dims == 3 else networks_2d
    G = nets.GeneratorVAE_nb(opt)
    for _ in range(2):
        G.init_next_stage()
    assert list(G.state_dict().keys()) == list(fx["G_init"].keys())
    G.load_state_dict(fx["G_init"])
    G.to(dev)
    G.noise_source = NoiseFeed(fx["draws"], dev)
    x, vae_out, (mu, logvar, bern) = G(fx["video"].to(dev), fx["amps"], mode=fx["mode"])
    for k, v in (("x", x), ("vae_out", vae_out), ("mu", mu), ("logvar", logvar), ("bern", bern)):
        assert_close(v, fx[k], RTOL, "%s.%s" % (case, k))
    assert_close(kl_bern_criterion(bern), fx["kl_bern"], RTOL, case + ".kl_bern")
    loss = _loss(x, vae_out, mu, logvar, bern, fx["real"].to(dev), fx["video"].to(dev), mse_loss, kl_criterion, kl_bern_criterion)
    assert_close(loss, fx["loss"], RTOL, case + ".loss")
    loss.backward()
    for n, p in G.named_parameters():
        want = fx["grads"][n]
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, n
        else:
            assert_close(p.grad, want, RTOL, "%s.grad.%s" % (case, n), atol=bn_bias_atol(n, fx["grads"], 1e-7))
    sd = G.state_dict()
    for k, v in fx["G_after"].items():
        if k.endswith(("running_mean", "running_var", "weight_u", "weight_v")):
            assert_close(sd[k], v, RTOL, "%s.buffer.%s" % (case, k), atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["enc1x1_3d", "enc1x1_2d"])
def test_hip_encoder_1x1(case):
    import hp_vae_gan_amd  # noqa: F401
    from hp_vae_gan_amd.modules import networks_2d, networks_3d
    fx = load_golden("variants.pt")[case]
    dev = "cuda"
    nets = networks_3d if fx["dims"] == 3 else networks_2d
    opt = opt_from(fx["opt"], ker_size=3, enc_blocks=2)
    E = nets.Encode3DVAE1x1(opt, out_dim=opt.latent_dim)
    assert list(E.state_dict().keys()) == list(fx["E_init"].keys())
    E.load_state_dict(fx["E_init"])
    E.to(dev)
    x = fx["x"].to(dev).requires_grad_(True)
    mu, logvar = E(x)
    assert_close(mu, fx["mu"], RTOL, case + ".mu")
    assert_close(logvar, fx["logvar"], RTOL, case + ".logvar")
    params = dict(E.named_parameters())
    grads = torch.autograd.grad([mu, logvar], [x] + list(params.values()), [fx["gmu"].to(dev), fx["glogvar"].to(dev)])
    assert_close(grads[0], fx["dx"], RTOL, case + ".dx")
    for (n, _), g in zip(params.items(), grads[1:]):
        assert_close(g, fx["dparams"][n], RTOL, "%s.d%s" % (case, n), atol=1e-7)
