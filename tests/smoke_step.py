"""smoke(): one VAE-stage and one GAN-stage train iteration of the hot path on cuda:0, checked against the CPU oracle
fed with the same recorded noise (inputs come from the committed golden fixtures; the reference itself is not needed)."""
import torch

from helpers import assert_close, bn_bias_atol, load_golden, opt_from, oracle_state, run_hip_stage, flat_to_named
from oracle import hpvg_oracle as O


def _oracle_first_iter(fx):
    opt = opt_from(fx["opt"])
    O.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    dims, s = fx["dims"], fx["scale_idx"]
    PG = oracle_state(fx["G_init"])
    PD = oracle_state(fx["D_init"]) if fx["D_init"] is not None else None
    amps = list(fx["noise_amps_init"])
    rec = fx["iters"][0]
    noises = iter(rec["noises"])
    O.noise_amp_for_stage(PG, opt, dims, s, fx["real"], fx["real_zero"], amps, noises)
    alpha = rec["alpha"].reshape(()) if rec["alpha"] is not None else None
    return O.train_step(PG, PD, opt, dims, s, fx["real"], fx["real_zero"], rec["noise_init"], noises, alpha, amps, {}, {})


def run_smoke():
    torch.set_num_threads(1)  # the golden vectors and the oracle pinning were produced single-threaded
    for fname in ("step3d_vae_s1.pt", "step3d_gan_s3.pt"):
        fx = load_golden(fname)
        want = _oracle_first_iter(fx)
        rec, out, netG, netD, trainer = next(run_hip_stage(fx))
        torch.cuda.synchronize()
        for k in ("total_loss", "rec_vae_loss", "kl_loss", "errD_real", "errD_fake", "gradient_penalty", "rec_loss", "errG"):
            if k in want:
                assert_close(out[k], want[k], 3e-3 if k in ("errG", "total_loss") and fx["D_init"] is not None else 1e-3,
                             "smoke." + fname + "." + k)
        got = flat_to_named(out["gradG_flat"], trainer.arenaG, netG)
        # GAN stage: the G gradients flow through D AFTER its Adam step; Adam moves near-zero-gradient weights by
        # +-lr on a sign flip, so post-optimizer quantities are chaotic across CPUs/thread counts: the oracle run on
        # the GPU box's host differs from the oracle run in the build container by up to 6e-3 there (measured), while
        # the HIP path is within 1e-4 of the reference-generated golden vectors (tests/test_hip_train_step.py).
        rtol = 2e-2 if fx["D_init"] is not None else 1e-3
        for k, g in want["gradsG"].items():
            if g is not None:
                assert_close(got[k], g, rtol, "smoke." + fname + ".grad." + k, atol=bn_bias_atol(k, want["gradsG"], 1e-6))
        print("smoke ok:", fname, {k: float(v) for k, v in out.items() if torch.is_tensor(v) and v.numel() == 1})


if __name__ == "__main__":
    run_smoke()
