"""gloo tests (CPU) of the level pipeline (hp_vae_gan_amd.pipeline.LevelPipelineTrainer): contiguous pyramid levels per
rank, level outputs sent forward and their gradients back, global clip norm by one scalar all-reduce - against the
single-process oracle step on the same golden fixture.  Torch-CPU backend built on the oracle, as in test_multigpu_gloo."""
import os
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

from test_multigpu_gloo import OracleNet, TorchBackend, _check_updates, _free_port, _groups, _single_process  # noqa: E402


class TorchPipeBackend(TorchBackend):
    def g_head(self, netG, video, amps, noise_init, mode, stop):
        return netG(video, amps, noise_init=noise_init, mode=mode, stop_idx=stop)

    def g_levels(self, netG, start, x, amps, mode, stop):
        z = torch.zeros(self.opt.Z_init_size)      # the oracle's forward always runs level 0; its result is not used
        return netG(None, amps, noise_init=z, sample_init=(start, x), mode=mode, stop_idx=stop)[0]

    def level_shape(self, level, dims):
        return self.O.level_shape(level, self.opt, dims)

    def level_tensors(self, netG, level):
        pre = ("encode.", "decoder.") if level == 0 else ("body.%d." % (level - 1),)
        P = netG.P()
        params = [v for k, v in P.items() if k.startswith(pre) and self.O.is_param(k)]
        bufs = [v for k, v in P.items() if k.startswith(pre) and not self.O.is_param(k)]
        return params, bufs

    def g_optimizer(self, netG, owned, g_groups, beta1):
        groups = [{"params": [p for p in ps if id(p) in owned], "lr": lr} for ps, lr in g_groups]
        groups = [g for g in groups if g["params"]]
        adam = torch.optim.Adam(groups, betas=(beta1, 0.999)) if groups else None
        allp = list(netG.parameters())

        class _O:
            @staticmethod
            def zero():
                netG.zero_grad(set_to_none=True)

            @staticmethod
            def sqsum():
                s = torch.zeros(1)
                for p in allp:
                    if p.grad is not None and id(p) in owned_all:
                        s += (p.grad.double() ** 2).sum().float()
                return s

            @staticmethod
            def clip_step(sq_total, max_norm):
                coef = min(1.0, max_norm / (float(sq_total.sqrt()) + 1e-6))
                for p in allp:
                    if p.grad is not None:
                        p.grad.mul_(coef)
                if adam is not None:
                    adam.step()
        # gradients exist only for parameters of levels this rank ran, i.e. its own: count them all
        owned_all = set(id(p) for p in allp)
        return _O

    def d_optimizer(self, netD, lr_d, beta1):
        adam = torch.optim.Adam(list(netD.parameters()), lr=lr_d, betas=(beta1, 0.999))

        class _O:
            @staticmethod
            def zero():
                netD.zero_grad(set_to_none=True)
            step = staticmethod(adam.step)
        return _O


def _worker(rank, world, port, fname, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from helpers import NoiseFeed, load_golden, opt_from
    from hp_vae_gan_amd import pipeline
    from oracle import hpvg_oracle as O
    fx = load_golden(fname)
    opt = opt_from(fx["opt"], scale_idx=fx["scale_idx"], Noise_Amps=list(fx["noise_amps_init"]))
    O.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    dims = fx["dims"]
    netG = OracleNet(fx["G_init"], opt, dims, "G")
    netD = OracleNet(fx["D_init"], opt, dims, "D") if fx["D_init"] is not None else None
    rec = fx["iters"][0]
    opt.Z_init_size = list(rec["noise_init"].shape)
    tr = pipeline.LevelPipelineTrainer(opt, netG, netD, TorchPipeBackend(opt), _groups(opt, netG), dims=dims)
    noises = list(rec["noises"])
    gan = netD is not None
    if tr.active:
        # reference draw order: [calibration eps], rec eps, then one tensor per noisy level of the rand pass; a rank draws
        # the eps if it holds level 0 and the level noises of the levels it holds
        feed = []
        n_eps = 2 if fx["scale_idx"] > 0 and not opt.const_amp else 1
        if tr.first:
            feed += noises[:n_eps]
        level_noise = noises[n_eps:]
        if gan:
            noisy = [k for k in range(1, fx["scale_idx"] + 1) if dims == 2 or k >= opt.vae_levels]
            assert len(noisy) == len(level_noise), (noisy, len(level_noise))
            feed += [t for k, t in zip(noisy, level_noise) if tr.a <= k <= tr.b]
        netG.noise_source = NoiseFeed(feed, "cpu")
    out = tr.step(fx["real"], fx["real_zero"], noise_init=rec["noise_init"], alpha=rec["alpha"])
    tr.broadcast_levels()
    torch.save({"out": out, "amps": opt.Noise_Amps, "parts": tr.parts,
                "G": {k: v.detach().clone() for k, v in netG.P().items()},
                "D": {k: v.detach().clone() for k, v in netD.P().items()} if gan else None},
               os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fname,world", [("step3d_gan_s3.pt", 2), ("step3d_gan_s3.pt", 3), ("step3d_gan_s3_td2.pt", 4),
                                         ("step3d_gan_s2_all.pt", 3), ("step2d_gan_s2.pt", 2), ("step3d_vae_s1.pt", 2),
                                         # BASELINE configs[3]: 8 pyramid levels, one per rank on 8 ranks
                                         ("step3d_gan_s7.pt", 8), ("step3d_gan_s7.pt", 5)])
def test_level_pipeline_matches_single_process(fname, world):
    from helpers import assert_close
    from oracle import hpvg_oracle as O
    fx, want, PG, PD, amps = _single_process(fname)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), fname, d), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(world)]
    assert len(got[0]["parts"]) == min(world, fx["scale_idx"] + 1)
    if world == fx["scale_idx"] + 1:
        assert [tuple(p) for p in got[0]["parts"]] == [(k, k) for k in range(world)], "one pyramid level per rank"
    for r in range(world):
        assert got[r]["amps"] == pytest.approx(amps, rel=1e-5)
        for k in ("errD_real", "errD_fake", "gradient_penalty", "errG", "rec_loss", "rec_vae_loss", "kl_loss"):
            if k in want and k in got[r]["out"]:
                assert_close(torch.tensor(got[r]["out"][k]), want[k].float().reshape(()), 2e-4, "%s.rank%d.%s" % (fname, r, k))
        # after broadcast_levels every rank holds the owners' parameters
        _check_updates("%s.rank%d" % (fname, r), fx, got[r], PG, PD)


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: the SinGAN-3D baseline (GeneratorSG) with one stage per rank
class OracleSG(torch.nn.Module):
    """GeneratorSG stand-in whose forward is the oracle's restatement (supports the pipeline's stage ranges)."""

    def __init__(self, sd, opt):
        super().__init__()
        from oracle import hpvg_oracle as O
        self.O, self.opt = O, opt
        self.keys = list(sd.keys())
        for k, v in sd.items():
            name = k.replace(".", "__")
            if O.is_param(k):
                self.register_parameter(name, torch.nn.Parameter(v.clone()))
            else:
                self.register_buffer(name, v.clone())
        self.noise_source = None
        nb = O.num_body(sd)

        class _Block:
            def __init__(blk, k):
                blk.k = k

            def parameters(blk):
                return [getattr(self, key.replace(".", "__")) for key in self.keys if key.startswith("body.%d." % blk.k) and O.is_param(key)]
        self.body = [_Block(k) for k in range(nb)]

    def P(self):
        return {k: getattr(self, k.replace(".", "__")) for k in self.keys}

    def forward(self, x, noise_amp, mode="rand", start=0, stop=None):
        src = self.noise_source
        return self.O.generator_sg_forward(self.P(), self.opt, x, noise_amp, mode, lambda shape: src(torch.empty(shape)), start, stop)


class TorchBaselineBackend(TorchPipeBackend):
    def sg_levels(self, netG, x, amps, mode, start, stop):
        return netG(x, amps, mode=mode, start=start, stop=stop)

    def level_tensors(self, netG, level):
        P = netG.P()
        pre = "body.%d." % level
        return ([v for k, v in P.items() if k.startswith(pre) and self.O.is_param(k)],
                [v for k, v in P.items() if k.startswith(pre) and not self.O.is_param(k)])

    def g_optimizer(self, netG, owned, g_groups, beta1):
        groups = [{"params": [p for p in ps if id(p) in owned], "lr": lr} for ps, lr in g_groups]
        groups = [g for g in groups if g["params"]]
        adam = torch.optim.Adam(groups, betas=(beta1, 0.999)) if groups else None

        class _O:
            @staticmethod
            def zero():
                netG.zero_grad(set_to_none=True)

            @staticmethod
            def step():
                if adam is not None:
                    adam.step()
        return _O


def _baseline_worker(rank, world, port, fname, outdir, train_depth):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from helpers import NoiseFeed, load_golden, opt_from
    from hp_vae_gan_amd import pipeline
    from oracle import hpvg_oracle as O
    fx = load_golden(fname)
    opt = opt_from(fx["opt"], scale_idx=fx["scale_idx"], Noise_Amps=list(fx["noise_amps_init"]), train_depth=train_depth)
    O.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    opt.Z_init = fx["Z_init"]
    netG = OracleSG(fx["G_init"], opt)
    netD = OracleNet(fx["D_init"], opt, 3, "D")
    tr = pipeline.BaselinePipelineTrainer(opt, netG, netD, TorchBaselineBackend(opt))
    rec = fx["iters"][0]
    if tr.active:
        # draw order of the rand pass: one padded-volume noise per stage >= 1; a rank draws those of the stages it holds
        netG.noise_source = NoiseFeed([t for k, t in enumerate(rec["noises"], 1) if tr.a <= k <= tr.b], "cpu")
    out = tr.step(fx["real"], noise_init=rec["noise_init"], alphas=rec["alphas"])
    tr.broadcast_levels()
    torch.save({"out": out, "amps": opt.Noise_Amps, "parts": tr.parts,
                "G": {k: v.detach().clone() for k, v in netG.P().items()},
                "D": {k: v.detach().clone() for k, v in netD.P().items()}}, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,train_depth", [(8, 1), (3, 1), (8, 3)])
def test_baseline_pipeline_matches_single_process(world, train_depth):
    """train_video_baselines.py's iteration with GeneratorSG's 8 stages over the ranks (one stage per rank at world 8;
    train_depth 3: trained stages on three ranks, gradients handed down) == the single-process oracle step."""
    from helpers import _bn_fed_bias, compare_update, load_golden, opt_from, oracle_state
    from oracle import hpvg_oracle as O
    fname = "baseline3d_sg_s7.pt"
    fx = load_golden(fname)
    opt = opt_from(fx["opt"], train_depth=train_depth)
    O.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    s = fx["scale_idx"]
    rec = fx["iters"][0]
    PG, PD = oracle_state(fx["G_init"]), oracle_state(fx["D_init"])
    amps = list(fx["noise_amps_init"]) + [0]
    with torch.no_grad():
        z = O.generator_sg_forward(PG, opt, fx["Z_init"], amps, "rec", None)
    amps[-1] = opt.noise_amp_init * float(torch.sqrt(O.mse(fx["real"], z))) / opt.batch_size
    want = O.baseline_train_step(PG, PD, opt, s, fx["real"], fx["Z_init"], rec["noise_init"], iter(rec["noises"]),
                                 [a.reshape(()) for a in rec["alphas"]], amps, {}, {})
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_baseline_worker, args=(world, _free_port(), fname, d, train_depth), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(world)]
    assert len(got[0]["parts"]) == min(world, s + 1)
    if world == s + 1:
        assert [tuple(p) for p in got[0]["parts"]] == [(k, k) for k in range(world)], "one stage per rank"
    groups = O.baseline_g_groups(PG, opt, s)
    names = set(k for k in PG if O.is_param(k))
    for r in range(world):
        assert got[r]["amps"] == pytest.approx(amps, rel=1e-5)
        for k in ("errD_real", "errD_fake", "gradient_penalty", "errG", "rec_loss"):
            assert got[r]["out"][k] == pytest.approx(float(want[k]), rel=2e-4, abs=1e-7), (r, k)
        for k, v in PG.items():
            if O.is_param(k):
                lr = next((l for pre, l in groups if k.startswith(pre)), None)
                compare_update("baseline.rank%d.G.%s" % (r, k), fx["G_init"][k], v.detach(), got[r]["G"][k], lr, 0.0, _bn_fed_bias(k, names))
            elif not k.endswith("num_batches_tracked"):
                torch.testing.assert_close(got[r]["G"][k], v.detach(), rtol=1e-4, atol=1e-6)
        for k, v in PD.items():
            if O.is_param(k):
                compare_update("baseline.rank%d.D.%s" % (r, k), fx["D_init"][k], v.detach(), got[r]["D"][k], opt.lr_d)
            else:
                torch.testing.assert_close(got[r]["D"][k], v.detach(), rtol=1e-4, atol=1e-6)
