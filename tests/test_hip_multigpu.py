"""GPU rehearsal of the N>1 path with the real kernels: two (four, in quad mode) processes share the box's single MI355X
(gloo rendezvous, messages staged through the host) and run hp_vae_gan_amd.multigpu.DistStageTrainer with the HIP backend on a golden
GAN-stage fixture; both replicas must land on the reference's post-step parameters."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, fname, outdir, quad=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import NoiseFeed, hip_opt, load_golden
    from hp_vae_gan_amd import multigpu, train as hp_train
    from hp_vae_gan_amd.modules import networks_2d, networks_3d
    fx = load_golden(fname)
    dims, s = fx["dims"], fx["scale_idx"]
    dev = "cuda"
    opt = hip_opt(fx["opt"], dims, s, dev)
    nets = networks_3d if dims == 3 else networks_2d
    netG = nets.GeneratorHPVAEGAN(opt)
    for _ in range(s):
        netG.init_next_stage()
    netG.load_state_dict(fx["G_init"])
    netG.to(dev)
    netD = getattr(nets, opt.discriminator)(opt)
    netD.load_state_dict(fx["D_init"])
    netD.to(dev)
    opt.Noise_Amps = list(fx["noise_amps_init"])
    rec = fx["iters"][0]
    opt.Z_init_size = list(rec["noise_init"].shape)
    tr = multigpu.DistStageTrainer(opt, netG, netD, multigpu.HipBackend(opt), hp_train.generator_param_groups(opt, netG), quad=quad)
    noises = rec["noises"]
    if quad:  # rec ranks {0, 1}: calibration eps + rec eps, rand ranks {2, 3}: the level noises
        netG.noise_source = NoiseFeed(noises[:2] if rank < 2 else noises[2:], dev)
    else:
        netG.noise_source = NoiseFeed(noises[:2] if rank == 0 else noises[2:], dev)
    out = tr.step(fx["real"].to(dev), fx["real_zero"].to(dev), noise_init=rec["noise_init"].to(dev), alpha=rec["alpha"])
    torch.cuda.synchronize()
    torch.save({"out": {k: v.cpu() for k, v in out.items()}, "amps": opt.Noise_Amps,
                "G": {k: v.detach().cpu() for k, v in netG.state_dict().items()},
                "D": {k: v.detach().cpu() for k, v in netD.state_dict().items()}}, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fname", ["step3d_gan_s3.pt", "step2d_gan_s2.pt"])
def test_two_rank_hip_step_matches_reference(fname):
    from helpers import assert_close, load_golden
    from oracle import hpvg_oracle as O
    fx = load_golden(fname)
    rec = fx["iters"][0]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(2, _free_port(), fname, d), nprocs=2, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(2)]
    lr = fx["opt"]["lr_g"]
    for r in range(2):
        assert got[r]["amps"] == pytest.approx(rec["noise_amps"], rel=1e-4)
        for k in ("errD_real", "errD_fake", "gradient_penalty"):
            assert_close(got[r]["out"][k], rec[k], 1e-3, "%s.rank%d.%s" % (fname, r, k))
        assert_close(got[r]["out"]["errG"], rec["errG"], 3e-3, "%s.rank%d.errG" % (fname, r))
        for k, v in rec["G_after"].items():
            if O.is_param(k):
                assert_close(got[r]["G"][k], v, 1e-3, "%s.rank%d.G.%s" % (fname, r, k), atol=2 * lr)
        for k, v in rec["D_after"].items():
            if O.is_param(k) or k.endswith(("weight_u", "weight_v")):
                assert_close(got[r]["D"][k], v, 1e-3, "%s.rank%d.D.%s" % (fname, r, k), atol=2 * lr)
    for k, v in got[0]["G"].items():
        if O.is_param(k):
            assert torch.equal(v, got[1]["G"][k]), "replicas diverged: " + k


@pytest.mark.parametrize("fname", ["step3d_gan_s3.pt", "step2d_gan_s2.pt"])
def test_four_rank_hip_step_matches_reference(fname):
    """Quad mode with the real kernels (ops.BNActSync inside the pass pairs, spectral-norm replay): every replica lands on
    the reference's post-step parameters and D u/v buffers."""
    from helpers import assert_close, load_golden
    from oracle import hpvg_oracle as O
    fx = load_golden(fname)
    rec = fx["iters"][0]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(4, _free_port(), fname, d, True), nprocs=4, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(4)]
    lr = fx["opt"]["lr_g"]
    for r in range(4):
        assert got[r]["amps"] == pytest.approx(rec["noise_amps"], rel=1e-4)
        for k in ("errD_real", "errD_fake", "gradient_penalty", "rec_loss"):
            assert_close(got[r]["out"][k], rec[k], 1e-3, "%s.rank%d.%s" % (fname, r, k))
        assert_close(got[r]["out"]["errG"], rec["errG"], 3e-3, "%s.rank%d.errG" % (fname, r))
        for k, v in rec["G_after"].items():
            if O.is_param(k):
                assert_close(got[r]["G"][k], v, 1e-3, "%s.rank%d.G.%s" % (fname, r, k), atol=2 * lr)
        for k, v in rec["D_after"].items():
            if O.is_param(k) or k.endswith(("weight_u", "weight_v")):
                assert_close(got[r]["D"][k], v, 1e-3, "%s.rank%d.D.%s" % (fname, r, k), atol=2 * lr)
    for r in range(1, 4):
        for k, v in got[0]["G"].items():
            if O.is_param(k):
                assert torch.equal(v, got[r]["G"][k]), "replicas diverged: " + k


def _slab_worker(rank, world, port, fname, outdir):
    """Both ranks evaluate generator pass + MSE backward and D(real) + gradient penalty backward twice with the gfx950
    kernels: on the whole image, then on their row slab (halo swaps, BatchNorm sums and level re-assembly over the two
    ranks); the slab results must be the rows / the rank-summed gradients of the whole-image results."""
    import copy
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import NoiseFeed, hip_opt, load_golden
    from hp_vae_gan_amd import multigpu
    from hp_vae_gan_amd.modules import networks_2d, networks_3d
    from hp_vae_gan_amd.slab import Halo, SlabPlan, slab_rows
    fx = load_golden(fname)
    dims, s = fx["dims"], fx["scale_idx"]
    dev = "cuda"
    opt = hip_opt(fx["opt"], dims, s, dev)
    nets = networks_3d if dims == 3 else networks_2d
    netG = nets.GeneratorHPVAEGAN(opt)
    for _ in range(s):
        netG.init_next_stage()
    netG.load_state_dict(fx["G_init"])
    netG.to(dev)
    netD = getattr(nets, opt.discriminator)(opt)
    netD.load_state_dict(fx["D_init"])
    netD.to(dev)
    amps = list(fx["iters"][0]["noise_amps"])
    rec = fx["iters"][0]
    be = multigpu.HipBackend(opt)
    real = fx["real"].to(dev)
    z = rec["noise_init"].to(dev)
    alpha = rec["alpha"].reshape(1).to(dev, torch.float32)
    H = real.shape[-2]
    r0, r1 = slab_rows(H, rank, 2)
    state = (copy.deepcopy(netG.state_dict()), copy.deepcopy(netD.state_dict()))

    def run(slab):
        netG.load_state_dict(state[0])
        netD.load_state_dict(state[1])
        netG.zero_grad()
        netD.zero_grad()
        netG.noise_source = NoiseFeed(rec["noises"][2:], dev)
        rows = (r0, r1) if slab else (0, H)
        frac = (rows[1] - rows[0]) / H
        real_s = real.narrow(real.dim() - 2, rows[0], rows[1] - rows[0]).contiguous()
        fake, _ = netG(z, amps, noise_init=z, mode="rand")
        (be.mse(fake, real_s) * frac).backward()
        fake_d = fake.detach().contiguous()
        errD = be.wgan_mean(netD(real_s), -1.0) * frac + be.grad_penalty(netD, real_s, fake_d, opt.lambda_grad, alpha) * frac
        errD.backward()
        gG = {k: p.grad.detach().clone() for k, p in netG.named_parameters() if p.grad is not None}
        gD = {k: p.grad.detach().clone() for k, p in netD.named_parameters() if p.grad is not None}
        return fake_d, errD.detach().clone(), gG, gD

    whole = run(False)
    swap = multigpu.pair_swap(1 - rank)
    plan = SlabPlan(rank, 2, max(opt.vae_levels, s - 1), Halo(up=swap if rank == 1 else None, down=swap if rank == 0 else None),
                    multigpu.all_reduce)

    def level_sync(level):
        from hp_vae_gan_amd import utils as hu
        size = hu.images.level_shape_3d(level, opt) if dims == 3 else hu.images.level_shape_2d(level, opt)
        n = opt.batch_size
        for d in size:
            n *= int(d)
        return (multigpu.all_reduce, 2, n)
    be.set_slab(netG, netD, plan, level_sync)
    part = run(True)
    be.set_slab(netG, netD, None)
    be.set_sync_bn(netG, None)
    errD = part[1].clone()
    multigpu.all_reduce(errD)
    for g in (part[2], part[3]):
        for k in g:
            multigpu.all_reduce(g[k])
    torch.cuda.synchronize()
    torch.save({"fake_whole": whole[0][..., r0:r1, :].cpu(), "fake_slab": part[0].cpu(), "errD": (whole[1].cpu(), errD.cpu()),
                "gG": ({k: v.cpu() for k, v in whole[2].items()}, {k: v.cpu() for k, v in part[2].items()}),
                "gD": ({k: v.cpu() for k, v in whole[3].items()}, {k: v.cpu() for k, v in part[3].items()})},
               os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fname", ["step3d_gan_s3.pt", "step2d_gan_s2.pt"])
def test_two_rank_row_slabs_match_whole_image(fname):
    """slab.py with the real kernels (the 8-rank oct mode cannot be rehearsed on a one-GPU box: at most 6 processes):
    generator pass, discriminator, first- and second-order backward on two row slabs == the whole image."""
    from helpers import assert_close, bn_bias_atol
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_slab_worker, args=(2, _free_port(), fname, d), nprocs=2, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(2)]
    for r in range(2):
        assert_close(got[r]["fake_slab"], got[r]["fake_whole"], 1e-3, "%s.rank%d.fake" % (fname, r))
        assert_close(got[r]["errD"][1], got[r]["errD"][0], 1e-3, "%s.rank%d.errD" % (fname, r))
        for which in ("gG", "gD"):
            whole, part = got[r][which]
            assert set(whole) == set(part)
            for k in whole:
                assert_close(part[k], whole[k], 1e-3, "%s.rank%d.%s.%s" % (fname, r, which, k), atol=bn_bias_atol(k, whole))


def _pipe_worker(rank, world, port, fname, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import NoiseFeed, hip_opt, load_golden
    from hp_vae_gan_amd import pipeline, train as hp_train
    from hp_vae_gan_amd.modules import networks_2d, networks_3d
    fx = load_golden(fname)
    dims, s = fx["dims"], fx["scale_idx"]
    dev = "cuda"
    opt = hip_opt(fx["opt"], dims, s, dev)
    nets = networks_3d if dims == 3 else networks_2d
    netG = nets.GeneratorHPVAEGAN(opt)
    for _ in range(s):
        netG.init_next_stage()
    netG.load_state_dict(fx["G_init"])
    netG.to(dev)
    netD = getattr(nets, opt.discriminator)(opt)
    netD.load_state_dict(fx["D_init"])
    netD.to(dev)
    opt.Noise_Amps = list(fx["noise_amps_init"])
    rec = fx["iters"][0]
    opt.Z_init_size = list(rec["noise_init"].shape)
    tr = pipeline.LevelPipelineTrainer(opt, netG, netD, pipeline.HipPipeBackend(opt), hp_train.generator_param_groups(opt, netG), dims=dims)
    noises = list(rec["noises"])
    feed = list(noises[:2]) if tr.first else []
    noisy = [k for k in range(1, s + 1) if dims == 2 or k >= opt.vae_levels]
    feed += [t for k, t in zip(noisy, noises[2:]) if tr.a <= k <= tr.b]
    netG.noise_source = NoiseFeed(feed, dev)
    out = tr.step(fx["real"].to(dev), fx["real_zero"].to(dev), noise_init=rec["noise_init"].to(dev), alpha=rec["alpha"])
    tr.broadcast_levels()
    torch.cuda.synchronize()
    torch.save({"out": out, "amps": opt.Noise_Amps, "parts": tr.parts,
                "G": {k: v.detach().cpu() for k, v in netG.state_dict().items()},
                "D": {k: v.detach().cpu() for k, v in netD.state_dict().items()}}, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("fname,world", [("step3d_gan_s3.pt", 2), ("step3d_gan_s3_td2.pt", 3), ("step2d_gan_s2.pt", 2),
                                         # the 8-level pyramid of BASELINE configs[3] on 5 processes (a box admits 6 on its
                                         # card, and the test runner itself holds the GPU)
                                         ("step3d_gan_s7.pt", 5)])
def test_level_pipeline_hip_matches_reference(fname, world):
    """pipeline.LevelPipelineTrainer with the real kernels: levels spread over 2 / 3 processes (sharing the box's GPU),
    level outputs forward, their gradients back, global clip norm - the reference's post-step parameters on every rank."""
    from helpers import assert_close, load_golden
    from oracle import hpvg_oracle as O
    fx = load_golden(fname)
    rec = fx["iters"][0]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_pipe_worker, args=(world, _free_port(), fname, d), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(world)]
    lr = fx["opt"]["lr_g"]
    assert len(got[0]["parts"]) == world
    for r in range(world):
        assert got[r]["amps"] == pytest.approx(rec["noise_amps"], rel=1e-4)
        for k in ("errD_real", "errD_fake", "gradient_penalty", "rec_loss"):
            assert_close(torch.tensor(got[r]["out"][k]), rec[k].float().reshape(()), 1e-3, "%s.rank%d.%s" % (fname, r, k))
        assert_close(torch.tensor(got[r]["out"]["errG"]), rec["errG"].float().reshape(()), 3e-3, "%s.rank%d.errG" % (fname, r))
        for k, v in rec["G_after"].items():
            if O.is_param(k):
                assert_close(got[r]["G"][k], v, 1e-3, "%s.rank%d.G.%s" % (fname, r, k), atol=2 * lr)
        for k, v in rec["D_after"].items():
            if O.is_param(k) or k.endswith(("weight_u", "weight_v")):
                assert_close(got[r]["D"][k], v, 1e-3, "%s.rank%d.D.%s" % (fname, r, k), atol=2 * lr)



def _baseline_pipe_worker(rank, world, port, fname, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from helpers import NoiseFeed, hip_opt, load_golden
    from hp_vae_gan_amd import pipeline
    from hp_vae_gan_amd.modules import networks_3d
    fx = load_golden(fname)
    s = fx["scale_idx"]
    dev = "cuda"
    opt = hip_opt(fx["opt"], 3, s, dev)
    netG = networks_3d.GeneratorSG(opt)
    for _ in range(s):
        netG.init_next_stage()
    netG.load_state_dict(fx["G_init"])
    netG.to(dev)
    netD = networks_3d.WDiscriminator3D(opt)
    netD.load_state_dict(fx["D_init"])
    netD.to(dev)
    opt.Noise_Amps = list(fx["noise_amps_init"])
    opt.Z_init = fx["Z_init"].to(dev)
    rec = fx["iters"][0]
    tr = pipeline.BaselinePipelineTrainer(opt, netG, netD, pipeline.HipBaselinePipeBackend(opt))
    netG.noise_source = NoiseFeed([t for k, t in enumerate(rec["noises"], 1) if tr.a <= k <= tr.b], dev)
    out = tr.step(fx["real"].to(dev), noise_init=rec["noise_init"].to(dev), alphas=rec["alphas"])
    tr.broadcast_levels()
    torch.cuda.synchronize()
    torch.save({"out": out, "amps": opt.Noise_Amps, "parts": tr.parts,
                "G": {k: v.detach().cpu() for k, v in netG.state_dict().items()},
                "D": {k: v.detach().cpu() for k, v in netD.state_dict().items()}}, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 5])
def test_baseline_stage_pipeline_hip_matches_reference(world):
    """BASELINE configs[4] with the real kernels: pipeline.BaselinePipelineTrainer spreads GeneratorSG's 8 stages over 2 / 5
    processes (sharing the box's GPU; frozen stages forward-only, the newest stage + critic on the last rank) and must land
    on the reference's losses and post-step parameters on every rank (fixture baseline3d_sg_s7.pt)."""
    from helpers import _bn_fed_bias, compare_update, load_golden
    fname = "baseline3d_sg_s7.pt"
    fx = load_golden(fname)
    rec, spread = fx["iters"][0], fx["spread"][0]
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_baseline_pipe_worker, args=(world, _free_port(), fname, d), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(world)]
    s = fx["scale_idx"]
    lr_g, lr_d = fx["opt"]["lr_g"], fx["opt"]["lr_d"]
    assert len(got[0]["parts"]) == world
    pnames = set(k for k in rec["G_after"] if k.endswith((".weight", ".bias")))
    for r in range(world):
        assert got[r]["amps"] == pytest.approx(rec["noise_amps"], rel=1e-4)
        for k in ("errD_real", "errD_fake", "gradient_penalty", "errG", "rec_loss"):
            want = float(rec[k])
            assert got[r]["out"][k] == pytest.approx(want, rel=1e-3, abs=max(1e-7, 2 * spread.get(k, 0.0))), (r, k)
        for k, v in rec["G_after"].items():
            if k in pnames:
                lr = lr_g if k.startswith("body.%d." % s) else None      # train_depth 1: only the newest stage moves
                compare_update("rank%d.G.%s" % (r, k), fx["G_init"][k], v, got[r]["G"][k], lr, 0.0, _bn_fed_bias(k, pnames))
        for k, v in rec["D_after"].items():
            if k.endswith((".weight", ".bias", ".weight_orig")):
                compare_update("rank%d.D.%s" % (r, k), fx["D_init"][k], v, got[r]["D"][k], lr_d)


def _bcast_pack_worker(rank, world, port, outdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import types
    from hp_vae_gan_amd import multigpu
    from hp_vae_gan_amd.modules import _nets
    torch.manual_seed(10 + rank)                      # different weights on the two ranks
    conv = _nets.Conv(3, 8, 8).to("cuda")
    x = torch.randn(1, 8, 3, 6, 7, generator=torch.Generator().manual_seed(5)).to("cuda")
    with torch.no_grad():
        before = conv(x).cpu()                        # packs this rank's weight (ops.pack_weight cache)
        multigpu.broadcast_module(conv, src=0)        # overwrites the parameters through .data: no version counter moves
        after = conv(x).cpu()
    torch.save({"before": before, "after": after}, os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_module_drops_packed_weights():
    """Parameters written through `.data` (multigpu.broadcast_module, pipeline.broadcast_levels) leave torch's version
    counter alone; the conv must not keep using the weights it packed before the broadcast."""
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_bcast_pack_worker, args=(2, _free_port(), d), nprocs=2, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(2)]
    assert not torch.allclose(got[0]["before"], got[1]["before"])
    assert torch.equal(got[0]["after"], got[0]["before"])
    assert torch.equal(got[1]["after"], got[0]["after"]), "rank 1 still convolves with its pre-broadcast packed weights"
