"""world_size-2/3 (pair mode) and 4/5 (quad mode) gloo tests of the multi-GPU step logic (hp_vae_gan_amd.multigpu.DistStageTrainer) on CPU.

The distributed logic is backend-agnostic; here it is driven with a torch-CPU backend built on the oracle (tests may
use the oracle) and compared with the single-process oracle train step on the same golden fixture: same losses, and
post-step G / D parameters equal up to Adam's +-lr sign-flip slack.  Rendezvous on 127.0.0.1."""
import os
import socket
import sys
import tempfile

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

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


class OracleNet(nn.Module):
    """torch-CPU network whose forward is the oracle's functional restatement (test-only compute engine)."""

    def __init__(self, sd, opt, dims, kind):
        super().__init__()
        from oracle import hpvg_oracle as O
        self.O, self.opt, self.dims, self.kind = O, opt, dims, kind
        self.keys = list(sd.keys())
        for k, v in sd.items():
            name = k.replace(".", "__")
            if O.is_param(k):
                self.register_parameter(name, nn.Parameter(v.clone()))
            else:
                self.register_buffer(name, v.clone())
        self.noise_source = None
        self.halo = None    # D: slab.Halo while the discriminator runs on row slabs
        self.slab = None    # G: (slab.SlabPlan, level_sync, backend) while the upper levels run on row slabs

    def P(self):
        return {k: getattr(self, k.replace(".", "__")) for k in self.keys}

    def forward(self, x, noise_amp=None, noise_init=None, sample_init=None, mode="rand", stop_idx=None):
        O = self.O
        if self.kind == "D":
            if self.halo is None:
                return O.discriminator_forward(x, self.P(), self.opt)
            with _halo_convs(O, self.halo):
                return O.discriminator_forward(x, self.P(), self.opt)
        src = self.noise_source
        level_fn = None
        if self.slab is not None:
            plan, level_sync, be = self.slab
            nbody = O.num_body(self.P())

            def level_fn(idx, inp, up, f):
                if not plan.covers(idx + 1):
                    return f(inp, up)
                H = up.shape[-2]
                whole_batch_sync = be._sync
                be.set_sync_bn(None, level_sync(idx + 1))
                try:
                    with _halo_convs(O, plan.halo):
                        y = f(plan.cut(inp), plan.cut(up))
                finally:
                    be.set_sync_bn(None, whole_batch_sync)
                return plan.gather(y, H) if idx + 1 < nbody else y
        return O.generator_forward(self.P(), self.opt, self.dims, x, noise_amp, noise_init=noise_init, mode=mode,
                                   noises=lambda shape: src(torch.empty(shape)), level_fn=level_fn, sample_init=sample_init,
                                   stop=stop_idx)


class _halo_convs:
    """While active, every oracle convolution runs on a row slab with the neighbour's boundary rows (slab.conv_with_halo
    around the oracle's own conv)."""

    def __init__(self, O, halo):
        self.O, self.halo = O, halo

    def __enter__(self):
        from hp_vae_gan_amd.slab import conv_with_halo
        O, halo = self.O, self.halo
        self.plain = plain = O.conv
        O.conv = lambda x, w, b=None: conv_with_halo(x, halo, lambda xe: plain(xe, w, b))

    def __exit__(self, *exc):
        self.O.conv = self.plain


class TorchBackend:
    def __init__(self, opt):
        self.opt = opt
        from oracle import hpvg_oracle as O
        self.O = O
        self._sync = None

    def mse(self, a, b):
        return self.O.mse(a, b)

    def kl(self, mu, lv):
        return self.O.kl_criterion(mu, lv)

    def wgan_mean(self, x, sign):
        return sign * x.mean()

    def grad_penalty(self, netD, real, fake, lam, alpha):
        a = alpha.reshape(())
        xhat = (a * real + (1 - a) * fake).detach().requires_grad_(True)
        out = netD(xhat)
        g = torch.autograd.grad(out, xhat, torch.ones_like(out), create_graph=True, retain_graph=True)[0]
        return ((g.norm(2, dim=1) - 1) ** 2).mean() * lam

    def noise(self, ref):
        return torch.randn(ref.shape)

    def advance_sn(self, net, n):
        P = net.P()
        with torch.no_grad():
            for _ in range(n):
                for k in P:
                    if k.endswith("weight_orig"):
                        self.O.spectral_norm_weight(P[k], P[k[:-4] + "u"], P[k[:-4] + "v"], training=True)

    def set_sync_bn(self, netG, sync):
        """Batch-split BatchNorm for the oracle-backed stand-in: the oracle's batch_norm_train with the per-channel sums
        exchanged through a differentiable all-reduce (test infrastructure; the product path is ops.BNActSync)."""
        O = self.O
        self._sync = sync
        if not hasattr(O, "_bn_whole_batch"):
            O._bn_whole_batch = O.batch_norm_train
        if sync is None:
            O.batch_norm_train = O._bn_whole_batch
            return
        allreduce, nranks = sync[0], sync[1]
        total = sync[2] if len(sync) > 2 else None

        class Sum(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                y = x.detach().clone()
                allreduce(y)
                return y

            @staticmethod
            def backward(ctx, g):
                g = g.detach().clone()
                allreduce(g)
                return g

        def bn(x, gamma, beta, running_mean=None, running_var=None):
            dimsr = [0] + list(range(2, x.dim()))
            shape = (1, -1) + (1,) * (x.dim() - 2)
            n = total if total is not None else (x.numel() // x.shape[1]) * nranks
            xd = x.double()
            mean = Sum.apply(xd.sum(dim=dimsr)) / n
            var = Sum.apply(((xd - mean.view(shape)) ** 2).sum(dim=dimsr)) / n
            y = ((xd - mean.view(shape)) / torch.sqrt(var.view(shape) + O.BN_EPS)).float() * gamma.view(shape) + beta.view(shape)
            if running_mean is not None:
                with torch.no_grad():
                    running_mean.mul_(1 - O.BN_MOMENTUM).add_(O.BN_MOMENTUM * mean.detach().float())
                    running_var.mul_(1 - O.BN_MOMENTUM).add_(O.BN_MOMENTUM * var.detach().float() * (n / max(n - 1, 1)))
            return y
        O.batch_norm_train = bn

    def set_slab(self, netG, netD, plan, level_sync=None):
        netG.slab = (plan, level_sync, self) if plan is not None else None
        if netD is not None:
            netD.halo = plan.halo if plan is not None else None

    def optimizers(self, netG, netD, g_groups, lr_d, beta1):
        class _O:
            pass
        o = _O()
        gparams = list(netG.parameters())
        o.optG = torch.optim.Adam([{"params": list(p), "lr": lr} for p, lr in g_groups], betas=(beta1, 0.999))

        def ar(params, group):
            for p in params:
                if p.grad is None:
                    p.grad = torch.zeros_like(p)
                dist.all_reduce(p.grad, group=group)
        o.allreduce_G = lambda group=None: ar(gparams, group)
        o.zero_G = lambda: netG.zero_grad(set_to_none=True)

        def clip_step(max_norm):
            torch.nn.utils.clip_grad_norm_(gparams, max_norm)
            o.optG.step()
        o.clip_step_G = clip_step
        if netD is not None:
            dparams = list(netD.parameters())
            o.optD = torch.optim.Adam(dparams, lr=lr_d, betas=(beta1, 0.999))
            o.allreduce_D = lambda group=None: ar(dparams, group)
            o.zero_D = lambda: netD.zero_grad(set_to_none=True)
            o.step_D = o.optD.step
        return o


def _groups(opt, netG):
    """Adam groups of the generator as (parameter list, lr), from the oracle's rule table."""
    from oracle import hpvg_oracle as O
    named = dict(netG.named_parameters())
    out = []
    for prefix, lr in O.g_param_groups(netG.P(), opt, opt.scale_idx):
        out.append(([p for k, p in named.items() if k.replace("__", ".").startswith(prefix)], lr))
    return out


def _worker(rank, world, port, fname, outdir, quad=False, slabs=1, slab_levels=2):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    from helpers import NoiseFeed, load_golden, opt_from
    from hp_vae_gan_amd import multigpu
    from oracle import hpvg_oracle as O
    fx = load_golden(fname)
    opt = opt_from(fx["opt"], scale_idx=fx["scale_idx"], Noise_Amps=list(fx["noise_amps_init"]))
    O.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    dims = fx["dims"]
    netG = OracleNet(fx["G_init"], opt, dims, "G")
    netD = OracleNet(fx["D_init"], opt, dims, "D") if fx["D_init"] is not None else None
    opt.Z_init_size = list(fx["iters"][0]["noise_init"].shape)
    tr = multigpu.DistStageTrainer(opt, netG, netD, TorchBackend(opt), _groups(opt, netG), quad=quad, slabs=slabs,
                                   slab_levels=slab_levels)
    assert tr.nh == slabs and tr.quad == bool(quad)
    rec = fx["iters"][0]
    gan = netD is not None
    noises = rec["noises"]
    if gan and quad:
        # every rec rank repeats the calibration pass (first eps) and draws the rec eps; rand ranks draw the level noises
        netG.noise_source = NoiseFeed(noises[:2] if rank < 2 * slabs else noises[2:], "cpu")
    elif gan:
        # reference draw order: [calibration eps], rec eps, then the level noises of the rand pass
        netG.noise_source = NoiseFeed(noises[:2] if rank == 0 else noises[2:], "cpu")
    else:
        netG.noise_source = NoiseFeed(noises, "cpu")
        tr._bcast_noise = netG.noise_source  # identical recorded noise on every rank (what the broadcast guarantees)
    alpha = rec["alpha"] if rec["alpha"] is not None else None
    out = tr.step(fx["real"], fx["real_zero"], noise_init=rec["noise_init"], alpha=alpha)
    tr.sync_buffers()
    if rank < (4 * slabs if quad else 2):
        torch.save({"out": {k: v for k, v in out.items()}, "amps": opt.Noise_Amps,
                    "G": {k: v.detach().clone() for k, v in netG.P().items()},
                    "D": {k: v.detach().clone() for k, v in netD.P().items()} if gan else None},
                   os.path.join(outdir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _single_process(fname):
    from helpers import load_golden, opt_from, oracle_state
    from oracle import hpvg_oracle as O
    fx = load_golden(fname)
    opt = opt_from(fx["opt"])
    O.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    PG = oracle_state(fx["G_init"])
    PD = oracle_state(fx["D_init"]) if fx["D_init"] is not None else None
    amps = list(fx["noise_amps_init"])
    rec = fx["iters"][0]
    noises = iter(rec["noises"])
    O.noise_amp_for_stage(PG, opt, fx["dims"], fx["scale_idx"], fx["real"], fx["real_zero"], amps, noises)
    alpha = rec["alpha"].reshape(()) if rec["alpha"] is not None else None
    out = O.train_step(PG, PD, opt, fx["dims"], fx["scale_idx"], fx["real"], fx["real_zero"], rec["noise_init"], noises, alpha, amps, {}, {})
    return fx, out, PG, PD, amps


def _check_updates(tag, fx, got, PG, PD):
    """Optimizer steps are judged by their UPDATE against the single-process step's (helpers.compare_update): a parameter the
    single-process step leaves alone must stay bit-identical, a trained one must move by the same amount - its first Adam
    step is +-lr wherever the gradient is not zero, so max |update| of the single-process run IS the parameter's learning
    rate.  Spectral-norm u / v (buffers) are compared directly."""
    from helpers import _bn_fed_bias, assert_close, compare_update, oracle_state
    from oracle import hpvg_oracle as O
    for which, ref, init in (("G", PG, fx["G_init"]), ("D", PD, fx["D_init"])):
        if ref is None:
            continue
        before = oracle_state(init)
        names = set(k for k in ref if O.is_param(k))
        for k, v in ref.items():
            if O.is_param(k):
                # (a bias feeding a BatchNorm has an exactly-zero true gradient: its step size is the sibling weight's)
                kk = k[:-len("bias")] + "weight" if _bn_fed_bias(k, names) else k
                step = float((ref[kk].detach().double() - before[kk].double()).abs().max())
                compare_update("%s.%s.%s" % (tag, which, k), before[k].detach(), v.detach(), got[which][k], step if step > 0 else None, 0.0,
                               _bn_fed_bias(k, names))
            elif which == "D" and k.endswith(("weight_u", "weight_v")):
                # (the G step's critic forward runs its power iteration on the UPDATED weights, where the few sign-flipped
                # Adam steps of near-zero gradients differ between any two evaluation orders: 1e-3, north_star's tolerance)
                assert_close(got[which][k], v, 1e-3, "%s.%s.%s" % (tag, which, k), atol=1e-6)


def _check_generator_buffers(fname, got, PG, nranks):
    """After sync_buffers every rank holds the SINGLE-GPU generator buffers: BatchNorm running statistics follow the
    reference's rec-then-rand update sequence, the encoder's spectral-norm u / v are the rec pass's (unit vectors),
    num_batches_tracked counts both passes."""
    from helpers import assert_close
    from oracle import hpvg_oracle as O
    for r in range(nranks):
        for k, v in PG.items():
            if O.is_param(k):
                continue
            mine = got[r]["G"][k]
            if k.endswith("num_batches_tracked"):
                assert int(mine) == int(v), "%s.rank%d.%s: %d vs %d" % (fname, r, k, int(mine), int(v))
            else:
                assert_close(mine, v.detach(), 1e-5, "%s.rank%d.G.%s" % (fname, r, k), atol=1e-6)
            if k.endswith(("weight_u", "weight_v")):
                assert abs(float(mine.norm()) - 1.0) < 1e-5, "%s.rank%d.%s is not a unit vector" % (fname, r, k)


@pytest.mark.parametrize("fname,world", [("step3d_gan_s3.pt", 2), ("step2d_gan_s2.pt", 3), ("step3d_vae_s1.pt", 2)])
def test_distributed_step_matches_single_process(fname, world):
    from helpers import assert_close
    from oracle import hpvg_oracle as O
    fx, want, PG, PD, amps = _single_process(fname)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), fname, d), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(2)]
    for r in range(2):
        assert got[r]["amps"] == pytest.approx(amps, rel=1e-5)
        for k in ("errD_real", "errD_fake", "gradient_penalty", "errG"):
            if k in want:
                assert_close(got[r]["out"][k], want[k], 2e-4, "%s.rank%d.%s" % (fname, r, k))
        _check_updates("%s.rank%d" % (fname, r), fx, got[r], PG, PD)
    _check_generator_buffers(fname, got, PG, 2)
    # the two working ranks hold bit-identical replicas after the step
    for k in got[0]["G"]:
        if O.is_param(k):
            assert torch.equal(got[0]["G"][k], got[1]["G"][k]), k
    if PD is not None:
        for k in got[0]["D"]:
            if O.is_param(k):
                assert torch.equal(got[0]["D"][k], got[1]["D"][k]), k


@pytest.mark.parametrize("fname,world", [("step3d_gan_s3.pt", 4), ("step2d_gan_s2.pt", 5)])
def test_quad_step_matches_single_process(fname, world):
    """Four working ranks: generator passes split by sample with batch-split BatchNorm, discriminator work by sample and
    task, spectral-norm power iterations replayed - same losses, parameters and D u/v buffers as the single process."""
    from helpers import assert_close
    from oracle import hpvg_oracle as O
    fx, want, PG, PD, amps = _single_process(fname)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, _free_port(), fname, d, True), nprocs=world, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(4)]
    for r in range(4):
        assert got[r]["amps"] == pytest.approx(amps, rel=1e-5)
        for k in ("errD_real", "errD_fake", "gradient_penalty", "errG", "rec_loss"):
            assert_close(got[r]["out"][k], want[k], 2e-4, "%s.rank%d.%s" % (fname, r, k))
        _check_updates("%s.rank%d" % (fname, r), fx, got[r], PG, PD)
    _check_generator_buffers(fname, got, PG, 4)
    for r in range(1, 4):
        for k in got[0]["G"]:
            if O.is_param(k):
                assert torch.equal(got[0]["G"][k], got[r]["G"][k]), k
        for k in got[0]["D"]:
            if O.is_param(k):
                assert torch.equal(got[0]["D"][k], got[r]["D"][k]), k


@pytest.mark.parametrize("fname,slab_levels", [("step3d_gan_s3.pt", 2), ("step2d_gan_s2.pt", 1), ("step3d_gan_s2_all.pt", 1)])
def test_oct_step_matches_single_process(fname, slab_levels):
    """Eight working ranks: the four (pass, sample) jobs on two row slabs each - boundary-row swaps on every conv of the
    slabbed generator levels and of the discriminator (first- and second-order backward), BatchNorm over samples and
    slabs, level outputs re-assembled between slabbed levels, lower levels replicated - same losses, parameters and D u/v
    buffers as the single process."""
    from helpers import assert_close
    from oracle import hpvg_oracle as O
    fx, want, PG, PD, amps = _single_process(fname)
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(8, _free_port(), fname, d, True, 2, slab_levels), nprocs=8, join=True)
        got = [torch.load(os.path.join(d, "rank%d.pt" % r), weights_only=True) for r in range(8)]
    for r in range(8):
        assert got[r]["amps"] == pytest.approx(amps, rel=1e-5)
        for k in ("errD_real", "errD_fake", "gradient_penalty", "errG", "rec_loss"):
            assert_close(got[r]["out"][k], want[k], 2e-4, "%s.rank%d.%s" % (fname, r, k))
        _check_updates("%s.rank%d" % (fname, r), fx, got[r], PG, PD)
    _check_generator_buffers(fname, got, PG, 8)
    for r in range(1, 8):
        for k in got[0]["G"]:
            if O.is_param(k):
                assert torch.equal(got[0]["G"][k], got[r]["G"][k]), k
        for k in got[0]["D"]:
            if O.is_param(k):
                assert torch.equal(got[0]["D"][k], got[r]["D"][k]), k
