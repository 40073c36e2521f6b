"""Multi-GPU execution of one train iteration: one process per GPU, torch.distributed (backend "nccl" = RCCL over
xGMI on the GPU node, "gloo" in the CPU tests).

What shards.  Stages are strictly sequential and, inside an iteration, pyramid levels form a chain, so
the path does not split into independent units.  What does split in a GAN stage:

  * the two generator passes of an iteration are independent: rank 0 runs the reconstruction pass (mode="rec") and its
    backward, rank 1 the random pass (mode="rand") and its backward;
  * the discriminator has no BatchNorm (spectral norm only), so every D evaluation is separable over the batch
    dimension (batch_size 2): rank b evaluates D(real[b]), D(fake[b]), the gradient penalty of sample b and, in the G
    step, D(fake[b]).  The WGAN means are over (batch, voxels), i.e. 0.5 * (per-sample mean) each.

Exchange steps (point-to-point over one xGMI link, latency bound, <= 3.9 MB): fake[0] rank 1 -> rank 0, and its
gradient dfake[0] rank 0 -> rank 1 (this is the level output that BASELINE.json's north_star ships between GPUs).
Collectives: one all-reduce of the flat D gradient arena (2.2 MB) and one of the flat G gradient arena per
iteration, a 1-float broadcast of the GP alpha and of the calibrated noise amplitude.  Both ranks then apply the
identical Adam update, so the replicas stay bit-identical.  With both ranks busy the step costs
max(rec pass, rand pass) + half the D work: <= 2x over one GPU.  More than two ranks add nothing here (batch is 2,
the last level holds ~81 % of the work): ranks >= 2 idle, and bench.py reports that honestly.

VAE stages (scale_idx < vae_levels) have a single generator pass with BatchNorm over the batch: every rank runs the
identical step (noise broadcast from rank 0), which keeps the replicas in sync at no extra speed.

BatchNorm running statistics / SN u,v of the generator are advanced only by the pass a rank executes (they never
influence training: the reference keeps netG in train mode for every forward) - but they are saved in checkpoints, so
the single-GPU SEQUENCE is reproduced exactly: the reference updates every BatchNorm twice per iteration, rec pass first
(r <- 0.9 r + 0.1 a), then rand pass (r <- 0.9 r + 0.1 b).  The generator's running statistics live in one flat buffer
(optim.BufferArena); a rand-pass rank measures its own update 0.1 b = r_after - 0.9 r_before and sends it (one ~18 KB
message per iteration) to its rec-pass peer, which applies r <- 0.9 r + 0.1 b after its own pass: the rec ranks hold the
single-GPU running statistics at all times.  The encoder's spectral-norm u / v only ever move on the rec ranks (the rand
pass does not encode).  `sync_buffers()` at the end of a stage broadcasts both from rank 0 and adds up the
num_batches_tracked counts of the two passes.

Four ranks ("quad" mode, late stages): the generator passes are ALSO split over the batch - rank q = 2*pass + sample,
rec pass on ranks {0, 1}, rand pass on ranks {2, 3}, each on its own sample - with BatchNorm statistics exchanged inside
each pass pair (ops.BNActSync: a 2 x C double all-reduce per BatchNorm forward and backward, the only new traffic).
Rank (1, b) makes fake[b], keeps the gradient-penalty and the G-step critic work of sample b, and ships fake[b] once to
rank (0, b), which evaluates D(real[b]) and D(fake[b]).  No gradient crosses ranks except inside the all-reduces.
Per-rank work at the finest stage drops from ~36 to ~19.5 conv-units of 72 (SURVEY 8e numbers), i.e. <= 3.7x over one
GPU; early stages (launch / latency bound) stay on two ranks.

Eight ranks ("oct" mode, the finest stages): each of the four (pass, sample) jobs is additionally cut into two ROW SLABS
of the image (slab.py) - rank = 2*(2*pass + sample) + slab.  The slabbed part is where the work is: the discriminator
(all of its evaluations) and the upper `slab_levels` generator levels; the lower generator levels are small and are
replicated on the two slab ranks (their backward splits by linearity, so the weight-gradient all-reduce still adds up to
the whole gradient).  New traffic: one boundary row per conv layer and direction to the slab neighbour (point to point,
<= 0.85 MB), BatchNorm sums over the four ranks of a pass on slabbed levels, and a 3-channel all-reduce per slabbed
level boundary.  Losses are means over (batch, voxels): every rank contributes mean(own slab) * (own rows / rows).

The arithmetic is delegated to a `backend` (HipBackend below: the gfx950 kernels; the gloo tests plug in a torch-CPU
backend built on the oracle) so that the distributed logic is testable without a GPU."""
import os

import torch
import torch.distributed as dist


def _host_staged():
    """gloo cannot move device tensors point-to-point: stage through the host (rehearsal runs of the N>1 path on a
    one-GPU box use gloo; the real multi-GPU run uses nccl = RCCL and never takes this branch)."""
    return dist.get_backend() == "gloo"


def send(t, dst):
    if _host_staged() and t.is_cuda:
        dist.send(t.cpu(), dst=dst)
    else:
        dist.send(t, dst=dst)


def recv(t, src):
    if _host_staged() and t.is_cuda:
        h = torch.empty(t.shape, dtype=t.dtype)
        dist.recv(h, src=src)
        t.copy_(h)
    else:
        dist.recv(t, src=src)


def all_reduce(t, group=None):
    if _host_staged() and t.is_cuda:
        h = t.cpu()
        dist.all_reduce(h, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, group=group)


def broadcast(t, src, group=None):
    if _host_staged() and t.is_cuda:
        h = t.cpu()
        dist.broadcast(h, src=src, group=group)
        t.copy_(h)
    else:
        dist.broadcast(t, src=src, group=group)


def pair_swap(peer, group=None):
    """swap(t) -> the tensor `peer` (global rank) passed to its matching swap (same shape): one send + one receive, point
    to point, issued together (no ordering deadlock).  `group`: the two-rank process group of the pair, so that on RCCL
    the communicator behind the swap is set up by these two ranks alone."""
    def swap(t):
        staged = _host_staged() and t.is_cuda
        src = t.cpu() if staged else t
        got = torch.empty_like(src)
        for w in dist.batch_isend_irecv([dist.P2POp(dist.isend, src, peer, group), dist.P2POp(dist.irecv, got, peer, group)]):
            w.wait()
        return got.to(t.device) if staged else got
    return swap


def _next_iteration(be, device):
    """noise-stream bookkeeping of a backend, once per train iteration (the CPU test backends have none)"""
    f = getattr(be, "next_iteration", None)
    if f is not None:
        f(device)


class HipBackend:
    """Losses / optimisers of the product path (gfx950 kernels)."""

    def __init__(self, opt):
        self.opt = opt

    def mse(self, a, b):
        from .modules.losses import mse_loss
        return mse_loss(a, b)

    def kl(self, mu, logvar):
        from .modules.losses import kl_criterion
        return kl_criterion(mu, logvar)

    def wgan_mean(self, x, sign):
        from .modules.losses import wgan_mean
        return wgan_mean(x, sign)

    def grad_penalty(self, netD, real, fake, lam, alpha):
        from .modules.utils import calc_gradient_penalty
        return calc_gradient_penalty(netD, real, fake, lam, real.device, alpha=alpha)

    def next_iteration(self, device):
        """Start the library noise stream of the next train iteration (ops.rng_next_iteration, as StageTrainer.step does):
        the stream is keyed (torch seed, iteration counter on the device, call index inside the iteration), so the call index
        restarts every iteration instead of growing for the whole run.  Every rank seeds alike and draws the FULL-batch noise
        at the same (iteration, call) - identical tensors on all ranks, which the schedules then slice per rank; a draw that is
        meant to differ between ranks must be sliced from such a full-batch draw (nothing folds the rank into the key)."""
        from . import ops
        if torch.device(device).type == "cuda":
            ops.rng_next_iteration(device)

    def optimizers(self, netG, netD, g_groups, lr_d, beta1):
        from . import optim as hp_optim

        class _Opt:
            pass

        o = _Opt()
        o.arenaG = hp_optim.ParamArena(netG)
        o.optG = hp_optim.FlatAdam(o.arenaG, g_groups, betas=(beta1, 0.999))
        o.allreduce_G = lambda group=None: all_reduce(o.arenaG.grad, group=group)  # ONE collective for all of G
        o.zero_G = o.arenaG.zero_grad
        o.clip_step_G = lambda max_norm: (o.arenaG.clip_grad_norm_(max_norm), o.optG.step())
        if netD is not None:
            o.arenaD = hp_optim.ParamArena(netD)
            o.optD = hp_optim.FlatAdam(o.arenaD, [(netD.parameters(), lr_d)], betas=(beta1, 0.999))
            o.allreduce_D = lambda group=None: all_reduce(o.arenaD.grad, group=group)
            o.zero_D = o.arenaD.zero_grad
            o.step_D = o.optD.step
        return o

    def noise(self, ref):
        from . import utils
        return utils.generate_noise(ref=ref)

    def advance_sn(self, net, n):
        """Run n spectral-norm power iterations on every SN conv of `net` without using the weight: keeps the u/v
        buffers in step with the single-GPU sequence of D forwards when a rank skips some of them."""
        from .modules._nets import SNConv
        with torch.no_grad():
            for _ in range(n):
                for m in net.modules():
                    if isinstance(m, SNConv) and m.training:
                        m.effective_weight()

    def set_sync_bn(self, netG, sync):
        """sync = (allreduce callable, ranks) or None: batch-split BatchNorm on every BatchNorm of the generator."""
        from .modules._nets import BatchNorm
        for m in netG.modules():
            if isinstance(m, BatchNorm):
                m.sync = sync


    def set_slab(self, netG, netD, plan, level_sync=None):
        """plan = slab.SlabPlan or None: row slabs on the generator levels the plan covers (halo swaps on their convs,
        BatchNorm sums per `level_sync(level)`) and on every conv of the discriminator."""
        from .modules._nets import BatchNorm, Conv, SNConv
        netG.slab = plan
        for k, block in enumerate(netG.body):
            on = plan is not None and plan.covers(k + 1)
            for m in block.modules():
                if isinstance(m, (Conv, SNConv)):
                    m.halo = plan.halo if on else None
                elif isinstance(m, BatchNorm) and on:
                    m.sync = level_sync(k + 1)
        if netD is not None:
            for m in netD.modules():
                if isinstance(m, (Conv, SNConv)):
                    m.halo = plan.halo if plan is not None else None


_GROUPS = {}


def subgroup(ranks):
    """Process group of `ranks` (None = the whole world), created once per process and reused by every stage trainer.
    dist.new_group is collective: every rank calls this with the same lists in the same order."""
    ranks = tuple(ranks)
    if len(ranks) == dist.get_world_size():
        return None
    if ranks not in _GROUPS:
        _GROUPS[ranks] = dist.new_group(ranks=list(ranks))
    return _GROUPS[ranks]


class DistStageTrainer:
    """One pyramid stage on `world` ranks (see module docstring).  Mirrors train.StageTrainer.step().
    quad=True (needs world >= 4): generator passes and discriminator work over four ranks."""

    def __init__(self, opt, netG, netD, backend, g_groups, group=None, quad=False, slabs=1, slab_levels=2):
        self.opt, self.netG, self.netD, self.be = opt, netG, netD, backend
        self.rank = dist.get_rank()
        self.world = dist.get_world_size()
        self.is_gan = opt.vae_levels < opt.scale_idx + 1
        if self.is_gan and opt.batch_size != 2:
            raise NotImplementedError("the batch split of the discriminator work assumes batch_size == 2 (reference default)")
        self.quad = bool(quad) and self.is_gan and self.world >= 4
        # oct mode: the four (pass, sample) jobs of quad mode, each on `slabs` row slabs (rank = slabs*job + slab)
        self.nh = int(slabs) if (self.quad and int(slabs) > 1 and self.world >= 4 * int(slabs)) else 1
        if self.nh not in (1, 2):
            raise NotImplementedError("row slabs are wired for two slabs per (pass, sample) job (8 ranks)")
        self.slab_levels = int(slab_levels)
        nwork = (4 * self.nh) if self.quad else 2
        self.nwork = nwork
        self.active = self.rank < nwork or not self.is_gan     # GAN stages: two (four) working ranks
        # collectives of a GAN stage run over the working ranks only (every rank creates every group, in this order)
        self.pair = subgroup([0, 1]) if self.world > 2 else None
        if self.world >= 4:
            self.g_rand = subgroup([2, 3])
            self.g_quad = subgroup([0, 1, 2, 3])
        if self.world >= 8:
            self.g_oct = subgroup(range(8))
            self.g_pass4 = [subgroup(range(4 * p, 4 * p + 4)) for p in range(2)]                     # one generator pass
            self.g_samples = [[subgroup([4 * p + h, 4 * p + 2 + h]) for h in range(2)] for p in range(2)]  # same slab, b=0/1
            self.g_slabs = [subgroup([2 * q, 2 * q + 1]) for q in range(4)]                          # the slabs of one job
        self.work = (self.g_quad if self.quad else self.pair) if self.world >= 4 else self.pair
        if self.nh == 2:
            self.work = self.g_oct
        self.o = backend.optimizers(netG, netD if self.is_gan else None, g_groups, opt.lr_d, opt.beta1)
        self.iteration = 0
        self.dev = next(netG.parameters()).device
        # running statistics of the generator's BatchNorms in one flat tensor (exchanged between the passes' ranks), and the
        # forward counts at the start of the stage (see module docstring / sync_buffers)
        from .optim import BufferArena
        self.bn_stats = BufferArena(netG) if self.is_gan else None
        self._flush_counters()
        self._nbt0 = {n: b.clone() for n, b in netG.named_buffers() if not b.dtype.is_floating_point}
        self.rand_rank = nwork // 2            # first rank of the rand pass (pair: 1, quad: 2, oct: 4)

    # ---- running statistics: the rand pass's update, measured where it happens and applied where the sequence is kept
    BN_KEEP = 0.9   # 1 - momentum of every BatchNorm on the path (networks_3d.py:54: torch default momentum 0.1)

    def _flush_counters(self):
        for m in self.netG.modules():
            if hasattr(m, "flush_counter"):
                m.flush_counter()

    def _rand_stats_before(self):
        return self.bn_stats.flat.clone() if (self.bn_stats is not None and self.bn_stats.flat is not None) else None

    def _rand_stats_send(self, before, dst):
        if before is not None:
            send(self.bn_stats.flat - self.BN_KEEP * before, dst=dst)      # = momentum * (batch statistics of the rand pass)

    def _rec_stats_merge(self, src):
        if self.bn_stats is not None and self.bn_stats.flat is not None:
            upd = torch.empty_like(self.bn_stats.flat)
            recv(upd, src=src)
            with torch.no_grad():
                self.bn_stats.flat.mul_(self.BN_KEEP).add_(upd)

    # ---- tiny helpers
    def _bcast_float(self, value, src=0, group=None):
        t = torch.tensor([float(value) if value is not None else 0.0], dtype=torch.float64, device=self.dev)
        broadcast(t, src=src, group=group)
        return float(t.item())

    def _alpha_to_device(self, alpha):
        if alpha.is_cuda:
            return alpha.reshape(1).to(self.dev, torch.float32).clone()
        if self.dev.type == "cuda":
            from . import ops
            return ops.host_floats_to_device([float(alpha.reshape(-1)[0])], self.dev)   # no host stall
        return alpha.reshape(1).to(self.dev, torch.float32).clone()

    def _bcast_noise(self, ref):
        n = self.be.noise(ref) if self.rank == 0 else torch.empty_like(ref)
        broadcast(n, src=0)
        return n

    def calibrate_noise_amp(self, real, real_zero):
        opt = self.opt
        if opt.const_amp:
            opt.Noise_Amps.append(1)
            return
        if opt.scale_idx == 0:
            opt.noise_amp = 1
            opt.Noise_Amps.append(1)
            return
        opt.Noise_Amps.append(0)
        amp = 0.0
        # rank 0 measures the amplitude; in quad / oct mode every rank of the rec pass repeats that forward (they would
        # only wait for the broadcast otherwise) so that the encoder's spectral-norm u/v AND the noise generator state -
        # the reparameterisation eps drawn here - stay identical on the ranks that later share the rec pass
        if self.rank == 0 or not self.is_gan or (self.quad and self.rank < 2 * self.nh):
            with torch.no_grad():
                rec, _, _ = self.netG(real_zero, opt.Noise_Amps, mode="rec")
                amp = opt.noise_amp_init * float(torch.sqrt(self.be.mse(real, rec)).item()) / opt.batch_size
        if self.is_gan:
            amp = self._bcast_float(amp, 0)
        opt.noise_amp = amp
        opt.Noise_Amps[-1] = amp

    # ---- one iteration
    def step(self, real, real_zero, noise_init=None, alpha=None):
        opt = self.opt
        _next_iteration(self.be, real.device)
        if not self.is_gan:
            return self._vae_step(real, real_zero)
        if self.iteration == 0:
            self.calibrate_noise_amp(real, real_zero)
        self.iteration += 1
        if not self.active:
            return {}
        if self.quad:
            return self._quad_step(real, real_zero, noise_init, alpha)
        r, g = self.rank, self.pair
        netG, netD, be, o = self.netG, self.netD, self.be, self.o
        out = {}
        # -- generator passes: rec on rank 0, rand on rank 1
        if r == 0:
            generated, _, _ = netG(real_zero, opt.Noise_Amps, mode="rec")
            fake_b = torch.empty_like(real[0:1])
            recv(fake_b, src=1)
            self._rec_stats_merge(1)
        else:
            if noise_init is None:
                noise_init = be.noise(torch.empty(opt.Z_init_size, device=self.dev))
            before = self._rand_stats_before()
            fake, _ = netG(noise_init, opt.Noise_Amps, noise_init=noise_init, mode="rand")
            send(fake[0:1].detach().contiguous(), dst=0)
            self._rand_stats_send(before, 0)
            fake_b = fake[1:2].detach().contiguous()
        real_b = real[r:r + 1].contiguous()
        # -- D step on this rank's batch sample (same alpha on both ranks, drawn on rank 0's CPU generator)
        if alpha is None:
            alpha = torch.rand(1, 1) if r == 0 else torch.zeros(1, 1)
        a = self._alpha_to_device(alpha)
        broadcast(a, src=0, group=g)
        o.zero_D()
        errD_real = be.wgan_mean(netD(real_b), -1.0) * 0.5
        errD_fake = be.wgan_mean(netD(fake_b), 1.0) * 0.5
        gp = be.grad_penalty(netD, real_b, fake_b, opt.lambda_grad, a) * 0.5
        (errD_real + errD_fake + gp).backward()
        o.allreduce_D(g)
        o.step_D()
        # -- G step: critic term of this rank's sample, gradient w.r.t. the sample goes back to the rand pass owner
        leaf = fake_b.detach().requires_grad_(True)
        for p in netD.parameters():
            p.requires_grad_(False)
        errG_b = be.wgan_mean(netD(leaf), -1.0) * (0.5 * opt.disc_loss_weight)
        (dfake_b,) = torch.autograd.grad(errG_b, leaf)
        for p in netD.parameters():
            p.requires_grad_(True)
        o.zero_G()
        if r == 0:
            send(dfake_b.contiguous(), dst=1)
            rec_loss = be.mse(generated, real)
            (opt.rec_weight * rec_loss).backward()
            out["rec_loss"] = rec_loss.detach()
        else:
            dfake0 = torch.empty_like(dfake_b)
            recv(dfake0, src=0)
            fake.backward(torch.cat([dfake0, dfake_b], dim=0))
        o.allreduce_G(g)
        o.clip_step_G(opt.grad_clip)
        stats = torch.stack([errD_real.detach(), errD_fake.detach(), gp.detach(), errG_b.detach()]).reshape(4).clone()
        all_reduce(stats, group=g)
        out.update(errD_real=stats[0], errD_fake=stats[1], gradient_penalty=stats[2], errG=stats[3])
        return out

    def _quad_step(self, real, real_zero, noise_init, alpha):
        """GAN-stage iteration on four ranks: rank q = 2*p + b runs pass p (0 rec, 1 rand) on batch sample b.
        With row slabs (oct mode, nh = 2): rank = nh*q + h, and job q's upper levels / discriminator work run on slab h."""
        from .slab import Halo, SlabPlan, slab_rows
        opt, netG, netD, be, o = self.opt, self.netG, self.netD, self.be, self.o
        nh = self.nh
        q, h = divmod(self.rank, nh)
        p, b = divmod(q, 2)
        if nh == 1:
            g_pass = self.pair if p == 0 else self.g_rand     # the two samples of this pass
            g_all = self.g_quad
        else:
            g_pass = self.g_samples[p][h]                     # replicated levels: same slab index, the two samples
            g_all = self.g_oct
        peer = lambda pp: nh * (2 * pp + b) + h               # same sample and slab in the other pass
        out = {}
        # batch-split BatchNorm inside the pass pair; every noise tensor is drawn for the whole batch (identical generator
        # state on the ranks of a pass) and this rank keeps its sample
        be.set_sync_bn(netG, (lambda t: all_reduce(t, group=g_pass), 2))
        H = real.shape[-2]
        r0, r1 = 0, H
        if nh > 1:
            nbr = nh * q + (1 - h)
            g_slab, g_p4 = self.g_slabs[q], self.g_pass4[p]
            swap = pair_swap(nbr, g_slab)
            halo = Halo(up=swap if h == 1 else None, down=swap if h == 0 else None)
            plan = SlabPlan(h, nh, max(opt.vae_levels, opt.scale_idx - self.slab_levels + 1), halo,
                            lambda t: all_reduce(t, group=g_slab))

            def level_sync(level):  # BatchNorm over both samples and both slabs of the level
                shape = self._level_shape(level, real.dim() - 2)
                n = opt.batch_size
                for d in shape:
                    n *= int(d)
                return (lambda t: all_reduce(t, group=g_p4), 2 * nh, n)
            be.set_slab(netG, netD, plan, level_sync)
            r0, r1 = slab_rows(H, h, nh)
        frac = 0.5 * (r1 - r0) / H                            # this rank's share of a mean over (batch, voxels)
        prev_src = netG.noise_source
        draw = prev_src if prev_src is not None else be.noise

        def sliced(ref):
            full = draw(torch.empty((opt.batch_size,) + tuple(ref.shape[1:]), device=ref.device))
            return full[b:b + 1].contiguous()
        netG.noise_source = sliced
        try:
            real_b = real[b:b + 1].narrow(real.dim() - 2, r0, r1 - r0).contiguous()
            if alpha is None:
                alpha = torch.rand(1, 1) if self.rank == 0 else torch.zeros(1, 1)
            a = self._alpha_to_device(alpha)
            broadcast(a, src=0, group=g_all)
            o.zero_D()
            zero = torch.zeros((), device=self.dev)
            errD_real = errD_fake = gp = errG_b = rec_loss = zero
            if p == 0:
                generated, _, _ = netG(real_zero[b:b + 1].contiguous(), opt.Noise_Amps, mode="rec")
                fake_b = torch.empty_like(real_b)
                recv(fake_b, src=peer(1))
                self._rec_stats_merge(peer(1))
                errD_real = be.wgan_mean(netD(real_b), -1.0) * frac   # D forward 1 of the reference sequence
                errD_fake = be.wgan_mean(netD(fake_b), 1.0) * frac    # D forward 2
                (errD_real + errD_fake).backward()
                be.advance_sn(netD, 1)                               # forward 3 (gradient penalty) runs on the rand ranks
            else:
                if noise_init is None:
                    noise_init = be.noise(torch.empty(opt.Z_init_size, device=self.dev))
                z = noise_init[b:b + 1].contiguous()
                before = self._rand_stats_before()
                fake, _ = netG(z, opt.Noise_Amps, noise_init=z, mode="rand")
                fake_b = fake.detach().contiguous()
                send(fake_b, dst=peer(0))
                self._rand_stats_send(before, peer(0))
                be.advance_sn(netD, 2)                               # forwards 1 and 2 run on the rec ranks
                gp = be.grad_penalty(netD, real_b, fake_b, opt.lambda_grad, a) * frac   # D forward 3
                gp.backward()
            o.allreduce_D(g_all)
            o.step_D()
            # -- G step: the critic term of sample b stays on the rank that made fake[b]; the rec ranks take the MSE term
            o.zero_G()
            if p == 1:
                leaf = fake_b.detach().requires_grad_(True)
                for w in netD.parameters():
                    w.requires_grad_(False)
                errG_b = be.wgan_mean(netD(leaf), -1.0) * (frac * opt.disc_loss_weight)
                (dfake_b,) = torch.autograd.grad(errG_b, leaf)
                for w in netD.parameters():
                    w.requires_grad_(True)
                fake.backward(dfake_b)
            else:
                be.advance_sn(netD, 1)                               # forward 4 (critic term of the G step), new weights
                rec_loss = be.mse(generated, real_b) * frac
                (opt.rec_weight * rec_loss).backward()
            o.allreduce_G(g_all)
            o.clip_step_G(opt.grad_clip)
            stats = torch.stack([errD_real.detach(), errD_fake.detach(), gp.detach(), errG_b.detach(), rec_loss.detach()])
            stats = stats.reshape(5).to(torch.float32).clone()
            all_reduce(stats, group=g_all)
            out.update(errD_real=stats[0], errD_fake=stats[1], gradient_penalty=stats[2], errG=stats[3], rec_loss=stats[4])
        finally:
            netG.noise_source = prev_src
            be.set_sync_bn(netG, None)
            if nh > 1:
                be.set_slab(netG, netD, None)
        return out

    def _level_shape(self, level, dims):
        from . import utils as hu
        return hu.images.level_shape_3d(level, self.opt) if dims == 3 else hu.images.level_shape_2d(level, self.opt)

    def _vae_step(self, real, real_zero):
        """Single generator pass with BatchNorm over the batch: every rank runs the identical step."""
        opt, netG, be, o = self.opt, self.netG, self.be, self.o
        prev = netG.noise_source
        netG.noise_source = self._bcast_noise
        try:
            if self.iteration == 0:
                self.calibrate_noise_amp(real, real_zero)
            self.iteration += 1
            generated, generated_vae, (mu, logvar) = netG(real_zero, opt.Noise_Amps, mode="rec")
            rec_vae_loss = be.mse(generated, real) + be.mse(generated_vae, real_zero)
            kl_loss = be.kl(mu, logvar)
            total = opt.rec_weight * rec_vae_loss + opt.kl_weight * kl_loss
            o.zero_G()
            total.backward()
            o.clip_step_G(opt.grad_clip)
        finally:
            netG.noise_source = prev
        return {"rec_vae_loss": rec_vae_loss.detach(), "kl_loss": kl_loss.detach(), "total_loss": total.detach()}

    def sync_buffers(self):
        """End of a stage: every rank gets the single-GPU generator buffers.  BatchNorm running statistics and the
        encoder's spectral-norm u / v are those of rank 0 (a rec-pass rank: it holds the sequential running statistics,
        see the module docstring, and the only ranks that run the encoder are the rec ranks) - broadcast, never averaged
        (the mean of an updated and a stale unit vector is not a unit vector); num_batches_tracked = start of the stage +
        the forwards of the rec pass + those of the rand pass."""
        if not self.is_gan or self.world < 2:
            return
        self._flush_counters()
        for n, b in self.netG.named_buffers():
            if b.dtype.is_floating_point:
                broadcast(b, src=0)
            else:
                d = (b - self._nbt0[n]) if self.rank in (0, self.rand_rank) else torch.zeros_like(b)
                all_reduce(d)
                with torch.no_grad():
                    b.copy_(self._nbt0[n] + d)

    def finish_stage(self):
        """sync_buffers() + hand the trained parameters to the ranks that idled through a GAN stage (world > working ranks)
        and drop the packed-weight cache (parameters are written through .data)."""
        self.sync_buffers()
        if self.is_gan and self.world > self.nwork:
            broadcast_module(self.netG, src=0)
            if self.netD is not None:
                broadcast_module(self.netD, src=0)


def build_bench_runner(make_opt, stages, device, rank, world, config="video", mode="schedules"):
    """bench.py's N>1 path: one trainer per pyramid stage with synthetic resident inputs.
    config "baseline": pipeline.BaselinePipelineTrainer (GeneratorSG stages over the ranks); otherwise mode "levels":
    pipeline.LevelPipelineTrainer, mode "schedules": DistStageTrainer (pass / sample / row-slab splits)."""
    import copy
    from . import train as hp_train
    from . import utils as hu
    from .modules import networks_3d
    torch.manual_seed(0)  # identical replicas on every rank
    base = make_opt(device)
    hu.adjust_scales2image(base.img_size, base)
    base.stop_scale_time = base.stop_scale
    proto = getattr(networks_3d, getattr(base, "generator", "GeneratorHPVAEGAN"))(base)
    built = []
    z_init = None
    for s in range(base.stop_scale + 1):
        if s > 0:
            proto.init_next_stage()
        if s not in stages:
            continue
        opt = make_opt(device)
        hu.adjust_scales2image(opt.img_size, opt)
        opt.stop_scale_time = opt.stop_scale
        opt.scale_idx = s
        opt.Noise_Amps = [1] + [0.05] * max(0, s - 1)
        netG = copy.deepcopy(proto)
        netG.opt = opt
        netG.to(device)
        netD = None
        if opt.vae_levels < s + 1:
            torch.manual_seed(1000 + s)
            netD = networks_3d.WDiscriminator3D(opt).to(device)
        shapes = [hu.images.level_shape_3d(i, opt) for i in range(s + 1)]
        g = torch.Generator().manual_seed(100 + s)
        real = (torch.rand(opt.batch_size, 3, *shapes[s], generator=g) * 2 - 1).to(device)
        real_zero = (torch.rand(opt.batch_size, 3, *shapes[0], generator=g) * 2 - 1).to(device) if s > 0 else real
        opt.Z_init_size = [opt.batch_size, opt.latent_dim, *shapes[0]]
        if config == "baseline":
            from . import pipeline
            if netD is None:
                torch.manual_seed(1000 + s)
                netD = networks_3d.WDiscriminator3D(opt).to(device)
            if z_init is None:
                z_init = torch.randn(opt.batch_size, 3, *shapes[0], generator=torch.Generator().manual_seed(99)).to(device)
            opt.Z_init = z_init
            trainer = pipeline.BaselinePipelineTrainer(opt, netG, netD, pipeline.HipBaselinePipeBackend(opt))
            built.append((s, trainer, real, None))
            continue
        if mode == "levels":
            # the north_star's partition (pipeline.py): contiguous levels per rank; bounded by the rank holding the finest
            # level + D (<= ~1.2x), which is why it is not the default schedule
            from . import pipeline
            trainer = pipeline.LevelPipelineTrainer(opt, netG, netD, pipeline.HipPipeBackend(opt),
                                                    hp_train.generator_param_groups(opt, netG), dims=3)
            built.append((s, trainer, real, real_zero))
            continue
        if netD is None and os.environ.get("HPVG_VAE_ON_RANK0", "1") != "0":
            # VAE stages: one BatchNorm-coupled generator pass on a <= 11 K-voxel volume, a few ms of latency-bound
            # launches - nothing to shard.  Rank 0 trains them alone as a replayed hipGraph (the single-GPU path); a
            # training run would broadcast the stage's parameters once at its end (broadcast_module).  The other ranks
            # wait at the next collective.  (DistStageTrainer._vae_step, every rank in lock-step with broadcast noise,
            # stays available: HPVG_VAE_ON_RANK0=0.)
            trainer = None
            if rank == 0:
                trainer = hp_train.StageTrainer(opt, netG)
                trainer.step(real, real_zero)
                trainer.enable_graph(real, real_zero)
            built.append((s, trainer, real, real_zero))
            continue
        # four working ranks from the first stage whose iteration is long enough to pay for the BatchNorm exchanges
        # (~130 small all-reduces per iteration); earlier GAN stages run on two ranks
        quad = world >= 4 and s >= int(os.environ.get("HPVG_QUAD_MIN_STAGE", "5"))
        # eight: two row slabs per job once a level is big enough that halving its convs outweighs ~30 boundary-row swaps
        oct_ = world >= 8 and s >= int(os.environ.get("HPVG_OCT_MIN_STAGE", "7"))
        trainer = DistStageTrainer(opt, netG, netD, HipBackend(opt), hp_train.generator_param_groups(opt, netG), quad=quad,
                                   slabs=2 if oct_ else 1, slab_levels=int(os.environ.get("HPVG_SLAB_LEVELS", "2")))
        built.append((s, trainer, real, real_zero))

    class Runner:
        n = len(built)

        def timed_stage(self, idx):
            s, trainer, real, real_zero = built[idx]
            if trainer is not None:
                self.last[s] = trainer.step(real) if real_zero is None else trainer.step(real, real_zero)
            return s

        last = {}
        items = built      # [(stage, trainer or None, real, real_zero)]

        def check_finite(self):
            for s, out in self.last.items():
                for k, v in (out or {}).items():
                    t = v if torch.is_tensor(v) else torch.tensor(float(v))
                    if t.numel() <= 2 and not bool(torch.isfinite(t).all()):
                        raise RuntimeError("stage %d: %s is not finite after the timed iterations: %s" % (s, k, v))

    return Runner()


def broadcast_module(net, src=0, group=None):
    """Hand a module trained on one rank (VAE stages) to the others: parameters and buffers, in state_dict order."""
    for t in list(net.parameters()) + list(net.buffers()):
        broadcast(t.data, src=src, group=group)
    # written through `.data`: no version counter moved, so the conv kernels' packed-weight cache must be dropped by hand
    if any(p.is_cuda for p in net.parameters()):
        from . import ops
        ops.weights_changed()
