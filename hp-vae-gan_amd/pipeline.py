"""Level pipeline: the partition BASELINE.json's north_star names - contiguous pyramid levels per GPU, the level OUTPUT
(3 channels, 57 KB ... 3.9 MB, before upsampling) sent point to point to the GPU holding the next level, its gradient
sent back.  Level 0 = encoder + VAE decoder, level k >= 1 = body[k-1]; the rank holding the finest level
also holds the discriminator.

What it buys (stated up front): BatchNorm needs no exchange (a level sees the whole batch), parameters and
optimizer state are sharded by level - but the levels form a chain and the finest level + D hold ~81 % of a late stage's
work, so the iteration is bounded by the last rank: <= ~1.2x over one GPU.  The two generator passes of an iteration
(rec, rand) follow each other through the chain, so ranks overlap on different passes; that is all the overlap there is.
The schedules in multigpu.py (pass / sample / row-slab splits) are what bench.py uses; this one is selected with
HPVG_PARALLELISM=levels.

Per iteration (GAN stage): rec pass forward rank 0 -> R-1, rand pass forward rank 0 -> R-1, discriminator step on rank
R-1, generator loss on rank R-1, backward rank R-1 -> ... down to the level after the detach (networks_3d.py:391-392), one
scalar all-reduce for the global clip norm (train_video.py:201 clips over ALL generator parameters), Adam on every rank
for its own levels.  A rank's copies of the levels it does not own are never used (broadcast_levels() hands the owners'
parameters round at the end of a stage).

Backend-agnostic like multigpu.DistStageTrainer: HipPipeBackend (gfx950 kernels) below, a torch-CPU one in the tests."""
import torch
import torch.distributed as dist

from .multigpu import _next_iteration, all_reduce, broadcast, recv, send


def level_costs(shapes, batch, has_d):
    """Relative cost of each level (voxels; the finest level carries the discriminator: 15 D-forwards vs 6 G-forwards)."""
    c = []
    for k, sh in enumerate(shapes):
        v = batch
        for d in sh:
            v *= int(d)
        c.append(float(v))
    if has_d:
        c[-1] *= 3.5
    return c


def partition_levels(costs, nranks):
    """Contiguous split of levels 0..len-1 over min(nranks, len) ranks minimising the largest share (exact, tiny DP).
    Returns [(first, last)] per rank."""
    n = len(costs)
    r = min(nranks, n)
    pre = [0.0]
    for c in costs:
        pre.append(pre[-1] + c)
    best = {}

    def solve(i, k):  # levels i.. on k ranks -> (max share, cuts)
        if k == 1:
            return pre[n] - pre[i], [(i, n - 1)]
        key = (i, k)
        if key in best:
            return best[key]
        out = None
        for j in range(i, n - k + 1):
            head = pre[j + 1] - pre[i]
            tail, cuts = solve(j + 1, k - 1)
            m = max(head, tail)
            if out is None or m < out[0]:
                out = (m, [(i, j)] + cuts)
        best[key] = out
        return out
    return solve(0, r)[1]


class HipPipeBackend:
    """The arithmetic of the pipeline on the gfx950 kernels."""

    def __init__(self, opt):
        from .multigpu import HipBackend
        self.opt = opt
        self.base = HipBackend(opt)
        for name in ("mse", "kl", "wgan_mean", "grad_penalty", "noise", "next_iteration"):
            setattr(self, name, getattr(self.base, name))

    def g_head(self, netG, video, amps, noise_init, mode, stop):
        return netG(video, amps, noise_init=noise_init, mode=mode, stop_idx=stop)

    def g_levels(self, netG, start, x, amps, mode, stop):
        return netG.refinement_layers(start, x, amps, mode, stop)

    def level_tensors(self, netG, level):
        """(parameters, buffers) of pyramid level `level`: 0 = encoder + VAE decoder, k >= 1 = body[k-1]."""
        mods = [netG.encode, netG.decoder] if level == 0 else [netG.body[level - 1]]
        return [p for m in mods for p in m.parameters()], [b for m in mods for b in m.buffers()]

    def level_shape(self, level, dims):
        from . import utils as hu
        return hu.images.level_shape_3d(level, self.opt) if dims == 3 else hu.images.level_shape_2d(level, self.opt)

    def g_optimizer(self, netG, owned, g_groups, beta1):
        """Adam over the owned levels' groups; the gradient arena covers the whole generator (unowned slots stay zero)."""
        from . import ops
        from . import optim as hp_optim
        arena = hp_optim.ParamArena(netG)
        groups = []
        for params, lr in g_groups:
            params = [p for p in params if id(p) in owned]
            if params:
                groups.append((params, lr))
        adam = hp_optim.FlatAdam(arena, groups, betas=(beta1, 0.999))

        class _O:
            zero = staticmethod(arena.zero_grad)
            sqsum = staticmethod(lambda: ops.sqsum(arena.grad))

            @staticmethod
            def clip_step(sq_total, max_norm):
                ops.clip_scale_(arena.grad, sq_total, max_norm)
                adam.step()

            step = staticmethod(adam.step)     # no clipping (the baseline trainer)
        return _O

    def d_optimizer(self, netD, lr_d, beta1):
        from . import optim as hp_optim
        arena = hp_optim.ParamArena(netD)
        adam = hp_optim.FlatAdam(arena, [(netD.parameters(), lr_d)], betas=(beta1, 0.999))

        class _O:
            zero = staticmethod(arena.zero_grad)
            step = staticmethod(adam.step)
        return _O


class LevelPipelineTrainer:
    """One pyramid stage with its levels spread over the ranks (see module docstring).  step() mirrors
    train.StageTrainer.step(); every rank returns the same dict of (broadcast) loss scalars."""

    def __init__(self, opt, netG, netD, backend, g_groups, dims=3):
        self.opt, self.netG, self.netD, self.be = opt, netG, netD, backend
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.dims = dims
        s = opt.scale_idx
        self.is_gan = opt.vae_levels < s + 1
        self.shapes = [backend.level_shape(k, dims) for k in range(s + 1)]
        self.parts = partition_levels(level_costs(self.shapes, opt.batch_size, self.is_gan), self.world)
        self.R = len(self.parts)
        self.active = self.rank < self.R
        self.dev = next(netG.parameters()).device
        self.iteration = 0
        if not self.active:
            return
        self.a, self.b = self.parts[self.rank]
        self.first, self.last = self.rank == 0, self.rank == self.R - 1
        owned = set()
        for k in range(self.a, self.b + 1):
            owned.update(id(p) for p in backend.level_tensors(netG, k)[0])
        self.optG = backend.g_optimizer(netG, owned, [(list(ps), lr) for ps, lr in g_groups], opt.beta1)
        self.optD = backend.d_optimizer(netD, opt.lr_d, opt.beta1) if (self.is_gan and self.last) else None

    # gradient crosses the boundary below level `a` unless the reference's detach sits at or above it
    def _flows(self, a):
        if a <= 0:
            return False
        return (not self.is_gan) or self.opt.train_all or a > self.opt.vae_levels

    def _io_shape(self, level):
        return (self.opt.batch_size, self.opt.nc_im, *self.shapes[level])

    def _g_pass(self, mode, real_zero, noise_init, grad=True):
        """This rank's part of one generator pass: returns (output of level b, the received leaf or None, head extras)."""
        be, netG, amps = self.be, self.netG, self.opt.Noise_Amps
        extras = None
        xin = None
        if self.first:
            if mode == "rec":
                x, vae_out, mulv = be.g_head(netG, real_zero, amps, None, "rec", self.b)
                extras = (vae_out, mulv)
            else:
                x, _ = be.g_head(netG, noise_init, amps, noise_init, "rand", self.b)
        else:
            xin = torch.empty(self._io_shape(self.a - 1), dtype=torch.float32, device=self.dev)
            recv(xin, src=self.rank - 1)
            if grad and self._flows(self.a):
                xin.requires_grad_(True)
            x = be.g_levels(netG, self.a - 1, xin, amps, mode, self.b)
        if not self.last:
            send(x.detach().contiguous(), dst=self.rank + 1)
        return x, xin, extras

    def _bcast_scalars(self, vals, src):
        if self.rank == src:
            t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device=self.dev)
        else:
            t = torch.zeros(len(vals), dtype=torch.float64, device=self.dev)
        broadcast(t, src=src)
        return [float(v) for v in t.tolist()]

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
        if self.active:
            with torch.no_grad():
                x, _, _ = self._g_pass("rec", real_zero, None, grad=False)
                if self.last:
                    amp = opt.noise_amp_init * float(torch.sqrt(self.be.mse(real, x)).item()) / opt.batch_size
        amp = self._bcast_scalars([amp], self.R - 1)[0]
        opt.noise_amp = amp
        opt.Noise_Amps[-1] = amp

    def step(self, real, real_zero, noise_init=None, alpha=None):
        opt, be = self.opt, self.be
        _next_iteration(be, real.device)
        if self.iteration == 0:
            self.calibrate_noise_amp(real, real_zero)
        self.iteration += 1
        names = ["rec_vae_loss", "kl_loss"] if not self.is_gan else ["errD_real", "errD_fake", "gradient_penalty", "errG", "rec_loss"]
        vals = [0.0] * len(names)
        if self.active:
            if noise_init is None and self.first and self.is_gan:
                noise_init = be.noise(torch.empty(opt.Z_init_size, device=self.dev))
            gen, gen_in, extras = self._g_pass("rec", real_zero, None)
            outs, leaves = [gen], [gen_in]
            if self.is_gan:
                fake, fake_in, _ = self._g_pass("rand", real_zero, noise_init)
                outs.append(fake)
                leaves.append(fake_in)
            self.optG.zero()
            flows_up = (not self.last) and self._flows(self.b + 1)   # will the next rank send gradients for my outputs?
            heads, head_grads = [], []
            if self.last:
                if self.is_gan:
                    netD = self.netD
                    self.optD.zero()
                    errD_real = be.wgan_mean(netD(real), -1.0)
                    errD_fake = be.wgan_mean(netD(fake.detach()), 1.0)
                    gp = be.grad_penalty(netD, real, fake, opt.lambda_grad, alpha if alpha is not None else torch.rand(1, 1))
                    (errD_real + errD_fake + gp).backward()
                    self.optD.step()
                    rec_loss = be.mse(gen, real)
                    for p in netD.parameters():
                        p.requires_grad_(False)
                    errG = be.wgan_mean(netD(fake), -1.0) * opt.disc_loss_weight
                    for p in netD.parameters():
                        p.requires_grad_(True)
                    heads.append(opt.rec_weight * rec_loss + errG)
                    vals = [errD_real, errD_fake, gp, errG, rec_loss]
                else:
                    top = be.mse(gen, real)
                    heads.append(opt.rec_weight * top)
                    vals[0] = top
                head_grads.append(None)
            elif flows_up:
                for o in outs:
                    g = torch.empty_like(o)
                    recv(g, src=self.rank + 1)
                    heads.append(o)
                    head_grads.append(g)
            if self.first and not self.is_gan:
                vae_out, (mu, logvar) = extras
                low = be.mse(vae_out, real_zero)
                kl = be.kl(mu, logvar)
                heads.append(opt.rec_weight * low + opt.kl_weight * kl)
                head_grads.append(None)
                vals_low = (low, kl)
            if heads:
                torch.autograd.backward(heads, head_grads)
            if not self.first and self._flows(self.a):
                for leaf in leaves:
                    send(leaf.grad if leaf.grad is not None else torch.zeros_like(leaf), dst=self.rank - 1)
        # global clip norm over every generator parameter, wherever it lives; then each rank steps its own levels
        sq = self.optG.sqsum() if self.active else torch.zeros(1, dtype=torch.float32, device=self.dev)
        sq = sq.reshape(1).clone()
        all_reduce(sq)
        if self.active:
            self.optG.clip_step(sq, opt.grad_clip)
        # loss scalars: everything from the last rank, the VAE-stage low-level terms from the first
        out = {}
        if self.is_gan:
            got = self._bcast_scalars(vals, self.R - 1)
            out = dict(zip(names, got))
        else:
            top = self._bcast_scalars([vals[0]], self.R - 1)[0]
            low, kl = self._bcast_scalars(list(vals_low) if (self.active and self.first) else [0.0, 0.0], 0)
            out = {"rec_vae_loss": top + low, "kl_loss": kl}
        return out

    def broadcast_levels(self):
        """End of a stage: every rank receives the owners' parameters and buffers (checkpointing, next stage's deepcopy)."""
        for r, (a, b) in enumerate(self.parts):
            for k in range(a, b + 1):
                params, buffers = self.be.level_tensors(self.netG, k)
                for t in params + buffers:
                    broadcast(t.data, src=r)
        if self.netD is not None:
            for t in list(self.netD.parameters()) + list(self.netD.buffers()):
                broadcast(t.data, src=self.R - 1)
        _weights_rewritten()


def _weights_rewritten():
    """Parameters were overwritten through `.data` (a broadcast): torch's version counters did not move, so the packed
    copies the conv kernels cache per weight version (ops.pack_weight) must be dropped by hand."""
    try:
        from . import ops
    except Exception:   # CPU-only test processes that never load the kernels library
        return
    ops.weights_changed()


# ---------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4]: train_video_baselines.py --generator GeneratorSG with one pyramid stage per GPU.
class HipBaselinePipeBackend(HipPipeBackend):
    """Arithmetic of the baseline pipeline on the gfx950 kernels: level k = GeneratorSG.body[k]."""

    def sg_levels(self, netG, x, amps, mode, start, stop):
        return netG(x, amps, mode=mode, start=start, stop=stop)

    def level_tensors(self, netG, level):
        m = netG.body[level]
        return list(m.parameters()), list(m.buffers())


class BaselinePipelineTrainer:
    """One stage of the SinGAN-3D baseline trainer (train.BaselineStageTrainer = train_video_baselines.py:93-173) with
    GeneratorSG's stages spread over the ranks: contiguous stages per rank, the newest stage and the discriminator on the
    last.  The stages below the newest `train_depth` are FROZEN (train_video_baselines.py:55-57), so their ranks only run
    forwards - the 3-channel stage output (before the tanh) goes point to point to the next rank, twice per iteration
    (random pass, reconstruction pass), and no gradient comes back unless a trained stage lives further down
    (`--train-depth` > the last rank's share).  No gradient clipping in this trainer; every rank steps Adam for the trained
    stages it owns.  step() returns the same loss scalars on every rank."""

    def __init__(self, opt, netG, netD, backend):
        self.opt, self.netG, self.netD, self.be = opt, netG, netD, backend
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        s = opt.scale_idx
        if len(netG.body) != s + 1:
            raise ValueError("generator has %d stages, stage index is %d" % (len(netG.body), s))
        pad = opt.num_layer + 2
        self.shapes = [backend.level_shape(k, 3) for k in range(s + 1)]
        padded = [[d + 2 * pad for d in sh] for sh in self.shapes]   # the valid convs of a stage run on the padded volume
        self.parts = partition_levels(level_costs(padded, opt.batch_size, True), self.world)
        self.R = len(self.parts)
        self.active = self.rank < self.R
        self.dev = next(netG.parameters()).device
        self.iteration = 0
        self.lo_t = max(0, s + 1 - opt.train_depth)      # first trained stage
        for block in netG.body[:-opt.train_depth]:
            for p in block.parameters():
                p.requires_grad = False
        if not self.active:
            return
        self.a, self.b = self.parts[self.rank]
        self.first, self.last = self.rank == 0, self.rank == self.R - 1
        owned = set()
        for k in range(self.a, self.b + 1):
            owned.update(id(p) for p in backend.level_tensors(netG, k)[0])
        blocks = list(netG.body[-opt.train_depth:])
        groups = [(list(blk.parameters()), opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))) for i, blk in enumerate(blocks)]
        self.optG = backend.g_optimizer(netG, owned, groups, opt.beta1)
        self.optD = backend.d_optimizer(netD, opt.lr_d, opt.beta1) if self.last else None

    def _flows(self, a):
        """does a gradient cross the boundary below stage a?  Only when a trained stage lives below it."""
        return a > 0 and self.lo_t < a

    def _io_shape(self, level):
        return (self.opt.batch_size, self.opt.nc_im, *self.shapes[level])

    def _g_pass(self, mode, z, grad=True):
        be, netG, amps = self.be, self.netG, self.opt.Noise_Amps
        xin = None
        if self.first:
            x = be.sg_levels(netG, z, amps, mode, 0, self.b + 1)
        else:
            xin = torch.empty(self._io_shape(self.a - 1), dtype=torch.float32, device=self.dev)
            recv(xin, src=self.rank - 1)
            if grad and self._flows(self.a):
                xin.requires_grad_(True)
            x = be.sg_levels(netG, xin, amps, mode, self.a, self.b + 1)
        if not self.last:
            send(x.detach().contiguous(), dst=self.rank + 1)
        return x, xin

    def _bcast_scalars(self, vals, src):
        if self.rank == src:
            t = torch.tensor([float(v) for v in vals], dtype=torch.float64, device=self.dev)
        else:
            t = torch.zeros(len(vals), dtype=torch.float64, device=self.dev)
        broadcast(t, src=src)
        return [float(v) for v in t.tolist()]

    def step(self, real, noise_init=None, alphas=None):
        opt, be = self.opt, self.be
        _next_iteration(be, real.device)
        if self.iteration == 0:
            if opt.scale_idx == 0:
                opt.noise_amp = 1
                opt.Noise_Amps.append(1)
            else:
                opt.Noise_Amps.append(0)
                amp = 0.0
                if self.active:
                    # (the reference runs this pass with autograd enabled, train_video_baselines.py:117-122; nothing is
                    # differentiated afterwards, so the numbers are those of a no_grad pass)
                    with torch.no_grad():
                        x, _ = self._g_pass("rec", opt.Z_init, grad=False)
                        if self.last:
                            amp = opt.noise_amp_init * float(torch.sqrt(be.mse(real, x)).item()) / opt.batch_size
                amp = self._bcast_scalars([amp], self.R - 1)[0]
                opt.noise_amp = amp
                opt.Noise_Amps[-1] = amp
        self.iteration += 1
        names = ["errD_real", "errD_fake", "gradient_penalty", "errG", "rec_loss"]
        vals = [0.0] * len(names)
        if self.active:
            if noise_init is None and self.first:
                noise_init = be.noise(opt.Z_init)
            trains = self.lo_t <= self.b           # this rank holds a trained stage or hands gradients further down
            netD = self.netD
            fake = fake_in = None
            for j in range(opt.Dsteps):
                keep = (j == opt.Dsteps - 1)
                if self.last:
                    self.optD.zero()
                    errD_real = be.wgan_mean(netD(real), -1.0)
                if keep:
                    fake, fake_in = self._g_pass("rand", noise_init, grad=trains)
                else:
                    with torch.no_grad():
                        fake, fake_in = self._g_pass("rand", noise_init, grad=False)
                if self.last:
                    errD_fake = be.wgan_mean(netD(fake.detach()), 1.0)
                    a = None if alphas is None else alphas[j]
                    gp = be.grad_penalty(netD, real, fake, opt.lambda_grad, a if a is not None else torch.rand(1, 1))
                    (errD_real + errD_fake + gp).backward()
                    self.optD.step()
            outs, leaves = [fake], [fake_in]
            gen = None
            if opt.alpha > 0:
                gen, gen_in = self._g_pass("rec", opt.Z_init, grad=trains)
                outs.append(gen)
                leaves.append(gen_in)
            self.optG.zero()
            if self.last:
                for p in netD.parameters():
                    p.requires_grad_(False)
                errG = be.wgan_mean(netD(fake), -1.0) * opt.disc_loss_weight
                for p in netD.parameters():
                    p.requires_grad_(True)
                total = errG
                rec_loss = 0.0
                if gen is not None:
                    rec_loss = opt.alpha * be.mse(gen, real)
                    total = total + rec_loss
                total.backward()
                vals = [errD_real, errD_fake, gp, errG, rec_loss]
            elif trains:
                grads = []
                for o in outs:
                    g = torch.empty_like(o)
                    recv(g, src=self.rank + 1)
                    grads.append(g)
                torch.autograd.backward(outs, grads)
            if trains and not self.first and self._flows(self.a):
                for leaf in leaves:
                    send(leaf.grad if leaf.grad is not None else torch.zeros_like(leaf), dst=self.rank - 1)
            if trains:
                for _ in range(opt.Gsteps):
                    self.optG.step()
        got = self._bcast_scalars(vals, self.R - 1)
        return dict(zip(names, got))

    def broadcast_levels(self):
        """End of a stage: every rank receives the owners' parameters and buffers."""
        for r, (a, b) in enumerate(self.parts):
            for k in range(a, b + 1):
                params, buffers = self.be.level_tensors(self.netG, k)
                for t in params + buffers:
                    broadcast(t.data, src=r)
        for t in list(self.netD.parameters()) + list(self.netD.buffers()):
            broadcast(t.data, src=self.R - 1)
        _weights_rewritten()
