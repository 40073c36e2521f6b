"""Per-scale trainer: the hot loop of the reference's train(opt, netG) (train_video.py:25-258, train_image.py:39-272)
on the MI355X kernels.  Same `opt` blackboard fields, same step sequence and the same quirks (SURVEY.md 3.1 a-j):
netG stays in train mode for every forward, noise amplitude calibrated at iteration 0 and divided by batch_size,
clip over ALL generator gradients, a fresh Adam per stage, D gradients of the G step discarded.

Data loading, logging, tensorboard and checkpoint writing are out of scope: `data` is any iterable
yielding `real` (stage 0) or `(real, real_zero)` device tensors."""

import torch

from . import ops
from . import optim as hp_optim
from . import utils
from .modules import networks_2d, networks_3d
from .modules.losses import kl_criterion, mse_loss, wgan_mean
from .modules.utils import calc_gradient_penalty


_NODE_TYPES = ["kernel", "memcpy", "memset", "host", "graph", "empty", "wait_event", "event_record", "ext_sem_signal",
               "ext_sem_wait", "mem_alloc", "mem_free", "memcpy_from_symbol", "memcpy_to_symbol", "batch_mem_op", "other"]


def graph_node_census(graph):
    """{node type: count} of a torch.cuda.CUDAGraph captured with keep_graph=True (hpvg_graph_node_census)."""
    import ctypes
    from .lib import call
    counts = (ctypes.c_int * len(_NODE_TYPES))()
    call("hpvg_graph_node_census", ctypes.c_void_p(graph.raw_cuda_graph()), counts, len(_NODE_TYPES))
    return {name: int(c) for name, c in zip(_NODE_TYPES, counts)}


def _networks(opt):
    return networks_3d if getattr(opt, 'dims', 3) == 3 else networks_2d


def generator_param_groups(opt, netG):
    """Adam parameter groups and learning rates of the generator for stage opt.scale_idx (train_video.py:57-86)."""
    groups = []
    body = netG.body
    if not opt.train_all:
        if opt.vae_levels < opt.scale_idx + 1:
            depth = min(opt.train_depth, len(body) - opt.vae_levels + 1)
            blocks = list(body[-depth:])
            groups += [(b.parameters(), opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))) for i, b in enumerate(blocks)]
        else:
            lr0 = opt.lr_g * (opt.lr_scale ** opt.scale_idx)
            groups += [(netG.encode.parameters(), lr0), (netG.decoder.parameters(), lr0)]
            blocks = list(body[-opt.train_depth:])
            groups += [(b.parameters(), opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))) for i, b in enumerate(blocks)]
    else:
        if len(body) < opt.train_depth:
            lr0 = opt.lr_g * (opt.lr_scale ** opt.scale_idx)
            groups += [(netG.encode.parameters(), lr0), (netG.decoder.parameters(), lr0)]
            groups += [(b.parameters(), opt.lr_g * (opt.lr_scale ** (len(body) - 1 - i))) for i, b in enumerate(body)]
        else:
            blocks = list(body[-opt.train_depth:])
            groups += [(b.parameters(), opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))) for i, b in enumerate(blocks)]
    return groups


def _capture_iteration(trainer, run, nets):
    """Capture `run()` - one eager iteration of `trainer` on its static input buffers - into a hipGraph.  Sets
    trainer._graph / _g_out / _graph_bn / graph_nodes; leaves trainer.iteration where it was (capture records the launches,
    it does not execute the iteration).  Shared by StageTrainer and BaselineStageTrainer."""
    ops.pin_workspaces()  # the graph bakes in scratch addresses: they must outlive later (larger) stages' buffers
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        run()  # warm-up on the capture stream's side (allocator, workspaces)
    torch.cuda.current_stream().wait_stream(side)
    it = trainer.iteration
    graph = torch.cuda.CUDAGraph(keep_graph=True)   # keep the hipGraph_t: its nodes are inspected below
    # host-side state that python advances while it records the iteration: BatchNorm forward counts (replays must add
    # the same amounts, the capture itself must not count), the optimizers' host step counts, the noise stream's call index
    bns = [m for net in nets if net is not None for m in net.modules() if hasattr(m, 'pending_batches')]
    before = [m.pending_batches for m in bns]
    host = _host_state(trainer)
    ops.weights_changed()  # packed weights made outside the capture must not be baked into it, nor its buffers reused after
    ok = False
    try:
        with torch.cuda.graph(graph):
            trainer._g_out = run()
        ops.weights_changed()
        trainer.graph_nodes = graph_node_census(graph)
        bad = {k: v for k, v in trainer.graph_nodes.items() if k not in ("kernel", "empty", "event_record", "wait_event") and v}
        if bad:
            # memcpy / memset nodes are not reliably ordered against kernel nodes on this runtime (replays
            # trained NaNs); whatever put them there (a torch fill / slice-backward / pad lowering) must become a kernel
            raise GraphCaptureRefused("the captured iteration holds non-kernel graph nodes %s (of %s): refusing to replay it"
                                      % (bad, trainer.graph_nodes))
        graph.instantiate()
        trainer._graph = graph
        trainer._graph_bn = [(m, m.pending_batches - b) for m, b in zip(bns, before) if m.pending_batches != b]
        ok = True
    finally:
        # the capture recorded launches, it did not execute an iteration: whatever the host advanced while recording goes
        # back - also when the capture is refused or fails, so that a caller who catches the error continues eagerly from a
        # consistent state
        for m, b in zip(bns, before):
            m.pending_batches = b
        trainer.iteration = it
        if not ok:
            trainer._graph = None
            trainer._g_out = None
            _restore_host_state(trainer, host)


class GraphCaptureRefused(RuntimeError):
    """the captured iteration cannot be replayed safely (non-kernel graph nodes); the trainer is left in its eager state"""


def _host_state(trainer):
    """host-side counters a recorded (not executed) iteration advances: the optimizers' step counts and the noise stream's
    call index of the trainer's device"""
    st = {"opt": [(o, o.t) for o in (getattr(trainer, n, None) for n in ("optimizerG", "optimizerD")) if o is not None and hasattr(o, "t")]}
    dev = getattr(getattr(trainer, "opt", None), "device", None)
    try:
        rs = ops._rng(torch.device(dev)) if dev is not None and torch.device(dev).type == "cuda" else None
    except Exception:   # (no device: the CPU unit test of this path)
        rs = None
    st["rng"] = (rs, rs.call) if rs is not None else None
    return st


def _restore_host_state(trainer, st):
    for o, t in st["opt"]:
        o.t = t
    if st["rng"] is not None:
        st["rng"][0].call = st["rng"][1]


def _replay_iteration(trainer):
    trainer._graph.replay()
    ops.weights_changed()  # the replayed Adam kernels moved the weights without passing through python
    for m, d in trainer._graph_bn:
        m.pending_batches += d
    trainer.iteration += 1
    return trainer._g_out


class StageTrainer:
    """State of one pyramid stage: discriminator, arenas and optimisers (train_video.py:38-96), plus `step()` =
    one iteration of the hot loop (train_video.py:111-202)."""

    def __init__(self, opt, netG, netD=None):
        self.opt = opt
        self.netG = netG
        self.dims = getattr(opt, 'dims', 3)
        nets = _networks(opt)
        self.is_gan = opt.vae_levels < opt.scale_idx + 1
        self.netD = None
        if self.is_gan:
            self.netD = netD if netD is not None else getattr(nets, opt.discriminator)(opt).to(opt.device)
            self.arenaD = hp_optim.ParamArena(self.netD)
            self.optimizerD = hp_optim.FlatAdam(self.arenaD, [(self.netD.parameters(), opt.lr_d)], betas=(opt.beta1, 0.999))
        self.arenaG = hp_optim.ParamArena(netG)
        self.optimizerG = hp_optim.FlatAdam(self.arenaG, generator_param_groups(opt, netG), betas=(opt.beta1, 0.999))
        level0 = utils.images.level_shape_3d(0, opt) if self.dims == 3 else utils.images.level_shape_2d(0, opt)
        # latent size = the scale-0 size (train_video.py:39-42: set once at stage 0; train_image.py:137-139: every iteration)
        opt.Z_init_size = [opt.batch_size, opt.latent_dim, *level0]
        self.iteration = 0
        self.clip_info = torch.zeros(2, dtype=torch.float32, device=opt.device)
        self.last = {}
        # test hook: callable(trainer) run right after the discriminator's optimizer step (the parity tests put the reference's
        # post-step critic in place there, so that the generator step is judged from an identical state: see tests/helpers.py)
        self.after_d_step = None

    def calibrate_noise_amp(self, real, real_zero):
        """Iteration-0 noise amplitude (train_video.py:131-145)."""
        opt = self.opt
        if opt.const_amp:
            opt.Noise_Amps.append(1)
            return
        with torch.no_grad():
            if opt.scale_idx == 0:
                opt.noise_amp = 1
                opt.Noise_Amps.append(opt.noise_amp)
            else:
                opt.Noise_Amps.append(0)
                z_reconstruction, _, _ = self.netG(real_zero, opt.Noise_Amps, mode="rec")
                rmse = torch.sqrt(mse_loss(real, z_reconstruction))
                opt.noise_amp = opt.noise_amp_init * rmse.item() / opt.batch_size
                opt.Noise_Amps[-1] = opt.noise_amp

    # ---- hipGraph execution of the iteration (launch-bound small stages: ~350-900 launches per iteration)
    def enable_graph(self, real, real_zero):
        """Capture one iteration into a hipGraph and replay it from then on (call after >= 1 eager iteration, i.e.
        after the noise-amplitude calibration and once every workspace has its final size).  Inside the graph the GP
        alpha is drawn from the DEVICE generator (a fresh host->device copy cannot be replayed); Adam reads its step
        count from device memory.  Inputs are copied into static buffers before each replay."""
        if self.iteration < 1:
            raise RuntimeError("run one eager iteration first (noise-amplitude calibration is not capturable)")
        self._g_real = real.clone()
        self._g_rz = self._g_real if real_zero is real else real_zero.clone()
        self._graph_alpha = True
        _capture_iteration(self, lambda: self._step_eager(self._g_real, self._g_rz), (self.netG, self.netD))
        return self

    def step(self, real, real_zero, noise_init=None, alpha=None):
        """One training iteration.  `noise_init` / `alpha` may be injected (parity tests); otherwise drawn like the
        reference does (utils.generate_noise on the device; torch.rand(1,1) on the CPU generator)."""
        if getattr(self, '_graph', None) is not None and noise_init is None and alpha is None:
            if real.data_ptr() != self._g_real.data_ptr():
                self._g_real.copy_(real)
            if real_zero.data_ptr() != self._g_rz.data_ptr() and self._g_rz is not self._g_real:
                self._g_rz.copy_(real_zero)
            self.last = _replay_iteration(self)
            return self.last
        return self._step_eager(real, real_zero, noise_init, alpha)

    def _step_eager(self, real, real_zero, noise_init=None, alpha=None):
        opt, netG = self.opt, self.netG
        if real.is_cuda:
            ops.rng_next_iteration(real.device)   # the library's noise stream: (seed, iteration on the device, call index)
        if alpha is None and getattr(self, '_graph_alpha', False):
            alpha = torch.rand(1, device=real.device)
        if noise_init is None:
            noise_init = utils.generate_noise(size=opt.Z_init_size, device=opt.device)
        if self.iteration == 0:
            self.calibrate_noise_amp(real, real_zero)

        out = {}
        merged = self.is_gan and getattr(opt, 'merge_passes', True) and hasattr(netG, 'forward_pair')
        if merged:
            # rec and rand pass in one (netG.forward_pair): same draws, same arithmetic per sample, half the launches
            generated, fake, generated_vae, (mu, logvar) = netG.forward_pair(real_zero, opt.Noise_Amps, noise_init)
        else:
            generated, generated_vae, (mu, logvar) = netG(real_zero, opt.Noise_Amps, mode="rec")
        if not self.is_gan:
            rec_vae_loss = mse_loss(generated, real) + mse_loss(generated_vae, real_zero)
            kl_loss = kl_criterion(mu, logvar)
            total_loss = opt.rec_weight * rec_vae_loss + opt.kl_weight * kl_loss
            out.update(rec_vae_loss=rec_vae_loss.detach(), kl_loss=kl_loss.detach())
        else:
            netD = self.netD
            self.arenaD.zero_grad()
            errD_real = wgan_mean(netD(real), -1.0)
            if not merged:
                fake, _ = netG(noise_init, opt.Noise_Amps, noise_init=noise_init, mode="rand")
            errD_fake = wgan_mean(netD(fake.detach()), 1.0)
            gradient_penalty = calc_gradient_penalty(netD, real, fake, opt.lambda_grad, opt.device, alpha=alpha)
            errD_total = errD_real + errD_fake + gradient_penalty
            errD_total.backward()
            if getattr(opt, 'record_grads', False):
                out['gradD_flat'] = self.arenaD.grad.clone()
            self.optimizerD.step()
            if self.after_d_step is not None:
                self.after_d_step(self)

            rec_loss = mse_loss(generated, real)
            # D's own weight gradients of this pass are discarded by the reference (D.zero_grad() next iteration);
            # freezing D here skips that wasted work without changing any result
            for p in netD.parameters():
                p.requires_grad_(False)
            errG = wgan_mean(netD(fake), -1.0) * opt.disc_loss_weight
            for p in netD.parameters():
                p.requires_grad_(True)
            total_loss = opt.rec_weight * rec_loss + errG
            out.update(errD_real=errD_real.detach(), errD_fake=errD_fake.detach(), gradient_penalty=gradient_penalty.detach(),
                       rec_loss=rec_loss.detach(), errG=errG.detach(), fake=fake.detach())

        self.arenaG.zero_grad()
        total_loss.backward()
        if getattr(opt, 'record_grads', False):
            out['gradG_flat'] = self.arenaG.grad.clone()
        self.arenaG.clip_grad_norm_(opt.grad_clip, self.clip_info)
        self.optimizerG.step()
        self.iteration += 1
        out.update(total_loss=total_loss.detach(), generated=generated.detach(), generated_vae=generated_vae.detach(),
                   mu=mu.detach(), logvar=logvar.detach(), clip_info=self.clip_info)
        self.last = out
        return out


def train(opt, netG, data, netD=None, niter=None):
    """Train stage opt.scale_idx for opt.niter iterations (reference: train(opt, netG)).  Returns the StageTrainer
    (holding netD and the last losses) so that the caller can checkpoint exactly what the reference saves."""
    if getattr(opt, 'dims', 3) == 3:
        fps, td, fps_index = utils.get_fps_td_by_index(opt.scale_idx, opt)
        opt.fps, opt.td, opt.fps_index = fps, td, fps_index
    trainer = StageTrainer(opt, netG, netD)
    iterator = iter(data)
    n = opt.niter if niter is None else niter
    while trainer.iteration < n:
        try:
            item = next(iterator)
        except StopIteration:
            iterator = iter(data)
            item = next(iterator)
        if opt.scale_idx > 0:
            real, real_zero = item
        else:
            real = item
            real_zero = real
        trainer.step(real, real_zero)
        # after two eager iterations (noise-amplitude calibration done, every workspace at its final size) the iteration
        # is captured once and replayed as a hipGraph: the host leaves the critical path (opt.hip_graph = False: stay
        # eager).  Capturing runs one more real iteration on this batch first (side-stream warm-up), which counts.
        if (trainer.iteration == 2 and n - trainer.iteration >= 2 and getattr(opt, 'hip_graph', True)
                and getattr(trainer, '_graph', None) is None and real.is_cuda):
            try:
                trainer.enable_graph(real, real_zero)
            except GraphCaptureRefused as e:
                # a torch lowering change put a memcpy / memset node into the iteration: say so and train on eagerly (the
                # trainer's host state was put back by _capture_iteration)
                print("hp-vae-gan_amd: hipGraph replay off for stage %d (%s; node census %s)"
                      % (opt.scale_idx, e, getattr(trainer, "graph_nodes", None)))
                opt.hip_graph = False
    return trainer


class BaselineStageTrainer:
    """One pyramid stage of the SinGAN-3D baseline trainer (reference: train_video_baselines.py:24-213, BASELINE
    config 5): fixed reconstruction noise `opt.Z_init`, older stages frozen (requires_grad=False), `Dsteps`
    discriminator updates (the generator graph is kept only for the last one), then the adversarial + alpha * MSE
    generator loss and `Gsteps` optimizer steps; no gradient clipping.  Quirk kept: the iteration-0 noise-amplitude
    pass runs WITH autograd enabled (train_video_baselines.py:117-122)."""

    def __init__(self, opt, netG, netD=None):
        self.opt, self.netG = opt, netG
        nets = _networks(opt)
        if not hasattr(opt, 'Z_init'):
            level0 = utils.images.level_shape_3d(0, opt)
            opt.Z_init = utils.generate_noise(size=[opt.batch_size, 3, *level0], device=opt.device)
        self.netD = netD if netD is not None else getattr(nets, opt.discriminator)(opt).to(opt.device)
        for block in netG.body[:-opt.train_depth]:
            for p in block.parameters():
                p.requires_grad = False
        blocks = list(netG.body[-opt.train_depth:])
        groups = [(b.parameters(), opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))) for i, b in enumerate(blocks)]
        # GeneratorCSG: the shared head trains only while every stage still does, the tail always (train_video_baselines.py:67-72)
        if hasattr(netG, 'head') and opt.scale_idx - opt.train_depth < 0:
            groups.append((netG.head.parameters(), opt.lr_g * (opt.lr_scale ** opt.scale_idx)))
        if hasattr(netG, 'tail'):
            groups.append((netG.tail.parameters(), opt.lr_g))
        self.arenaD = hp_optim.ParamArena(self.netD)
        self.optimizerD = hp_optim.FlatAdam(self.arenaD, [(self.netD.parameters(), opt.lr_d)], betas=(opt.beta1, 0.999))
        self.arenaG = hp_optim.ParamArena(netG)
        self.optimizerG = hp_optim.FlatAdam(self.arenaG, groups, betas=(opt.beta1, 0.999))
        self.iteration = 0
        self.after_d_step = None   # test hook: callable(trainer, j) after the j-th discriminator update (see StageTrainer)

    def enable_graph(self, real, real_zero=None):
        """Capture one iteration into a hipGraph and replay it from then on (see StageTrainer.enable_graph: call after >= 1
        eager iteration; the gradient penalties' alphas then come from the device generator)."""
        if self.iteration < 1:
            raise RuntimeError("run one eager iteration first (noise-amplitude calibration is not capturable)")
        self._g_real = real.clone()
        self._graph_alpha = True
        _capture_iteration(self, lambda: self._step_eager(self._g_real), (self.netG, self.netD))
        return self

    def step(self, real, noise_init=None, alphas=None):
        if getattr(self, '_graph', None) is not None and noise_init is None and alphas is None:
            if real.data_ptr() != self._g_real.data_ptr():
                self._g_real.copy_(real)
            return _replay_iteration(self)
        return self._step_eager(real, noise_init, alphas)

    def _step_eager(self, real, noise_init=None, alphas=None):
        opt, netG, netD = self.opt, self.netG, self.netD
        if real.is_cuda:
            ops.rng_next_iteration(real.device)
        if alphas is None and getattr(self, '_graph_alpha', False):
            alphas = [torch.rand(1, device=real.device) for _ in range(opt.Dsteps)]
        if noise_init is None:
            noise_init = utils.generate_noise(ref=opt.Z_init)
        if self.iteration == 0:
            if opt.scale_idx == 0:
                opt.noise_amp = 1
                opt.Noise_Amps.append(opt.noise_amp)
            else:
                opt.Noise_Amps.append(0)
                z_reconstruction = netG(opt.Z_init, opt.Noise_Amps, mode="rec")
                rmse = torch.sqrt(mse_loss(real, z_reconstruction))
                opt.noise_amp = opt.noise_amp_init * rmse.item() / opt.batch_size
                opt.Noise_Amps[-1] = opt.noise_amp
        out = {}
        for j in range(opt.Dsteps):
            self.arenaD.zero_grad()
            errD_real = wgan_mean(netD(real), -1.0)
            if j == opt.Dsteps - 1:
                fake = netG(noise_init, opt.Noise_Amps, mode="rand")
            else:
                with torch.no_grad():
                    fake = netG(noise_init, opt.Noise_Amps, mode="rand")
            errD_fake = wgan_mean(netD(fake.detach()), 1.0)
            gradient_penalty = calc_gradient_penalty(netD, real, fake, opt.lambda_grad, opt.device,
                                                     alpha=None if alphas is None else alphas[j])
            (errD_real + errD_fake + gradient_penalty).backward()
            if getattr(opt, 'record_grads', False):
                out['gradD_flat'] = self.arenaD.grad.clone()
            self.optimizerD.step()
            if self.after_d_step is not None:
                self.after_d_step(self, j)
        for p in netD.parameters():
            p.requires_grad_(False)
        errG = wgan_mean(netD(fake), -1.0) * opt.disc_loss_weight
        for p in netD.parameters():
            p.requires_grad_(True)
        errG_total = errG
        generated = None
        if opt.alpha > 0:
            generated = netG(opt.Z_init, opt.Noise_Amps, mode="rec")
            rec_loss = opt.alpha * mse_loss(generated, real)
            errG_total = errG_total + rec_loss
            out['rec_loss'] = rec_loss.detach()
        self.arenaG.zero_grad()
        errG_total.backward()
        if getattr(opt, 'record_grads', False):
            out['gradG_flat'] = self.arenaG.grad.clone()
        for _ in range(opt.Gsteps):
            self.optimizerG.step()
        self.iteration += 1
        out.update(errD_real=errD_real.detach(), errD_fake=errD_fake.detach(), gradient_penalty=gradient_penalty.detach(),
                   errG=errG.detach(), fake=fake.detach(), generated=None if generated is None else generated.detach())
        return out
