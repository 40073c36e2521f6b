"""Generate the golden vectors under tests/golden/ by IMPORTING THE REFERENCE (build container only).

Run:  python tests/golden/make_golden.py        (needs /root/reference; never runs on the GPU box)

The reference's hot-path modules are loaded by file path exactly as SURVEY.md Appendix E describes
(`utils/images.py`, `modules/networks_3d.py`, `networks_2d.py`, `losses.py`, `utils.py`; a stub `utils` module
re-exports utils.images.__all__ so that `import utils` inside the networks does not execute utils/__init__.py,
which needs cv2/torchvision).  The three train_*.py scripts cannot be imported here (cv2, kornia, tensorboard,
colorama, neptune are absent), so the step sequence of train() (train_video.py:111-202 / train_image.py:122-217)
is driven by this script around the reference's own classes and functions.  Every random draw made during a step
(Tensor.normal_, torch.rand) is recorded so that the oracle and the HIP path can be fed the same numbers.

Only DATA is written (inputs + expected outputs, as torch tensors / python scalars); no reference source text."""
import importlib.util
import json
import os
import sys
import types

import torch
import torch.nn.functional as F
import torch.optim as optim

REF = '/root/reference'
OUT = os.path.dirname(os.path.abspath(__file__))


def load_reference():
    def by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    images = by_path('utils.images', 'utils/images.py')
    stub = types.ModuleType('utils')
    for n in images.__all__:
        setattr(stub, n, getattr(images, n))
    stub.images = images
    sys.modules['utils'] = stub
    n3 = by_path('ref_networks_3d', 'modules/networks_3d.py')
    n2 = by_path('ref_networks_2d', 'modules/networks_2d.py')
    losses = by_path('ref_losses', 'modules/losses.py')
    mutils = by_path('ref_modules_utils', 'modules/utils.py')
    return images, n3, n2, losses, mutils


def make_opt(**kw):
    o = types.SimpleNamespace(
        nc_im=3, nfc=8, latent_dim=8, enc_blocks=2, ker_size=3, num_layer=5, padd_size=1, stride=1,
        vae_levels=2, train_all=False, train_depth=1, scale_factor_init=0.75, min_size=16, max_size=40, img_size=40,
        ar=0.75, sampling_rates=[4, 3, 2, 1], org_fps=24, fps_lcm=12, batch_size=2, lr_g=5e-4, lr_d=5e-4, beta1=0.5,
        lambda_grad=0.1, rec_weight=10.0, kl_weight=1.0, disc_loss_weight=1.0, lr_scale=0.2, grad_clip=5.0,
        noise_amp_init=0.1, const_amp=False, device='cpu')
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class Recorder:
    """Record every Tensor.normal_() and torch.rand() result, in call order."""

    def __init__(self):
        self.normals = []
        self.rands = []
        self.draws = []       # every normal_ / uniform_ result in call order (the variant generators mix the two)

    def __enter__(self):
        self._normal = torch.Tensor.normal_
        self._uniform = torch.Tensor.uniform_
        self._rand = torch.rand
        rec = self

        def uniform_(t, *a, **k):
            r = rec._uniform(t, *a, **k)
            rec.draws.append(r.detach().clone())
            return r
        torch.Tensor.uniform_ = uniform_

        def normal_(t, *a, **k):
            r = rec._normal(t, *a, **k)
            rec.normals.append(r.detach().clone())
            rec.draws.append(r.detach().clone())
            return r

        def rand(*a, **k):
            r = rec._rand(*a, **k)
            rec.rands.append(r.detach().clone())
            return r

        torch.Tensor.normal_ = normal_
        torch.rand = rand
        return self

    def __exit__(self, *exc):
        torch.Tensor.normal_ = self._normal
        torch.Tensor.uniform_ = self._uniform
        torch.rand = self._rand


class KinkMargin:
    """Smallest |z| seen at the input of any LeakyReLU of the given modules (forward_pre_hooks; the reference's
    activations are in-place).  LeakyReLU'(z) jumps at z = 0, so two fp32 implementations legitimately disagree on the
    gradient mask of an activation with |z| of the order of their rounding difference (~1e-6): a draw that lands there
    is ill-posed as a parity vector, and the generators below reject it."""

    def __init__(self, *modules):
        self.margin = float('inf')
        self.handles = []
        for m in modules:
            if m is None:
                continue
            for sub in m.modules():
                if isinstance(sub, torch.nn.LeakyReLU):
                    self.handles.append(sub.register_forward_pre_hook(self._hook))

    def _hook(self, mod, inp):
        self.margin = min(self.margin, float(inp[0].detach().abs().min()))

    def close(self):
        for h in self.handles:
            h.remove()


class KinkFlip:
    """Re-run helper for fixtures with millions of activations, where no draw can stay clear of the LeakyReLU kinks: every
    LeakyReLU input with |z| < tau has its sign flipped (forward_pre_hook; the value moves by < 2 tau, the gradient mask
    jumps).  The difference to the plain run is what the kink elements can do to each recorded quantity - part of the
    fixture's spread, so that a correct implementation whose rounding lands on the other side of a kink still passes."""

    def __init__(self, tau, *modules):
        self.tau, self.flipped, self.handles = tau, 0, []
        for m in modules:
            if m is None:
                continue
            for sub in m.modules():
                if isinstance(sub, torch.nn.LeakyReLU):
                    self.handles.append(sub.register_forward_pre_hook(self._hook))

    def _hook(self, mod, inp):
        z = inp[0]
        near = z.detach().abs() < self.tau
        self.flipped += int(near.sum())
        return (torch.where(near, z - 2 * z.detach(), z),)   # value -z, derivative of the identity

    def close(self):
        for h in self.handles:
            h.remove()


KINK_FLIP = [0.0]        # wide_fixture(): tau of the kink-flipped re-run (0 = off)
SEED_OVERRIDE = [None]   # main(): re-run a job on exactly the seed its first run settled on
# main(): relative perturbation applied to the input clip of a re-run (2^-20: the size of the difference between two fp32
# implementations after a few dozen layers).  The spread between the plain and the perturbed run is how much the reference
# itself moves when its rounding is disturbed - after an optimizer step that includes Adam's sign flips of ~0 gradients.
INPUT_EPS = [0.0]
# main(): relative size (of a tensor's largest gradient) of Gaussian noise added to every gradient just before its optimizer
# step in a re-run.  Adam's first step is lr * g / (|g| + 1e-8) ~ lr * sign(g): any two fp32 implementations disagree on the
# sign of gradients smaller than their rounding difference (1e-7 ... 1e-4 of the tensor's scale after 10^5-term sums), each
# such weight then lands 2 lr apart, and everything computed AFTER the step (the G-step critic term and the generator's
# gradients in a GAN stage) moves with it.  This run measures that movement in the reference itself.
GRAD_NOISE = [0.0]


def noisy_grads(params, seed):
    if not GRAD_NOISE[0]:
        return
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for p in params:
            if p.grad is not None and p.grad.numel():
                p.grad.add_(GRAD_NOISE[0] * float(p.grad.abs().max()) * torch.randn(p.grad.shape, generator=g))


def iter_spread(a, b):
    """max |a - b| per recorded quantity of one iteration (tensors; dicts of tensors -> dict)."""
    out = {}
    for k, va in a.items():
        vb = b.get(k)
        if torch.is_tensor(va) and va.dtype.is_floating_point:
            out[k] = float((va.double() - vb.double()).abs().max()) if va.numel() else 0.0
        elif isinstance(va, dict):
            out[k] = {n: (float((t.double() - vb[n].double()).abs().max()) if (torch.is_tensor(t) and t.dtype.is_floating_point and t.numel()) else 0.0)
                      for n, t in va.items() if t is not None}
        elif isinstance(va, list) and va and isinstance(va[0], (int, float)):
            out[k] = max(abs(x - y) for x, y in zip(va, vb))
    return out


def merge_spread(a, b):
    if isinstance(a, dict):
        return {k: merge_spread(a[k], b[k]) for k in a}
    if a is None or b is None:
        return a if b is None else b
    if isinstance(a, list):
        return [max(x, y) for x, y in zip(a, b)]
    return max(a, b)


def with_kink_margin(make, seed, min_margin, tries=200):
    """Run make(seed') for seed' = seed, seed+1000, ... until the LeakyReLU kink margin of the draw is >= min_margin."""
    if SEED_OVERRIDE[0] is not None:
        return make(SEED_OVERRIDE[0])
    best = None
    for k in range(tries):
        fx = make(seed + 1000 * k)
        if best is None or fx['kink_margin'] > best['kink_margin']:
            best = fx
        if fx['kink_margin'] >= min_margin:
            return fx
    print('warning: no draw reached margin %g; keeping the best (%g)' % (min_margin, best['kink_margin']))
    return best


def perturb(module, gen):
    """Break the symmetry of fresh / deep-copied blocks: jitter every parameter and BN buffer."""
    with torch.no_grad():
        for n, p in module.named_parameters():
            p.add_(0.05 * torch.randn(p.shape, generator=gen))
        for n, b in module.named_buffers():
            if n.endswith('running_mean'):
                b.add_(0.1 * torch.randn(b.shape, generator=gen))
            elif n.endswith('running_var'):
                b.mul_(1.0 + 0.2 * torch.rand(b.shape, generator=gen))


def sd_clone(module):
    return {k: v.detach().clone() for k, v in module.state_dict().items()}


def g_param_list(netG, opt, scale_idx):
    """Parameter groups exactly as train_video.py:57-86 builds them (this driver's own restatement)."""
    plist = []
    if not opt.train_all:
        if opt.vae_levels < scale_idx + 1:
            train_depth = min(opt.train_depth, len(netG.body) - opt.vae_levels + 1)
            blocks = netG.body[-train_depth:]
            plist += [{"params": b.parameters(), "lr": opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))} for i, b in enumerate(blocks)]
        else:
            plist += [{"params": netG.encode.parameters(), "lr": opt.lr_g * (opt.lr_scale ** scale_idx)},
                      {"params": netG.decoder.parameters(), "lr": opt.lr_g * (opt.lr_scale ** scale_idx)}]
            blocks = netG.body[-opt.train_depth:]
            plist += [{"params": b.parameters(), "lr": opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))} for i, b in enumerate(blocks)]
    else:
        if len(netG.body) < opt.train_depth:
            plist += [{"params": netG.encode.parameters(), "lr": opt.lr_g * (opt.lr_scale ** scale_idx)},
                      {"params": netG.decoder.parameters(), "lr": opt.lr_g * (opt.lr_scale ** scale_idx)}]
            plist += [{"params": b.parameters(), "lr": opt.lr_g * (opt.lr_scale ** (len(netG.body) - 1 - i))} for i, b in enumerate(netG.body)]
        else:
            blocks = netG.body[-opt.train_depth:]
            plist += [{"params": b.parameters(), "lr": opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))} for i, b in enumerate(blocks)]
    return plist


def run_stage_steps(images, nets, losses, mutils, opt, dims, scale_idx, n_iters, seed):
    """Drive `n_iters` iterations of train() at stage `scale_idx` on the reference modules; record everything."""
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    images.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    netG = nets.GeneratorHPVAEGAN(opt)
    for _ in range(scale_idx):
        netG.init_next_stage()
    perturb(netG, gen)
    gan = opt.vae_levels < scale_idx + 1
    D = None
    if gan:
        D = (nets.WDiscriminator3D if dims == 3 else nets.WDiscriminator2D)(opt)
        perturb(D, gen)
        optimizerD = optim.Adam(D.parameters(), lr=opt.lr_d, betas=(opt.beta1, 0.999))
    optimizerG = optim.Adam(g_param_list(netG, opt, scale_idx), lr=opt.lr_g, betas=(opt.beta1, 0.999))
    kinks = KinkMargin(netG, D)

    def shape(i):
        w = images.get_scales_by_index(i, opt.scale_factor, opt.stop_scale, opt.img_size)
        if dims == 3:
            _, td, _ = images.get_fps_td_by_index(i, opt)
            return [td, int(w * opt.ar), w]
        return [int(w * opt.ar), w]

    real = torch.rand(opt.batch_size, 3, *shape(scale_idx), generator=gen) * 2 - 1
    real_zero = torch.rand(opt.batch_size, 3, *shape(0), generator=gen) * 2 - 1 if scale_idx > 0 else real
    if INPUT_EPS[0]:
        real_zero = real_zero * (1 + INPUT_EPS[0]) if scale_idx > 0 else real * (1 + INPUT_EPS[0])
        real = real * (1 + INPUT_EPS[0]) if scale_idx > 0 else real_zero
    z_size = [opt.batch_size, opt.latent_dim, *shape(0)]
    # earlier stages' amplitudes (any plausible values; the last one is calibrated at iteration 0)
    noise_amps = [1] + [0.05 + 0.01 * k for k in range(1, scale_idx)]
    rec_loss_fn = torch.nn.MSELoss()

    fx = {'opt': {k: v for k, v in vars(opt).items() if isinstance(v, (int, float, bool, list, str))},
          'dims': dims, 'scale_idx': scale_idx, 'real': real, 'real_zero': real_zero,
          'G_init': sd_clone(netG), 'D_init': sd_clone(D) if gan else None,
          'noise_amps_init': list(noise_amps), 'iters': []}

    for it in range(n_iters):
        with Recorder() as rec:
            noise_init = images.generate_noise(size=z_size, device='cpu')
            if it == 0:
                with torch.no_grad():
                    if scale_idx == 0:
                        noise_amps.append(1)
                    else:
                        noise_amps.append(0)
                        z_rec, _, _ = netG(real_zero, noise_amps, mode="rec")
                        rmse = torch.sqrt(F.mse_loss(real, z_rec))
                        noise_amps[-1] = opt.noise_amp_init * rmse.item() / opt.batch_size
            rec_i = {}
            generated, generated_vae, (mu, logvar) = netG(real_zero, noise_amps, mode="rec")
            if not gan:
                rec_vae_loss = rec_loss_fn(generated, real) + rec_loss_fn(generated_vae, real_zero)
                kl_loss = losses.kl_criterion(mu, logvar)
                total_loss = opt.rec_weight * rec_vae_loss + opt.kl_weight * kl_loss
                rec_i.update(rec_vae_loss=rec_vae_loss.detach().clone(), kl_loss=kl_loss.detach().clone())
            else:
                D.zero_grad()
                errD_real = -D(real).mean()
                fake, _ = netG(noise_init, noise_amps, noise_init=noise_init, mode="rand")
                errD_fake = D(fake.detach()).mean()
                gp = mutils.calc_gradient_penalty(D, real, fake, opt.lambda_grad, 'cpu')
                errD_total = errD_real + errD_fake + gp
                errD_total.backward()
                rec_i['gradsD'] = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in D.named_parameters()}
                noisy_grads(D.parameters(), 7000 + it)
                optimizerD.step()
                rec_loss = rec_loss_fn(generated, real)
                errG = -D(fake).mean() * opt.disc_loss_weight
                total_loss = opt.rec_weight * rec_loss + errG
                rec_i.update(errD_real=errD_real.detach().clone(), errD_fake=errD_fake.detach().clone(),
                             gradient_penalty=gp.detach().clone(), rec_loss=rec_loss.detach().clone(),
                             errG=errG.detach().clone(), fake=fake.detach().clone())
            netG.zero_grad()
            total_loss.backward()
            rec_i['gradsG'] = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in netG.named_parameters()}
            total_norm = torch.nn.utils.clip_grad_norm_(netG.parameters(), opt.grad_clip)
            noisy_grads(netG.parameters(), 8000 + it)
            optimizerG.step()
        rec_i.update(total_loss=total_loss.detach().clone(), total_norm=total_norm.detach().clone(),
                     generated=generated.detach().clone(), generated_vae=generated_vae.detach().clone(),
                     mu=mu.detach().clone(), logvar=logvar.detach().clone(),
                     noise_init=rec.normals[0], noises=rec.normals[1:], alpha=(rec.rands[0] if rec.rands else None),
                     noise_amps=list(noise_amps), G_after=sd_clone(netG), D_after=sd_clone(D) if gan else None)
        fx['iters'].append(rec_i)
    kinks.close()
    fx['kink_margin'] = kinks.margin
    fx['seed'] = seed
    return fx


class DetNoise:
    """Replace every Tensor.normal_() by tests/detfill.det_normal(shape, 'n<k>') (k = call index) and torch.rand by a
    constant: a step whose inputs are all closed-form (nothing but output summaries needs storing)."""
    ALPHA = 0.37

    def __init__(self):
        self.shapes = []

    def __enter__(self):
        import detfill
        self._normal = torch.Tensor.normal_
        self._rand = torch.rand
        rec = self

        def normal_(t, *a, **k):
            with torch.no_grad():
                t.copy_(detfill.det_normal(tuple(t.shape), 'n%d' % len(rec.shapes)))
            rec.shapes.append(list(t.shape))
            return t

        def rand(*a, **k):
            return torch.full(a if not (len(a) == 1 and isinstance(a[0], (tuple, list))) else tuple(a[0]), DetNoise.ALPHA)

        torch.Tensor.normal_ = normal_
        torch.rand = rand
        return self

    def __exit__(self, *exc):
        torch.Tensor.normal_ = self._normal
        torch.rand = self._rand


def run_wide_step(images, nets, losses, mutils, opt, dims, scale_idx, threads):
    """ONE iteration of train() (train_video.py:111-202) at the BASELINE widths (nfc 64, latent 128, the 256-wide
    pyramid) on the reference modules.  Weights, inputs and noise come from tests/detfill.py (closed form), so the fixture
    holds output SUMMARIES only (norm, abs-max, strided sample per tensor)."""
    import detfill
    torch.set_num_threads(threads)
    torch.manual_seed(0)
    images.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    netG = nets.GeneratorHPVAEGAN(opt)
    for _ in range(scale_idx):
        netG.init_next_stage()
    netG.load_state_dict(detfill.fill_state(netG.state_dict(), 'G'))
    gan = opt.vae_levels < scale_idx + 1
    D = None
    if gan:
        D = (nets.WDiscriminator3D if dims == 3 else nets.WDiscriminator2D)(opt)
        D.load_state_dict(detfill.fill_state(D.state_dict(), 'D'))
        optimizerD = optim.Adam(D.parameters(), lr=opt.lr_d, betas=(opt.beta1, 0.999))
    optimizerG = optim.Adam(g_param_list(netG, opt, scale_idx), lr=opt.lr_g, betas=(opt.beta1, 0.999))
    G0, D0 = sd_clone(netG), (sd_clone(D) if gan else None)
    flipper = KinkFlip(KINK_FLIP[0], netG, D) if KINK_FLIP[0] else None

    def shape(i):
        w = images.get_scales_by_index(i, opt.scale_factor, opt.stop_scale, opt.img_size)
        if dims == 3:
            _, td, _ = images.get_fps_td_by_index(i, opt)
            return [td, int(w * opt.ar), w]
        return [int(w * opt.ar), w]

    real = detfill.det_uniform([opt.batch_size, 3, *shape(scale_idx)], 'real')
    real_zero = detfill.det_uniform([opt.batch_size, 3, *shape(0)], 'real_zero') if scale_idx > 0 else real
    if INPUT_EPS[0]:
        real_zero = real_zero * (1 + INPUT_EPS[0]) if scale_idx > 0 else real * (1 + INPUT_EPS[0])
        real = real * (1 + INPUT_EPS[0]) if scale_idx > 0 else real_zero
    z_size = [opt.batch_size, opt.latent_dim, *shape(0)]
    noise_amps = [1] + [0.05 + 0.01 * k for k in range(1, scale_idx)]
    rec_loss_fn = torch.nn.MSELoss()
    out = {}
    with DetNoise() as rec:
        noise_init = images.generate_noise(size=z_size, device='cpu')
        with torch.no_grad():
            if scale_idx == 0:
                noise_amps.append(1)
            else:
                noise_amps.append(0)
                z_rec, _, _ = netG(real_zero, noise_amps, mode="rec")
                rmse = torch.sqrt(F.mse_loss(real, z_rec))
                noise_amps[-1] = opt.noise_amp_init * rmse.item() / opt.batch_size
        generated, generated_vae, (mu, logvar) = netG(real_zero, noise_amps, mode="rec")
        if not gan:
            rec_vae_loss = rec_loss_fn(generated, real) + rec_loss_fn(generated_vae, real_zero)
            kl_loss = losses.kl_criterion(mu, logvar)
            total_loss = opt.rec_weight * rec_vae_loss + opt.kl_weight * kl_loss
            out.update(rec_vae_loss=float(rec_vae_loss), kl_loss=float(kl_loss))
        else:
            D.zero_grad()
            errD_real = -D(real).mean()
            fake, _ = netG(noise_init, noise_amps, noise_init=noise_init, mode="rand")
            errD_fake = D(fake.detach()).mean()
            gp = mutils.calc_gradient_penalty(D, real, fake, opt.lambda_grad, 'cpu')
            (errD_real + errD_fake + gp).backward()
            out['gradsD'] = {n: detfill.summarize(p.grad, 256) for n, p in D.named_parameters()}
            noisy_grads(D.parameters(), 7000)
            optimizerD.step()
            rec_loss = rec_loss_fn(generated, real)
            errG = -D(fake).mean() * opt.disc_loss_weight
            total_loss = opt.rec_weight * rec_loss + errG
            out.update(errD_real=float(errD_real), errD_fake=float(errD_fake), gradient_penalty=float(gp), rec_loss=float(rec_loss),
                       errG=float(errG), fake=detfill.summarize(fake, 4096))
        netG.zero_grad()
        total_loss.backward()
        out['gradsG'] = {n: (detfill.summarize(p.grad, 256) if p.grad is not None else None) for n, p in netG.named_parameters()}
        total_norm = torch.nn.utils.clip_grad_norm_(netG.parameters(), opt.grad_clip)
        noisy_grads(netG.parameters(), 8000)
        optimizerG.step()
    out.update(total_loss=float(total_loss), total_norm=float(total_norm), noise_amps=list(noise_amps), noise_shapes=rec.shapes,
               alpha=DetNoise.ALPHA, generated=detfill.summarize(generated, 4096), generated_vae=detfill.summarize(generated_vae, 4096),
               mu=detfill.summarize(mu, 4096), logvar=detfill.summarize(logvar, 4096))
    if flipper is not None:
        flipper.close()
        out['kink_flipped'] = flipper.flipped
    G1 = sd_clone(netG)
    out['G_delta'] = {k: detfill.summarize(G1[k].float() - G0[k].float(), 256) for k, _ in netG.named_parameters()}
    out['G_buffers'] = {k: v.clone() for k, v in G1.items() if k.endswith(('running_mean', 'running_var', 'weight_u', 'weight_v', 'num_batches_tracked'))}
    if gan:
        D1 = sd_clone(D)
        out['D_delta'] = {k: detfill.summarize(D1[k].float() - D0[k].float(), 256) for k, _ in D.named_parameters()}
        out['D_buffers'] = {k: v.clone() for k, v in D1.items() if k.endswith(('weight_u', 'weight_v'))}
    return out


def _spread(a, b):
    """Per-entry difference of two runs of the same step (1 thread vs 8 threads of the reference's own CPU kernels): the
    yardstick for how far two correct fp32 implementations drift apart on each quantity."""
    if isinstance(a, dict) and 'sample' in a:
        return {'norm': abs(a['norm'] - b['norm']), 'sample': float((a['sample'].double() - b['sample'].double()).abs().max()),
                'frac_lr10': None}
    if isinstance(a, dict):
        return {k: _spread(a[k], b[k]) for k in a if a[k] is not None}
    if isinstance(a, float):
        return abs(a - b)
    if torch.is_tensor(a) and a.dtype.is_floating_point:
        return float((a.double() - b.double()).abs().max())
    if isinstance(a, list) and a and isinstance(a[0], float):
        return [abs(x - y) for x, y in zip(a, b)]
    return None


def wide_fixture(images, nets, losses, mutils, dims, scale_idx, ar=144.0 / 256.0, **okw):
    def opt():
        return make_opt(**dict(dict(nfc=64, latent_dim=128, vae_levels=3, min_size=32, max_size=256, img_size=256, ar=ar), **okw))
    one = run_wide_step(images, nets, losses, mutils, opt(), dims, scale_idx, threads=1)
    with torch.backends.mkldnn.flags(enabled=False):   # 8 threads AND ATen's native conv kernels instead of oneDNN
        many = run_wide_step(images, nets, losses, mutils, opt(), dims, scale_idx, threads=8)
    INPUT_EPS[0] = 2.0 ** -20
    try:
        pert = run_wide_step(images, nets, losses, mutils, opt(), dims, scale_idx, threads=1)
    finally:
        INPUT_EPS[0] = 0.0
    GRAD_NOISE[0] = 1e-5
    try:
        noisy = run_wide_step(images, nets, losses, mutils, opt(), dims, scale_idx, threads=1)
    finally:
        GRAD_NOISE[0] = 0.0
    KINK_FLIP[0] = 2e-6      # activations are O(1); two fp32 evaluations differ by ~1e-6 after a few dozen layers
    try:
        flipped = run_wide_step(images, nets, losses, mutils, opt(), dims, scale_idx, threads=1)
    finally:
        KINK_FLIP[0] = 0.0
    nflip = flipped.pop('kink_flipped')
    torch.set_num_threads(1)
    o = opt()
    images.adjust_scales2image(o.img_size, o)
    lr = o.lr_g
    spread = merge_spread(merge_spread(merge_spread(_spread(one, many), _spread(one, pert)), _spread(one, noisy)), _spread(one, flipped))
    for name in ('G_delta', 'D_delta'):   # fraction of sampled weights whose update differs by more than lr/10 between the runs
        if name in one:
            for k in one[name]:
                fr = 0.0
                for other in (many, pert, noisy, flipped):
                    d = (one[name][k]['sample'].double() - other[name][k]['sample'].double()).abs()
                    fr = max(fr, float((d > lr / 10).double().mean()))
                spread[name][k]['frac_lr10'] = fr
    return {'opt': {k: v for k, v in vars(o).items() if isinstance(v, (int, float, bool, list, str))}, 'dims': dims,
            'scale_idx': scale_idx, 'expected': one, 'spread': spread, 'kink_flipped': nflip}


def run_baseline_steps(images, n3, mutils, opt, scale_idx, n_iters, seed, generator='GeneratorSG', discriminator='WDiscriminator3D'):
    """train_video_baselines.py:93-173 driven around the reference's GeneratorSG / GeneratorCSG + WDiscriminator3D."""
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    images.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    netG = getattr(n3, generator)(opt)
    for _ in range(scale_idx):
        netG.init_next_stage()
    perturb(netG, gen)
    D = getattr(n3, discriminator)(opt)
    perturb(D, gen)
    optimizerD = optim.Adam(D.parameters(), lr=opt.lr_d, betas=(opt.beta1, 0.999))
    for block in netG.body[:-opt.train_depth]:
        for p in block.parameters():
            p.requires_grad = False
    blocks = netG.body[-opt.train_depth:]
    parameter_list = [{"params": b.parameters(), "lr": opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))}
                      for i, b in enumerate(blocks)]
    if hasattr(netG, 'head') and scale_idx - opt.train_depth < 0:        # train_video_baselines.py:67-71
        parameter_list += [{"params": netG.head.parameters(), "lr": opt.lr_g * (opt.lr_scale ** scale_idx)}]
    if hasattr(netG, 'tail'):
        parameter_list += [{"params": netG.tail.parameters(), "lr": opt.lr_g}]
    optimizerG = optim.Adam(parameter_list, lr=opt.lr_g, betas=(opt.beta1, 0.999))

    def shape(i):
        w = images.get_scales_by_index(i, opt.scale_factor, opt.stop_scale, opt.img_size)
        _, td, _ = images.get_fps_td_by_index(i, opt)
        return [td, int(w * opt.ar), w]

    real = torch.rand(opt.batch_size, 3, *shape(scale_idx), generator=gen) * 2 - 1
    if INPUT_EPS[0]:
        real = real * (1 + INPUT_EPS[0])
    Z_init = torch.randn(opt.batch_size, 3, *shape(0), generator=gen)
    noise_amps = [1] + [0.05 + 0.01 * k for k in range(1, scale_idx)]
    rec_loss_fn = torch.nn.MSELoss()
    fx = {'opt': {k: v for k, v in vars(opt).items() if isinstance(v, (int, float, bool, list, str))}, 'scale_idx': scale_idx,
          'real': real, 'Z_init': Z_init, 'G_init': sd_clone(netG), 'D_init': sd_clone(D), 'noise_amps_init': list(noise_amps),
          'iters': []}
    km = KinkMargin(netG, D)
    for it in range(n_iters):
        with Recorder() as rec:
            noise_init = images.generate_noise(ref=Z_init)
            if it == 0:
                if scale_idx == 0:
                    noise_amps.append(1)
                else:
                    noise_amps.append(0)
                    z_rec = netG(Z_init, noise_amps, mode="rec")
                    rmse = torch.sqrt(F.mse_loss(real, z_rec))
                    noise_amps[-1] = opt.noise_amp_init * rmse.item() / opt.batch_size
            d_steps = []      # the critic after each of its Dsteps updates
            for j in range(opt.Dsteps):
                D.zero_grad()
                errD_real = -D(real).mean()
                if j == opt.Dsteps - 1:
                    fake = netG(noise_init, noise_amps, mode="rand")
                else:
                    with torch.no_grad():
                        fake = netG(noise_init, noise_amps, mode="rand")
                errD_fake = D(fake.detach()).mean()
                gp = mutils.calc_gradient_penalty(D, real, fake, opt.lambda_grad, 'cpu')
                (errD_real + errD_fake + gp).backward()
                gradsD = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in D.named_parameters()}
                noisy_grads(D.parameters(), 7000 + 10 * it + j)
                optimizerD.step()
                d_steps.append(sd_clone(D))
            errG = -D(fake).mean() * opt.disc_loss_weight
            generated = netG(Z_init, noise_amps, mode="rec")
            rec_loss = opt.alpha * rec_loss_fn(generated, real)
            netG.zero_grad()
            (errG + rec_loss).backward()
            gradsG = {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in netG.named_parameters()}
            noisy_grads(netG.parameters(), 8000 + it)
            for _ in range(opt.Gsteps):
                optimizerG.step()
        fx['iters'].append({'noise_init': rec.normals[0], 'noises': rec.normals[1:], 'alphas': list(rec.rands),
                            'errD_real': errD_real.detach().clone(), 'errD_fake': errD_fake.detach().clone(),
                            'gradient_penalty': gp.detach().clone(), 'errG': errG.detach().clone(), 'rec_loss': rec_loss.detach().clone(),
                            'fake': fake.detach().clone(), 'generated': generated.detach().clone(), 'gradsD': gradsD, 'gradsG': gradsG,
                            'noise_amps': list(noise_amps), 'G_after': sd_clone(netG), 'D_after': sd_clone(D), 'D_steps': d_steps})
    km.close()
    fx['kink_margin'] = km.margin
    fx['seed'] = seed
    return fx


def run_sampling(images, nets, opt, dims, scale_idx, seed):
    """The generation path of networks_3d.py:367-387 as the trainers' preview code drives it (train_video.py:225-241:
    no_grad, modules left in train mode): (a) a full random pass from a latent, (b) a restart from an intermediate
    level through sample_init=(index, tensor)."""
    torch.manual_seed(seed)
    gen = torch.Generator().manual_seed(seed + 1)
    images.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    netG = nets.GeneratorHPVAEGAN(opt)
    for _ in range(scale_idx):
        netG.init_next_stage()
    perturb(netG, gen)

    def shape(i):
        w = images.get_scales_by_index(i, opt.scale_factor, opt.stop_scale, opt.img_size)
        if dims == 3:
            _, td, _ = images.get_fps_td_by_index(i, opt)
            return [td, int(w * opt.ar), w]
        return [int(w * opt.ar), w]

    noise_amps = [1] + [0.05 + 0.01 * k for k in range(1, scale_idx + 1)]
    fx = {'opt': {k: v for k, v in vars(opt).items() if isinstance(v, (int, float, bool, list, str))},
          'dims': dims, 'scale_idx': scale_idx, 'G_init': sd_clone(netG), 'noise_amps': list(noise_amps), 'calls': []}
    z_size = [opt.batch_size, opt.latent_dim, *shape(0)]
    start = 1
    prev = torch.rand(opt.batch_size, 3, *shape(start), generator=gen) * 2 - 1
    for sample_init in (None, (start, prev)):
        with Recorder() as rec, torch.no_grad():
            noise_init = images.generate_noise(size=z_size, device='cpu')
            x, vae_out = netG(noise_init, noise_amps, noise_init=noise_init,
                              sample_init=None if sample_init is None else (sample_init[0], sample_init[1].clone()), mode="rand")
        fx['calls'].append({'sample_start': None if sample_init is None else sample_init[0],
                            'sample_tensor': None if sample_init is None else sample_init[1],
                            'noise_init': rec.normals[0], 'noises': rec.normals[1:], 'x': x.clone(), 'vae_out': vae_out.clone(),
                            'G_after': sd_clone(netG)})
    return fx


def op_fixtures(images, n3, n2, losses, mutils):
    """Per-op fixtures on odd shapes (halo bugs), produced by the reference's own block classes."""
    gen = torch.Generator().manual_seed(7)
    out = {}

    def block_case(name, blk, x):
        perturb(blk, gen)
        x = x.clone().requires_grad_(True)
        sd0 = sd_clone(blk)
        y = blk(x)
        gy = torch.randn(y.shape, generator=gen)
        grads = torch.autograd.grad(y, [x] + list(blk.parameters()), grad_outputs=gy)
        out[name] = {'x': x.detach().clone(), 'sd_before': sd0, 'y': y.detach().clone(), 'gy': gy,
                     'dx': grads[0].clone(), 'dparams': {n: g.clone() for (n, _), g in zip(blk.named_parameters(), grads[1:])},
                     'sd_after': {k: v for k, v in sd_clone(blk).items() if not torch.equal(v, sd0[k])}}

    torch.manual_seed(11)
    block_case('convblock3d_3_8', n3.ConvBlock3D(3, 8, 3, 1, 1), torch.randn(2, 3, 5, 7, 9, generator=gen))
    block_case('convblock3d_64_64', n3.ConvBlock3D(64, 64, 3, 1, 1), torch.randn(2, 64, 3, 5, 6, generator=gen))
    block_case('convblock3d_128_8', n3.ConvBlock3D(128, 8, 3, 1, 1), torch.randn(1, 128, 2, 4, 5, generator=gen))
    block_case('convblock3d_8_128_plain', n3.ConvBlock3D(8, 128, 3, 1, 1, bn=False, act=None), torch.randn(1, 8, 2, 5, 4, generator=gen))
    block_case('convblock3dsn_3_64', n3.ConvBlock3DSN(3, 64, 3, 1, 1), torch.randn(2, 3, 4, 6, 7, generator=gen))
    block_case('convblock3dsn_16_24', n3.ConvBlock3DSN(16, 24, 3, 1, 1), torch.randn(1, 16, 3, 4, 7, generator=gen))
    block_case('convblock2d_3_64', n2.ConvBlock2D(3, 64, 3, 1, 1), torch.randn(2, 3, 11, 13, generator=gen))
    block_case('convblock2d_64_64', n2.ConvBlock2D(64, 64, 3, 1, 1), torch.randn(2, 64, 9, 10, generator=gen))
    block_case('convblock2dsn_64_64', n2.ConvBlock2DSN(64, 64, 3, 1, 1), torch.randn(2, 64, 7, 12, generator=gen))
    block_case('tail3d_64_3', torch.nn.Conv3d(64, 3, 3, 1, 1), torch.randn(2, 64, 3, 6, 5, generator=gen))
    block_case('tail3d_64_1', torch.nn.Conv3d(64, 1, 3, padding=1, stride=1), torch.randn(2, 64, 2, 5, 7, generator=gen))

    # pyramid resize through the reference's upscale / upscale_2d (non-integer ratios)
    opt = make_opt(img_size=256, min_size=32, max_size=256, ar=0.5625)
    images.adjust_scales2image(opt.img_size, opt)
    opt.stop_scale_time = opt.stop_scale
    for name, idx, shp in (('upscale3d_l1', 1, (2, 3, 4, 18, 33)), ('upscale3d_l6', 6, (1, 1, 5, 57, 102))):
        x = torch.randn(*shp, generator=gen, requires_grad=True)
        y = images.upscale(x, idx, opt)
        gy = torch.randn(y.shape, generator=gen)
        (dx,) = torch.autograd.grad(y, x, gy)
        out[name] = {'x': x.detach().clone(), 'index': idx, 'opt': {'img_size': 256, 'min_size': 32, 'max_size': 256, 'ar': 0.5625},
                     'y': y.detach().clone(), 'gy': gy, 'dx': dx.clone()}
    opt2 = make_opt(img_size=256, min_size=32, max_size=256, ar=0.75)
    images.adjust_scales2image(opt2.img_size, opt2)
    x = torch.randn(2, 3, 24, 33, generator=gen, requires_grad=True)
    y = images.upscale_2d(x, 1, opt2)
    gy = torch.randn(y.shape, generator=gen)
    (dx,) = torch.autograd.grad(y, x, gy)
    out['upscale2d_l1'] = {'x': x.detach().clone(), 'index': 1, 'opt': {'img_size': 256, 'min_size': 32, 'max_size': 256, 'ar': 0.75},
                           'y': y.detach().clone(), 'gy': gy, 'dx': dx.clone()}

    # KL + reparameterize
    mu = torch.randn(2, 8, 2, 3, 4, generator=gen, requires_grad=True)
    lv = (0.3 * torch.randn(2, 8, 2, 3, 4, generator=gen)).requires_grad_(True)
    kl = losses.kl_criterion(mu, lv)
    dmu, dlv = torch.autograd.grad(kl, [mu, lv])
    out['kl'] = {'mu': mu.detach().clone(), 'logvar': lv.detach().clone(), 'kl': kl.detach().clone(), 'dmu': dmu.clone(), 'dlogvar': dlv.clone()}
    with Recorder() as rec:
        z = n3.reparameterize(mu, lv, True)
    gz = torch.randn(z.shape, generator=gen)
    dmu, dlv = torch.autograd.grad(z, [mu, lv], gz)
    out['reparam'] = {'mu': mu.detach().clone(), 'logvar': lv.detach().clone(), 'eps': rec.normals[0], 'z': z.detach().clone(), 'gz': gz,
                      'dmu': dmu.clone(), 'dlogvar': dlv.clone()}

    # gradient penalty with second-order gradients on every D parameter
    dopt = make_opt(nfc=8)
    torch.manual_seed(5)
    D = n3.WDiscriminator3D(dopt)
    perturb(D, gen)
    sd0 = sd_clone(D)
    real = torch.randn(2, 3, 3, 6, 7, generator=gen)
    fake = torch.randn(2, 3, 3, 6, 7, generator=gen)
    with Recorder() as rec:
        gp = mutils.calc_gradient_penalty(D, real, fake, 0.1, 'cpu')
    D.zero_grad()
    gp.backward()
    out['gp3d'] = {'opt': {'nfc': 8, 'num_layer': 5, 'nc_im': 3, 'ker_size': 3}, 'D_before': sd0, 'real': real, 'fake': fake,
                   'alpha': rec.rands[0], 'gp': gp.detach().clone(),
                   'grads': {n: (p.grad.clone() if p.grad is not None else None) for n, p in D.named_parameters()},
                   'D_after': sd_clone(D)}
    return out


def variant_fixtures(images, n3, n2, losses):
    """The _nb / 1x1 variants (networks_3d.py:110-160,409-485, networks_2d.py:115-165,272-348, losses.py:12-14): no trainer
    drives them, so the vectors are module-level: forward values and the gradients of a VAE-style loss."""
    gen = torch.Generator().manual_seed(21)
    out = {}
    for dims, nets in ((3, n3), (2, n2)):
        opt = make_opt(vae_levels=2, ar=0.75)
        images.adjust_scales2image(opt.img_size, opt)
        opt.stop_scale_time = opt.stop_scale

        def shape(i):
            w = images.get_scales_by_index(i, opt.scale_factor, opt.stop_scale, opt.img_size)
            if dims == 3:
                _, td, _ = images.get_fps_td_by_index(i, opt)
                return [td, int(w * opt.ar), w]
            return [int(w * opt.ar), w]
        torch.manual_seed(31 + dims)
        G = nets.GeneratorVAE_nb(opt)
        for _ in range(2):
            G.init_next_stage()
        perturb(G, gen)
        video = torch.rand(2, 3, *shape(0), generator=gen) * 2 - 1
        real = torch.rand(2, 3, *shape(2), generator=gen) * 2 - 1
        amps = [1, 0.05, 0.06]
        sd0 = sd_clone(G)
        for mode in ('rec', 'rand'):
            G.load_state_dict(sd0)
            G.zero_grad()
            with Recorder() as rec:
                x, vae_out, (mu, logvar, bern) = G(video, amps, mode=mode)
            loss = 10.0 * (F.mse_loss(x, real) + F.mse_loss(vae_out, video)) + losses.kl_criterion(mu, logvar) + losses.kl_bern_criterion(bern)
            loss.backward()
            out['gen%dd_%s' % (dims, mode)] = {
                'opt': {k: v for k, v in vars(opt).items() if isinstance(v, (int, float, bool, list, str))}, 'dims': dims, 'G_init': sd0,
                'video': video, 'real': real, 'amps': amps, 'mode': mode, 'draws': rec.draws, 'x': x.detach().clone(),
                'vae_out': vae_out.detach().clone(), 'mu': mu.detach().clone(), 'logvar': logvar.detach().clone(),
                'bern': bern.detach().clone(), 'loss': loss.detach().clone(), 'kl_bern': losses.kl_bern_criterion(bern).detach().clone(),
                'grads': {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in G.named_parameters()},
                'G_after': sd_clone(G)}
        # the 1x1 encoder (dead code in the reference, same treatment)
        torch.manual_seed(41 + dims)
        E = nets.Encode3DVAE1x1(opt, out_dim=opt.latent_dim)
        perturb(E, gen)
        e0 = sd_clone(E)
        xin = (torch.rand(2, 3, *shape(0), generator=gen) * 2 - 1).requires_grad_(True)
        mu, logvar = E(xin)
        gm, gl = torch.randn(mu.shape, generator=gen), torch.randn(logvar.shape, generator=gen)
        grads = torch.autograd.grad([mu, logvar], [xin] + list(E.parameters()), [gm, gl])
        out['enc1x1_%dd' % dims] = {'opt': {'nc_im': 3, 'nfc': opt.nfc, 'latent_dim': opt.latent_dim}, 'dims': dims, 'E_init': e0,
                                    'x': xin.detach().clone(), 'mu': mu.detach().clone(), 'logvar': logvar.detach().clone(), 'gmu': gm, 'glogvar': gl,
                                    'dx': grads[0].clone(), 'dparams': {n: g.clone() for (n, _), g in zip(E.named_parameters(), grads[1:])},
                                    'E_after': sd_clone(E)}
    return out


def table_fixtures(images):
    rows = []
    for img_size, min_size, max_size, ar in ((256, 32, 256, 0.5625), (256, 32, 256, 0.75), (256, 48, 256, 0.5625), (64, 32, 256, 0.75),
                                             (40, 16, 40, 0.75), (128, 32, 256, 0.5625)):
        opt = make_opt(img_size=img_size, min_size=min_size, max_size=max_size, ar=ar)
        images.adjust_scales2image(opt.img_size, opt)
        opt.stop_scale_time = opt.stop_scale
        levels = []
        for i in range(opt.stop_scale + 1):
            w = images.get_scales_by_index(i, opt.scale_factor, opt.stop_scale, opt.img_size)
            fps, td, fi = images.get_fps_td_by_index(i, opt)
            levels.append({'index': i, 'w': w, 'h': int(w * ar), 'td': td, 'fps': fps, 'fps_index': fi})
        rows.append({'img_size': img_size, 'min_size': min_size, 'max_size': max_size, 'ar': ar, 'num_scales': opt.num_scales,
                     'stop_scale': opt.stop_scale, 'scale1': opt.scale1, 'scale_factor': opt.scale_factor, 'levels': levels})
    return rows


def main():
    """usage: make_golden.py [fixture-name ...]   (default: all)"""
    images, n3, n2, losses, mutils = load_reference()
    torch.set_num_threads(1)
    jobs = {
        'ops.pt': lambda: op_fixtures(images, n3, n2, losses, mutils),
        'variants.pt': lambda: variant_fixtures(images, n3, n2, losses),
        'step3d_vae_s1.pt': lambda: with_kink_margin(lambda sd: run_stage_steps(images, n3, losses, mutils, make_opt(vae_levels=2), 3, 1, 1, seed=sd), 100, 2e-5),
        'step3d_vae_s0.pt': lambda: with_kink_margin(lambda sd: run_stage_steps(images, n3, losses, mutils, make_opt(vae_levels=2), 3, 0, 1, seed=sd), 101, 2e-5),
        'step3d_gan_s3.pt': lambda: run_stage_steps(images, n3, losses, mutils, make_opt(vae_levels=2), 3, 3, 1, seed=102),  # 1.5 M kinks: no draw clears a useful margin
        'step2d_gan_s2.pt': lambda: run_stage_steps(images, n2, losses, mutils, make_opt(vae_levels=1), 2, 2, 2, seed=103),
        'step2d_vae_s1.pt': lambda: with_kink_margin(lambda sd: run_stage_steps(images, n2, losses, mutils, make_opt(vae_levels=3), 2, 1, 1, seed=sd), 104, 2e-5),
        'baseline3d_s2.pt': lambda: run_baseline_steps(images, n3, mutils, make_opt(Dsteps=2, Gsteps=1, alpha=10.0, train_depth=1), 2, 1, seed=105),
        # the baselines script's DEFAULT generator (64-channel features between stages, head/tail trained along)
        'baseline3d_csg_s2.pt': lambda: run_baseline_steps(images, n3, mutils, make_opt(Dsteps=1, Gsteps=1, alpha=10.0, train_depth=3), 2, 1, seed=109,
                                                           generator='GeneratorCSG'),
        # the baselines' own critic: zero-padded input, BatchNorm inside (its gradient penalty differentiates BatchNorm twice)
        'baseline3d_dbl_s1.pt': lambda: with_kink_margin(lambda sd: run_baseline_steps(
            images, n3, mutils, make_opt(Dsteps=1, Gsteps=1, alpha=10.0, train_depth=1), 1, 1, seed=sd,
            discriminator='WDiscriminatorBaselines'), 110, 6e-6),
        'sample3d_s3.pt': lambda: run_sampling(images, n3, make_opt(vae_levels=2), 3, 3, seed=106),
        # non-default training depth: the last two blocks train (no detach between them, scaled learning rates), and
        # --train-all with every level open (train_video.py:74-86, networks_3d.py:391-392)
        'step3d_gan_s3_td2.pt': lambda: run_stage_steps(images, n3, losses, mutils, make_opt(vae_levels=2, train_depth=2), 3, 3, 1, seed=107),
        'step3d_gan_s2_all.pt': lambda: run_stage_steps(images, n3, losses, mutils, make_opt(vae_levels=1, train_all=True, train_depth=8), 3, 2, 1, seed=108),
    }
    sys.path.insert(0, os.path.dirname(OUT))   # tests/detfill.py
    small8 = dict(nfc=4, latent_dim=4, vae_levels=3, img_size=48, min_size=7, max_size=48, sampling_rates=[2, 2, 1, 1], fps_lcm=2)
    jobs.update({
        # whole steps at the BASELINE widths (nfc 64, latent 128, 256-wide pyramid): closed-form inputs, output summaries
        'wide3d_vae_s0.pt': lambda: wide_fixture(images, n3, losses, mutils, 3, 0),
        'wide3d_gan_s3.pt': lambda: wide_fixture(images, n3, losses, mutils, 3, 3),
        # the 2-D path at the same widths (BASELINE configs[1]: train_image.py air_balloons.jpg, 248x186 -> ar 0.75)
        'wide2d_vae_s1.pt': lambda: wide_fixture(images, n2, losses, mutils, 2, 1, ar=186.0 / 248.0),
        'wide2d_gan_s4.pt': lambda: wide_fixture(images, n2, losses, mutils, 2, 4, ar=186.0 / 248.0),
        # EVEN-width pyramids at the BASELINE channel widths: the kernels that own bench stages 8-9 (two-axis Winograd conv: even
        # W; 16-byte eight-wave Winograd weight gradient: W % 4 == 0; `stage_tail`: H * W = 2 mod 4) inside whole steps.
        # configs[3]'s geometry (--min-size 48: widths 48 / 61 / 78 / 99 ...): VAE stage 0 at 27 x 48, GAN stage 2 at 43 x 78
        # (vae_levels 2 so that the 78-wide level is a GAN stage); a 128-wide pyramid whose GAN stage 2 is 40 x 72 (W % 4 == 0)
        'wide3d_e48_vae_s0.pt': lambda: wide_fixture(images, n3, losses, mutils, 3, 0, min_size=48),
        'wide3d_e78_gan_s2.pt': lambda: wide_fixture(images, n3, losses, mutils, 3, 2, min_size=48, vae_levels=2),
        'wide3d_e72_gan_s2.pt': lambda: wide_fixture(images, n3, losses, mutils, 3, 2, min_size=48, max_size=128, img_size=128, vae_levels=2),
        # 8-level pyramids (BASELINE configs[3] / configs[4]: one level per GPU on 8 GPUs), tiny widths: the world-8 gloo tests
        'step3d_gan_s7.pt': lambda: run_stage_steps(images, n3, losses, mutils, make_opt(**small8), 3, 7, 1, seed=111),
        'baseline3d_sg_s7.pt': lambda: run_baseline_steps(images, n3, mutils, make_opt(Dsteps=1, Gsteps=1, alpha=10.0, train_depth=1, **small8),
                                                          7, 1, seed=112),
    })
    want = sys.argv[1:] or ['tables.json'] + list(jobs)
    if 'tables.json' in want:
        with open(os.path.join(OUT, 'tables.json'), 'w') as f:
            json.dump(table_fixtures(images), f, indent=1)
    for name in want:
        if name in jobs:
            fx = jobs[name]()
            if isinstance(fx, dict) and 'iters' in fx and 'seed' in fx:
                # the same draw through the reference once more, on 8 threads and with oneDNN switched off (ATen's native
                # convolution kernels): per-quantity spread between two correct fp32 evaluations of the reference
                # (summation order; after an optimizer step also Adam's sign flips) - the tests' tolerance yardstick
                torch.set_num_threads(8)
                SEED_OVERRIDE[0] = fx['seed']
                try:
                    with torch.backends.mkldnn.flags(enabled=False):   # ATen's native conv kernels: another summation order
                        fx8 = jobs[name]()
                finally:
                    SEED_OVERRIDE[0] = None
                    torch.set_num_threads(1)
                SEED_OVERRIDE[0], INPUT_EPS[0] = fx['seed'], 2.0 ** -20
                try:
                    fxp = jobs[name]()
                finally:
                    SEED_OVERRIDE[0], INPUT_EPS[0] = None, 0.0
                SEED_OVERRIDE[0], GRAD_NOISE[0] = fx['seed'], 1e-5
                try:
                    fxn = jobs[name]()
                finally:
                    SEED_OVERRIDE[0], GRAD_NOISE[0] = None, 0.0
                fx['spread'] = [merge_spread(merge_spread(iter_spread(a, b), iter_spread(a, c)), iter_spread(a, d))
                                for a, b, c, d in zip(fx['iters'], fx8['iters'], fxp['iters'], fxn['iters'])]
            torch.save(fx, os.path.join(OUT, name))
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)))


if __name__ == '__main__':
    main()
