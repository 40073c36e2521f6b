"""CPU ORACLE for the HP-VAE-GAN train-step hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file; the product
package (hp-vae-gan_amd/) never does.  It is a from-scratch functional restatement, on plain torch CPU fp32
tensors, of the arithmetic the reference (lior1990/hp-vae-gan, /root/reference) performs on this path.  The
reference's arithmetic lives in a third-party dependency - torch ATen (reference pins pytorch==1.4.0, env.sh:3;
this image has torch 2.10.0) - so every step below is written out from the published definition of the op and
cites the reference call site it stands for:

  conv                      nn.Conv3d/2d k3 s1 p1        modules/networks_3d.py:51,63,175,341,362
  batch_norm_train          nn.BatchNorm3d/2d (train)    modules/networks_3d.py:54
  leaky_relu                nn.LeakyReLU(0.2)            modules/networks_3d.py:21
  spectral_norm_weight      nn.utils.spectral_norm       modules/networks_3d.py:63
  resize_linear_ac          F.interpolate(align_corners) utils/images.py:13,17,24
  generator_forward         GeneratorHPVAEGAN.forward    modules/networks_3d.py:367-406, networks_2d.py:230-269
  discriminator_forward     WDiscriminator3D.forward     modules/networks_3d.py:177-181
  kl_criterion              kl_criterion                 modules/losses.py:7-9
  gradient_penalty          calc_gradient_penalty        modules/utils.py:4-19
  train_step                train() loop body            train_video.py:111-202, train_image.py:122-217
  adam_step / clip_grad     optim.Adam / clip_grad_norm_ train_video.py:55,88,201

Convolution itself is evaluated with torch's CPU conv (the same third-party kernel the reference calls);
`conv_direct` (pure loops) and oracle/conv_direct.c pin its semantics (cross-correlation, zero padding) on small
cases.  Derivatives come from torch.autograd over these restated forward formulas.

PARITY PINNING: this oracle is checked in tests/test_oracle_vs_golden.py against golden vectors produced by
importing the reference's own modules in the build container (tests/golden/make_golden.py, Appendix E of
SURVEY.md).  The reference ships no tests or fixtures of its own."""
import math

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2
BN_EPS = 1e-5
BN_MOMENTUM = 0.1
SN_EPS = 1e-12


# ----------------------------------------------------------------------------------------------- geometry
def adjust_scales2image(size, opt):
    """utils/images.py:29-36."""
    opt.num_scales = math.ceil(math.log(math.pow(opt.min_size / size, 1), opt.scale_factor_init)) + 1
    scale2stop = math.ceil(math.log(min([opt.max_size, size]) / size, opt.scale_factor_init))
    opt.stop_scale = opt.num_scales - scale2stop
    opt.scale1 = min(opt.max_size / size, 1)
    opt.scale_factor = math.pow(opt.min_size / size, 1 / opt.stop_scale)


def level_width(index, opt):
    """utils/images.py:60-64."""
    return math.ceil(math.pow(opt.scale_factor, opt.stop_scale - index) * opt.img_size)


def level_td(index, opt):
    """utils/images.py:67-80."""
    fps_index = int((index / opt.stop_scale_time) * (len(opt.sampling_rates) - 1))
    return opt.fps_lcm // opt.sampling_rates[fps_index] + 1


def level_shape(index, opt, dims):
    w = level_width(index, opt)
    if dims == 3:
        return [level_td(index, opt), int(w * opt.ar), w]
    return [int(w * opt.ar), w]


# ----------------------------------------------------------------------------------------------- primitive ops
def conv(x, w, b=None):
    """Cross-correlation, stride 1, zero padding 1 on every spatial/temporal side."""
    if x.dim() == 5:
        return F.conv3d(x, w, b, stride=1, padding=1)
    return F.conv2d(x, w, b, stride=1, padding=1)


def conv_direct(x, w, b=None):
    """Pure-loop definition of `conv` (small cases only): y[n,o,p] = b[o] + sum_{c,tap} w[o,c,tap]*x[n,c,p+tap-1]."""
    x64, w64 = x.double(), w.double()
    nd = x.dim() - 2
    pad = [1] * (2 * nd)
    xp = F.pad(x64, pad)
    sp = x.shape[2:]
    y = torch.zeros((x.shape[0], w.shape[0], *sp), dtype=torch.float64)
    if nd == 3:
        for dt in range(3):
            for dh in range(3):
                for dw in range(3):
                    patch = xp[:, :, dt:dt + sp[0], dh:dh + sp[1], dw:dw + sp[2]]
                    y += torch.einsum('ncthw,oc->nothw', patch, w64[:, :, dt, dh, dw])
    else:
        for dh in range(3):
            for dw in range(3):
                patch = xp[:, :, dh:dh + sp[0], dw:dw + sp[1]]
                y += torch.einsum('nchw,oc->nohw', patch, w64[:, :, dh, dw])
    if b is not None:
        y += b.double().view(1, -1, *([1] * nd))
    return y.float()


def leaky_relu(x):
    return torch.where(x > 0, x, LRELU_SLOPE * x)


def _bn_backward(x, gamma, dy):
    """The batch-norm backward in its fused form (the formula torch's native kernel evaluates, sums accumulated in double
    as ATen's CPU kernel does):  dx = gamma * invstd / N * (N dy - sum dy - xhat sum(dy xhat)),  dgamma = sum(dy xhat),
    dbeta = sum dy.  Written with differentiable ops on the INPUTS, so a gradient penalty can differentiate it again
    (WDiscriminatorBaselines, modules/networks_3d.py:216-247 under modules/utils.py:4-19)."""
    dimsr = [0] + list(range(2, x.dim()))
    shape = (1, -1) + (1,) * (x.dim() - 2)
    n = x.numel() // x.shape[1]
    mean = x.mean(dim=dimsr)
    var = ((x - mean.view(shape)) ** 2).mean(dim=dimsr)
    invstd = 1.0 / torch.sqrt(var + BN_EPS)
    xhat = (x - mean.view(shape)) * invstd.view(shape)
    sdy = dy.sum(dim=dimsr, dtype=torch.float64)
    sdyx = (dy * xhat).sum(dim=dimsr, dtype=torch.float64)
    dx = (gamma * invstd / n).view(shape) * (n * dy - sdy.float().view(shape) - xhat * sdyx.float().view(shape))
    return dx, sdyx.float(), sdy.float()


class _BatchNormTrain(torch.autograd.Function):
    """Forward = the definition; backward = the fused formula above.  (Letting autograd differentiate the forward
    expression term by term gives the same derivative in exact arithmetic but amplifies fp32 rounding by invstd on
    nearly constant channels - 4e-3 relative between two thread counts on tests/golden/wide2d_vae_s1.pt, where the
    reference's own evaluations agree to 1e-8.)"""

    @staticmethod
    def forward(ctx, x, gamma, beta):
        dimsr = [0] + list(range(2, x.dim()))
        shape = (1, -1) + (1,) * (x.dim() - 2)
        mean = x.mean(dim=dimsr)
        var = ((x - mean.view(shape)) ** 2).mean(dim=dimsr)
        y = (x - mean.view(shape)) / torch.sqrt(var.view(shape) + BN_EPS) * gamma.view(shape) + beta.view(shape)
        ctx.save_for_backward(x, gamma)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dmean, _dvar):
        x, gamma = ctx.saved_tensors
        return _bn_backward(x, gamma, dy)


def batch_norm_train(x, gamma, beta, running_mean=None, running_var=None):
    """Per-channel batch statistics over (B, spatial); biased variance for normalisation, unbiased for the
    running_var update; running <- 0.9*running + 0.1*stat (in place)."""
    n = x.numel() // x.shape[1]
    y, mean, var = _BatchNormTrain.apply(x, gamma, beta)
    if running_mean is not None:
        with torch.no_grad():
            running_mean.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * mean.detach())
            running_var.mul_(1 - BN_MOMENTUM).add_(BN_MOMENTUM * var.detach() * (n / max(n - 1, 1)))
    return y


def spectral_norm_weight(w_orig, u, v, training=True):
    """torch hook semantics (n_power_iterations=1, dim=0): under no-grad v <- normalize(W^T u), u <- normalize(W v)
    in place on the buffers; then sigma = u^T W v with u, v constants; W = W_orig / sigma."""
    w_mat = w_orig.reshape(w_orig.shape[0], -1)
    if training:
        with torch.no_grad():
            vn = torch.mv(w_mat.t(), u)
            vn = vn / max(float(vn.norm()), SN_EPS)
            un = torch.mv(w_mat, vn)
            un = un / max(float(un.norm()), SN_EPS)
            v.copy_(vn)
            u.copy_(un)
    uu, vv = u.clone(), v.clone()
    sigma = torch.dot(uu, torch.mv(w_mat, vv))
    return w_orig / sigma


def resize_linear_ac(x, size):
    """(Tri/bi)linear resize with align_corners=True: src = dst*(in-1)/(out-1) (fp32), lerp between floor(src) and
    min(floor(src)+1, in-1), separably over the resized dims (the last len(size) dims)."""
    nd = len(size)
    out = x
    for k in range(nd):
        dim = x.dim() - nd + k
        n_in, n_out = out.shape[dim], int(size[k])
        scale = torch.tensor((n_in - 1) / (n_out - 1) if n_out > 1 else 0.0, dtype=torch.float32)
        src = scale * torch.arange(n_out, dtype=torch.float32)
        i0 = src.floor().long().clamp(max=n_in - 1)
        i1 = (i0 + 1).clamp(max=n_in - 1)
        w1 = (src - i0.float())
        w0 = 1.0 - w1
        shape = [1] * out.dim()
        shape[dim] = n_out
        out = out.index_select(dim, i0) * w0.view(shape) + out.index_select(dim, i1) * w1.view(shape)
    return out


def kl_criterion(mu, logvar):
    return (-0.5 * (1 + logvar - mu.pow(2) - logvar.exp())).mean()


def mse(a, b):
    return ((a - b) ** 2).mean()


# ----------------------------------------------------------------------------------------------- networks
# Parameters/buffers are addressed by the reference's state_dict keys.
def _sn_block(x, P, prefix, training):
    w = spectral_norm_weight(P[prefix + '.conv.weight_orig'], P[prefix + '.conv.weight_u'], P[prefix + '.conv.weight_v'], training)
    return leaky_relu(conv(x, w, P[prefix + '.conv.bias']))


def _bn_block(x, P, prefix):
    r = conv(x, P[prefix + '.conv.weight'], P[prefix + '.conv.bias'])
    if prefix + '.norm.num_batches_tracked' in P:
        P[prefix + '.norm.num_batches_tracked'] += 1
    y = batch_norm_train(r, P[prefix + '.norm.weight'], P[prefix + '.norm.bias'], P[prefix + '.norm.running_mean'],
                         P[prefix + '.norm.running_var'])
    return leaky_relu(y)


def conv_valid(x, w, b=None):
    """padding=0 convolution of the SinGAN baselines (networks_3d.py:285-290)."""
    return F.conv3d(x, w, b, stride=1, padding=0)


def _bn_block_valid(x, P, prefix):
    r = conv_valid(x, P[prefix + '.conv.weight'], P[prefix + '.conv.bias'])
    if prefix + '.norm.num_batches_tracked' in P:
        P[prefix + '.norm.num_batches_tracked'] += 1
    return leaky_relu(batch_norm_train(r, P[prefix + '.norm.weight'], P[prefix + '.norm.bias'], P[prefix + '.norm.running_mean'],
                                       P[prefix + '.norm.running_var']))


def generator_sg_forward(P, opt, noise_init, noise_amp, mode='rand', noises=None, start=0, stop=None):
    """GeneratorSG.forward (networks_3d.py:298-322): valid 7-conv stacks on volumes padded by num_layer+2.
    start / stop (not in the reference; defaults = the whole generator): run stages [start, stop) only - `noise_init` is
    then the previous stage's raw output (before the tanh) for start > 0, and the raw output is returned unless the last
    stage is included (the level pipeline keeps the other stages on other ranks)."""
    pad = opt.num_layer + 2
    p6 = (pad,) * 6
    nb = num_body(P)
    stop = nb if stop is None else stop

    def stack(x, k):
        h = _bn_block_valid(x, P, 'body.%d.head' % k)
        for i in range(opt.num_layer):
            h = _bn_block_valid(h, P, 'body.%d.block%d' % (k, i))
        return conv_valid(h, P['body.%d.tail.weight' % k], P['body.%d.tail.bias' % k])

    x = stack(F.pad(noise_init, p6), 0) if start == 0 else noise_init
    for idx in range(max(start, 1), stop):
        x = torch.tanh(x)
        size = level_shape(idx, opt, 3)
        up = resize_linear_ac(x, size)
        if mode == 'rand':
            up2 = resize_linear_ac(x, [s_ + 2 * pad for s_ in size])
            nz = noises(tuple(up2.shape)) if callable(noises) else next(noises)
            xp = stack(up2 + nz * noise_amp[idx], idx)
        else:
            xp = stack(F.pad(up, p6), idx)
        x = xp + up
    return torch.tanh(x) if stop == nb else x


def generator_csg_forward(P, opt, noise_init, noise_amp, mode='rand', noises=None):
    """GeneratorCSG.forward (networks_3d.py:246-269): head block on the noise, then per stage num_layer VALID conv blocks
    on a volume padded by num_layer; the nfc-channel FEATURES (not images) are upsampled between stages and added
    back without tanh; one tail conv + tanh at the end.  Keys: head.*, body.k.blockI.*, tail.0.{weight,bias}."""
    nl = opt.num_layer
    pn, p1 = (nl,) * 6, (1,) * 6

    def stack(x, k):
        for i in range(nl):
            x = _bn_block_valid(x, P, 'body.%d.block%d' % (k, i))
        return x

    x = stack(F.pad(_bn_block_valid(F.pad(noise_init, p1), P, 'head'), pn), 0)
    k = 1
    while 'body.%d.block0.conv.weight' % k in P:
        size = level_shape(k, opt, 3)
        up = resize_linear_ac(x, size)
        if mode == 'rand':
            up2 = resize_linear_ac(x, [s_ + 2 * nl for s_ in size])
            nz = noises(tuple(up2.shape)) if callable(noises) else next(noises)
            xp = stack(up2 + nz * noise_amp[k], k)
        else:
            xp = stack(F.pad(up, pn), k)
        x = xp + up
        k += 1
    return torch.tanh(conv_valid(F.pad(x, p1), P['tail.0.weight'], P['tail.0.bias']))


def baseline_g_groups(PG, opt, scale_idx):
    """[(key prefix, lr)] of the baseline generator optimizer (train_video_baselines.py:55-73): the last train_depth
    body blocks with lr_g * lr_scale^(distance from the newest); `head` (if the generator has one) only while every
    block is still trained, at lr_g * lr_scale^scale_idx; `tail` (if any) at lr_g."""
    nb = 0
    while any(k.startswith('body.%d.' % nb) for k in PG):
        nb += 1
    blocks = list(range(nb))[-opt.train_depth:]
    groups = [('body.%d.' % kb, opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))) for i, kb in enumerate(blocks)]
    if any(k.startswith('head.') for k in PG) and scale_idx - opt.train_depth < 0:
        groups.append(('head.', opt.lr_g * (opt.lr_scale ** scale_idx)))
    if any(k.startswith('tail.') for k in PG):
        groups.append(('tail.', opt.lr_g))
    return groups


def baseline_train_step(PG, PD, opt, scale_idx, real, Z_init, noise_init, noises, alphas, noise_amps, adam_g, adam_d):
    """One iteration of the baseline train() (train_video_baselines.py:93-173) with all random draws injected.
    The generator is GeneratorCSG when the state dict has a top-level `head`, else GeneratorSG."""
    out = {}
    csg = 'head.conv.weight' in PG
    gen_forward = generator_csg_forward if csg else generator_sg_forward
    groups = baseline_g_groups(PG, opt, scale_idx)
    # blocks outside the optimizer are frozen (requires_grad False); head / tail always keep requires_grad (only `body`
    # blocks are frozen, train_video_baselines.py:55-58), so they receive gradients even when head is not optimised
    nb = 0
    while any(k.startswith('body.%d.' % nb) for k in PG):
        nb += 1
    trained = ['body.%d.' % k for k in range(nb)][-opt.train_depth:] + ['head.', 'tail.']
    gparams = {k: v for k, v in PG.items() if is_param(k) and any(k.startswith(t) for t in trained)}
    dparams = {k: v for k, v in PD.items() if is_param(k)}
    for j in range(opt.Dsteps):
        errD_real = -any_discriminator_forward(real, PD, opt).mean()
        if j == opt.Dsteps - 1:
            fake = gen_forward(PG, opt, noise_init, noise_amps, 'rand', noises)
        else:
            with torch.no_grad():
                fake = gen_forward(PG, opt, noise_init, noise_amps, 'rand', noises)
        errD_fake = any_discriminator_forward(fake.detach(), PD, opt).mean()
        gp = gradient_penalty(PD, opt, real, fake, opt.lambda_grad, alphas[j])
        dgrads = torch.autograd.grad(errD_real + errD_fake + gp, list(dparams.values()), allow_unused=True)
        out['gradsD'] = {k: (g.clone() if g is not None else None) for k, g in zip(dparams.keys(), dgrads)}
        for (k, p), g in zip(dparams.items(), dgrads):
            if g is not None:
                with torch.no_grad():
                    adam_step(p, g, adam_d.setdefault(k, {}), opt.lr_d, opt.beta1)
    errG = -any_discriminator_forward(fake, PD, opt).mean() * opt.disc_loss_weight
    total = errG
    if opt.alpha > 0:
        generated = gen_forward(PG, opt, Z_init, noise_amps, 'rec', None)
        rec_loss = opt.alpha * mse(generated, real)
        total = total + rec_loss
        out['rec_loss'] = rec_loss.detach()
    keys = list(gparams.keys())
    grads = torch.autograd.grad(total, [gparams[k] for k in keys], allow_unused=True)
    out['gradsG'] = {k: (g.clone() if g is not None else None) for k, g in zip(keys, grads)}
    for prefix, lr in groups:
        for k, g in zip(keys, grads):
            if k.startswith(prefix) and g is not None:
                for _ in range(opt.Gsteps):
                    with torch.no_grad():
                        adam_step(gparams[k], g, adam_g.setdefault(k, {}), lr, opt.beta1)
    out.update(errD_real=errD_real.detach(), errD_fake=errD_fake.detach(), gradient_penalty=gp.detach(), errG=errG.detach())
    return out


def _stack7(x, P, prefix, num_layer):
    h = _bn_block(x, P, prefix + '.head')
    for i in range(num_layer):
        h = _bn_block(h, P, prefix + '.block%d' % i)
    return conv(h, P[prefix + '.tail.weight'], P[prefix + '.tail.bias'])


def encoder_forward(x, P, opt, training=True):
    h = x
    for i in range(opt.enc_blocks + 1):
        h = _sn_block(h, P, 'encode.features.conv_block_%d' % i, training)
    mu = conv(h, P['encode.mu.conv.weight'], P['encode.mu.conv.bias'])
    logvar = conv(h, P['encode.logvar.conv.weight'], P['encode.logvar.conv.bias'])
    return mu, logvar


def num_body(P):
    k = 0
    while 'body.%d.tail.weight' % k in P:
        k += 1
    return k


def generator_forward(P, opt, dims, video, noise_amp, noise_init=None, mode='rand', noises=None, training=True,
                      sample_init=None, level_fn=None, stop=None):
    """GeneratorHPVAEGAN.forward (networks_3d.py:367-406).  `noises` is an iterator yielding the N(0,1) draws in
    reference order (reparameterisation eps first, then one tensor per noisy level), or a callable(shape) -> tensor.
    sample_init = (start_index, tensor): the refinement restarts from that level's tensor (generation path).
    stop: run refinement levels [start, stop) only (the level-pipeline tests keep the rest on other ranks).
    level_fn(idx, inp, up, f) -> x, optional: wraps the evaluation f(inp, up) of refinement level idx+1 (the multi-GPU
    tests cut the level into row slabs there); None = f(inp, up)."""
    def draw(shape):
        return noises(tuple(shape)) if callable(noises) else next(noises)

    if sample_init is not None:
        assert num_body(P) > sample_init[0], "Strating index must be lower than # of body blocks"
    if noise_init is None:
        mu, logvar = encoder_forward(video, P, opt, training)
        eps = draw(mu.shape)
        z = eps * torch.exp(0.5 * logvar) + mu if training else eps
    else:
        z = noise_init
    vae_out = torch.tanh(_stack7(z, P, 'decoder', opt.num_layer))
    start, x = (0, vae_out) if sample_init is None else (sample_init[0], sample_init[1])
    for idx in range(start, num_body(P) if stop is None else stop):
        if opt.vae_levels == idx + 1 and not opt.train_all:
            x = x.detach()
            if idx == 0:
                vae_out = x  # reference detaches IN PLACE (networks_3d.py:392): the returned vae_out is cut too
        up = resize_linear_ac(x, level_shape(idx + 1, opt, dims))
        inject = mode == 'rand' and (dims == 2 or opt.vae_levels <= idx + 1)
        inp = up + draw(up.shape) * noise_amp[idx + 1] if inject else up
        def f(i_, u_, idx=idx):
            return torch.tanh(_stack7(i_, P, 'body.%d' % idx, opt.num_layer) + u_)
        x = f(inp, up) if level_fn is None else level_fn(idx, inp, up, f)
    if noise_init is None:
        return x, vae_out, (mu, logvar)
    return x, vae_out


# ----------------------------------------------------------------------------------------------- variant models
def encoder_nb_forward(x, P, opt, training=True):
    """Encode3DVAE_nb / Encode2DVAE_nb.forward (networks_3d.py:130-138, networks_2d.py:135-143): features gated by
    bern = sigmoid(conv), then mu / logvar = global average of a conv.  Keys: encode.features.*, encode.{mu,logvar}.0.conv.*,
    encode.bern.conv.*"""
    h = x
    for i in range(opt.enc_blocks + 1):
        h = _sn_block(h, P, 'encode.features.conv_block_%d' % i, training)
    bern = torch.sigmoid(conv(h, P['encode.bern.conv.weight'], P['encode.bern.conv.bias']))
    h = bern * h
    dims = tuple(range(2, x.dim()))
    mu = conv(h, P['encode.mu.0.conv.weight'], P['encode.mu.0.conv.bias']).mean(dim=dims, keepdim=True)
    logvar = conv(h, P['encode.logvar.0.conv.weight'], P['encode.logvar.0.conv.bias']).mean(dim=dims, keepdim=True)
    return mu, logvar, bern


def conv1x1(x, w, b=None):
    """kernel-size-1 convolution (Encode3DVAE1x1, networks_3d.py:141-160): a per-voxel matrix product."""
    y = torch.einsum('oc,bc...->bo...', w.reshape(w.shape[0], w.shape[1]), x)
    return y if b is None else y + b.view(1, -1, *([1] * (x.dim() - 2)))


def encoder_1x1_forward(x, P, training=True, prefix=''):
    """Encode3DVAE1x1.forward: three spectral-norm 1x1 blocks + LeakyReLU, then mu / logvar 1x1 convs."""
    h = x
    for i in range(3):
        k = prefix + 'features.conv_block_%d.conv.' % i
        w = spectral_norm_weight(P[k + 'weight_orig'], P[k + 'weight_u'], P[k + 'weight_v'], training)
        h = leaky_relu(conv1x1(h, w, P[k + 'bias']))
    return (conv1x1(h, P[prefix + 'mu.conv.weight'], P[prefix + 'mu.conv.bias']),
            conv1x1(h, P[prefix + 'logvar.conv.weight'], P[prefix + 'logvar.conv.bias']))


def reparameterize_bern(x, eps):
    """networks_3d.py:38-42 (training branch), eps ~ U(0,1) injected."""
    return torch.log(x + 1e-20) - torch.log(-torch.log(eps + 1e-20) + 1e-20)


def kl_bern_criterion(x):
    """modules/losses.py:12-14."""
    kld = x * (torch.log(x + 1e-20) - math.log(0.5)) + (1 - x) * (torch.log(1 - x + 1e-20) - math.log(1 - 0.5))
    return kld.mean()


def generator_vae_nb_forward(P, opt, dims, video, noise_amp, noises, mode='rec', training=True):
    """GeneratorVAE_nb.forward with the encoder (networks_3d.py:443-485): `noises` yields, in reference order, the normal eps,
    the uniform eps and then one N(0,1) tensor per level in 'rand' mode (EVERY level: no vae_levels condition here)."""
    mu, logvar, bern = encoder_nb_forward(video, P, opt, training)
    z_norm = next(noises) * torch.exp(0.5 * logvar) + mu
    z_bern = reparameterize_bern(bern, next(noises))
    vae_out = torch.tanh(_stack7(z_norm * z_bern, P, 'decoder', opt.num_layer))
    x = vae_out
    for idx in range(num_body(P)):
        if opt.vae_levels == idx + 1:
            x = x.detach()
            if idx == 0:
                vae_out = x
        up = resize_linear_ac(x, level_shape(idx + 1, opt, dims))
        inp = up + next(noises) * noise_amp[idx + 1] if mode == 'rand' else up
        x = torch.tanh(_stack7(inp, P, 'body.%d' % idx, opt.num_layer) + up)
    return x, vae_out, (mu, logvar, bern)


def discriminator_forward(x, P, opt, training=True):
    h = _sn_block(x, P, 'head', training)
    for i in range(opt.num_layer):
        h = _sn_block(h, P, 'body.block%d' % i, training)
    return conv(h, P['tail.weight'], P['tail.bias'])


def discriminator_baselines_forward(x, P, opt, training=True):
    """WDiscriminatorBaselines.forward (networks_3d.py:184-210): input zero-padded by num_layer + 2 voxels per side, head
    = conv (padding padd_size) + LeakyReLU (NO norm), num_layer x [conv + BatchNorm(batch stats) + LeakyReLU], tail conv.
    Keys: head.conv.*, body.blockI.{conv,norm}.*, tail.*"""
    p = opt.num_layer + 2
    same = opt.padd_size == 1
    cv = conv if same else conv_valid
    h = leaky_relu(cv(F.pad(x, (p,) * 6), P['head.conv.weight'], P['head.conv.bias']))
    for i in range(opt.num_layer):
        h = (_bn_block if same else _bn_block_valid)(h, P, 'body.block%d' % i)
    return cv(h, P['tail.weight'], P['tail.bias'])


def any_discriminator_forward(x, P, opt, training=True):
    """WDiscriminator3D/2D (spectral norm) or WDiscriminatorBaselines (BatchNorm), told apart by the state-dict keys."""
    if 'head.conv.weight_orig' in P:
        return discriminator_forward(x, P, opt, training)
    return discriminator_baselines_forward(x, P, opt, training)


def gradient_penalty(PD, opt, real, fake, lam, alpha):
    """modules/utils.py:4-19 with the scalar alpha injected."""
    xhat = (alpha * real + (1 - alpha) * fake).detach().requires_grad_(True)
    out = any_discriminator_forward(xhat, PD, opt)
    g = torch.autograd.grad(out, xhat, grad_outputs=torch.ones_like(out), create_graph=True, retain_graph=True)[0]
    return ((g.norm(2, dim=1) - 1) ** 2).mean() * lam


# ----------------------------------------------------------------------------------------------- optimiser
def clip_grad_norm(grads, max_norm):
    """torch.nn.utils.clip_grad_norm_ (L2): coef = min(1, max_norm/(total+1e-6)); returns (total_norm, coef)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads)).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads:
        g.mul_(coef)
    return total, coef


def adam_step(p, g, state, lr, beta1=0.5, beta2=0.999, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad), in place on p."""
    state['step'] = state.get('step', 0) + 1
    t = state['step']
    m = state.setdefault('m', torch.zeros_like(p))
    v = state.setdefault('v', torch.zeros_like(p))
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def g_param_groups(PG, opt, scale_idx):
    """Adam parameter groups of the generator at stage `scale_idx` (train_video.py:57-86): list of (prefix, lr)."""
    nb = num_body(PG)
    groups = []
    if not opt.train_all:
        if opt.vae_levels < scale_idx + 1:
            depth = min(opt.train_depth, nb - opt.vae_levels + 1)
            blocks = list(range(nb))[-depth:]
            groups += [('body.%d.' % k, opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))) for i, k in enumerate(blocks)]
        else:
            groups += [('encode.', opt.lr_g * (opt.lr_scale ** scale_idx)), ('decoder.', opt.lr_g * (opt.lr_scale ** scale_idx))]
            blocks = list(range(nb))[-opt.train_depth:] if nb else []
            groups += [('body.%d.' % k, opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))) for i, k in enumerate(blocks)]
    else:
        if nb < opt.train_depth:
            groups += [('encode.', opt.lr_g * (opt.lr_scale ** scale_idx)), ('decoder.', opt.lr_g * (opt.lr_scale ** scale_idx))]
            groups += [('body.%d.' % k, opt.lr_g * (opt.lr_scale ** (nb - 1 - k))) for k in range(nb)]
        else:
            blocks = list(range(nb))[-opt.train_depth:]
            groups += [('body.%d.' % k, opt.lr_g * (opt.lr_scale ** (len(blocks) - 1 - i))) for i, k in enumerate(blocks)]
    return groups


def is_param(key):
    return key.endswith(('.weight', '.bias', '.weight_orig'))


# ----------------------------------------------------------------------------------------------- one train iteration
def train_step(PG, PD, opt, dims, scale_idx, real, real_zero, noise_init, noises, alpha, noise_amps, adam_g, adam_d):
    """One iteration of train() (train_video.py:111-202) at stage `scale_idx`, with every random draw injected:
    `noises` = iterator over N(0,1) tensors in reference draw order, `alpha` = the GP scalar.
    PG / PD: dicts key -> tensor (parameters have requires_grad=True); updated in place.  Returns a dict of losses,
    gradients (before clipping) and the clip coefficient."""
    out = {}
    gparams = {k: v for k, v in PG.items() if is_param(k)}
    for v in gparams.values():
        v.grad = None
    generated, generated_vae, (mu, logvar) = generator_forward(PG, opt, dims, real_zero, noise_amps, mode='rec', noises=noises)
    if opt.vae_levels >= scale_idx + 1:
        rec_vae_loss = mse(generated, real) + mse(generated_vae, real_zero)
        kl_loss = kl_criterion(mu, logvar)
        total = opt.rec_weight * rec_vae_loss + opt.kl_weight * kl_loss
        out.update(rec_vae_loss=rec_vae_loss.detach(), kl_loss=kl_loss.detach())
    else:
        dparams = {k: v for k, v in PD.items() if is_param(k)}
        for v in dparams.values():
            v.grad = None
        errD_real = -discriminator_forward(real, PD, opt).mean()
        fake, _ = generator_forward(PG, opt, dims, noise_init, noise_amps, noise_init=noise_init, mode='rand', noises=noises)
        errD_fake = discriminator_forward(fake.detach(), PD, opt).mean()
        gp = gradient_penalty(PD, opt, real, fake, opt.lambda_grad, alpha)
        errD_total = errD_real + errD_fake + gp
        dgrads = torch.autograd.grad(errD_total, list(dparams.values()), allow_unused=True)
        out['gradsD'] = {k: (g.clone() if g is not None else None) for k, g in zip(dparams.keys(), dgrads)}
        for (k, p), g in zip(dparams.items(), dgrads):
            if g is not None:
                with torch.no_grad():
                    adam_step(p, g, adam_d.setdefault(k, {}), opt.lr_d, opt.beta1)
        rec_loss = mse(generated, real)
        errG = -discriminator_forward(fake, PD, opt).mean() * opt.disc_loss_weight
        total = opt.rec_weight * rec_loss + errG
        out.update(errD_real=errD_real.detach(), errD_fake=errD_fake.detach(), gradient_penalty=gp.detach(),
                   rec_loss=rec_loss.detach(), errG=errG.detach(), fake=fake.detach())
    out.update(generated=generated.detach(), generated_vae=generated_vae.detach(), mu=mu.detach(), logvar=logvar.detach())
    keys = list(gparams.keys())
    grads = torch.autograd.grad(total, [gparams[k] for k in keys], allow_unused=True)
    out['total_loss'] = total.detach()
    out['gradsG'] = {k: (g.clone() if g is not None else None) for k, g in zip(keys, grads)}
    present = [g for g in grads if g is not None]
    total_norm, coef = clip_grad_norm(present, opt.grad_clip)
    out['total_norm'], out['clip_coef'] = total_norm, coef
    gmap = dict(zip(keys, grads))
    for prefix, lr in g_param_groups(PG, opt, scale_idx):
        for k in keys:
            if k.startswith(prefix) and gmap[k] is not None:
                with torch.no_grad():
                    adam_step(gparams[k], gmap[k], adam_g.setdefault(k, {}), lr, opt.beta1)
    return out


def noise_amp_for_stage(PG, opt, dims, scale_idx, real, real_zero, noise_amps, noises):
    """Noise-amplitude calibration at iteration 0 (train_video.py:131-145): appends to noise_amps in place."""
    if scale_idx == 0:
        noise_amps.append(1)
        return
    noise_amps.append(0)
    with torch.no_grad():
        rec, _, _ = generator_forward(PG, opt, dims, real_zero, noise_amps, mode='rec', noises=noises)
        rmse = torch.sqrt(mse(real, rec))
    noise_amps[-1] = opt.noise_amp_init * float(rmse) / opt.batch_size
