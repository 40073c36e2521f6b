"""Dimension-generic building blocks behind modules/networks_3d.py and modules/networks_2d.py.

Same module tree, child names, parameter/buffer names and registration order as the reference classes, so
state_dicts are interchangeable with the reference's; every forward runs on the gfx950
kernels through ops.*.  Parameters are created with the same torch initialisers, in the same order, as
nn.ConvNd.reset_parameters / nn.utils.spectral_norm / nn.BatchNormNd, so a seeded construction consumes the CPU
generator exactly like the reference does."""
import copy
import math

import torch
import torch.nn as nn
from torch.nn import init

from .. import ops
from .. import utils as hp_utils
from ..slab import conv_with_halo


def _kernel_shape(dims):
    return (3,) * dims


class Conv(nn.Module):
    """nn.Conv3d / nn.Conv2d(k=3, s=1, p=1) with bias (reference: networks_3d.py:51,175,341)."""

    def __init__(self, dims, in_channel, out_channel, ker_size=3, padding=1, stride=1):
        super().__init__()
        if stride != 1 or not ((ker_size == 3 and padding in (0, 1)) or (ker_size == 1 and padding == 0)):
            raise NotImplementedError("the MI355X path implements ker_size=3 with padding in {0, 1} and ker_size=1 with padding 0, stride=1")
        self.dims = dims
        self.ker_size = ker_size
        # a 1x1(x1) convolution (Encode3DVAE1x1, networks_3d.py:141-160) = the 3x3(x3) 'same' conv of its centre-embedded weight
        self.padding = padding if ker_size == 3 else 1
        self.weight = nn.Parameter(torch.empty(out_channel, in_channel, *((ker_size,) * dims)))
        self.bias = nn.Parameter(torch.empty(out_channel))
        init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        fan_in = in_channel * ker_size ** dims
        bound = 1 / math.sqrt(fan_in)
        init.uniform_(self.bias, -bound, bound)

    # multi-GPU: slab.Halo when this layer's level is cut into row slabs over neighbouring ranks (multigpu.py sets it)
    halo = None

    def forward(self, x, act=False, in_act=False, in_bits=None):
        """in_act: x is the activated output of a spectral-norm block that left its LeakyReLU backward to this conv
        (ops.Conv: the mask rides in this layer's backward-data epilogue); in_bits: its 1-bit form from the producer."""
        w = self.weight if self.ker_size == 3 else embed_center(self.weight, self.dims)
        if self.halo is not None:
            assert self.padding == 1, "row slabs are implemented for the 'same' convolutions of the HP-VAE-GAN path"
            return conv_with_halo(x, self.halo, lambda xe: ops.Conv.apply(xe, w, self.bias, act, in_act))
        y = ops.Conv.apply(x, w, self.bias, act, in_act, False, in_bits if self.padding == 1 else None)
        return y if self.padding == 1 else crop_border(y)


def embed_center(w, dims):
    """[Co, Ci, 1, 1(, 1)] -> [Co, Ci, 3, 3(, 3)], zero except the centre tap (data movement; its backward is the slice)."""
    return torch.nn.functional.pad(w, (1, 1) * dims)


def crop_border(y):
    """padding=0 ("valid") convolution of the SinGAN baselines (networks_3d.py:285-290) = the zero-padded conv with
    its outermost voxel shell removed (every kept output only ever saw real inputs).  Pure data movement (torch
    slicing; autograd's backward of it is the zero pad that the backward-data pass needs)."""
    idx = (slice(None), slice(None)) + (slice(1, -1),) * (y.dim() - 2)
    return y[idx].contiguous()


class SNConv(nn.Module):
    """nn.utils.spectral_norm(nn.ConvNd(k=3, p=1)) (reference: networks_3d.py:63): parameters `bias`, `weight_orig`,
    buffers `weight_u`, `weight_v`; one power iteration per training-mode forward, eps 1e-12."""

    def __init__(self, dims, in_channel, out_channel, ker_size=3, padding=1, stride=1):
        super().__init__()
        if stride != 1 or not ((ker_size == 3 and padding == 1) or (ker_size == 1 and padding == 0)):
            raise NotImplementedError("the MI355X path implements ker_size=3 / padding=1 (reference defaults) and ker_size=1 / padding=0, stride=1")
        self.dims = dims
        self.ker_size = ker_size
        weight = torch.empty(out_channel, in_channel, *((ker_size,) * dims))
        init.kaiming_uniform_(weight, a=math.sqrt(5))
        bias = torch.empty(out_channel)
        fan_in = in_channel * ker_size ** dims
        bound = 1 / math.sqrt(fan_in)
        init.uniform_(bias, -bound, bound)
        self.bias = nn.Parameter(bias)
        self.weight_orig = nn.Parameter(weight)
        h, w = out_channel, weight.numel() // out_channel
        u = torch.nn.functional.normalize(weight.new_empty(h).normal_(0, 1), dim=0, eps=1e-12)
        v = torch.nn.functional.normalize(weight.new_empty(w).normal_(0, 1), dim=0, eps=1e-12)
        self.register_buffer('weight_u', u)
        self.register_buffer('weight_v', v)

    def effective_weight(self):
        return ops.SpectralNormWeight.apply(self.weight_orig, self.weight_u, self.weight_v, self.training, 1e-12)

    halo = None  # as Conv.halo

    def forward(self, x, act=False, weight=None, in_act=False, mask_by_consumer=False, in_bits=None):
        """weight: the effective weight when the caller computed it for all of its spectral-norm layers at once
        (sn_weights below); None: this layer runs its own power iteration.  in_act / mask_by_consumer: see ops.Conv
        (the LeakyReLU backward of a chain of activated convs rides in the consumer's backward-data epilogue).
        Returns (y, bits): bits = the 1-bit mask of y for the consumer (None on row slabs or without mask_by_consumer)."""
        w = weight if weight is not None else self.effective_weight()
        if self.ker_size == 1:
            w = embed_center(w, self.dims)
        if self.halo is not None:
            return conv_with_halo(x, self.halo, lambda xe: ops.Conv.apply(xe, w, self.bias, act, in_act, mask_by_consumer)), None
        return ops.Conv.apply_bits(x, w, self.bias, act, in_act, mask_by_consumer, in_bits)


def sn_weights(convs, x=None):
    """Effective weights of a list of SNConv layers through ONE launch (ops.SpectralNormWeightBatch): the same arithmetic
    and buffer updates as calling effective_weight() on each, in order.  x: the activation the chain runs on (its
    geometry decides which operand forms the packs carry, ops.pack_weight)."""
    out = []
    for i in range(0, len(convs), ops.SN_BATCH_MAX):
        part = convs[i:i + ops.SN_BATCH_MAX]
        args = []
        for m in part:
            args += [m.weight_orig, m.weight_u, m.weight_v]
        out += list(ops.SpectralNormWeightBatch.apply(part[0].training, 1e-12, *args))
    if all(tuple(w.shape[2:]) == (3,) * (w.dim() - 2) for w in out):
        geom_k = None
        if x is not None:
            B, _, T, H, W = ops.geom(x)
            geom_k = (B, T, H, W)
        ops.prepack_weights(out, geom_k=geom_k)   # forward and backward-data packs of the equal-shaped layers: one launch
    return out


class BatchNorm(nn.Module):
    """nn.BatchNorm3d / 2d parameters and buffers (eps 1e-5, momentum 0.1, affine, track_running_stats)."""

    def __init__(self, num_features):
        super().__init__()
        self.eps = 1e-5
        self.momentum = 0.1
        self.weight = nn.Parameter(torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer('running_mean', torch.zeros(num_features))
        self.register_buffer('running_var', torch.ones(num_features))
        self.register_buffer('num_batches_tracked', torch.tensor(0, dtype=torch.long))
        # train-mode forwards not yet added to the buffer above.  Nothing on the path reads the count (momentum is a
        # constant), so instead of one int64 add kernel per BatchNorm per forward (~75 launches an iteration) the count
        # is kept on the host and written to the buffer when somebody asks for it (state_dict / flush_counters).
        self.pending_batches = 0
        self.register_state_dict_pre_hook(BatchNorm._flush_hook)
        self._register_load_state_dict_pre_hook(self._loading_hook)

    @staticmethod
    def _flush_hook(module, prefix, keep_vars):
        module.flush_counter()

    def _loading_hook(self, *args):
        self.pending_batches = 0

    def flush_counter(self):
        if self.pending_batches:
            self.num_batches_tracked.add_(self.pending_batches)
            self.pending_batches = 0

    # multi-GPU: (allreduce callable, number of ranks, total count or None) when the batch - and, on row slabs, the
    # image - is split over a process group (multigpu.py sets it on the BatchNorms of a sharded generator pass): the
    # statistics are over `total count` elements per channel (None: ranks x the local count); None = everything is local
    sync = None
    # > 1 while the batch holds that many independent passes back to back (GeneratorHPVAEGAN.forward_pair): statistics per pass
    groups = 1

    def forward(self, r, lrelu=True):
        if self.training:
            self.pending_batches += self.groups
            if self.sync is not None:
                return ops.BNActSync.apply(r, self.weight, self.bias, self.running_mean, self.running_var, self.momentum,
                                           self.eps, lrelu, self.sync[0], self.sync[1],
                                           self.sync[2] if len(self.sync) > 2 else None)
            return ops.BNAct.apply(r, self.weight, self.bias, self.running_mean, self.running_var, self.momentum, self.eps,
                                   lrelu, self.groups)
        with torch.no_grad():
            scale = self.weight / torch.sqrt(self.running_var + self.eps)
            shift = self.bias - self.running_mean * scale
        return ops.AffineAct.apply(r, scale, shift, lrelu)


class ConvBlock(nn.Module):
    """Conv -> BatchNorm (batch statistics) -> LeakyReLU(0.2) (reference ConvBlock3D/2D: networks_3d.py:48-56).
    bn=False / act=None give the plain conv used for the encoder's mu / logvar heads (networks_3d.py:99-100)."""

    def __init__(self, dims, in_channel, out_channel, ker_size, padding, stride, bn=True, act='lrelu'):
        super().__init__()
        if act not in ('lrelu', None):
            raise NotImplementedError("only LeakyReLU(0.2) ('lrelu') is implemented on the MI355X path")
        self.conv = Conv(dims, in_channel, out_channel, ker_size, padding, stride)
        if bn:
            self.norm = BatchNorm(out_channel)
        self.has_bn = bn
        self.act = act

    def forward(self, x, in_act=False, in_bits=None):
        if self.has_bn:
            return self.norm(self.conv(x, in_act=in_act, in_bits=in_bits), lrelu=self.act is not None)
        return self.conv(x, act=self.act is not None, in_act=in_act, in_bits=in_bits)


class ConvBlockSN(nn.Module):
    """spectral_norm(Conv) -> LeakyReLU(0.2); NO BatchNorm (reference ConvBlock3DSN/2DSN: networks_3d.py:59-70;
    its `bn` flag selects spectral norm).  The bn=False reflect-padding branch is dead code in the reference."""

    def __init__(self, dims, in_channel, out_channel, ker_size, padding, stride, bn=True, act='lrelu'):
        super().__init__()
        if not bn:
            raise NotImplementedError("ConvBlockSN(bn=False) (reflect padding) is never reached on the reference path")
        if act not in ('lrelu', None):
            raise NotImplementedError("only LeakyReLU(0.2) ('lrelu') is implemented on the MI355X path")
        self.conv = SNConv(dims, in_channel, out_channel, ker_size, padding, stride)
        self.act = act

    def forward(self, x, weight=None, in_act=False, mask_by_consumer=False, in_bits=None, with_bits=False):
        """with_bits: return (y, bits) - the 1-bit LeakyReLU mask for the conv that consumes y (see SNConv.forward)."""
        y, bits = self.conv(x, act=self.act is not None, weight=weight, in_act=in_act,
                            mask_by_consumer=mask_by_consumer and self.act is not None, in_bits=in_bits)
        return (y, bits) if with_bits else y


class FeatureExtractor(nn.Sequential):
    """num_blocks+1 spectral-norm blocks (reference: networks_3d.py:73-85)."""

    def __init__(self, dims, in_channel, out_channel, ker_size, padding, stride, num_blocks=2, return_linear=False):
        super().__init__()
        if return_linear:
            raise NotImplementedError("return_linear=True is never used on the reference path")
        self.add_module('conv_block_0', ConvBlockSN(dims, in_channel, out_channel, ker_size, padding, stride))
        for i in range(num_blocks - 1):
            self.add_module('conv_block_{}'.format(i + 1), ConvBlockSN(dims, out_channel, out_channel, ker_size, padding, stride))
        self.add_module('conv_block_{}'.format(num_blocks), ConvBlockSN(dims, out_channel, out_channel, ker_size, padding, stride))

    def forward(self, x, mask_by_consumer=False):
        """mask_by_consumer: every consumer of the returned features is a conv called with in_act=True (EncodeVAE's mu /
        logvar heads), so the last block leaves its LeakyReLU backward to them like the inner blocks do."""
        blocks = list(self)
        last = len(blocks) - 1
        bits = None
        for i, (blk, w) in enumerate(zip(blocks, sn_weights([b.conv for b in blocks], x))):   # all power iterations in one launch
            x, bits = blk(x, weight=w, in_act=i > 0, mask_by_consumer=(i < last or mask_by_consumer), in_bits=bits, with_bits=True)
        return (x, bits) if mask_by_consumer else x


class EncodeVAE(nn.Module):
    """features -> (mu, logvar) (reference Encode3DVAE/2DVAE: networks_3d.py:88-107)."""

    def __init__(self, dims, opt, out_dim=None, num_blocks=2):
        super().__init__()
        if out_dim is None:
            output_dim = opt.nfc
        else:
            assert type(out_dim) is int
            output_dim = out_dim
        self.features = FeatureExtractor(dims, opt.nc_im, opt.nfc, opt.ker_size, opt.ker_size // 2, 1, num_blocks=num_blocks)
        self.mu = ConvBlock(dims, opt.nfc, output_dim, opt.ker_size, opt.ker_size // 2, 1, bn=False, act=None)
        self.logvar = ConvBlock(dims, opt.nfc, output_dim, opt.ker_size, opt.ker_size // 2, 1, bn=False, act=None)

    def forward(self, x):
        features, bits = self.features(x, mask_by_consumer=True)
        return self.mu(features, in_act=True, in_bits=bits), self.logvar(features, in_act=True, in_bits=bits)


def _seven_conv_stack(dims, in_channel, N, opt, padding):
    """head ConvBlock(in->N) + num_layer ConvBlock(N->N) + tail conv(N->nc_im) (decoder and per-level body blocks)."""
    seq = nn.Sequential()
    seq.add_module('head', ConvBlock(dims, in_channel, N, opt.ker_size, padding, stride=1))
    for i in range(opt.num_layer):
        seq.add_module('block%d' % i, ConvBlock(dims, N, N, opt.ker_size, padding, stride=1))
    seq.add_module('tail', Conv(dims, N, opt.nc_im, opt.ker_size, opt.ker_size // 2, 1))
    return seq


class WDiscriminator(nn.Module):
    """Patch critic: SN(3->N)+LReLU, num_layer x SN(N->N)+LReLU, conv(N->1) (reference: networks_3d.py:163-181)."""

    def __init__(self, dims, opt):
        super().__init__()
        self.opt = opt
        N = int(opt.nfc)
        self.head = ConvBlockSN(dims, opt.nc_im, N, opt.ker_size, opt.ker_size // 2, stride=1, bn=True, act='lrelu')
        self.body = nn.Sequential()
        for i in range(opt.num_layer):
            self.body.add_module('block%d' % i, ConvBlockSN(dims, N, N, opt.ker_size, opt.ker_size // 2, stride=1, bn=True,
                                                            act='lrelu'))
        self.tail = Conv(dims, N, 1, opt.ker_size, 1, 1)

    def forward(self, x):
        blocks = [self.head] + list(self.body)
        # every activation of the chain has exactly one consumer, the next conv: its LeakyReLU backward rides in that conv's
        # backward-data epilogue (ops.Conv in_act / mask_by_consumer) instead of being a pass of its own
        bits = None
        for i, (blk, w) in enumerate(zip(blocks, sn_weights([b.conv for b in blocks], x))):   # all power iterations in one launch
            x, bits = blk(x, weight=w, in_act=i > 0, mask_by_consumer=True, in_bits=bits, with_bits=True)
        return self.tail(x, in_act=True, in_bits=bits)


class GeneratorHPVAEGAN(nn.Module):
    """Encoder + VAE decoder + growing list of per-level refinement blocks (reference: networks_3d.py:325-406,
    networks_2d.py:188-269).  Differences 3-D vs 2-D kept exactly: the 3-D path injects level noise only for levels
    idx+1 >= vae_levels (networks_3d.py:398), the 2-D path at every level in 'rand' mode (networks_2d.py:261).

    `noise_source`: optional callable(ref_tensor) -> N(0,1) tensor of ref's shape, used for the reparameterisation
    eps and the per-level noise (tests inject recorded noise; default = utils.generate_noise on the device)."""

    def __init__(self, dims, opt):
        super().__init__()
        self.dims = dims
        self.opt = opt
        N = int(opt.nfc)
        self.N = N
        self.encode = EncodeVAE(dims, opt, out_dim=opt.latent_dim, num_blocks=opt.enc_blocks)
        self.decoder = _seven_conv_stack(dims, opt.latent_dim, N, opt, opt.padd_size)
        self.body = nn.ModuleList([])
        self.noise_source = None
        self.slab = None  # multi-GPU: slab.SlabPlan when the upper levels of this pass are cut into row slabs

    def init_next_stage(self):
        if len(self.body) == 0:
            self.body.append(_seven_conv_stack(self.dims, self.opt.nc_im, self.N, self.opt, self.opt.padd_size))
        else:
            self.body.append(copy.deepcopy(self.body[-1]))

    def _noise_like(self, ref):
        if self.noise_source is not None:
            return self.noise_source(ref)
        return hp_utils.generate_noise(ref=ref)

    def _reparameterize(self, mu, logvar):
        if self.training:
            return ops.Reparam.apply(mu, logvar, self._noise_like(mu))
        return self._noise_like(mu)

    def forward(self, video, noise_amp, noise_init=None, sample_init=None, mode='rand', stop_idx=None):
        """stop_idx (not in the reference; default None = all levels): run the refinement only up to body[stop_idx - 1] -
        the level-pipeline schedule (pipeline.py) keeps the remaining levels on other GPUs."""
        if sample_init is not None:
            assert len(self.body) > sample_init[0], "Strating index must be lower than # of body blocks"

        if noise_init is None:
            mu, logvar = self.encode(video)
            z_vae = self._reparameterize(mu, logvar)
        else:
            z_vae = noise_init

        vae_out = ops.TanhRes.apply(self.decoder(z_vae), None)

        if sample_init is not None:
            x_prev_out = self.refinement_layers(sample_init[0], sample_init[1], noise_amp, mode, stop_idx)
        else:
            x_prev_out = self.refinement_layers(0, vae_out, noise_amp, mode, stop_idx)

        if noise_init is None:
            return x_prev_out, vae_out, (mu, logvar)
        return x_prev_out, vae_out

    def forward_pair(self, video, noise_amp, noise_init):
        """The two generator passes of a GAN-stage iteration - forward(video, amp, mode='rec') and forward(noise_init, amp,
        noise_init=noise_init, mode='rand') (train_video.py:147,175) - as ONE pass over the concatenated batch: both use the
        same weights and differ only in their inputs, in the level noise (zero for the rec half) and in BatchNorm's batch
        (each half is normalised with its own statistics, running statistics updated rec first, then rand: BNAct groups).
        Half the launches and twice the GEMM size where an iteration is a chain of small kernels.
        Returns (generated, fake, vae_out of the rec half, (mu, logvar)); same random draws in the same order."""
        assert self.training and self.slab is None
        B = video.shape[0]
        mu, logvar = self.encode(video)
        z = ops.Concat2.apply(self._reparameterize(mu, logvar), noise_init)
        bns = [m for part in (self.decoder, self.body) for m in part.modules() if isinstance(m, BatchNorm)]
        for m in bns:
            m.groups = 2
        try:
            vae_out = ops.TanhRes.apply(self.decoder(z), None)
            x = vae_out
            for idx, block in enumerate(self.body):
                if self.opt.vae_levels == idx + 1 and not self.opt.train_all:
                    x.detach_()
                size = self._level_size(idx + 1)
                if self.dims == 2 or self.opt.vae_levels <= idx + 1:   # levels where the rand pass injects noise
                    if self.noise_source is None:
                        # noise generated inside the resize kernel, for the rand half only
                        up, up_noisy = ops.UpsampleACNoise.apply(x, tuple(size), float(noise_amp[idx + 1]), B)
                    else:
                        ref = x.new_empty((B, x.shape[1], *size))
                        noise = ops.concat_batch([B, self._noise_like(ref)])   # zero noise for the rec half
                        up, up_noisy = ops.UpsampleAC.apply(x, tuple(size), noise, float(noise_amp[idx + 1]))
                else:
                    up = up_noisy = ops.UpsampleAC.apply(x, tuple(size), None, 0.0)
                x = ops.TanhRes.apply(block(up_noisy), up)
        finally:
            for m in bns:
                m.groups = 1
        generated, fake = ops.SplitBatch.apply(x, B)
        return generated, fake, vae_out.detach()[:B], (mu, logvar)

    def _level_size(self, index):
        if self.dims == 3:
            return hp_utils.images.level_shape_3d(index, self.opt)
        return hp_utils.images.level_shape_2d(index, self.opt)

    def _detach_before(self, idx):
        """does the reference cut the graph in front of body[idx]?  (networks_3d.py:391-392)"""
        return self.opt.vae_levels == idx + 1 and not self.opt.train_all

    def _inject_noise(self, idx, mode):
        """does body[idx] get level noise?  3-D: only from the first GAN level on (networks_3d.py:398); 2-D: every level"""
        return mode == 'rand' and (self.dims == 2 or self.opt.vae_levels <= idx + 1)

    def refinement_layers(self, start_idx, x_prev_out, noise_amp, mode, stop_idx=None):
        for idx, block in enumerate(self.body[start_idx:stop_idx], start_idx):
            if self._detach_before(idx):
                x_prev_out.detach_()
            size = self._level_size(idx + 1)
            inject = self._inject_noise(idx, mode)
            if inject and self.noise_source is None:
                up, up_noisy = ops.UpsampleACNoise.apply(x_prev_out, tuple(size), float(noise_amp[idx + 1]), 0)   # noise made in the kernel
            elif inject:
                ref = x_prev_out.new_empty((x_prev_out.shape[0], x_prev_out.shape[1], *size))
                noise = self._noise_like(ref)
                up, up_noisy = ops.UpsampleAC.apply(x_prev_out, tuple(size), noise, float(noise_amp[idx + 1]))
            else:
                up = up_noisy = ops.UpsampleAC.apply(x_prev_out, tuple(size), None, 0.0)
            slab = self.slab if (self.slab is not None and self.slab.covers(idx + 1)) else None
            if slab is not None:
                # this level runs on this rank's row slab (its convs carry the halo swaps, its BatchNorms sum over the
                # slab ranks); the 3-channel upsampled input was cheap to make whole, and the whole output is put back
                # together for the next level's upsample.  The last level stays a slab: the losses are separable.
                up = slab.cut(up)
                up_noisy = slab.cut(up_noisy) if inject else up
            x_prev_out = ops.TanhRes.apply(block(up_noisy), up)
            if slab is not None and idx + 1 < len(self.body):
                x_prev_out = slab.gather(x_prev_out, size[-2])
        return x_prev_out


class _GlobalAvgPool(nn.Module):
    """nn.AdaptiveAvgPool3d(1) / nn.AdaptiveAvgPool2d(1) (no parameters; keeps the reference's Sequential numbering)."""

    def forward(self, x):
        return ops.GlobalAvgPool.apply(x)


class EncodeVAE_nb(nn.Module):
    """Encode3DVAE_nb / Encode2DVAE_nb (reference: networks_3d.py:110-138, networks_2d.py:115-143): features gated by a
    sigmoid map `bern`, then mu / logvar convs followed by GLOBAL average pooling ([B, latent, 1, 1, 1] codes)."""

    def __init__(self, dims, opt, out_dim=None, num_blocks=2):
        super().__init__()
        if out_dim is None:
            output_dim = opt.nfc
        else:
            assert type(out_dim) is int
            output_dim = out_dim
        k, p = opt.ker_size, opt.ker_size // 2
        self.features = FeatureExtractor(dims, opt.nc_im, opt.nfc, k, p, 1, num_blocks=num_blocks)
        self.mu = nn.Sequential(ConvBlock(dims, opt.nfc, output_dim, k, p, 1, bn=False, act=None), _GlobalAvgPool())
        self.logvar = nn.Sequential(ConvBlock(dims, opt.nfc, output_dim, k, p, 1, bn=False, act=None), _GlobalAvgPool())
        self.bern = ConvBlock(dims, opt.nfc, 1, k, p, 1, bn=False, act=None)

    def forward(self, x):
        features = self.features(x)
        features, bern = ops.Gate.apply(features, self.bern(features))     # bern = sigmoid(conv), features *= bern
        return self.mu(features), self.logvar(features), bern


class EncodeVAE1x1(nn.Module):
    """Encode3DVAE1x1 (reference: networks_3d.py:141-160, networks_2d.py:146-165; instantiated by nothing in the reference):
    the encoder with 1x1(x1) kernels - run as centre-embedded 3x3(x3) convs."""

    def __init__(self, dims, opt, out_dim=None):
        super().__init__()
        if out_dim is None:
            output_dim = opt.nfc
        else:
            assert type(out_dim) is int
            output_dim = out_dim
        self.features = FeatureExtractor(dims, opt.nc_im, opt.nfc, 1, 0, 1, num_blocks=2)
        self.mu = ConvBlock(dims, opt.nfc, output_dim, 1, 0, 1, bn=False, act=None)
        self.logvar = ConvBlock(dims, opt.nfc, output_dim, 1, 0, 1, bn=False, act=None)

    def forward(self, x):
        features = self.features(x)
        return self.mu(features), self.logvar(features)


class GeneratorVAE_nb(GeneratorHPVAEGAN):
    """GeneratorVAE_nb (reference: networks_3d.py:409-485, networks_2d.py:272-348): the HP-VAE-GAN generator with the
    normal x relaxed-Bernoulli latent of EncodeVAE_nb.  Differences to GeneratorHPVAEGAN kept exactly: forward takes
    (noise_init_norm, noise_init_bern) and returns (mu, logvar, bern); the decoder input is code [B, C, 1, ..] x map
    [B, 1, ...]; the refinement detaches at vae_levels regardless of train_all and injects noise at EVERY level in 'rand' mode.
    No trainer of the reference can drive it (train_video.py:140 unpacks two statistics)."""

    def __init__(self, dims, opt):
        super().__init__(dims, opt)
        self.encode = EncodeVAE_nb(dims, opt, out_dim=opt.latent_dim, num_blocks=opt.enc_blocks)

    def _detach_before(self, idx):
        return self.opt.vae_levels == idx + 1

    def _inject_noise(self, idx, mode):
        return mode == 'rand'

    def _uniform_like(self, ref):
        if self.noise_source is not None:
            return self.noise_source(ref)
        return ops.uniform_(torch.empty_like(ref))

    def _reparameterize_bern(self, x):
        """reference: networks_3d.py:38-45"""
        if self.training:
            return ops.ReparamBern.apply(x, self._uniform_like(x))
        return (self._uniform_like(x) < 0.5).to(x.dtype)      # zeros_like(x).bernoulli_(): p = 0.5

    def forward(self, video, noise_amp, noise_init_norm=None, noise_init_bern=None, sample_init=None, mode='rand'):
        if sample_init is not None:
            assert len(self.body) > sample_init[0], "Strating index must be lower than # of body blocks"
        if noise_init_norm is None:
            mu, logvar, bern = self.encode(video)
            z_vae_norm = self._reparameterize(mu, logvar)
            z_vae_bern = self._reparameterize_bern(bern)
        else:
            z_vae_norm = noise_init_norm
            z_vae_bern = noise_init_bern
        vae_out = ops.TanhRes.apply(self.decoder(ops.CodeTimesMap.apply(z_vae_norm, z_vae_bern)), None)
        if sample_init is not None:
            x_prev_out = self.refinement_layers(sample_init[0], sample_init[1], noise_amp, mode)
        else:
            x_prev_out = self.refinement_layers(0, vae_out, noise_amp, mode)
        if noise_init_norm is None:
            return x_prev_out, vae_out, (mu, logvar, bern)
        return x_prev_out, vae_out

    def forward_pair(self, *a, **k):
        raise NotImplementedError("the merged rec + rand pass is built for GeneratorHPVAEGAN")


def weights_init(m):
    """N(0, 0.02) conv weights, N(1, 0.02) BatchNorm scale, zero BatchNorm shift (reference: networks_3d.py:9-15;
    applied to the SinGAN baselines only)."""
    if isinstance(m, Conv):
        m.weight.data.normal_(0.0, 0.02)
    elif isinstance(m, (BatchNorm, nn.modules.batchnorm._BatchNorm)):
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)


class GeneratorSG(nn.Module):
    """SinGAN-3D baseline generator (reference: networks_3d.py:272-322, BASELINE config 5): every stage is a 7-conv
    stack of VALID convolutions on a volume padded by num_layer+2 = 7 voxels per side; stage 0 maps noise to video,
    stage k >= 1 refines the upsampled previous output (rand: the previous output is resized straight to the padded
    size and noise is added; rec: zero padding).  `noise_source` as in GeneratorHPVAEGAN."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        N = int(opt.nfc)
        self.pad = opt.num_layer + 2
        self.body = nn.ModuleList([])
        first = nn.Sequential()
        first.add_module('head', ConvBlock(3, opt.nc_im, N, opt.ker_size, 0, 1))
        for i in range(opt.num_layer):
            first.add_module('block%d' % i, ConvBlock(3, N, N, opt.ker_size, 0, 1))
        first.add_module('tail', Conv(3, N, opt.nc_im, opt.ker_size, 0, 1))
        self.body.append(first)
        self.apply(weights_init)
        self.noise_source = None

    def init_next_stage(self):
        self.body.append(copy.deepcopy(self.body[-1]))

    def _zero_pad(self, x):
        p = self.pad
        return torch.nn.functional.pad(x, (p, p, p, p, p, p))

    def forward(self, noise_init, noise_amp, mode='rand', start=0, stop=None):
        """start / stop (not in the reference; defaults = every stage): run body[start:stop] only.  For start > 0
        `noise_init` is the previous stage's raw output (before the tanh); the raw output is returned unless the last
        stage is included.  The level pipeline (pipeline.BaselinePipelineTrainer) keeps the other stages on other GPUs."""
        nb = len(self.body)
        stop = nb if stop is None else stop
        x_prev_out = self.body[0](self._zero_pad(noise_init)) if start == 0 else noise_init
        for idx in range(max(start, 1), stop):
            block = self.body[idx]
            x_prev_out = ops.TanhRes.apply(x_prev_out, None)
            size = hp_utils.images.level_shape_3d(idx, self.opt)
            up = ops.UpsampleAC.apply(x_prev_out, tuple(size), None, 0.0)
            if mode == 'rand':
                big = tuple(s + 2 * self.pad for s in size)
                if self.noise_source is None:
                    _, up2_noisy = ops.UpsampleACNoise.apply(x_prev_out, big, float(noise_amp[idx]), 0)
                else:
                    ref = x_prev_out.new_empty((x_prev_out.shape[0], x_prev_out.shape[1], *big))
                    _, up2_noisy = ops.UpsampleAC.apply(x_prev_out, big, self.noise_source(ref), float(noise_amp[idx]))
                x_prev = block(up2_noisy)
            else:
                x_prev = block(self._zero_pad(up))
            x_prev_out = ops.Add.apply(x_prev, up)
        return ops.TanhRes.apply(x_prev_out, None) if stop == nb else x_prev_out


class GeneratorCSG(nn.Module):
    """The baselines script's default generator (reference: networks_3d.py:213-269; `--generator GeneratorCSG`,
    train_video_baselines.py:232): ONE head block (noise -> nfc features) and ONE tail conv + tanh around a growing list
    of stages, each num_layer VALID conv blocks on a volume padded by num_layer voxels per side; the nfc-channel
    features - not images - are upsampled from stage to stage and added back without a tanh.  `noise_source` as in
    GeneratorHPVAEGAN.  State-dict keys: head.*, body.k.blockI.*, tail.0.{weight,bias} (the reference's tail is
    Sequential(Conv3d, Tanh))."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        N = int(opt.nfc)
        self.pad = opt.num_layer
        self.head = ConvBlock(3, opt.nc_im, N, opt.ker_size, 0, 1)
        self.body = nn.ModuleList([])
        first = nn.Sequential()
        for i in range(opt.num_layer):
            first.add_module('block%d' % i, ConvBlock(3, N, N, opt.ker_size, 0, 1))
        self.body.append(first)
        self.tail = nn.Sequential(Conv(3, N, opt.nc_im, opt.ker_size, 0, 1))
        self.apply(weights_init)
        self.noise_source = None

    def init_next_stage(self):
        self.body.append(copy.deepcopy(self.body[-1]))

    @staticmethod
    def _zero_pad(x, p):
        return torch.nn.functional.pad(x, (p,) * 6)

    def forward(self, noise_init, noise_amp, mode='rand'):
        p = self.pad
        x_prev_out = self.body[0](self._zero_pad(self.head(self._zero_pad(noise_init, 1)), p))
        for idx, block in enumerate(self.body[1:], 1):
            size = hp_utils.images.level_shape_3d(idx, self.opt)
            up = ops.UpsampleAC.apply(x_prev_out, tuple(size), None, 0.0)
            if mode == 'rand':
                big = tuple(s + 2 * p for s in size)
                if self.noise_source is None:
                    _, up2_noisy = ops.UpsampleACNoise.apply(x_prev_out, big, float(noise_amp[idx]), 0)
                else:
                    ref = x_prev_out.new_empty((x_prev_out.shape[0], x_prev_out.shape[1], *big))
                    _, up2_noisy = ops.UpsampleAC.apply(x_prev_out, big, self.noise_source(ref), float(noise_amp[idx]))
                x_prev = block(up2_noisy)
            else:
                x_prev = block(self._zero_pad(up, p))
            x_prev_out = ops.Add.apply(x_prev, up)
        return ops.TanhRes.apply(self.tail(self._zero_pad(x_prev_out, 1)), None)


class WDiscriminatorBaselines(nn.Module):
    """The baselines' own critic (reference: networks_3d.py:184-210; `--discriminator WDiscriminatorBaselines`): input
    zero-padded by num_layer + 2 voxels per side, head conv + LeakyReLU (no norm), num_layer x [conv + BatchNorm +
    LeakyReLU], tail conv, weights_init.  The one place on the path where the WGAN-GP needs BatchNorm's SECOND derivative:
    ops.BNAct's backward runs as the differentiable ops.BNActBwd while the penalty's graph is recorded, and its double
    backward is the gfx950 kernel pair hpvg_bn_act_bwd2_f32 - every arithmetic op of this critic is a kernel of libhpvg."""

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        N = int(opt.nfc)
        self.pad = opt.num_layer + 2
        self.head = ConvBlock(3, opt.nc_im, N, opt.ker_size, opt.padd_size, 1, bn=False, act='lrelu')
        self.body = nn.Sequential()
        for i in range(opt.num_layer):
            self.body.add_module('block%d' % i, ConvBlock(3, N, N, opt.ker_size, opt.padd_size, 1, bn=True, act='lrelu'))
        self.tail = Conv(3, N, 1, opt.ker_size, opt.padd_size, 1)
        self.apply(weights_init)

    def forward(self, x):
        x = torch.nn.functional.pad(x, (self.pad,) * 6)
        return self.tail(self.body(self.head(x)))
