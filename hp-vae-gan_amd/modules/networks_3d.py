"""3-D (video) networks - mirror of the reference's modules/networks_3d.py surface for the hot path:
`getattr(networks_3d, opt.generator)(opt)` / `getattr(networks_3d, opt.discriminator)(opt)` (train_video.py:45,396-397)."""
from . import _nets
from .. import ops

__all__ = ['ConvBlock3D', 'ConvBlock3DSN', 'FeatureExtractor', 'Encode3DVAE', 'WDiscriminator3D', 'GeneratorHPVAEGAN',
           'GeneratorSG', 'GeneratorCSG', 'WDiscriminatorBaselines', 'weights_init', 'reparameterize', 'reparameterize_bern',
           'Encode3DVAE_nb', 'Encode3DVAE1x1', 'GeneratorVAE_nb']

weights_init = _nets.weights_init


def reparameterize(mu, logvar, training, eps=None):
    """eps*exp(0.5*logvar)+mu when training, pure N(0,1) otherwise (reference: networks_3d.py:29-35)."""
    from .. import utils
    if eps is None:
        eps = utils.generate_noise(ref=mu)
    return ops.Reparam.apply(mu, logvar, eps) if training else eps


class ConvBlock3D(_nets.ConvBlock):
    def __init__(self, in_channel, out_channel, ker_size, padding, stride, bn=True, act='lrelu'):
        super().__init__(3, in_channel, out_channel, ker_size, padding, stride, bn=bn, act=act)


class ConvBlock3DSN(_nets.ConvBlockSN):
    def __init__(self, in_channel, out_channel, ker_size, padding, stride, bn=True, act='lrelu'):
        super().__init__(3, in_channel, out_channel, ker_size, padding, stride, bn=bn, act=act)


class FeatureExtractor(_nets.FeatureExtractor):
    def __init__(self, in_channel, out_channel, ker_size, padding, stride, num_blocks=2, return_linear=False):
        super().__init__(3, in_channel, out_channel, ker_size, padding, stride, num_blocks=num_blocks,
                         return_linear=return_linear)


class Encode3DVAE(_nets.EncodeVAE):
    def __init__(self, opt, out_dim=None, num_blocks=2):
        super().__init__(3, opt, out_dim=out_dim, num_blocks=num_blocks)


class WDiscriminator3D(_nets.WDiscriminator):
    def __init__(self, opt):
        super().__init__(3, opt)


class GeneratorHPVAEGAN(_nets.GeneratorHPVAEGAN):
    def __init__(self, opt):
        super().__init__(3, opt)


class GeneratorSG(_nets.GeneratorSG):
    """SinGAN-3D baseline (BASELINE config 5; reference: networks_3d.py:272-322)."""


class GeneratorCSG(_nets.GeneratorCSG):
    """SinGAN-3D baseline with shared head / tail, the default of train_video_baselines.py (reference: networks_3d.py:213-269)."""


class WDiscriminatorBaselines(_nets.WDiscriminatorBaselines):
    """The baselines' BatchNorm critic on a zero-padded volume (reference: networks_3d.py:184-210)."""


def reparameterize_bern(x, training, eps=None):
    """log(x+1e-20) - log(-log(eps+1e-20)+1e-20), eps ~ U(0,1) when training; Bernoulli(0.5) otherwise (networks_3d.py:38-45)."""
    import torch
    if eps is None:
        eps = ops.uniform_(torch.empty_like(x))
    return ops.ReparamBern.apply(x, eps) if training else (eps < 0.5).to(x.dtype)


class Encode3DVAE_nb(_nets.EncodeVAE_nb):
    def __init__(self, opt, out_dim=None, num_blocks=2):
        super().__init__(3, opt, out_dim=out_dim, num_blocks=num_blocks)


class Encode3DVAE1x1(_nets.EncodeVAE1x1):
    def __init__(self, opt, out_dim=None):
        super().__init__(3, opt, out_dim=out_dim)


class GeneratorVAE_nb(_nets.GeneratorVAE_nb):
    def __init__(self, opt):
        super().__init__(3, opt)
