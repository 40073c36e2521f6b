"""2-D (image) networks - mirror of the reference's modules/networks_2d.py surface for the hot path:
`getattr(networks_2d, opt.generator)(opt)` / `getattr(networks_2d, opt.discriminator)(opt)` (train_image.py:41,418-419)."""
from . import _nets
from .networks_3d import reparameterize, reparameterize_bern  # noqa: F401  (same functions in the reference, networks_2d.py:36-50)

__all__ = ['ConvBlock2D', 'ConvBlock2DSN', 'FeatureExtractor', 'Encode2DVAE', 'WDiscriminator2D', 'GeneratorHPVAEGAN',
           'reparameterize', 'reparameterize_bern', 'Encode2DVAE_nb', 'Encode3DVAE1x1', 'GeneratorVAE_nb']


class ConvBlock2D(_nets.ConvBlock):
    def __init__(self, in_channel, out_channel, ker_size, padding, stride, bn=True, act='lrelu'):
        super().__init__(2, in_channel, out_channel, ker_size, padding, stride, bn=bn, act=act)


class ConvBlock2DSN(_nets.ConvBlockSN):
    def __init__(self, in_channel, out_channel, ker_size, padding, stride, bn=True, act='lrelu'):
        super().__init__(2, in_channel, out_channel, ker_size, padding, stride, bn=bn, act=act)


class FeatureExtractor(_nets.FeatureExtractor):
    def __init__(self, in_channel, out_channel, ker_size, padding, stride, num_blocks=2, return_linear=False):
        super().__init__(2, in_channel, out_channel, ker_size, padding, stride, num_blocks=num_blocks,
                         return_linear=return_linear)


class Encode2DVAE(_nets.EncodeVAE):
    def __init__(self, opt, out_dim=None, num_blocks=2):
        super().__init__(2, opt, out_dim=out_dim, num_blocks=num_blocks)


class WDiscriminator2D(_nets.WDiscriminator):
    def __init__(self, opt):
        super().__init__(2, opt)


class GeneratorHPVAEGAN(_nets.GeneratorHPVAEGAN):
    def __init__(self, opt):
        super().__init__(2, opt)


class Encode2DVAE_nb(_nets.EncodeVAE_nb):
    def __init__(self, opt, out_dim=None, num_blocks=2):
        super().__init__(2, opt, out_dim=out_dim, num_blocks=num_blocks)


class Encode3DVAE1x1(_nets.EncodeVAE1x1):
    """(the reference's 2-D file keeps the 3-D name: networks_2d.py:146)"""

    def __init__(self, opt, out_dim=None):
        super().__init__(2, opt, out_dim=out_dim)


class GeneratorVAE_nb(_nets.GeneratorVAE_nb):
    def __init__(self, opt):
        super().__init__(2, opt)
