"""hp-vae-gan_amd: MI355X-native (gfx950) implementation of the HP-VAE-GAN train-step hot path.

Mirrors the reference's Python module surface for this path (modules/networks_3d.py, networks_2d.py,
losses.py, utils.py, utils/images.py and the per-scale train() loop); all arithmetic runs in
hand-written HIP kernels behind the C ABI of include/hpvg.h (libhpvg.so).  No CPU fallback."""
from . import lib  # noqa: F401
