"""Pyramid geometry, resize and noise helpers - mirror of the reference's utils/images.py (same names, argument
meaning and assertion behaviour).  Geometry is host-side float64 math reproduced expression by expression
(e.g. scale 0 of a 256-wide pyramid is 33, not 32, because ceil(0.7937^9 * 256) - SURVEY.md 8a15); the resize and
noise injection run in the gfx950 kernels of libhpvg.so."""
import math

import torch

from .. import ops

__all__ = ['interpolate', 'interpolate_3D', 'adjust_scales2image', 'generate_noise', 'get_scales_by_index',
           'get_fps_td_by_index', 'get_fps_by_index', 'upscale', 'upscale_2d']


def _resolve_size(in_sizes, size, scale_factor):
    if size is not None:
        if isinstance(size, int):
            size = [size] * len(in_sizes)
        return [int(s) for s in size]
    if scale_factor is None:
        raise ValueError("either size or scale_factor should be defined")
    if not isinstance(scale_factor, (list, tuple)):
        scale_factor = [scale_factor] * len(in_sizes)
    return [int(math.floor(float(i) * float(f))) for i, f in zip(in_sizes, scale_factor)]


def interpolate(input, size=None, scale_factor=None, interpolation='bilinear'):
    """Bilinear align_corners=True resize of H, W (reference: utils/images.py:9-19).  A 5-D input is resized
    frame by frame in H, W only (T kept)."""
    if interpolation != 'bilinear':
        raise NotImplementedError("MI355X path implements bilinear (align_corners=True) only")
    if input.dim() == 5:
        t = input.shape[2]
        h, w = _resolve_size(input.shape[3:], size, scale_factor)
        return ops.UpsampleAC.apply(input, (t, h, w), None, 0.0)
    h, w = _resolve_size(input.shape[2:], size, scale_factor)
    return ops.UpsampleAC.apply(input, (h, w), None, 0.0)


def interpolate_3D(input, size=None, scale_factor=None, interpolation='trilinear'):
    """Trilinear align_corners=True resize of T, H, W (reference: utils/images.py:22-26)."""
    assert input.dim() == 5, "input must be 5D"
    if interpolation != 'trilinear':
        raise NotImplementedError("MI355X path implements trilinear (align_corners=True) only")
    t, h, w = _resolve_size(input.shape[2:], size, scale_factor)
    return ops.UpsampleAC.apply(input, (t, h, w), None, 0.0)


def adjust_scales2image(size, opt):
    """Derive num_scales / stop_scale / scale1 / scale_factor from the image size (reference: utils/images.py:29-36)."""
    capped = min([opt.max_size, size])
    levels_to_min = math.log(math.pow(opt.min_size / size, 1), opt.scale_factor_init)
    levels_to_cap = math.ceil(math.log(capped / size, opt.scale_factor_init))
    opt.num_scales = math.ceil(levels_to_min) + 1
    opt.stop_scale = opt.num_scales - levels_to_cap
    opt.scale1 = min(opt.max_size / size, 1)
    opt.scale_factor = math.pow(opt.min_size / size, 1 / opt.stop_scale)


def generate_noise(ref=None, size=None, type='normal', emb_size=None, device=None):
    """Noise tensor shaped like `ref` or `size` (reference: utils/images.py:39-57).  N(0, 1) on the device comes from the
    library's own counter-based kernel (hpvg_normal_f32; seeded by torch.manual_seed, advanced per train iteration on the
    device so that hipGraph replays draw fresh noise); the other types and host tensors use torch's generators.  No device
    stream can reproduce the reference's CPU mt19937 draws: parity tests inject recorded noise instead."""
    # (the reference fills zeros first; every element is overwritten by the draw below, and torch.zeros is a
    # hipMemsetAsync, whose node inside a captured hipGraph is not reliably ordered on this runtime - see elementwise.hip)
    if ref is not None:
        noise = torch.empty_like(ref)
    elif size is not None:
        noise = torch.empty(*size, device=device)  # allocated on the target device (no host tensor, no H2D copy)
    else:
        raise Exception("ref or size must be applied")

    if type == 'normal':
        if noise.is_cuda and noise.dtype == torch.float32:
            return ops.normal_(noise)      # Philox4x32-10 + Box-Muller kernel of libhpvg (stream follows torch.manual_seed)
        return noise.normal_(0, 1)
    elif type == 'benoulli':
        return noise.bernoulli_(0.5)

    if type == 'int':
        assert (emb_size is not None) and (size is not None) and (device is not None)
        return torch.randint(0, emb_size, size=size, device=device)

    return noise.uniform_(0, 1)  # default: uniform


def get_scales_by_index(index, scale_factor, stop_scale, img_size):
    """Width of pyramid level `index` (reference: utils/images.py:60-64)."""
    scale = math.pow(scale_factor, stop_scale - index)
    return math.ceil(scale * img_size)


def get_fps_by_index(index, opt):
    """(fps, index into sampling_rates) of level `index` (reference: utils/images.py:67-71)."""
    fps_index = int((index / opt.stop_scale_time) * (len(opt.sampling_rates) - 1))
    return opt.org_fps / opt.sampling_rates[fps_index], fps_index


def get_fps_td_by_index(index, opt):
    """(fps, time depth, fps index) of level `index` (reference: utils/images.py:74-80)."""
    fps, fps_index = get_fps_by_index(index, opt)
    every = opt.sampling_rates[fps_index]
    time_depth = opt.fps_lcm // every + 1
    return fps, time_depth, fps_index


def level_shape_3d(index, opt):
    w = get_scales_by_index(index, opt.scale_factor, opt.stop_scale, opt.img_size)
    _, td, _ = get_fps_td_by_index(index, opt)
    return [td, int(w * opt.ar), w]


def level_shape_2d(index, opt):
    w = get_scales_by_index(index, opt.scale_factor, opt.stop_scale, opt.img_size)
    return [int(w * opt.ar), w]


def upscale(video, index, opt):
    """Trilinear resize of a level-(index-1) video to the (T, H, W) of level `index` (reference: utils/images.py:83-93)."""
    assert index > 0
    return interpolate_3D(video, size=level_shape_3d(index, opt))


def upscale_2d(image, index, opt):
    """Bilinear resize of a level-(index-1) image to the (H, W) of level `index` (reference: utils/images.py:96-105)."""
    assert index > 0
    return interpolate(image, size=level_shape_2d(index, opt))
