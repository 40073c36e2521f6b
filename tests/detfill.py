"""Deterministic, closed-form "random" tensors for fixtures that are too large to store (whole-step vectors at the
BASELINE widths: a stage-3 generator + critic at nfc = 64 is 16 MB of weights).  The same function fills the reference's
modules when tests/golden/make_golden.py records the expected outputs and the product / oracle modules when the tests
run, so only the OUTPUT summaries travel.  Integer arithmetic only up to the final conversion (exact on every machine);
normals come from Box-Muller in float64 and are rounded to fp32 once."""
import math
import zlib

import torch

_M = 1 << 31


def _tag_int(tag):
    return zlib.crc32(tag.encode()) & 0x7FFFFFFF


def det_uniform01(n, tag):
    """n values in [0, 1), float64, from three rounds of a 31-bit LCG over (index, tag)."""
    i = torch.arange(n, dtype=torch.int64)
    h = (i * 1103515245 + 12345 + _tag_int(tag)) % _M
    h = (h * 1103515245 + 12345 + (h >> 7)) % _M
    h = (h * 69069 + 1 + (h >> 11)) % _M
    h = (h * 1103515245 + 12345 + (h >> 5)) % _M
    return h.to(torch.float64) / float(_M)


def det_uniform(shape, tag, lo=-1.0, hi=1.0):
    n = int(math.prod(shape)) if len(shape) else 1
    return (det_uniform01(n, tag) * (hi - lo) + lo).to(torch.float32).reshape(shape)


def det_normal(shape, tag):
    n = int(math.prod(shape)) if len(shape) else 1
    u1 = det_uniform01(n, tag + "/a").clamp_min(1.0 / _M)
    u2 = det_uniform01(n, tag + "/b")
    z = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)
    return z.to(torch.float32).reshape(shape)


def fill_state(sd, prefix):
    """New values for every entry of a state_dict (reference key layout), by key name:
    conv weights / biases U(+-1/sqrt(fan_in)) (torch's default scale), BatchNorm gamma 1 +- 0.1, beta +- 0.1,
    running_mean +- 0.1, running_var 1 .. 1.2, spectral-norm u / v unit vectors; counters untouched."""
    out = {}
    for k, v in sd.items():
        tag = prefix + ":" + k
        if k.endswith("num_batches_tracked"):
            out[k] = v.clone()
        elif k.endswith(("weight_u", "weight_v")):
            t = det_normal(tuple(v.shape), tag)
            out[k] = t / t.norm()
        elif k.endswith("running_mean"):
            out[k] = det_uniform(tuple(v.shape), tag, -0.1, 0.1)
        elif k.endswith("running_var"):
            out[k] = det_uniform(tuple(v.shape), tag, 1.0, 1.2)
        elif ".norm." in k and k.endswith("weight"):
            out[k] = det_uniform(tuple(v.shape), tag, 0.9, 1.1)
        elif ".norm." in k and k.endswith("bias"):
            out[k] = det_uniform(tuple(v.shape), tag, -0.1, 0.1)
        elif v.dim() >= 4:      # conv weight / weight_orig
            fan_in = v[0].numel()
            b = 1.0 / math.sqrt(fan_in)
            out[k] = det_uniform(tuple(v.shape), tag, -b, b)
        elif v.dim() == 1:      # conv bias: fan_in of the sibling weight is not known here; a fixed small range
            out[k] = det_uniform(tuple(v.shape), tag, -0.05, 0.05)
        else:
            out[k] = v.clone()
    return out


def sample_idx(n, count=2048):
    """Indices of a strided sample of a flat tensor of n elements (<= count + 1 of them, first and last included)."""
    if n <= count:
        return torch.arange(n)
    step = n // count
    idx = torch.arange(0, n, step)
    if int(idx[-1]) != n - 1:
        idx = torch.cat([idx, torch.tensor([n - 1])])
    return idx


def summarize(t, count=2048):
    """What a wide fixture keeps of a tensor: its L2 norm, abs-max and a strided sample."""
    f = t.detach().double().reshape(-1)
    return {"shape": list(t.shape), "norm": float(f.norm()), "absmax": float(f.abs().max()) if f.numel() else 0.0,
            "sample": f[sample_idx(f.numel(), count)].float().clone()}
