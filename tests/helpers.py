"""Shared helpers for the parity tests (fixture loading, tolerances, oracle state handling)."""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# north_star tolerance: outputs within 1e-3 relative (fp32) of the reference's PyTorch-CPU path
RTOL = 1e-3


def load_golden(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=True)


def opt_from(d, **kw):
    o = types.SimpleNamespace(**d)
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def rel_err(a, b):
    """max |a-b| / max(|b|) - the relative error measure used for every tensor comparison."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    denom = max(float(b.abs().max()), 1e-30)
    return float((a - b).abs().max()) / denom


def assert_close(a, b, rtol=RTOL, what="", atol=0.0):
    assert tuple(a.shape) == tuple(b.shape), "%s: shape %s vs %s" % (what, tuple(a.shape), tuple(b.shape))
    a64 = a.detach().double().cpu()
    b64 = b.detach().double().cpu()
    err = float((a64 - b64).abs().max()) if a64.numel() else 0.0
    scale = float(b64.abs().max()) if b64.numel() else 0.0
    assert err <= rtol * scale + atol, "%s: max abs err %.3e vs scale %.3e (rel %.3e > %.1e)" % (
        what, err, scale, err / max(scale, 1e-30), rtol)


def oracle_state(sd, requires_grad=True):
    """state_dict (fixture) -> oracle parameter dict (clones; parameters get requires_grad)."""
    from oracle import hpvg_oracle as O
    P = {}
    for k, v in sd.items():
        t = v.clone()
        if requires_grad and O.is_param(k):
            t.requires_grad_(True)
        P[k] = t
    return P
