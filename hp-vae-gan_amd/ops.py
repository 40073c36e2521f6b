"""torch.autograd.Function wrappers over the C ABI (include/hpvg.h).

torch is used for device memory, streams and graph bookkeeping only; every arithmetic kernel below is a
hand-written gfx950 kernel in libhpvg.so.  The convolution family {Conv, ConvBwdData, ConvBwdWeight,
LReLUMaskMul} is closed under differentiation (each backward is expressed with the other Functions), which is
what the WGAN-GP double backward through the discriminator needs (reference: modules/utils.py:14-18)."""
import ctypes
import os

import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable

from .lib import call, load, ptr, stream

_ws_cache = {}
_kernel_timer = None


class KernelTimer:
    """Brackets matching conv launches with HIP events recorded on the launch stream (torch's current stream is the
    stream every kernel of this package is enqueued on), so bench.py can report the dominant kernel's average
    duration measured live inside its timed region."""

    def __init__(self, match):
        """match(desc) -> a family name (str) for launches to time, else None.  desc: {"op": "conv" | "wgrad", "Cin", "Cout",
        "KT", and for convs "flip", "var"} in the kernel's view."""
        self.match = match
        self.events = []

    def begin(self, desc):
        fam = self.match(desc)
        if not fam:
            return None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        return (fam, e0)

    def end(self, tok, key):
        if tok is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        self.events.append(((tok[0],) + tuple(key), tok[1], e1))

    def summary(self):
        """{(family, B, Cin, Cout, T, H, W): (average ms, launches)} - call after a device synchronise."""
        acc = {}
        for key, e0, e1 in self.events:
            ms = e0.elapsed_time(e1)
            tot, n = acc.get(key, (0.0, 0))
            acc[key] = (tot + ms, n + 1)
        return {k: (tot / n, n) for k, (tot, n) in acc.items()}


def set_kernel_timer(t):
    global _kernel_timer
    _kernel_timer = t


_ws_pinned = []


def pin_workspaces():
    """From now on a scratch buffer that is outgrown is kept alive instead of freed: a captured hipGraph holds the raw
    address of the buffer that was current at capture time, and replays it long after a larger stage replaced it."""
    if not _ws_pinned:
        _ws_pinned.append(None)
    _ws_pinned.extend(_ws_cache.values())


def workspace(nbytes, device):
    """Stream-ordered scratch buffer shared by all ops on `device` (grown on demand)."""
    key = (device.type, device.index)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
        if _ws_pinned:
            _ws_pinned.append(buf)
    return buf


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def geom(x, w_shape=None):
    """(B, C, T, H, W) of an NCDHW / NCHW activation."""
    if x.dim() == 5:
        B, C, T, H, W = x.shape
    elif x.dim() == 4:
        B, C, H, W = x.shape
        T = 1
    else:
        raise RuntimeError("expected a 4-D or 5-D activation, got %s" % (tuple(x.shape),))
    return B, C, T, H, W


def _kt(w_shape):
    if len(w_shape) == 5:
        if tuple(w_shape[2:]) != (3, 3, 3):
            raise RuntimeError("only 3x3x3 kernels are supported on the MI355X path, got %s" % (tuple(w_shape),))
        return 3
    if len(w_shape) == 4:
        if tuple(w_shape[2:]) != (3, 3):
            raise RuntimeError("only 3x3 kernels are supported on the MI355X path, got %s" % (tuple(w_shape),))
        return 1
    raise RuntimeError("bad weight shape %s" % (tuple(w_shape),))


# ------------------------------------------------------------------------------------------ raw launches
_pack_cache = {}


def weights_changed():
    """Forget every packed weight: called by whatever rewrites parameters behind torch's back (the Adam kernel writes
    through raw pointers, so no version counter moves) and around hipGraph capture."""
    _pack_cache.clear()


def _wants_2d(geom_k, cin_k, cout_k, KT):
    """Does a launch of this geometry read the pack's two-axis Winograd section?  (lib query; negative results are the
    common case and decide nothing else)"""
    if geom_k is None or KT != 3:
        return None if geom_k is None else False
    B, T, H, W = geom_k
    return load().hpvg_conv_wants_wino2d(B, cin_k, cout_k, T, H, W, KT) == 1


def pack_weight(w, flip, geom_k=None):
    """Natural [Co][Ci][taps] weight -> MFMA fragment order (forward, or backward-data when flip).

    The same weight is packed for several launches between two optimizer steps (rec and rand generator passes; forward,
    backward-data and the double-backward convs of one discriminator evaluation): the packed copy is kept until
    `weights_changed()`, keyed by the weight's storage and validated by its version counter.  The entry holds a
    detached alias of the weight, so the allocator cannot hand the address to another tensor while the entry lives.
    geom_k = (B, T, H, W) of the launch that will read the pack: the two-axis Winograd fragments (the largest section)
    are then only written when that launch runs the two-axis kernel; without it the pack serves every launch."""
    w = _c(w)
    Co, Ci = w.shape[0], w.shape[1]
    KT = _kt(w.shape)
    cin_k, cout_k = (Co, Ci) if flip else (Ci, Co)
    has2d = _wants_2d(geom_k, cin_k, cout_k, KT)
    key = (w.data_ptr(), bool(flip), tuple(w.shape))
    hit = _pack_cache.get(key)
    # (an entry packed with the section serves a launch that does not need it; None = packed without a geometry = all sections)
    if hit is not None and hit[0] == w._version and hit[1].device == w.device and (hit[3] is None or hit[3] is True or has2d is False):
        return hit[2]
    if geom_k is None:
        n = call("hpvg_conv_wpack_floats", cin_k, cout_k, KT)
        wp = torch.empty(n, dtype=torch.float32, device=w.device)
        call("hpvg_conv_pack_weight_f32", ptr(w), None, ptr(wp), Ci, Co, KT, 1 if flip else 0, stream())
    else:
        B, T, H, W = geom_k
        n = call("hpvg_conv_wpack_floats_for", cin_k, cout_k, KT, B, T, H, W)
        wp = torch.empty(n, dtype=torch.float32, device=w.device)
        call("hpvg_conv_pack_weight_for_f32", ptr(w), None, ptr(wp), Ci, Co, KT, 1 if flip else 0, B, T, H, W, stream())
    _pack_cache[key] = (w._version, w.detach(), wp, has2d)
    return wp


PACK_BATCH_MAX = 16


def prepack_weights(ws, flips=(False, True), geom_k=None):
    """Pack several weights of one square layer shape (C -> C, C > 4) in ONE launch and leave the results in the pack cache,
    where the convs that follow find them (a discriminator forward packs its six spectral-norm weights twice each - forward
    and backward-data - which used to be a dozen 4.5 us launches in a row).  Weights of other shapes are left to
    pack_weight.  geom_k = (B, T, H, W) of the launches that will read them (see pack_weight)."""
    items = []
    for w in ws:
        if w.dim() < 4 or w.shape[0] != w.shape[1] or w.shape[0] <= 4 or not w.is_contiguous():
            continue
        for f in flips:
            key = (w.data_ptr(), bool(f), tuple(w.shape))
            hit = _pack_cache.get(key)
            if hit is None or hit[0] != w._version:
                items.append((w, bool(f), key))
    if not items:
        return
    shape = tuple(items[0][0].shape)
    items = [it for it in items if tuple(it[0].shape) == shape][:PACK_BATCH_MAX]
    C, KT = shape[0], _kt(shape)
    has2d = _wants_2d(geom_k, C, C, KT)
    if geom_k is None:
        nfl = call("hpvg_conv_wpack_floats", C, C, KT)
    else:
        nfl = call("hpvg_conv_wpack_floats_for", C, C, KT, *geom_k)
    dev = items[0][0].device
    wps = [torch.empty(nfl, dtype=torch.float32, device=dev) for _ in items]
    m = len(items)
    PA, IA = ctypes.c_void_p * m, ctypes.c_int * m
    if geom_k is None:
        call("hpvg_conv_pack_weight_batch_f32", m, PA(*[ptr(it[0]) for it in items]), PA(*[ptr(t) for t in wps]),
             IA(*[1 if it[1] else 0 for it in items]), C, KT, stream())
    else:
        call("hpvg_conv_pack_weight_batch_for_f32", m, PA(*[ptr(it[0]) for it in items]), PA(*[ptr(t) for t in wps]),
             IA(*[1 if it[1] else 0 for it in items]), C, KT, *geom_k, stream())
    for (w, f, key), wp in zip(items, wps):
        _pack_cache[key] = (w._version, w.detach(), wp, has2d)


def conv_fwd_raw(x, w, bias, out_lrelu=False, flip=False, in_affine=None, in_lrelu=False, out_mask=None, mask_bits=None,
                 want_bits=False):
    """y = conv(f(x), w) (+bias); flip=True runs the backward-data conv of the layer weight w.
    out_mask (shaped like y): y *= (out_mask > 0 ? 1 : 0.2) in the kernel's epilogue (leaky_relu_backward of the layer below);
    mask_bits: the same mask in 1-bit form (int32 words, written by the producer of the activation: want_bits=True with
    out_lrelu returns (y, bits))."""
    x = _c(x)
    B, C, T, H, W = geom(x)
    KT = _kt(w.shape)
    Co_l, Ci_l = w.shape[0], w.shape[1]
    cin_k, cout_k = (Co_l, Ci_l) if flip else (Ci_l, Co_l)
    if C != cin_k:
        raise RuntimeError("conv: input has %d channels, weight expects %d" % (C, cin_k))
    wp = pack_weight(w, flip, (B, T, H, W))
    shape = (B, cout_k, T, H, W) if x.dim() == 5 else (B, cout_k, H, W)
    y = torch.empty(shape, dtype=torch.float32, device=x.device)
    sc = sh = None
    if in_affine is not None:
        sc, sh = in_affine
    timed = None
    if _kernel_timer is not None:
        var = "pro" if in_affine is not None else ("mask" if (out_mask is not None or mask_bits is not None) else ("bits" if want_bits else "plain"))
        timed = _kernel_timer.begin({"op": "conv", "Cin": cin_k, "Cout": cout_k, "KT": KT, "flip": flip, "var": var})
    nws = call("hpvg_conv_fwd_ws_bytes", B, cin_k, cout_k, T, H, W, KT)
    ws = workspace(nws, x.device) if nws else None
    if out_mask is not None and tuple(out_mask.shape) != tuple(shape):
        raise RuntimeError("conv: out_mask has shape %s, the output %s" % (tuple(out_mask.shape), tuple(shape)))
    bits = None
    if (mask_bits is not None or want_bits) and cout_k > 4 and in_affine is None:
        if want_bits:
            bits = torch.empty(call("hpvg_conv_mask_words", B, cout_k, T, H, W), dtype=torch.int32, device=x.device)
        if mask_bits is not None and mask_bits.numel() != call("hpvg_conv_mask_words", B, cout_k, T, H, W):
            raise RuntimeError("conv: mask_bits has %d words, the output needs %d" % (mask_bits.numel(), call("hpvg_conv_mask_words", B, cout_k, T, H, W)))
        call("hpvg_conv_fwd_bits_f32", ptr(x), ptr(wp), ptr(bias), ptr(y), 1 if out_lrelu else 0, ptr(mask_bits), ptr(bits), ptr(ws),
             ctypes.c_size_t(ws.numel() if ws is not None else 0), B, cin_k, cout_k, T, H, W, KT, stream())
    else:
        call("hpvg_conv_fwd_f32", ptr(x), ptr(wp), ptr(bias), ptr(sc), ptr(sh), 1 if in_lrelu else 0, ptr(y),
             1 if out_lrelu else 0, ptr(_c(out_mask)) if out_mask is not None else None, ptr(ws),
             ctypes.c_size_t(ws.numel() if ws is not None else 0), B, cin_k, cout_k, T, H, W, KT, stream())
    if timed is not None:
        _kernel_timer.end(timed, (B, cin_k, cout_k, T, H, W))
    return (y, bits) if want_bits else y


_DIRECT_GRAD = os.environ.get("HPVG_DIRECT_GRAD", "1") != "0"


def grad_slot(p):
    """The gradient buffer of parameter `p` when a backward kernel may add its result straight into it, else None.

    autograd would take the returned gradient and run `p.grad += it` as one more (4.7 us) kernel per parameter and
    contribution - ~210 per GAN iteration.  The kernels that produce parameter gradients (weight-gradient reduce, channel
    sum, BatchNorm finalize, spectral-norm backward) can do that addition themselves: when `p` is a leaf that already has
    a dense fp32 `.grad` (the trainers' ParamArena gives every parameter one, zeroed per step) and no graph is being
    recorded, the Function adds into `p.grad` and returns None for that input.  Same numbers as AccumulateGrad;
    tensor hooks on `p` do not fire for it (HPVG_DIRECT_GRAD=0 restores plain autograd accumulation)."""
    if not _DIRECT_GRAD or p is None or torch.is_grad_enabled() or not (p.is_leaf and p.requires_grad):
        return None
    g = p.grad
    if g is None or g.dtype != torch.float32 or g.shape != p.shape or g.device != p.device or not g.is_contiguous():
        return None
    return g


def conv_bwd_weight_raw(dy, x, w_shape, into=None):
    """dw of the conv; `into` (a gradient buffer of the weight's shape): add the result into it and return None."""
    dy = _c(dy)
    x = _c(x)
    B, Co, T, H, W = geom(dy)
    Ci = x.shape[1]
    KT = _kt(w_shape)
    if (Co, Ci) != (w_shape[0], w_shape[1]):
        raise RuntimeError("conv_bwd_weight: channel mismatch")
    nbytes = call("hpvg_conv_bwd_weight_ws_bytes", B, Ci, Co, T, H, W, KT)
    ws = workspace(nbytes, dy.device)
    dw = into if into is not None else torch.empty(tuple(w_shape), dtype=torch.float32, device=dy.device)
    timed = _kernel_timer.begin({"op": "wgrad", "Cin": Ci, "Cout": Co, "KT": KT, "bias": False}) if _kernel_timer is not None else None
    call("hpvg_conv_bwd_weight_f32", ptr(dy), ptr(x), None, None, 0, ptr(dw), 1 if into is not None else 0, ptr(ws),
         ctypes.c_size_t(ws.numel()), B, Ci, Co, T, H, W, KT, stream())
    if timed is not None:
        _kernel_timer.end(timed, (B, Ci, Co, T, H, W))
    return None if into is not None else dw


def conv_bwd_weight_bias_raw(dy, x, w_shape, into_w, into_b):
    """dw AND the bias gradient from one launch where the library can (wide layers on the Winograd weight-gradient kernel),
    both added into their gradient buffers; returns False when the layer is not one of those (nothing was launched)."""
    dy = _c(dy)
    x = _c(x)
    B, Co, T, H, W = geom(dy)
    Ci = x.shape[1]
    KT = _kt(w_shape)
    if load().hpvg_conv_bwd_weight_fuses_bias(B, Ci, Co, T, H, W, KT) != 1:
        return False
    nbytes = call("hpvg_conv_bwd_weight_ws_bytes", B, Ci, Co, T, H, W, KT)
    ws = workspace(nbytes, dy.device)
    timed = _kernel_timer.begin({"op": "wgrad", "Cin": Ci, "Cout": Co, "KT": KT, "bias": True}) if _kernel_timer is not None else None
    call("hpvg_conv_bwd_weight_bias_f32", ptr(dy), ptr(x), ptr(into_w), 1, ptr(into_b), 1, ptr(ws), ctypes.c_size_t(ws.numel()),
         B, Ci, Co, T, H, W, KT, stream())
    if timed is not None:
        _kernel_timer.end(timed, (B, Ci, Co, T, H, W))
    return True


def channel_sum_raw(dy, into=None):
    dy = _c(dy)
    B, C, T, H, W = geom(dy)
    out = into if into is not None else torch.empty(C, dtype=torch.float32, device=dy.device)
    ws = workspace(call("hpvg_channel_sum_ws_bytes", C), dy.device)
    call("hpvg_channel_sum_f32", ptr(dy), ptr(out), 1 if into is not None else 0, ptr(ws), ctypes.c_size_t(ws.numel()), B, C,
         ctypes.c_long(T * H * W), stream())
    return None if into is not None else out


def _weight_grad(dy, x, w):
    """dw for Conv / ConvBwdData backward: added straight into w.grad when allowed (returns None), else a Function."""
    slot = grad_slot(w)
    if slot is not None:
        return conv_bwd_weight_raw(dy, x, w.shape, into=slot)
    return ConvBwdWeight.apply(dy, x, w.shape)


def _scalar_out(device):
    return torch.empty(1, dtype=torch.float32, device=device)


def _reduce_ws(device):
    n = call("hpvg_reduce_ws_bytes")
    return workspace(n, device), n


class inputs_only:
    """with ops.inputs_only(): a backward pass that is known to want gradients w.r.t. activations only
    (torch.autograd.grad(D(x_hat), x_hat) of the gradient penalty).  A Python autograd Function cannot see which of its
    inputs the running graph task needs - ctx.needs_input_grad only says "requires grad" - so without this every conv
    of the critic would also launch its weight- and bias-gradient kernels there, for results the engine throws away."""
    active = False

    def __enter__(self):
        self.prev = inputs_only.active
        inputs_only.active = True

    def __exit__(self, *exc):
        inputs_only.active = self.prev


# ------------------------------------------------------------------------------------------ conv family
class LReLUMaskMul(Function):
    """dy * (h > 0 ? 1 : 0.2) - leaky_relu_backward on the activated tensor (reference: networks_3d.py:21)."""

    @staticmethod
    def forward(ctx, dy, h):
        dy = _c(dy)
        out = torch.empty_like(dy)
        call("hpvg_lrelu_mask_mul_f32", ptr(dy), ptr(h), ptr(out), ctypes.c_long(dy.numel()), stream())
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, g):
        (h,) = ctx.saved_tensors
        return LReLUMaskMul.apply(g, h), None


class ChannelSum(Function):
    @staticmethod
    def forward(ctx, dy):
        ctx.shape = dy.shape
        return channel_sum_raw(dy)

    @staticmethod
    def backward(ctx, g):
        view = [1, -1] + [1] * (len(ctx.shape) - 2)
        return g.view(*view).expand(ctx.shape).contiguous()


class Conv(Function):
    """y = conv3x3(x, w) + b, optionally followed by LeakyReLU(0.2) (ConvBlock3DSN / plain tails).

    Chains of activated convs (the spectral-norm critic and encoder: networks_3d.py:59-70,163-181) hand the LeakyReLU's
    backward mask from the PRODUCER of an activation to its CONSUMER, where it costs nothing: the consumer's backward-data
    conv computes the gradient w.r.t. the activated tensor - its own saved input - and multiplies by the mask in its
    epilogue (conv_fwd_raw(out_mask=)); leaky_relu_backward as a pass of its own (three tensor sweeps per layer and
    backward chain) is gone.
      in_act       : x is the output of a Conv(act=True, out_masked_by_consumer=True): dx is masked with lrelu'(x) here
      mask_by_consumer : (with act) every consumer of y is such a Conv, so this backward must NOT mask dy again."""

    @staticmethod
    def forward(ctx, x, w, b, act, in_act=False, mask_by_consumer=False, in_bits=None):
        """in_bits: the 1-bit LeakyReLU mask of x as its producer's epilogue wrote it (see Conv.apply_bits); the
        backward-data epilogue then reads 1/32 of the bytes.  Without it (row slabs: x was extended by halo rows) the mask
        is read from x itself."""
        want = act and mask_by_consumer and w.shape[0] > 4
        if want:
            y, bits = conv_fwd_raw(x, w, b, out_lrelu=True, want_bits=True)
            Conv.last_bits = bits          # handed to the caller by apply_bits (not an autograd output)
        else:
            y = conv_fwd_raw(x, w, b, out_lrelu=act)
            Conv.last_bits = None
        own_mask = act and not mask_by_consumer
        ctx.save_for_backward(x, w, y if own_mask else None, b, in_bits if in_act else None)
        ctx.own_mask, ctx.in_act = own_mask, in_act
        ctx.has_bias = b is not None
        return y

    last_bits = None

    @staticmethod
    def apply_bits(x, w, b, act, in_act=False, mask_by_consumer=False, in_bits=None):
        """Conv.apply that also returns the 1-bit mask of the activated output (None when this call made none)."""
        y = Conv.apply(x, w, b, act, in_act, mask_by_consumer, in_bits)
        bits, Conv.last_bits = Conv.last_bits, None
        return y, bits

    @staticmethod
    def backward(ctx, dy):
        x, w, y, b, in_bits = ctx.saved_tensors
        dy = _c(dy)
        if ctx.own_mask:
            dy = LReLUMaskMul.apply(dy, y)
        params = not inputs_only.active
        dx = ConvBwdData.apply(dy, w, x if ctx.in_act else None, in_bits) if ctx.needs_input_grad[0] else None
        want_w = ctx.needs_input_grad[1] and params
        want_b = ctx.has_bias and ctx.needs_input_grad[2] and params
        if want_w and want_b:
            # both straight into their gradient buffers from ONE launch where the library fuses them
            ws_, bs_ = grad_slot(w), grad_slot(b)
            if ws_ is not None and bs_ is not None and conv_bwd_weight_bias_raw(dy, x, w.shape, ws_, bs_):
                return dx, None, None, None, None, None, None
        dw = _weight_grad(dy, x, w) if want_w else None
        db = None
        if want_b:
            slot = grad_slot(b)
            db = channel_sum_raw(dy, into=slot) if slot is not None else ChannelSum.apply(dy)
        return dx, dw, db, None, None, None, None


class ConvBwdData(Function):
    """dx = conv(dy, flip/transpose(w)) [* lrelu'(h)]: backward-data of a stride-1 'same' conv; with h (the activated tensor
    the conv consumed) the LeakyReLU backward of the layer below is applied in the kernel's epilogue - from the 1-bit mask
    `bits` when the producer of h wrote one."""

    @staticmethod
    def forward(ctx, dy, w, h=None, bits=None):
        ctx.save_for_backward(dy, w, h)
        if h is not None and bits is not None and w.shape[1] > 4:
            return conv_fwd_raw(dy, w, None, flip=True, mask_bits=bits)
        return conv_fwd_raw(dy, w, None, flip=True, out_mask=h)

    @staticmethod
    def backward(ctx, g):
        dy, w, h = ctx.saved_tensors
        g = _c(g)
        if h is not None:
            g = LReLUMaskMul.apply(g, h)     # second order only (the gradient penalty's double backward): a pass of its own
        ddy = Conv.apply(g, w, None, False) if ctx.needs_input_grad[0] else None
        dw = _weight_grad(dy, g, w) if ctx.needs_input_grad[1] else None
        return ddy, dw, None, None


class ConvBwdWeight(Function):
    """dw[o][c][tap] = sum dy[o][n] * x[c][n + off(tap)]."""

    @staticmethod
    def forward(ctx, dy, x, w_shape):
        ctx.save_for_backward(dy, x)
        return conv_bwd_weight_raw(dy, x, w_shape)

    @staticmethod
    def backward(ctx, gw):
        dy, x = ctx.saved_tensors
        gw = _c(gw)
        ddy = Conv.apply(x, gw, None, False) if ctx.needs_input_grad[0] else None
        dx = ConvBwdData.apply(dy, gw) if ctx.needs_input_grad[1] else None
        return ddy, dx, None


# ------------------------------------------------------------------------------------------ BatchNorm + LeakyReLU
class BNAct(Function):
    """h = LeakyReLU_opt(BatchNorm_train(r)) with running-stat update (ConvBlock3D: networks_3d.py:54-56).

    groups > 1: the batch holds `groups` independent passes of the network one after the other (the merged rec + rand
    generator pass, GeneratorHPVAEGAN.forward_pair); every group is normalised with ITS OWN batch statistics and the
    running statistics are updated once per group, in order - exactly what the separate passes would do (the kernels take
    the group count: one launch pair for all groups)."""

    @staticmethod
    def forward(ctx, r, gamma, beta, running_mean, running_var, momentum, eps, lrelu, groups=1):
        r = _c(r)
        B, C, T, H, W = geom(r)
        S = T * H * W
        dev = r.device
        assert B % groups == 0
        stats = torch.empty(groups, 4, C, dtype=torch.float32, device=dev)  # per group: mean, invstd, scale, shift
        nws = call("hpvg_bn_ws_bytes", C * groups)
        ws = workspace(nws, dev)
        h = torch.empty_like(r)
        st = stats[0]
        call("hpvg_bn_train_fwd_f32", ptr(r), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
             float(momentum), float(eps), ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), ptr(h),
             1 if lrelu else 0, groups, ptr(ws), ctypes.c_size_t(ws.numel()), B, C, ctypes.c_long(S), stream())
        ctx.save_for_backward(r, stats, gamma, beta)
        ctx.lrelu, ctx.groups = lrelu, groups
        return h

    @staticmethod
    def backward(ctx, dh):
        r, stats, gamma, beta = ctx.saved_tensors
        dh = _c(dh)
        if torch.is_grad_enabled():
            # a graph is being recorded over this backward (create_graph=True: the gradient penalty of a critic that contains
            # BatchNorm, WDiscriminatorBaselines): run it as a differentiable op of its own
            if ctx.groups != 1:
                raise NotImplementedError("second-order BatchNorm is implemented for groups == 1")
            dr, dg, db = BNActBwd.apply(dh, r, gamma, stats, ctx.lrelu)
            return dr, dg, db, None, None, None, None, None, None
        B, C, T, H, W = geom(r)
        S = T * H * W
        dev = r.device
        groups = ctx.groups
        dr = torch.empty_like(r)
        sg, sb = grad_slot(gamma), grad_slot(beta)
        direct = sg is not None and sb is not None and ctx.needs_input_grad[1] and ctx.needs_input_grad[2]
        dgb = (sg, sb) if direct else torch.empty(2, C, dtype=torch.float32, device=dev)
        nws = call("hpvg_bn_ws_bytes", C * groups)
        ws = workspace(nws, dev)
        st = stats[0]
        call("hpvg_bn_act_bwd_f32", ptr(dh), ptr(r), ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]),
             1 if ctx.lrelu else 0, groups, ptr(dr), ptr(dgb[0]), ptr(dgb[1]), 1 if direct else 0, ptr(ws),
             ctypes.c_size_t(ws.numel()), B, C, ctypes.c_long(S), stream())
        if direct:
            return dr, None, None, None, None, None, None, None, None
        return dr, dgb[0], dgb[1], None, None, None, None, None, None


class BNActBwd(Function):
    """(dr, dgamma, dbeta) = backward of BNAct as a DIFFERENTIABLE op: what BNAct.backward runs while a graph is recorded
    over it.  Its own backward is the second-order BatchNorm kernel pair (hpvg_bn_act_bwd2_f32): gradients w.r.t. dh, r and
    gamma of <dr, g>; dgamma / dbeta are not differentiated on the path (the penalty differentiates input gradients only)."""

    @staticmethod
    def forward(ctx, dh, r, gamma, stats, lrelu):
        B, C, T, H, W = geom(r)
        S = T * H * W
        dev = r.device
        dr = torch.empty_like(r)
        dgb = torch.empty(2, C, dtype=torch.float32, device=dev)
        ws = workspace(call("hpvg_bn_ws_bytes", C), dev)
        st = stats[0]
        call("hpvg_bn_act_bwd_f32", ptr(dh), ptr(r), ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), 1 if lrelu else 0, 1,
             ptr(dr), ptr(dgb[0]), ptr(dgb[1]), 0, ptr(ws), ctypes.c_size_t(ws.numel()), B, C, ctypes.c_long(S), stream())
        ctx.save_for_backward(dh, r, gamma, stats)
        ctx.lrelu = lrelu
        dg, db = dgb[0], dgb[1]
        ctx.mark_non_differentiable(dg, db)
        return dr, dg, db

    @staticmethod
    @once_differentiable
    def backward(ctx, g, _g_dgamma, _g_dbeta):
        dh, r, gamma, stats = ctx.saved_tensors
        g = _c(g)
        B, C, T, H, W = geom(r)
        S = T * H * W
        dev = r.device
        g_dh = torch.empty_like(r) if ctx.needs_input_grad[0] else None
        g_r = torch.empty_like(r) if ctx.needs_input_grad[1] else None
        slot = grad_slot(gamma) if ctx.needs_input_grad[2] else None
        g_gamma = slot if slot is not None else (torch.empty(C, dtype=torch.float32, device=dev) if ctx.needs_input_grad[2] else None)
        ws = workspace(call("hpvg_bn_bwd2_ws_bytes", C), dev)
        st = stats[0]
        call("hpvg_bn_act_bwd2_f32", ptr(dh), ptr(g), ptr(r), ptr(st[0]), ptr(st[1]), ptr(st[2]), ptr(st[3]), 1 if ctx.lrelu else 0,
             ptr(g_dh), ptr(g_r), ptr(g_gamma), 1 if slot is not None else 0, ptr(ws), ctypes.c_size_t(ws.numel()), B, C,
             ctypes.c_long(S), stream())
        return g_dh, g_r, (None if slot is not None else g_gamma), None, None


def concat_batch(parts, like=None):
    """torch.cat(parts, dim=0) for contiguous fp32 tensors of equal trailing shape, with kernel copies: torch.cat (and
    torch.zeros) may lower to hipMemcpyAsync / hipMemsetAsync, i.e. memcpy / memset NODES when the iteration is captured
    into a hipGraph, and those are not reliably ordered on this runtime.  An entry that is an int n
    stands for n zero samples."""
    ref = like if like is not None else next(p for p in parts if torch.is_tensor(p))
    tail = tuple(ref.shape[1:])
    per = 1
    for d in tail:
        per *= int(d)
    total = sum(p if isinstance(p, int) else p.shape[0] for p in parts)
    out = torch.empty((total,) + tail, dtype=torch.float32, device=ref.device)
    o = 0
    for p in parts:
        n = p if isinstance(p, int) else p.shape[0]
        src = None if isinstance(p, int) else ptr(_c(p.detach()))
        call("hpvg_copy_f32", src, ptr(out[o:o + n]), ctypes.c_long(n * per), stream())
        o += n
    return out


class Concat2(Function):
    """Differentiable concat_batch of two tensors (the latent batch of the merged generator pass)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.n = a.shape[0]
        return concat_batch([a, b])

    @staticmethod
    def backward(ctx, g):
        g = _c(g)
        return g[:ctx.n], g[ctx.n:]


class SplitBatch(Function):
    """(x[:n], x[n:]) as two outputs.  torch's own slicing would do: but its backward builds the full-size gradient with
    at::zeros + copy, i.e. a hipMemsetAsync, which must not appear in an iteration that may be captured into a hipGraph
    (DESIGN.md section 4); this backward is one concatenation kernel."""

    @staticmethod
    def forward(ctx, x, n):
        x = _c(x)
        ctx.n, ctx.shape = n, x.shape
        return x[:n], x[n:]

    @staticmethod
    def backward(ctx, g0, g1):
        n, shape = ctx.n, ctx.shape
        if g0 is None and g1 is None:
            return None, None
        ref = g0 if g0 is not None else g1
        return concat_batch([g0 if g0 is not None else n, g1 if g1 is not None else shape[0] - n], like=ref), None


class BNActSync(Function):
    """BNAct with the batch split over the ranks of a process group (multi-GPU, one process per GPU): each rank reduces
    its own samples, `allreduce(t)` (a callable summing a float64 tensor over the group, in place) combines the
    per-channel pairs, and statistics / running buffers / dr follow from the global sums - the same arithmetic as BNAct
    over the whole batch.  dgamma / dbeta are this rank's share (the parameter all-reduce adds the shares up)."""

    @staticmethod
    def forward(ctx, r, gamma, beta, running_mean, running_var, momentum, eps, lrelu, allreduce, nranks, total=None):
        r = _c(r)
        B, C, T, H, W = geom(r)
        S = T * H * W
        dev = r.device
        ws = workspace(call("hpvg_bn_ws_bytes", C), dev)
        sums = torch.empty(C, 2, dtype=torch.float64, device=dev)
        call("hpvg_bn_sums_f32", ptr(r), ptr(sums), ptr(ws), ctypes.c_size_t(ws.numel()), B, C, ctypes.c_long(S), stream())
        allreduce(sums)
        # elements per channel behind the statistics: `total` when the ranks hold unequal shares (row slabs)
        count = float(total) if total is not None else float(B) * float(S) * float(nranks)
        stats = torch.empty(4, C, dtype=torch.float32, device=dev)  # mean, invstd, scale, shift
        call("hpvg_bn_finalize_f32", ptr(sums), ctypes.c_double(count), ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var),
             float(momentum), float(eps), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]), C, stream())
        h = torch.empty_like(r)
        call("hpvg_affine_act_f32", ptr(r), ptr(stats[2]), ptr(stats[3]), ptr(h), 1 if lrelu else 0, B, C, ctypes.c_long(S),
             stream())
        ctx.save_for_backward(r, stats)
        ctx.lrelu, ctx.allreduce, ctx.count = lrelu, allreduce, count
        return h

    @staticmethod
    @once_differentiable
    def backward(ctx, dh):
        r, stats = ctx.saved_tensors
        dh = _c(dh)
        B, C, T, H, W = geom(r)
        S = T * H * W
        dev = r.device
        ws = workspace(call("hpvg_bn_ws_bytes", C), dev)
        local = torch.empty(C, 2, dtype=torch.float64, device=dev)  # (sum dz, sum dz*xhat) of this rank's samples
        call("hpvg_bn_act_bwd_sums_f32", ptr(dh), ptr(r), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]),
             1 if ctx.lrelu else 0, ptr(local), ptr(ws), ctypes.c_size_t(ws.numel()), B, C, ctypes.c_long(S), stream())
        glob = local.clone()
        ctx.allreduce(glob)
        gsum = glob.to(torch.float32).contiguous()
        dr = torch.empty_like(r)
        call("hpvg_bn_act_bwd_apply_f32", ptr(dh), ptr(r), ptr(stats[0]), ptr(stats[1]), ptr(stats[2]), ptr(stats[3]),
             1 if ctx.lrelu else 0, ptr(gsum), float(1.0 / ctx.count), ptr(dr), B, C, ctypes.c_long(S), stream())
        loc32 = local.to(torch.float32)
        return dr, loc32[:, 1].contiguous(), loc32[:, 0].contiguous(), None, None, None, None, None, None, None, None


class AffineAct(Function):
    """y = LeakyReLU_opt(scale[c]*x + shift[c]) - eval-mode BatchNorm apply (running statistics)."""

    @staticmethod
    def forward(ctx, r, scale, shift, lrelu):
        r = _c(r)
        B, C, T, H, W = geom(r)
        h = torch.empty_like(r)
        call("hpvg_affine_act_f32", ptr(r), ptr(_c(scale)), ptr(_c(shift)), ptr(h), 1 if lrelu else 0, B, C,
             ctypes.c_long(T * H * W), stream())
        return h

    @staticmethod
    def backward(ctx, dh):
        raise NotImplementedError("eval-mode BatchNorm backward is not on the reference's train path")


# ------------------------------------------------------------------------------------------ spectral norm
class SpectralNormWeight(Function):
    """W = W_orig / sigma(W_orig), sigma = u^T W_mat v after one power iteration (u, v updated in place when
    training).  torch hook semantics: nn.utils.spectral_norm, n_power_iterations=1, eps=1e-12 (networks_3d.py:63)."""

    @staticmethod
    def forward(ctx, w_orig, u, v, do_iter, eps):
        w_orig = _c(w_orig)
        Co = w_orig.shape[0]
        K = w_orig.numel() // Co
        dev = w_orig.device
        sig = torch.empty(2, dtype=torch.float32, device=dev)  # sigma, 1/sigma
        ws = workspace(Co * 4, dev)
        # later forwards overwrite the u/v buffers before this call's backward runs: the kernel also writes the (u, v) its
        # sigma belongs to into `uv` (torch clones them for the same reason); [0, Co) = u, [Co, Co + K) = v
        uv = torch.empty(Co + K, dtype=torch.float32, device=dev) if w_orig.requires_grad else None
        call("hpvg_sn_power_iter_f32", ptr(w_orig), ptr(u), ptr(v), ptr(sig[0:1]), ptr(sig[1:2]), ptr(uv), Co, K,
             1 if do_iter else 0, float(eps), ptr(ws), ctypes.c_size_t(ws.numel()), stream())
        w = torch.empty_like(w_orig)
        call("hpvg_div_scalar_f32", ptr(w_orig), ptr(sig[0:1]), ptr(w), ctypes.c_long(w.numel()), stream())
        ctx.save_for_backward(w_orig, uv, sig)
        return w

    @staticmethod
    @once_differentiable
    def backward(ctx, dw):
        w_orig, uv, sig = ctx.saved_tensors
        dw = _c(dw)
        Co = w_orig.shape[0]
        K = w_orig.numel() // Co
        slot = grad_slot(w_orig)
        out = slot if slot is not None else torch.empty_like(w_orig)
        ws = workspace(call("hpvg_sn_bwd_ws_bytes", Co, K), dw.device)
        call("hpvg_sn_bwd_f32", ptr(dw), ptr(w_orig), ptr(uv[:Co]), ptr(uv[Co:]), ptr(sig[0:1]), ptr(out),
             1 if slot is not None else 0, ptr(ws), ctypes.c_size_t(ws.numel()), Co, K, stream())
        return (None if slot is not None else out), None, None, None, None


SN_BATCH_MAX = 8


class SpectralNormWeightBatch(Function):
    """SpectralNormWeight for up to 8 independent layers in ONE launch (one workgroup per layer: power iteration, sigma,
    the (u, v) copy for the backward and weight = weight_orig / sigma): the spectral-norm convs of a network all need
    their weight at the start of a forward, and six ~19 us single-workgroup launches in a row are pure latency.
    apply(do_iter, eps, w1, u1, v1, w2, u2, v2, ...) -> (weight1, weight2, ...).  The backward runs per layer (the
    gradients arrive layer by layer) with the same kernels as SpectralNormWeight."""

    @staticmethod
    def forward(ctx, do_iter, eps, *tensors):
        n = len(tensors) // 3
        assert 1 <= n <= SN_BATCH_MAX and len(tensors) == 3 * n
        ws_orig = [_c(tensors[3 * i]) for i in range(n)]
        us = [tensors[3 * i + 1] for i in range(n)]
        vs = [tensors[3 * i + 2] for i in range(n)]
        dev = ws_orig[0].device
        Cos = [w.shape[0] for w in ws_orig]
        Ks = [w.numel() // w.shape[0] for w in ws_orig]
        sig = torch.empty(n, 2, dtype=torch.float32, device=dev)
        need = [w.requires_grad for w in ws_orig]
        uvs = [torch.empty(Cos[i] + Ks[i], dtype=torch.float32, device=dev) if need[i] else None for i in range(n)]
        outs = [torch.empty_like(w) for w in ws_orig]
        ws = workspace(4 * sum(Cos), dev)
        PA, IA = ctypes.c_void_p * n, ctypes.c_int * n
        call("hpvg_sn_power_iter_batch_f32", n, PA(*[ptr(w) for w in ws_orig]), PA(*[ptr(u) for u in us]), PA(*[ptr(v) for v in vs]),
             PA(*[ptr(sig[i]) for i in range(n)]), PA(*[ptr(t) for t in uvs]), PA(*[ptr(o) for o in outs]), IA(*Cos), IA(*Ks),
             1 if do_iter else 0, float(eps), ptr(ws), ctypes.c_size_t(ws.numel()), stream())
        ctx.n = n
        ctx.save_for_backward(sig, *ws_orig, *uvs)
        return tuple(outs)

    @staticmethod
    @once_differentiable
    def backward(ctx, *dws):
        n = ctx.n
        saved = ctx.saved_tensors
        sig, ws_orig, uvs = saved[0], saved[1:1 + n], saved[1 + n:1 + 2 * n]
        grads = [None, None]
        live = [i for i in range(n) if dws[i] is not None and ctx.needs_input_grad[2 + 3 * i]]
        res = {}
        if live:
            m = len(live)
            dev = dws[live[0]].device
            dwc = [_c(dws[i]) for i in live]
            slots = [grad_slot(ws_orig[i]) for i in live]
            outs = [sl if sl is not None else torch.empty_like(ws_orig[i]) for sl, i in zip(slots, live)]
            Cos = [ws_orig[i].shape[0] for i in live]
            Ks = [ws_orig[i].numel() // ws_orig[i].shape[0] for i in live]
            ws = workspace(sum(call("hpvg_sn_bwd_ws_bytes", c, k) for c, k in zip(Cos, Ks)), dev)
            PA, IA = ctypes.c_void_p * m, ctypes.c_int * m
            call("hpvg_sn_bwd_batch_f32", m, PA(*[ptr(t) for t in dwc]), PA(*[ptr(ws_orig[i]) for i in live]),
                 PA(*[ptr(uvs[i]) for i in live]), PA(*[ptr(sig[i, 0:1]) for i in live]), PA(*[ptr(o) for o in outs]),
                 IA(*[1 if sl is not None else 0 for sl in slots]), IA(*Cos), IA(*Ks), ptr(ws), ctypes.c_size_t(ws.numel()), stream())
            for i, sl, o in zip(live, slots, outs):
                res[i] = None if sl is not None else o
        for i in range(n):
            grads += [res.get(i), None, None]
        return tuple(grads)


# ------------------------------------------------------------------------------------------ generator glue
class TanhRes(Function):
    """y = tanh(x + res)  (res optional): networks_3d.py:377 (vae_out) and :404 (residual refinement)."""

    @staticmethod
    def forward(ctx, x, res):
        x = _c(x)
        if res is not None:
            res = _c(res)
        y = torch.empty_like(x)
        call("hpvg_tanh_fwd_f32", ptr(x), ptr(res), ptr(y), ctypes.c_long(x.numel()), stream())
        ctx.save_for_backward(y)
        ctx.has_res = res is not None
        return y

    @staticmethod
    @once_differentiable
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(y)
        call("hpvg_tanh_bwd_f32", ptr(dy), ptr(y), ptr(dx), ctypes.c_long(y.numel()), stream())
        return dx, (dx if ctx.has_res else None)


class Add(Function):
    """a + b: the pre-tanh residual of the SinGAN baseline (networks_3d.py:319)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        if a.shape != b.shape:
            raise RuntimeError("add: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        out = torch.empty_like(a)
        call("hpvg_add_f32", ptr(a), ptr(b), ptr(out), ctypes.c_long(a.numel()), stream())
        return out

    @staticmethod
    def backward(ctx, g):
        return g, g


class UpsampleAC(Function):
    """Tri/bi-linear resize with align_corners=True to `size`; with `noise`, also returns up + amp*noise
    (utils/images.py:83-105 + networks_3d.py:395-400)."""

    @staticmethod
    def forward(ctx, x, size, noise, amp):
        x = _c(x)
        B, C, Ti, Hi, Wi = geom(x)
        if x.dim() == 5:
            To, Ho, Wo = size
            oshape = (B, C, To, Ho, Wo)
        else:
            Ho, Wo = size
            To = 1
            oshape = (B, C, Ho, Wo)
        y = torch.empty(oshape, dtype=torch.float32, device=x.device)
        yn = None
        if noise is not None:
            noise = _c(noise)
            if tuple(noise.shape) != tuple(oshape):
                raise RuntimeError("noise shape %s != upsampled shape %s" % (tuple(noise.shape), oshape))
            yn = torch.empty_like(y)
        call("hpvg_upsample_linear_ac_f32", ptr(x), ptr(y), ptr(noise), float(amp), ptr(yn), ctypes.c_long(B * C), Ti, Hi,
             Wi, To, Ho, Wo, stream())
        ctx.in_shape = x.shape
        ctx.dims = (B * C, Ti, Hi, Wi, To, Ho, Wo)
        if yn is None:
            return y
        return y, yn

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dyn=None):
        # both outputs (`up` and `up + amp*noise`) may carry a gradient: the kernel adds the two on load
        g, g2 = (dy, dyn) if dy is not None else (dyn, None)
        g = _c(g)
        g2 = _c(g2) if g2 is not None else None
        BC, Ti, Hi, Wi, To, Ho, Wo = ctx.dims
        dx = torch.empty(ctx.in_shape, dtype=torch.float32, device=g.device)
        call("hpvg_upsample_linear_ac_bwd_f32", ptr(g), ptr(g2), ptr(dx), ctypes.c_long(BC), Ti, Hi, Wi, To, Ho, Wo, stream())
        return dx, None, None, None


# ------------------------------------------------------------------------------------------ N(0,1) noise (Philox kernel)
class _RngState:
    """Counter-based noise stream of one device: (seed, iteration counter ON THE DEVICE, call index inside the iteration).
    The kernels read the iteration from device memory, so a hipGraph replay draws fresh noise although none of its launch
    arguments changes; the trainers bump it once per iteration (rng_next_iteration)."""

    def __init__(self, device):
        self.iter_dev = torch.zeros(1, dtype=torch.int32, device=device)
        self.call = 0


_rng_states = {}


def _rng(device):
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())   # "cuda" and "cuda:0" are one stream
    key = (device.type, device.index)
    st = _rng_states.get(key)
    if st is None:
        st = _rng_states[key] = _RngState(device)
    return st


def rng_next_iteration(device):
    """Start the noise stream of the next train iteration on `device` (one 1-thread kernel; captured with the iteration)."""
    st = _rng(torch.device(device) if not isinstance(device, torch.device) else device)
    counter_inc_(st.iter_dev)
    st.call = 0


def _seed():
    return ctypes.c_ulonglong(torch.initial_seed() & 0xFFFFFFFFFFFFFFFF)   # follows torch.manual_seed


def normal_(out):
    """out <- N(0, 1) (utils/images.py:49, networks_3d.py:32) with the Philox kernel of libhpvg; returns out."""
    st = _rng(out.device)
    call("hpvg_normal_f32", ptr(out), ctypes.c_long(out.numel()), _seed(), ctypes.c_uint(st.call & 0xFFFFFFFF), ptr(st.iter_dev), stream())
    st.call += 1
    return out


def uniform_(out):
    """out <- U[0, 1) from the library's Philox stream (reparameterize_bern's eps, networks_3d.py:40)."""
    st = _rng(out.device)
    call("hpvg_uniform_f32", ptr(out), ctypes.c_long(out.numel()), _seed(), ctypes.c_uint(st.call & 0xFFFFFFFF), ptr(st.iter_dev), stream())
    st.call += 1
    return out


# ------------------------------------------------------------------------------------------ variant models (the _nb family)
def _bcs(x):
    B, C = x.shape[0], x.shape[1]
    S = 1
    for d in x.shape[2:]:
        S *= int(d)
    return B, C, S


class Gate(Function):
    """(bern * f, bern) with bern = sigmoid(logit) broadcast over channels (Encode3DVAE_nb: networks_3d.py:131-133)."""

    @staticmethod
    def forward(ctx, f, logit):
        f, logit = _c(f), _c(logit)
        B, C, S = _bcs(f)
        out = torch.empty_like(f)
        bern = torch.empty_like(logit)
        call("hpvg_gate_fwd_f32", ptr(f), ptr(logit), ptr(out), ptr(bern), B, C, ctypes.c_long(S), stream())
        ctx.save_for_backward(f, bern)
        return out, bern

    @staticmethod
    @once_differentiable
    def backward(ctx, dout, dbern):
        f, bern = ctx.saved_tensors
        B, C, S = _bcs(f)
        dout = _c(dout) if dout is not None else None
        dbern = _c(dbern) if dbern is not None else None
        df = torch.empty_like(f) if (ctx.needs_input_grad[0] and dout is not None) else None
        dlogit = torch.empty_like(bern) if ctx.needs_input_grad[1] else None
        call("hpvg_gate_bwd_f32", ptr(dout), ptr(f), ptr(bern), ptr(dbern), ptr(df), ptr(dlogit), B, C, ctypes.c_long(S), stream())
        return df, dlogit


class GlobalAvgPool(Function):
    """nn.AdaptiveAvgPool3d(1) / AdaptiveAvgPool2d(1): [B, C, ...] -> [B, C, 1, (1,) 1] (networks_3d.py:121-128)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        B, C, S = _bcs(x)
        out = torch.empty((B, C) + (1,) * (x.dim() - 2), dtype=torch.float32, device=x.device)
        call("hpvg_rowsum_f32", ptr(x), None, ptr(out), float(1.0 / S), B, C, ctypes.c_long(S), stream())
        ctx.shape = x.shape
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _c(g)
        B, C, S = _bcs(torch.empty(ctx.shape, device="meta"))
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        call("hpvg_outer_f32", ptr(g), None, ptr(dx), float(1.0 / S), B, C, ctypes.c_long(S), stream())
        return dx


class CodeTimesMap(Function):
    """z[b,c,s] = code[b,c] * map[b,s]: z_vae_norm [B,C,1,1,1] x z_vae_bern [B,1,T,H,W] (networks_3d.py:456)."""

    @staticmethod
    def forward(ctx, code, zmap):
        code, zmap = _c(code), _c(zmap)
        B, C = code.shape[0], code.shape[1]
        S = zmap.numel() // B
        if code.numel() != B * C or zmap.shape[1] != 1:
            raise RuntimeError("CodeTimesMap: expected a [B,C,1,..] code and a [B,1,...] map, got %s, %s" % (tuple(code.shape), tuple(zmap.shape)))
        out = torch.empty((B, C) + tuple(zmap.shape[2:]), dtype=torch.float32, device=code.device)
        call("hpvg_outer_f32", ptr(code), ptr(zmap), ptr(out), 1.0, B, C, ctypes.c_long(S), stream())
        ctx.save_for_backward(code, zmap)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        code, zmap = ctx.saved_tensors
        dz = _c(dz)
        B, C = code.shape[0], code.shape[1]
        S = zmap.numel() // B
        dcode = dmap = None
        if ctx.needs_input_grad[0]:
            dcode = torch.empty_like(code)
            call("hpvg_rowsum_f32", ptr(dz), ptr(zmap), ptr(dcode), 1.0, B, C, ctypes.c_long(S), stream())
        if ctx.needs_input_grad[1]:
            dmap = torch.empty_like(zmap)
            call("hpvg_colsum_f32", ptr(dz), ptr(code), ptr(dmap), B, C, ctypes.c_long(S), stream())
        return dcode, dmap


class ReparamBern(Function):
    """log(x + 1e-20) - log(-log(eps + 1e-20) + 1e-20), eps ~ U(0,1) (reparameterize_bern, networks_3d.py:38-42)."""

    @staticmethod
    def forward(ctx, x, eps):
        x, eps = _c(x), _c(eps)
        z = torch.empty_like(x)
        call("hpvg_reparam_bern_fwd_f32", ptr(x), ptr(eps), ptr(z), ctypes.c_long(x.numel()), stream())
        ctx.save_for_backward(x)
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        call("hpvg_reparam_bern_bwd_f32", ptr(_c(dz)), ptr(x), ptr(dx), ctypes.c_long(x.numel()), stream())
        return dx, None


class KLBern(Function):
    """mean(x (log(x+1e-20) - log .5) + (1-x)(log(1-x+1e-20) - log .5))  (kl_bern_criterion, modules/losses.py:12-14)."""

    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        out = _scalar_out(x.device)
        ws, n = _reduce_ws(x.device)
        call("hpvg_kl_bern_fwd_f32", ptr(x), ptr(out), ptr(ws), ctypes.c_size_t(n), ctypes.c_long(x.numel()), stream())
        ctx.save_for_backward(x)
        return out.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        dx = torch.empty_like(x)
        call("hpvg_kl_bern_bwd_f32", ptr(_c(g).view(1)), ptr(x), ptr(dx), ctypes.c_long(x.numel()), stream())
        return dx


class UpsampleACNoise(Function):
    """(up, up + amp * N(0,1)) with the level noise generated inside the resize kernel (no noise tensor in memory);
    samples below `first_noisy` get no noise (the reconstruction half of a merged generator pass).  Same backward as
    UpsampleAC (the noise carries no gradient)."""

    @staticmethod
    def forward(ctx, x, size, amp, first_noisy):
        x = _c(x)
        B, C, Ti, Hi, Wi = geom(x)
        if x.dim() == 5:
            To, Ho, Wo = size
            oshape = (B, C, To, Ho, Wo)
        else:
            Ho, Wo = size
            To = 1
            oshape = (B, C, Ho, Wo)
        y = torch.empty(oshape, dtype=torch.float32, device=x.device)
        yn = torch.empty_like(y)
        st = _rng(x.device)
        call("hpvg_upsample_linear_ac_noise_f32", ptr(x), ptr(y), ptr(yn), float(amp), ctypes.c_long(B * C), C, int(first_noisy), Ti,
             Hi, Wi, To, Ho, Wo, _seed(), ctypes.c_uint(st.call & 0xFFFFFFFF), ptr(st.iter_dev), stream())
        st.call += 1
        ctx.in_shape = x.shape
        ctx.dims = (B * C, Ti, Hi, Wi, To, Ho, Wo)
        return y, yn

    @staticmethod
    @once_differentiable
    def backward(ctx, dy, dyn):
        g, g2 = (dy, dyn) if dy is not None else (dyn, None)
        g = _c(g)
        g2 = _c(g2) if g2 is not None else None
        BC, Ti, Hi, Wi, To, Ho, Wo = ctx.dims
        dx = torch.empty(ctx.in_shape, dtype=torch.float32, device=g.device)
        call("hpvg_upsample_linear_ac_bwd_f32", ptr(g), ptr(g2), ptr(dx), ctypes.c_long(BC), Ti, Hi, Wi, To, Ho, Wo, stream())
        return dx, None, None, None


class Reparam(Function):
    """z = eps * exp(0.5*logvar) + mu (networks_3d.py:29-33, training branch)."""

    @staticmethod
    def forward(ctx, mu, logvar, eps):
        mu, logvar, eps = _c(mu), _c(logvar), _c(eps)
        z = torch.empty_like(mu)
        call("hpvg_reparam_fwd_f32", ptr(mu), ptr(logvar), ptr(eps), ptr(z), ctypes.c_long(mu.numel()), stream())
        ctx.save_for_backward(logvar, eps)
        return z

    @staticmethod
    @once_differentiable
    def backward(ctx, dz):
        logvar, eps = ctx.saved_tensors
        dz = _c(dz)
        dlv = torch.empty_like(dz)
        call("hpvg_reparam_bwd_f32", ptr(dz), ptr(logvar), ptr(eps), ptr(dlv), ctypes.c_long(dz.numel()), stream())
        return dz, dlv, None


# ------------------------------------------------------------------------------------------ losses
class KL(Function):
    """mean(-0.5*(1 + logvar - mu^2 - exp(logvar)))  (modules/losses.py:7-9)."""

    @staticmethod
    def forward(ctx, mu, logvar):
        mu, logvar = _c(mu), _c(logvar)
        out = _scalar_out(mu.device)
        ws, n = _reduce_ws(mu.device)
        call("hpvg_kl_fwd_f32", ptr(mu), ptr(logvar), ptr(out), ptr(ws), ctypes.c_size_t(n), ctypes.c_long(mu.numel()), stream())
        ctx.save_for_backward(mu, logvar)
        return out.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        mu, logvar = ctx.saved_tensors
        g = _c(g).view(1)
        dmu = torch.empty_like(mu)
        dlv = torch.empty_like(mu)
        call("hpvg_kl_bwd_f32", ptr(g), ptr(mu), ptr(logvar), ptr(dmu), ptr(dlv), ctypes.c_long(mu.numel()), stream())
        return dmu, dlv


class MSE(Function):
    """mean((a-b)^2)  (nn.MSELoss: train_video.py:355)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        if a.shape != b.shape:
            raise RuntimeError("mse: shape mismatch %s vs %s" % (tuple(a.shape), tuple(b.shape)))
        out = _scalar_out(a.device)
        ws, n = _reduce_ws(a.device)
        call("hpvg_mse_fwd_f32", ptr(a), ptr(b), ptr(out), ptr(ws), ctypes.c_size_t(n), ctypes.c_long(a.numel()), stream())
        ctx.save_for_backward(a, b)
        return out.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        a, b = ctx.saved_tensors
        g = _c(g).view(1)
        da = db = None
        if ctx.needs_input_grad[0]:
            da = torch.empty_like(a)
            call("hpvg_mse_bwd_f32", ptr(g), ptr(a), ptr(b), ptr(da), ctypes.c_long(a.numel()), stream())
        if ctx.needs_input_grad[1]:
            db = torch.empty_like(a)
            call("hpvg_mse_bwd_f32", ptr(g), ptr(b), ptr(a), ptr(db), ctypes.c_long(a.numel()), stream())
        return da, db


class MeanScaled(Function):
    """sign * mean(x): the WGAN critic terms (train_video.py:170,178,194)."""

    @staticmethod
    def forward(ctx, x, sign):
        x = _c(x)
        out = _scalar_out(x.device)
        ws, n = _reduce_ws(x.device)
        call("hpvg_sum_scaled_f32", ptr(x), ptr(out), ctypes.c_double(sign / x.numel()), ptr(ws), ctypes.c_size_t(n),
             ctypes.c_long(x.numel()), stream())
        ctx.shape = x.shape
        ctx.coef = sign / x.numel()
        return out.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = _c(g).view(1)
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=g.device)
        call("hpvg_fill_scaled_f32", ptr(g), float(ctx.coef), ptr(dx), ctypes.c_long(dx.numel()), stream())
        return dx, None


class GradPenalty(Function):
    """lambda * mean_{b,voxel} (||g[b,:,voxel]||_2 - 1)^2   (modules/utils.py:18; norm over dim=1)."""

    @staticmethod
    def forward(ctx, g, lam):
        g = _c(g)
        B, C, T, H, W = geom(g)
        out = _scalar_out(g.device)
        ws, n = _reduce_ws(g.device)
        call("hpvg_gp_fwd_f32", ptr(g), ptr(out), float(lam), ptr(ws), ctypes.c_size_t(n), B, C, ctypes.c_long(T * H * W),
             stream())
        ctx.save_for_backward(g)
        ctx.lam = lam
        return out.view(())

    @staticmethod
    @once_differentiable
    def backward(ctx, gout):
        (g,) = ctx.saved_tensors
        gout = _c(gout).view(1)
        B, C, T, H, W = geom(g)
        dg = torch.empty_like(g)
        call("hpvg_gp_bwd_f32", ptr(gout), ptr(g), ptr(dg), float(ctx.lam), B, C, ctypes.c_long(T * H * W), stream())
        return dg, None


_pinned_ring = []


def host_floats_to_device(values, device):
    """A few host floats -> a device tensor WITHOUT stalling the host: a plain `.to(device)` from pageable memory is a
    synchronous copy on the compute stream, i.e. the host waits for every kernel queued so far (the whole iteration up to
    the gradient penalty's alpha) and then has to re-fill the queue.  Staged through a small ring of pinned buffers."""
    vals = [float(v) for v in values]
    if device.type != "cuda":
        return torch.tensor(vals, dtype=torch.float32, device=device)
    if not _pinned_ring:
        _pinned_ring.extend([0, [torch.empty(16, dtype=torch.float32).pin_memory() for _ in range(16)]])
    slot = _pinned_ring[1][_pinned_ring[0] % 16]
    _pinned_ring[0] += 1
    for i, v in enumerate(vals):
        slot[i] = v
    return slot[:len(vals)].to(device, non_blocking=True)


def lerp(a, b, alpha):
    """alpha*a + (1-alpha)*b with a device scalar alpha (no autograd: the result becomes a leaf, modules/utils.py:9-10)."""
    a, b = _c(a.detach()), _c(b.detach())
    out = torch.empty_like(a)
    call("hpvg_lerp_f32", ptr(a), ptr(b), ptr(alpha), ptr(out), ctypes.c_long(a.numel()), stream())
    return out


# ------------------------------------------------------------------------------------------ optimizer primitives
def sqsum(flat):
    out = _scalar_out(flat.device)
    ws, n = _reduce_ws(flat.device)
    call("hpvg_sqsum_f32", ptr(flat), ptr(out), ptr(ws), ctypes.c_size_t(n), ctypes.c_long(flat.numel()), stream())
    return out


def clip_scale_(flat_grad, sq, max_norm, coef_out=None):
    call("hpvg_clip_scale_f32", ptr(flat_grad), ctypes.c_long(flat_grad.numel()), ptr(sq), float(max_norm), ptr(coef_out),
         stream())


def adam_step_(p, g, m, v, lr, beta1, beta2, eps, step, step_dev=None):
    weights_changed()
    call("hpvg_adam_step_f32", ptr(p), ptr(g), ptr(m), ptr(v), ctypes.c_long(p.numel()), float(lr), float(beta1), float(beta2),
         float(eps), int(step), ptr(step_dev), stream())


def counter_inc_(counter):
    call("hpvg_counter_inc_i32", ptr(counter), stream())
