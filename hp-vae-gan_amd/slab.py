"""Row-slab decomposition of one pyramid level over neighbouring GPUs: the H axis of a level is
cut into contiguous slabs, one per rank; every 3x3(x3) 'same' convolution first swaps ONE boundary row with each
neighbour (a [B, C, T, 1, W] message: 0.85 MB for a 64-channel layer of the finest video level, one xGMI link,
point to point), runs the ordinary kernel on the slab extended by those halo rows and drops the halo rows of the
result.  BatchNorm sums, the 3-channel level outputs and the scalar losses are the only other things that cross
ranks (multigpu.py).

Everything here is communication glue in plain torch (no kernels): the arithmetic stays in ops.* / the oracle, which
is why the same pieces serve the gfx950 path and the gloo CPU tests.  The exchanges are autograd Functions whose
backward is again an exchange, so first- and second-order backward (the WGAN-GP double backward through the
discriminator) work on slabs: the graph is the same on both sides of a boundary, the autograd engine replays it in
the same order, and the swaps pair up."""
import torch
from torch.autograd import Function
from torch.autograd.function import once_differentiable


def slab_rows(H, h, nslab):
    """Rows [r0, r1) of slab h when H rows are cut into nslab contiguous slabs (the first H % nslab slabs get one more)."""
    base, extra = divmod(int(H), int(nslab))
    r0 = h * base + min(h, extra)
    return r0, r0 + base + (1 if h < extra else 0)


class Exchange(Function):
    """y = the tensor the neighbour passed to its matching Exchange (same shape).  `swap(t) -> tensor` performs the
    pairwise send/receive.  The operation is linear and its transpose is itself (what I sent is what the neighbour
    received), so backward(g) = Exchange(g): differentiable to any order."""

    @staticmethod
    def forward(ctx, t, swap):
        ctx.swap = swap
        return swap(t.contiguous())

    @staticmethod
    def backward(ctx, g):
        return Exchange.apply(g, ctx.swap), None


class RowGather(Function):
    """Whole level from its slabs: out[..., r0:r1, :] = x on every rank, combined by `allreduce` (in-place sum over the
    slab group; the level outputs are 3-channel, so this is a few MB at most).  backward: every rank holds the gradient
    contribution of ITS slab of the next level w.r.t. the whole tensor -> sum over the group, keep the own rows."""

    @staticmethod
    def forward(ctx, x, r0, H, allreduce):
        d = x.dim() - 2
        shape = list(x.shape)
        shape[d] = H
        full = x.new_zeros(shape)
        full.narrow(d, r0, x.shape[d]).copy_(x)
        allreduce(full)
        ctx.r0, ctx.n, ctx.allreduce = r0, x.shape[d], allreduce
        return full

    @staticmethod
    @once_differentiable
    def backward(ctx, g):
        g = g.contiguous().clone()
        ctx.allreduce(g)
        return g.narrow(g.dim() - 2, ctx.r0, ctx.n).contiguous(), None, None, None


class Halo:
    """Neighbours of this rank's slab: `up` / `down` are swap callables (tensor -> the neighbour's tensor) or None at
    the image border."""

    def __init__(self, up=None, down=None):
        self.up, self.down = up, down


def conv_with_halo(x, halo, conv):
    """conv = a stride-1 'same' 3-tap convolution over [..., H, W]; returns conv(whole image) restricted to this slab.
    The rows next to an interior boundary see the neighbour's row instead of zero padding; the outputs computed AT the
    halo rows (which would need a second neighbour row) are dropped."""
    d = x.dim() - 2
    n = x.shape[d]
    parts = []
    if halo.up is not None:
        parts.append(Exchange.apply(x.narrow(d, 0, 1), halo.up))
    parts.append(x)
    if halo.down is not None:
        parts.append(Exchange.apply(x.narrow(d, n - 1, 1), halo.down))
    y = conv(torch.cat(parts, dim=d))
    return y.narrow(d, 1 if halo.up is not None else 0, n).contiguous()


class SlabPlan:
    """How one generator pass is cut: this rank is slab `h` of `nslab`; levels >= `min_level` are computed on slabs
    (lower levels are small and are replicated on the slab ranks), `halo` carries the neighbour swaps, `allreduce`
    sums a tensor over the slab ranks of this (pass, sample)."""

    def __init__(self, h, nslab, min_level, halo, allreduce):
        self.h, self.nslab, self.min_level, self.halo, self.allreduce = h, nslab, min_level, halo, allreduce

    def covers(self, level):
        return level >= self.min_level

    def rows(self, H):
        return slab_rows(H, self.h, self.nslab)

    def cut(self, x):
        d = x.dim() - 2
        r0, r1 = self.rows(x.shape[d])
        return x.narrow(d, r0, r1 - r0).contiguous()

    def gather(self, x, H):
        return RowGather.apply(x, self.rows(H)[0], int(H), self.allreduce)
