"""Flat parameter / gradient arenas, Adam and gradient clipping on the gfx950 kernels.

Replaces, for the hot path, optim.Adam(parameter_list, betas=(beta1, 0.999)) (train_video.py:55,88) and
torch.nn.utils.clip_grad_norm_(G.parameters(), grad_clip) (train_video.py:201).  All parameters of a module live in
ONE contiguous fp32 buffer (and their gradients in a second one), so the global L2 norm is a single reduction and
an Adam step is one launch per parameter group instead of one per tensor."""
import torch

from . import ops

_ALIGN = 64  # floats: every parameter starts on a 256-byte boundary


class ParamArena:
    """Re-homes every parameter of `module` into one flat buffer and pre-binds .grad to views of a flat gradient
    buffer (autograd then accumulates in place).  Build it after .to(device) / init_next_stage(); values are kept."""

    def __init__(self, module):
        params = list(module.parameters())
        if not params:
            raise ValueError("module has no parameters")
        dev = params[0].device
        offs, total = [], 0
        for p in params:
            offs.append(total)
            total += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.range = {}
        with torch.no_grad():
            for p, o in zip(params, offs):
                n = p.numel()
                self.flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)
                self.range[id(p)] = (o, n)
        self.params = params
        self.total = total

    def span(self, params):
        """[lo, hi) flat range covering `params` (they must be contiguous in the arena)."""
        rs = sorted(self.range[id(p)] for p in params)
        lo = rs[0][0]
        hi = rs[-1][0] + (rs[-1][1] + _ALIGN - 1) // _ALIGN * _ALIGN
        covered = sum((n + _ALIGN - 1) // _ALIGN * _ALIGN for _, n in rs)
        if covered != hi - lo:
            raise ValueError("parameter group is not contiguous in the arena")
        return lo, hi

    def zero_grad(self):
        self.grad.zero_()
        for p in self.params:  # re-bind in case autograd replaced a .grad tensor
            o, n = self.range[id(p)]
            g = p.grad
            if g is None or g.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + n].view(p.shape)

    def clip_grad_norm_(self, max_norm, info=None):
        """g *= min(1, max_norm / (||g||_2 + 1e-6)) over the whole arena; `info` (2 floats) receives (coef, norm)."""
        sq = ops.sqsum(self.grad)
        ops.clip_scale_(self.grad, sq, max_norm, info)


class BufferArena:
    """Re-homes the floating-point buffers of `module` whose names end with `suffixes` (default: BatchNorm running
    statistics) into ONE flat tensor (`flat`); the module's buffers become views of it, so kernels that update them in place
    keep working and the whole set can be exchanged / combined between ranks as one tensor (multigpu.DistStageTrainer).
    Build it after .to(device); values are kept."""

    def __init__(self, module, suffixes=("running_mean", "running_var")):
        items = []
        for mod in module.modules():
            for name, buf in mod._buffers.items():
                if buf is not None and buf.dtype.is_floating_point and name.endswith(tuple(suffixes)):
                    items.append((mod, name, buf))
        self.items = items
        total = sum(b.numel() for _, _, b in items)
        self.flat = None
        if not items:
            return
        self.flat = torch.empty(total, dtype=items[0][2].dtype, device=items[0][2].device)
        o = 0
        with torch.no_grad():
            for mod, name, buf in items:
                n = buf.numel()
                self.flat[o:o + n].copy_(buf.reshape(-1))
                mod._buffers[name] = self.flat[o:o + n].view(buf.shape)
                o += n


class FlatAdam:
    """torch.optim.Adam semantics (eps 1e-8, no weight decay, no amsgrad) over arena ranges, one lr per group."""

    def __init__(self, arena, groups, betas=(0.5, 0.999), eps=1e-8):
        """groups: list of (iterable of parameters, lr)."""
        self.arena = arena
        self.betas = betas
        self.eps = eps
        self.groups = []
        for params, lr in groups:
            params = list(params)
            if not params:
                continue
            lo, hi = arena.span(params)
            # (offset inside the group, numel, shape) per parameter, in the group's own order: state_dict() hands the
            # moments out per parameter, as torch.optim.Adam does
            layout = [(arena.range[id(p)][0] - lo, p.numel(), tuple(p.shape)) for p in params]
            self.groups.append({"lo": lo, "hi": hi, "lr": lr, "layout": layout,
                                "m": torch.zeros(hi - lo, dtype=torch.float32, device=arena.flat.device),
                                "v": torch.zeros(hi - lo, dtype=torch.float32, device=arena.flat.device)})
        self.t = 0
        # device-resident copy of the step count: a hipGraph replay of step() must not bake the count into the launch
        self.t_dev = torch.zeros(1, dtype=torch.int32, device=arena.flat.device)

    def step(self):
        self.t += 1   # host mirror (eager steps only; replays of a captured step advance t_dev alone: see steps())
        ops.counter_inc_(self.t_dev)
        for g in self.groups:
            lo, hi = g["lo"], g["hi"]
            ops.adam_step_(self.arena.flat[lo:hi], self.arena.grad[lo:hi], g["m"], g["v"], g["lr"], self.betas[0], self.betas[1],
                           self.eps, self.t, self.t_dev)

    def steps(self):
        """Optimizer steps taken so far - read from the device counter (the only one hipGraph replays advance)."""
        return int(self.t_dev.item())

    def state_dict(self):
        """torch.optim.Adam's state_dict layout ({'state': {i: {'step', 'exp_avg', 'exp_avg_sq'}}, 'param_groups': [...]}),
        CPU tensors, parameters numbered group by group in the order the groups were given - what the reference writes
        into netG.pth / netD_<s>.pth (train_video.py:246-258); torch.optim.Adam.load_state_dict accepts it for an optimizer
        built over the same groups."""
        t = self.steps()
        state, param_groups, idx = {}, [], 0
        for g in self.groups:
            ids = []
            m, v = g["m"].detach().cpu(), g["v"].detach().cpu()
            for o, n, shape in g["layout"]:
                if t > 0:
                    state[idx] = {"step": torch.tensor(float(t)), "exp_avg": m[o:o + n].view(shape).clone(),
                                  "exp_avg_sq": v[o:o + n].view(shape).clone()}
                ids.append(idx)
                idx += 1
            param_groups.append({"lr": g["lr"], "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                                 "params": ids})
        return {"state": state, "param_groups": param_groups}
