"""Fused optimizers on flat fp32 buckets.

``FlatParams`` re-homes every parameter of a set of modules into ONE contiguous
fp32 buffer (parameters become views) with a matching flat gradient buffer, so
that (i) the optimizer step of an architecture is a single HIP kernel
(``pk_rmsprop_step`` / ``pk_sgd_step``: 16 B read + 8 B written per parameter)
and (ii) the data-parallel gradient exchange (dp.py) all-reduces slices of that
same buffer with no packing copies.

Semantics follow ``torch.optim.RMSprop`` / ``SGD`` exactly as
``utils.optimizer_init`` configures them (utils.py:2106-2164).  Parameters the
reference leaves without a gradient (its unused ``ln``/``bn`` sub-modules,
SURVEY.md 7.2) keep a zero gradient here; for momentum-free RMSprop/SGD without
weight decay a zero gradient leaves the parameter unchanged, which is what
torch's "skip grad=None" does.
"""
import ctypes

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class FlatParams:
    def __init__(self, module, align=64):
        self.module = module
        self.params = [p for p in module.parameters()]
        dev = self.params[0].device  # the flat layout itself is device-agnostic (CPU in the gloo tests)
        self.offsets = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + align - 1) // align * align
        self.numel = off
        self.flat = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grad = torch.zeros(off, device=dev, dtype=torch.float32)
        with torch.no_grad():
            for p, o in zip(self.params, self.offsets):
                n = p.numel()
                self.flat[o:o + n].copy_(p.detach().reshape(-1))
                p.data = self.flat[o:o + n].view(p.shape)
                p.grad = self.grad[o:o + n].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):  # keep .grad aliased to the flat buffer
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class FusedOptimizer:
    """One fused step per architecture.  kind: 'rmsprop' | 'sgd'."""

    def __init__(self, flat, kind, lr, alpha=0.99, eps=1e-8, momentum=0.0, weight_decay=0.0, centered=False,
                 dampening=0.0, nesterov=False):
        if kind not in ("rmsprop", "sgd"):
            raise _lib.PkError("fused optimizer covers rmsprop and sgd (got %s); use torch.optim for the rest" % kind)
        if kind == "rmsprop" and (momentum != 0.0 or centered):
            raise _lib.PkError("fused RMSprop is the momentum-free, non-centred form every shipped recipe uses")
        if kind == "sgd" and (dampening != 0.0 or nesterov):
            raise _lib.PkError("fused SGD does not implement dampening / nesterov")
        self.flat, self.kind, self.lr = flat, kind, lr
        self.alpha, self.eps, self.momentum, self.weight_decay = alpha, eps, momentum, weight_decay
        self.state = torch.zeros_like(flat.flat) if (kind == "rmsprop" or momentum != 0.0) else None
        self.steps = 0
        self.param_groups = [{"lr": lr}]  # run_nn overrides the LR through param_groups (core.py:533-535)

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self):
        if not self.flat.flat.is_cuda:
            raise _lib.PkError("fused optimizer step runs on the GPU only (no CPU fallback)")
        lib = _lib.load()
        lr = float(self.param_groups[0]["lr"])
        f = self.flat
        if self.kind == "rmsprop":
            rc = lib.pk_rmsprop_step(_stream(), f.flat.data_ptr(), f.grad.data_ptr(), self.state.data_ptr(), f.numel, lr,
                                     self.alpha, self.eps, self.weight_decay)
        else:
            st = self.state.data_ptr() if self.state is not None else None
            rc = lib.pk_sgd_step(_stream(), f.flat.data_ptr(), f.grad.data_ptr(), st, f.numel, lr, self.momentum,
                                 self.weight_decay, int(self.steps == 0))
        _lib.check(rc, "fused optimizer step")
        self.steps += 1

    def state_dict(self):
        return {"kind": self.kind, "steps": self.steps, "lr": self.param_groups[0]["lr"],
                "state": None if self.state is None else self.state.clone()}

    def load_state_dict(self, sd):
        self.steps = sd["steps"]
        self.param_groups[0]["lr"] = sd["lr"]
        if sd["state"] is not None:
            self.state.copy_(sd["state"])


def fused_optimizer_init(nns, config, arch_dict):
    """Same signature and hyper-parameter fields as utils.optimizer_init (utils.py:2106-2164),
    returning fused optimizers; falls back to an error (never silently to torch) when a recipe
    asks for a form the fused kernels do not cover."""
    from .utils import strtobool

    opts = {}
    for net in nns.keys():
        sec = config[arch_dict[net][0]]
        lr = float(sec["arch_lr"])
        kind = sec["arch_opt"]
        flat = FlatParams(nns[net])
        if kind == "rmsprop":
            opts[net] = FusedOptimizer(flat, "rmsprop", lr, alpha=float(sec["opt_alpha"]), eps=float(sec["opt_eps"]),
                                       momentum=float(sec["opt_momentum"]),
                                       weight_decay=float(sec["opt_weight_decay"]),
                                       centered=bool(strtobool(sec["opt_centered"])))
        elif kind == "sgd":
            opts[net] = FusedOptimizer(flat, "sgd", lr, momentum=float(sec["opt_momentum"]),
                                       weight_decay=float(sec["opt_weight_decay"]),
                                       dampening=float(sec["opt_dampening"]),
                                       nesterov=bool(strtobool(sec["opt_nesterov"])))
        else:
            raise _lib.PkError("fused optimizer: arch_opt=%s is not covered" % kind)
    return opts
