"""Fused optimizers on flat fp32 buckets.

``FlatParams`` re-homes every parameter of a set of modules into ONE contiguous
fp32 buffer (parameters become views) with a matching flat gradient buffer, so
that (i) the optimizer step of an architecture is a single HIP kernel
(``pk_rmsprop_step`` / ``pk_sgd_step`` / ``pk_adam_step``: RMSprop reads 16 B and writes 8 B per parameter)
and (ii) the data-parallel gradient exchange (dp.py) all-reduces slices of that
same buffer with no packing copies.

Semantics follow ``torch.optim.RMSprop`` / ``SGD`` / ``Adam`` exactly as
``utils.optimizer_init`` configures them (utils.py:2106-2164).  Parameters the
reference leaves without a gradient (its unused ``ln``/``bn`` sub-modules,
SURVEY.md 7.2; a module lists them in ``pk_unused_parameters()``) are laid out
BEHIND the active ones: the fused step covers the active range only, so they are
never touched - also with weight decay, momentum or Adam, where a zero gradient
would still move them - and ``state_dict()`` has no entry for them, exactly like
torch's "skip grad=None".
"""
import ctypes
import os
import weakref

import torch

from . import _lib


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class FlatParams:
    def __init__(self, module, align=64):
        """Layout of the flat buffer: the parameters forward actually uses first, the never-used ones behind them
        (`n_active`).  A module may state a layout for its active parameters through ``pk_flat_groups()`` - a list of
        parameter lists in buffer order: the members of a group are packed back to back (the gates of a recurrent layer:
        their concatenation is then a VIEW of the buffer, and so is the gradient the weight-gradient GEMM accumulates
        into), groups follow each other in the given order (layer-major: the buckets of dp.GradReducer, which walk the
        buffer from the top, then complete in the order backward produces gradients).  torch-format optimizer state
        is indexed by registration order and does not see the layout."""
        self.module = module
        self.params = [p for p in module.parameters()]  # registration order: what indexes torch optimizer state
        dev = self.params[0].device  # the flat layout itself is device-agnostic (CPU in the gloo tests)
        unused = {id(p) for p in module.pk_unused_parameters()} if hasattr(module, "pk_unused_parameters") else set()
        self.unused = [id(p) in unused for p in self.params]
        index = {id(p): i for i, p in enumerate(self.params)}
        units, placed = [], set()
        if hasattr(module, "pk_flat_groups") and _lib.experiment("flat_groups", "1") != "0":
            for grp in module.pk_flat_groups():
                ids = [index[id(q)] for q in grp if id(q) in index and id(q) not in unused and index[id(q)] not in placed]
                if ids:
                    units.append(ids)
                    placed.update(ids)
        for i in range(len(self.params)):  # everything the module did not place: registration order, one by one
            if not self.unused[i] and i not in placed:
                units.append([i])
        self.offsets = [0] * len(self.params)
        off = 0
        for ids in units:
            off = (off + align - 1) // align * align
            for i in ids:
                self.offsets[i] = off
                off += self.params[i].numel()
        off = (off + align - 1) // align * align
        self.n_active = off
        for i, p in enumerate(self.params):  # the never-used ones behind the active range
            if self.unused[i]:
                self.offsets[i] = off
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
                p._pk_flat = True  # functional.side_targets_ok: weight gradients may accumulate here from a side stream
                p._pk_owner = weakref.ref(self)  # functional.weight_bf16: who keeps this weight's bf16 copy fresh
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._clean = False      # the gradient buffer is known to be all zeros (the fused step zeroed it on its way out)
        self._shadow_req = set() # parameters whose bf16 copy the perf-mode GEMMs asked for (functional.weight_bf16)
        self._shadow_have = ()   # ... the ones a copy exists for, in buffer order
        self.shadow = self.shadow_segs = None

    # ---- bf16 copies of 2-D weights, refreshed by the fused optimizer step itself (pk_fused_step) -------------------
    def want_shadow(self, p):
        i = self._index.get(id(p))
        # (pk_fused_step's threads own four consecutive elements of the flat buffer: a weight's rows AND its offset in
        # the buffer must be multiples of 4, or a quad would straddle the segment start / two rows)
        if (i is not None and not self.unused[i] and p.dim() == 2 and p.shape[1] % 4 == 0 and self.offsets[i] % 4 == 0
                and p.is_cuda):
            self._shadow_req.add(i)

    def invalidate_shadows(self):
        """Call after writing the flat buffer through anything but torch in-place ops on the Parameters or the fused
        step (a broadcast into ``flat``, ``p.data.copy_``, a raw-pointer kernel): those do not move the Parameters'
        version counters, which is what functional.weight_bf16 trusts a bf16 copy by.  The copies are redone at their
        next use."""
        for p in self.params:
            sh = getattr(p, "_pk_shadow", None)
            if sh is not None:
                p._pk_shadow = (sh[0], -1, sh[2], -1)

    def build_shadows(self):
        """(Re)build the copies for everything asked for so far: one bf16 buffer, a weight's rows at a pitch of its
        columns rounded up to 64 (what functional.cvt_bf16 gives), weights in buffer order back to back; pad columns
        stay zero.  Allocates: call outside a HIP-graph capture (FusedOptimizer.step does, in the eager warm-up steps)."""
        from . import functional as F_
        want = tuple(sorted(self._shadow_req, key=lambda i: self.offsets[i]))
        if want == self._shadow_have:
            return
        rows = []
        soff = 0
        for i in want:
            p = self.params[i]
            pitch = (p.shape[1] + 63) // 64 * 64
            rows.append((self.offsets[i], p.shape[0], p.shape[1], pitch, soff))
            soff += p.shape[0] * pitch
        self.shadow = torch.zeros(max(soff, 8), device=self.flat.device, dtype=torch.bfloat16)
        self.shadow_segs = torch.tensor(rows, dtype=torch.int64, device=self.flat.device).reshape(-1, 5)
        self._shadow_have = want
        for i, (_o, r, c, pitch, so) in zip(want, rows):
            p = self.params[i]
            view = self.shadow[so:so + r * pitch].view(r, pitch)
            F_.cvt_bf16(p.detach(), out=view)
            p._pk_shadow = (view, p._version, weakref.ref(self.shadow), self.flat._version)

    def zero_grad(self):
        from .functional import join_side
        join_side()
        # (contract of zero_in_step: between step() and this call nothing writes .grad - one backward pass per step; a
        # loop that accumulates gradients over several backward passes must call mark_dirty() or leave zero_in_step off)
        if self._clean:
            self._clean = False  # (whatever comes next may write gradients)
        else:
            self.grad.zero_()
        for p, o in zip(self.params, self.offsets):  # keep .grad aliased to the flat buffer
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


def _flat_mark_dirty(self):
    """Somebody wrote (or is about to write) .grad outside the zero_grad() -> backward -> step() cycle: the next
    zero_grad() must really fill."""
    self._clean = False


FlatParams.mark_dirty = _flat_mark_dirty


class FusedOptimizer:
    """One fused step per architecture.  kind: 'rmsprop' | 'sgd' | 'adam'.

    ``state_dict()`` / ``load_state_dict()`` speak torch.optim's own format (per-parameter ``state`` entries,
    ``param_groups`` with the torch hyper-parameter names), so the ``optimizer_par`` of a ``.pkl`` checkpoint
    written by the reference loads here and vice versa (core.py:531-532, 708-722)."""

    _STATE_KEYS = {"rmsprop": ("square_avg",), "sgd": ("momentum_buffer",),
                   "adam": ("exp_avg", "exp_avg_sq", "max_exp_avg_sq")}

    def __init__(self, flat, kind, lr, alpha=0.99, eps=1e-8, momentum=0.0, weight_decay=0.0, centered=False,
                 dampening=0.0, nesterov=False, betas=(0.9, 0.999), amsgrad=False):
        if kind not in ("rmsprop", "sgd", "adam"):
            raise _lib.PkError("fused optimizer covers rmsprop, sgd and adam (got %s)" % kind)
        if kind == "rmsprop" and (momentum != 0.0 or centered):
            raise _lib.PkError("fused RMSprop is the momentum-free, non-centred form every shipped recipe uses")
        if kind == "sgd" and (dampening != 0.0 or nesterov):
            raise _lib.PkError("fused SGD does not implement dampening / nesterov")
        self.flat, self.kind, self.lr = flat, kind, lr
        self.alpha, self.eps, self.momentum, self.weight_decay = alpha, eps, momentum, weight_decay
        self.betas, self.amsgrad = tuple(betas), bool(amsgrad)
        z = lambda: torch.zeros_like(flat.flat)
        self.bufs = {}
        if kind == "rmsprop":
            self.bufs["square_avg"] = z()
        elif kind == "sgd" and momentum != 0.0:
            self.bufs["momentum_buffer"] = z()
        elif kind == "adam":
            self.bufs["exp_avg"], self.bufs["exp_avg_sq"] = z(), z()
            if self.amsgrad:
                self.bufs["max_exp_avg_sq"] = z()
        self.steps = 0
        self.param_groups = [{"lr": lr}]  # run_nn overrides the LR through param_groups (core.py:533-535)
        # True: step() leaves the flat gradient zeroed (and the following zero_grad() skips its fill launch).  For loops
        # that call zero_grad() in front of every backward pass anyway - core.run_nn_dp, bench.py - and never read .grad
        # behind step(); off by default (torch.optim leaves .grad alone).  PK_EXPERIMENT opt_zero_in_step=0 turns it off everywhere.
        self.zero_in_step = False

    @property
    def state(self):  # the (first) flat state buffer; kept for callers that only need "the" state
        return next(iter(self.bufs.values()), None)

    def zero_grad(self):
        self.flat.zero_grad()

    def step(self):
        if not self.flat.flat.is_cuda:
            raise _lib.PkError("fused optimizer step runs on the GPU only (no CPU fallback)")
        from .functional import join_side
        join_side()  # weight-gradient GEMMs still running on the side stream
        lib = _lib.load()
        lr = float(self.param_groups[0]["lr"])
        f = self.flat
        ptr = lambda k: self.bufs[k].data_ptr() if k in self.bufs else None
        if f.n_active % 4 == 0 and _lib.experiment("fused_step", "1") != "0":
            # one launch that also zeroes the gradient (zero_in_step) and refreshes the bf16 copies of the weights
            if f._shadow_req and not torch.cuda.is_current_stream_capturing():
                f.build_shadows()
            nseg = len(f._shadow_have)
            zero = bool(self.zero_in_step) and _lib.experiment("opt_zero_in_step", "1") != "0"
            kind = {"rmsprop": 0, "sgd": 1, "adam": 2}[self.kind]
            h = {"rmsprop": (self.alpha, self.eps, 0.0), "sgd": (self.momentum, 0.0, 0.0),
                 "adam": (self.betas[0], self.betas[1], self.eps)}[self.kind]
            s0 = ptr("square_avg") if kind == 0 else ptr("momentum_buffer") if kind == 1 else ptr("exp_avg")
            rc = lib.pk_fused_step(_stream(), kind, f.flat.data_ptr(), f.grad.data_ptr(), s0, ptr("exp_avg_sq"),
                                   ptr("max_exp_avg_sq"), f.n_active, lr, h[0], h[1], h[2], self.weight_decay, self.steps + 1,
                                   int(zero), f.shadow_segs.data_ptr() if nseg else None, nseg,
                                   f.shadow.data_ptr() if nseg else None)
            _lib.check(rc, "fused optimizer step")
            f._clean = zero
            self.steps += 1
            return
        if self.kind == "rmsprop":
            rc = lib.pk_rmsprop_step(_stream(), f.flat.data_ptr(), f.grad.data_ptr(), ptr("square_avg"), f.n_active, lr,
                                     self.alpha, self.eps, self.weight_decay)
        elif self.kind == "sgd":
            rc = lib.pk_sgd_step(_stream(), f.flat.data_ptr(), f.grad.data_ptr(), ptr("momentum_buffer"), f.n_active, lr,
                                 self.momentum, self.weight_decay, int(self.steps == 0))
        else:
            rc = lib.pk_adam_step(_stream(), f.flat.data_ptr(), f.grad.data_ptr(), ptr("exp_avg"), ptr("exp_avg_sq"),
                                  ptr("max_exp_avg_sq"), f.n_active, lr, self.betas[0], self.betas[1], self.eps,
                                  self.weight_decay, self.steps + 1)
        _lib.check(rc, "fused optimizer step")
        self.steps += 1

    def _torch_twin(self):
        """The torch optimizer utils.optimizer_init would have built (never stepped): source of the
        param_groups layout of the installed torch version."""
        import torch.optim as optim
        ps = self.flat.params
        lr = float(self.param_groups[0]["lr"])
        if self.kind == "rmsprop":
            return optim.RMSprop(ps, lr=lr, alpha=self.alpha, eps=self.eps, weight_decay=self.weight_decay)
        if self.kind == "sgd":
            return optim.SGD(ps, lr=lr, momentum=self.momentum, weight_decay=self.weight_decay)
        return optim.Adam(ps, lr=lr, betas=self.betas, eps=self.eps, weight_decay=self.weight_decay, amsgrad=self.amsgrad)

    def state_dict(self):
        sd = self._torch_twin().state_dict()
        state = {}
        if self.steps > 0:
            for i, (p, o) in enumerate(zip(self.flat.params, self.flat.offsets)):
                if self.flat.unused[i]:  # torch never steps a parameter whose grad stays None: no state entry
                    continue
                ent = {k: b[o:o + p.numel()].view(p.shape).clone() for k, b in self.bufs.items()}
                if self.kind != "sgd":
                    ent["step"] = torch.tensor(float(self.steps))
                if ent:
                    state[i] = ent
        sd["state"] = state
        return sd

    def load_state_dict(self, sd):
        groups = sd["param_groups"]
        if len(groups) != 1 or len(groups[0]["params"]) != len(self.flat.params):
            raise _lib.PkError("optimizer state does not match this architecture's parameter list")
        g = groups[0]
        self.param_groups[0]["lr"] = g["lr"]
        self.weight_decay = float(g.get("weight_decay", self.weight_decay))
        if self.kind == "rmsprop":
            self.alpha, self.eps = float(g.get("alpha", self.alpha)), float(g.get("eps", self.eps))
        elif self.kind == "sgd":
            self.momentum = float(g.get("momentum", self.momentum))
        else:
            self.betas, self.eps = tuple(g.get("betas", self.betas)), float(g.get("eps", self.eps))
        steps = 0
        for b in self.bufs.values():
            b.zero_()
        for i, ent in sd["state"].items():  # parameters torch never stepped (grad None) have no entry: zeros
            i = int(i)
            p, o = self.flat.params[i], self.flat.offsets[i]
            for k, b in self.bufs.items():
                if ent.get(k) is not None:
                    b[o:o + p.numel()].copy_(ent[k].reshape(-1))
            if "step" in ent:
                steps = max(steps, int(float(ent["step"])))
            elif ent.get("momentum_buffer") is not None:
                steps = max(steps, 1)
        self.steps = steps


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
        elif kind == "adam":
            opts[net] = FusedOptimizer(flat, "adam", lr, betas=tuple(map(float, sec["opt_betas"].split(","))),
                                       eps=float(sec["opt_eps"]), weight_decay=float(sec["opt_weight_decay"]),
                                       amsgrad=bool(strtobool(sec["opt_amsgrad"])))
        else:
            raise _lib.PkError("fused optimizer: arch_opt=%s is not covered" % kind)
    return opts
