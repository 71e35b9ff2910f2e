"""Drop-in ``arch_library`` for PyTorch-Kaldi recipes: the acoustic-model classes of
the reference's ``neural_networks.py`` re-built on the MI355X engine.

Use it by changing one cfg line per architecture::

    arch_library = pytorch-kaldi_amd.nn        # was: neural_networks
    arch_class   = liGRU                       # unchanged

Contract kept from the reference:
  * ``Class(options, inp_dim)`` where ``options`` is the cfg section (all values are
    strings, keys may be lower-cased by configparser) with the injected ``use_cuda``
    and ``to_do`` fields (utils.py:2051-2057); ``.out_dim``; ``forward(x)`` with
    ``x`` = (T, B, F) for sequence models and (N, F) otherwise;
  * parameter / buffer names and shapes identical to the reference, so ``.pkl``
    checkpoints interoperate (core.py:523-535, 710-722), including the unused
    ``ln``/``bn`` sub-modules the reference always creates;
  * sub-modules are created in the reference's order with the same torch
    initialisers, so the same seed gives the same initial weights;
  * recurrent drop masks are Bernoulli(1-p), sampled per layer per forward call in layer
    order, unscaled and constant over time like the reference's
    (e.g. neural_networks.py:1102-1107); PK_MASK_RNG=reference draws them with the
    reference's own CPU-RNG call (bit-identical stream), the default draws on the GPU.

The nn.Linear / nn.BatchNorm1d / nn.Conv1d sub-modules are parameter containers
only: ``forward`` never calls them, it hands their tensors to the HIP kernels
through ``functional``.  There is no CPU path: ``use_cuda`` must be True.
"""
import math
import os
import weakref

import numpy as np
import torch
import torch.nn as nn

from . import functional as F_
from ._lib import PkError


def strtobool(s):
    s = str(s).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return 1
    if s in ("n", "no", "f", "false", "off", "0"):
        return 0
    raise ValueError("invalid truth value %r" % (s,))


def _opt(options, key):
    try:
        return options[key]
    except KeyError:
        return options[key.lower()]


def _ints(s):
    return list(map(int, str(s).split(",")))


def _floats(s):
    return list(map(float, str(s).split(",")))


def _bools(s):
    return list(map(strtobool, str(s).split(",")))


class LayerNorm(nn.Module):
    """gamma*(x-mean)/(std_unbiased+eps)+beta over the last dim (neural_networks.py:23-33)."""

    def __init__(self, features, eps=1e-6):
        super().__init__()
        self.gamma = nn.Parameter(torch.ones(features))
        self.beta = nn.Parameter(torch.zeros(features))
        self.eps = eps

    def forward(self, x):
        if self.gamma.dim() == 1:
            return F_.layer_norm(x, self.gamma, self.beta, self.eps)
        return F_.layer_norm_last(x, self.gamma, self.beta, self.eps)


class _Act(nn.Module):
    """Placeholder kept so that ``self.act`` exists like in the reference (no parameters)."""

    def __init__(self, name):
        super().__init__()
        self.name = name


def act_fun(act_type):
    """neural_networks.py:36-57; returns a parameter-free marker module."""
    if act_type not in ("relu", "tanh", "sigmoid", "leaky_relu", "elu", "softmax", "linear"):
        raise ValueError("unknown activation %r" % (act_type,))
    return _Act(act_type)


def flip(x, dim):
    """Time reversal (neural_networks.py:1962-1970); the engine folds it into kernel indexing,
    this helper exists for API parity."""
    return torch.flip(x, dims=[dim])


def _apply_act_drop(x2, bn, use_bn, training, act, drop_p, eps=None):
    """drop(act(bn(x))) on a 2-D tensor; 'softmax' is LogSoftmax(dim=1)."""
    mask = None
    if training and drop_p > 0.0:
        mask = F_.dropout_mask(x2, drop_p)
    if act == "softmax":
        y = F_.norm_act_drop(x2, bn, use_bn, training, "linear", None, eps) if use_bn else x2
        y = F_.log_softmax(y)
        return y * mask if mask is not None else y
    return F_.norm_act_drop(x2, bn, use_bn, training, act, mask, eps)


class MLP(nn.Module):
    """neural_networks.py:60-150."""

    def __init__(self, options, inp_dim):
        super().__init__()
        self.input_dim = inp_dim
        self.dnn_lay = _ints(_opt(options, "dnn_lay"))
        self.dnn_drop = _floats(_opt(options, "dnn_drop"))
        self.dnn_use_batchnorm = _bools(_opt(options, "dnn_use_batchnorm"))
        self.dnn_use_laynorm = _bools(_opt(options, "dnn_use_laynorm"))
        self.dnn_use_laynorm_inp = strtobool(_opt(options, "dnn_use_laynorm_inp"))
        self.dnn_use_batchnorm_inp = strtobool(_opt(options, "dnn_use_batchnorm_inp"))
        self.dnn_act = str(_opt(options, "dnn_act")).split(",")
        self.wx = nn.ModuleList([])
        self.bn = nn.ModuleList([])
        self.ln = nn.ModuleList([])
        self.act = nn.ModuleList([])
        self.drop = nn.ModuleList([])
        if self.dnn_use_laynorm_inp:
            self.ln0 = LayerNorm(self.input_dim)
        if self.dnn_use_batchnorm_inp:
            self.bn0 = nn.BatchNorm1d(self.input_dim, momentum=0.05)
        self.N_dnn_lay = len(self.dnn_lay)
        cur = self.input_dim
        for i in range(self.N_dnn_lay):
            self.drop.append(nn.Dropout(p=self.dnn_drop[i]))
            self.act.append(act_fun(self.dnn_act[i]))
            self.ln.append(LayerNorm(self.dnn_lay[i]))
            self.bn.append(nn.BatchNorm1d(self.dnn_lay[i], momentum=0.05))
            add_bias = not (self.dnn_use_laynorm[i] or self.dnn_use_batchnorm[i])
            self.wx.append(nn.Linear(cur, self.dnn_lay[i], bias=add_bias))
            bound = np.sqrt(0.01 / (cur + self.dnn_lay[i]))
            self.wx[i].weight = nn.Parameter(torch.Tensor(self.dnn_lay[i], cur).uniform_(-bound, bound))
            self.wx[i].bias = nn.Parameter(torch.zeros(self.dnn_lay[i]))  # the reference always re-creates it (:120)
            cur = self.dnn_lay[i]
        self.out_dim = cur

    def pk_unused_parameters(self):
        """Parameters forward never touches (the reference creates ln / bn for every layer and calls what the flags
        select, :139-148): torch leaves their .grad None and its optimizers skip them; optim.FlatParams does the same."""
        out = []
        for i in range(self.N_dnn_lay):
            if not self.dnn_use_laynorm[i]:
                out += list(self.ln[i].parameters())
            if not self.dnn_use_batchnorm[i]:
                out += list(self.bn[i].parameters())
        return out

    def pk_flat_groups(self):
        """optim.FlatParams layout: layer by layer (registration order is kind by kind - all wx, then all bn, ... -
        which makes every gradient bucket of dp.GradReducer wait for the FIRST layer's backward)."""
        out = []
        if self.dnn_use_laynorm_inp:
            out.append(list(self.ln0.parameters()))
        if self.dnn_use_batchnorm_inp:
            out.append(list(self.bn0.parameters()))
        for i in range(self.N_dnn_lay):
            out.append(list(self.wx[i].parameters()))
            out.append(list(self.ln[i].parameters()))
            out.append(list(self.bn[i].parameters()))
        return out

    def forward(self, x):
        if self.dnn_use_laynorm_inp:
            x = self.ln0(x)
        if self.dnn_use_batchnorm_inp:
            x = F_.norm_act_drop(x, self.bn0, True, self.training, "linear")
        if self.training and x.dim() == 2 and x.is_cuda:
            # a 128-frame step is launch-bound: all dropout masks of the stack in two launches per distinct p, and the
            # batch counters of the one-launch layers in one
            F_.masks_ahead([(x.shape[0], self.dnn_lay[i], self.dnn_drop[i]) for i in range(self.N_dnn_lay)
                            if self.dnn_drop[i] > 0.0], x.device)
        counted = []
        for i in range(self.N_dnn_lay):
            if (self.dnn_act[i] == "softmax" and not self.dnn_use_laynorm[i] and not self.dnn_use_batchnorm[i]
                    and not (self.training and self.dnn_drop[i] > 0.0)
                    and F_.linear_log_softmax_ok(x, self.wx[i].weight)):
                x = F_.linear_log_softmax(x, self.wx[i].weight, self.wx[i].bias)  # perf-mode output layer: one node
                continue
            if (not self.dnn_use_laynorm[i] and self.dnn_act[i] != "softmax"
                    and F_.linear_bn_act_ok(x, self.wx[i].weight, self.training, bool(self.dnn_use_batchnorm[i]), self.dnn_act[i])):
                # perf mode, small batch: the whole layer - GEMM, batch statistics, BatchNorm, activation, drop mask, the
                # bf16 copy the next layer reads - is one launch (functional.LinearBnActFn)
                mask = None
                if self.training and self.dnn_drop[i] > 0.0:
                    mask = F_.dropout_mask(torch.empty(x.shape[0], self.dnn_lay[i], device=x.device), self.dnn_drop[i])
                x = F_.linear_bn_act(x, self.wx[i].weight, self.wx[i].bias, self.bn[i], self.dnn_act[i], mask, count=False)
                counted.append(self.bn[i].num_batches_tracked)
                continue
            z = F_.linear(x, self.wx[i].weight, self.wx[i].bias)
            if self.dnn_use_laynorm[i]:
                z = self.ln[i](z)
            x = _apply_act_drop(z, self.bn[i], bool(self.dnn_use_batchnorm[i]), self.training, self.dnn_act[i],
                                self.dnn_drop[i])
        if counted:
            with torch.no_grad():
                torch._foreach_add_(counted, 1)
        return x


# gate order of the concatenated projections (matches PK_CELL_* in include/pk_amd.h)
_REC_SPEC = {
    "liGRU": ("ligru", [("wz", "uz", "bn_wz"), ("wh", "uh", "bn_wh")], ["wh", "wz"], ["uh", "uz"], ["bn_wh", "bn_wz"]),
    "minimalGRU": ("minimalgru", [("wz", "uz", "bn_wz"), ("wh", "uh", "bn_wh")], ["wh", "wz"], ["uh", "uz"],
                   ["bn_wh", "bn_wz"]),
    "GRU": ("gru", [("wz", "uz", "bn_wz"), ("wr", "ur", "bn_wr"), ("wh", "uh", "bn_wh")], ["wh", "wz", "wr"],
            ["uh", "uz", "ur"], ["bn_wh", "bn_wz", "bn_wr"]),
    "LSTM": ("lstm", [("wfx", "ufh", "bn_wfx"), ("wix", "uih", "bn_wix"), ("wox", "uoh", "bn_wox"),
                      ("wcx", "uch", "bn_wcx")], ["wfx", "wix", "wox", "wcx"], ["ufh", "uih", "uoh", "uch"],
             ["bn_wfx", "bn_wix", "bn_wox", "bn_wcx"]),
    "RNN": ("rnn", [("wh", "uh", "bn_wh")], ["wh"], ["uh"], ["bn_wh"]),
}


class _MaskPrefetcher:
    """PK_MASK_RNG=reference without the host on the critical path.  The reference draws its drop masks with
    torch.bernoulli(torch.Tensor(rows, H).fill_(1 - p)) on the global CPU generator (e.g. neural_networks.py:1102-1107):
    a scalar loop that, called from a thread with an intra-op pool, also fights over the generator's lock (42 ms per
    256 x 550 mask on 8 threads, ~5 ms on one).  The masks do not depend on data - only on the ORDER in which the
    generator is consumed - so ONE long-lived, single-threaded helper draws the NEXT forward call's masks (same shapes
    as this call's, layer by layer, the very same calls) while the GPU works on this one; it starts as soon as this
    call has taken delivery of its own set.  The generator state from before every ahead-of-time set is kept: if the
    next call turns out to want other shapes (last batch of a chunk), or somebody else drew from / re-seeded the
    generator in the meantime (its state differs from the one the helper left), that state is restored and the masks
    are drawn on the spot - the stream stays exactly the reference's.  Nothing else in the engine's step touches the
    CPU generator (batch padding uses python's random, nn.Dropout masks the device generator).

    SEVERAL recurrent modules in one model (the reference ships such recipes, e.g.
    cfg/TIMIT_baselines/TIMIT_rev/TIMIT_joint_training_liGRU_fbank.cfg) consume the ONE generator in turn: module A's
    helper would be drawing A's next masks while module B draws its current ones.  Drawing ahead is therefore only done
    while a single prefetcher is alive; as soon as a second one asks for a mask every set drawn ahead is given back to
    the generator and all of them draw on the spot, in call order - the reference's stream, at the reference's speed."""

    live = weakref.WeakSet()  # every prefetcher that may hold ahead-of-time draws (drain_mask_prefetch); a module's
    _jobs = None              # prefetcher dies with the module.  _jobs: the helper's queue

    def __init__(self):
        _MaskPrefetcher.live.add(self)
        self._done = None     # event of the set being drawn
        self._ahead = None    # dict(sig, masks, state0, state1): the set drawn ahead (complete once _done is set)
        self._ready = None    # the set this forward call is being served from
        self._cur = []        # signature of the call in progress
        self._taken = 0

    @staticmethod
    def _draw(rows, H, p):
        return torch.bernoulli(torch.Tensor(rows, H).fill_(1 - p))  # the reference's own call

    @classmethod
    def _worker(cls):
        torch.set_num_threads(1)  # (a sequential stream; a fresh thread per call would also make OpenMP build a team each time)
        while True:
            fn, done = cls._jobs.get()
            try:
                fn()
            finally:
                done.set()

    def _start(self, sig):
        import queue
        import threading

        if _MaskPrefetcher._jobs is None:
            _MaskPrefetcher._jobs = queue.Queue()
            threading.Thread(target=_MaskPrefetcher._worker, daemon=True).start()
        box = {"sig": list(sig)}

        def run():
            try:
                box["state0"] = torch.get_rng_state()
                box["masks"] = [self._draw(*s_) for s_ in box["sig"]]
                box["state1"] = torch.get_rng_state()
            except BaseException as e:  # handed to the thread that joins (the helper itself lives on)
                box["exc"] = e

        self._ahead, self._done = box, threading.Event()
        _MaskPrefetcher._jobs.put((run, self._done))

    def _join(self):
        if self._done is not None:
            self._done.wait()
            self._done = None
            if self._ahead is not None and "exc" in self._ahead:
                exc, st0 = self._ahead["exc"], self._ahead.get("state0")
                self._ahead = None
                if st0 is not None:
                    torch.set_rng_state(st0)  # whatever the failed set consumed goes back
                raise exc

    @classmethod
    def _alone(cls, me):
        return all(pf is me for pf in cls.live)

    def _rewind(self, to_state, redraw):
        """Give the generator back every draw made ahead that nobody will use: back to `to_state`, then the masks of
        the current call that WERE used are drawn again (same values) so that the stream continues behind them."""
        self._join()
        torch.set_rng_state(to_state)
        for s_ in redraw:
            self._draw(*s_)
        self._ahead = self._ready = None

    def get(self, i, n_lay, rows, H, p):
        """Mask of layer i (0 .. n_lay - 1, asked for in layer order) of the current forward call."""
        if i == 0:
            self._cur, self._taken, self._ready = [], 0, None
            if not self._alone(self):  # another module shares the generator: nobody keeps draws made ahead
                for pf in list(_MaskPrefetcher.live):
                    if pf is not self:
                        pf.drain()
            self._join()
            if self._ahead is not None:
                if torch.equal(torch.get_rng_state(), self._ahead["state1"]):
                    self._ready, self._ahead = self._ahead, None
                    if self._alone(self):
                        self._start(self._ready["sig"])  # the call after this one, while this one runs
                else:
                    self._ahead = None  # the generator moved on without us (manual_seed, somebody else's draws): theirs now
        want = (rows, H, p)
        self._cur.append(want)
        rd = self._ready
        if rd is not None and self._taken == i and i < len(rd["sig"]) and rd["sig"][i] == want:
            self._taken = i + 1
            return rd["masks"][i]
        if rd is not None:  # other shapes than the ones drawn ahead: un-draw this set's unused masks and the next set
            self._rewind(rd["state0"], self._cur[:-1])
        m = self._draw(*want)
        if i == n_lay - 1 and self._done is None and self._ahead is None and self._alone(self):
            self._start(self._cur)  # first call, or the call after a mismatch: start drawing ahead again
        return m

    def drain(self):
        """Un-draw whatever was drawn ahead (before anything else reads or re-seeds the generator)."""
        self._join()
        if self._ahead is not None and torch.equal(torch.get_rng_state(), self._ahead["state1"]):
            torch.set_rng_state(self._ahead["state0"])
        self._ahead = self._ready = None


def drain_mask_prefetch():
    """Bring torch's CPU generator to where the reference's would be: un-draw every mask the host helper drew ahead of
    time (PK_MASK_RNG=reference_host) and forget the prefetchers; write the device mirror's state back
    (PK_MASK_RNG=reference).  Call before torch.manual_seed / before reading the CPU generator's state (core.run_nn_dp
    does, at both ends of a chunk)."""
    for pf in list(_MaskPrefetcher.live):
        pf.drain()
    _MaskPrefetcher.live = weakref.WeakSet()
    F_._RefRng.sync_back()


class _Recurrent(nn.Module):
    """Shared body of LSTM / GRU / liGRU / minimalGRU / RNN (neural_networks.py:300-655, 997-1461)."""

    KIND = None

    def __init__(self, options, inp_dim):
        super().__init__()
        pre, gates, w_order, u_order, bn_order = _REC_SPEC[self.KIND]
        self._pre, self._gates = pre, gates
        self.input_dim = inp_dim
        lay = _ints(_opt(options, pre + "_lay"))
        setattr(self, pre + "_lay", lay)
        self._lay = lay
        self._drop = _floats(_opt(options, pre + "_drop"))
        self._use_bn = _bools(_opt(options, pre + "_use_batchnorm"))
        self._use_ln = _bools(_opt(options, pre + "_use_laynorm"))
        self._use_ln_inp = strtobool(_opt(options, pre + "_use_laynorm_inp"))
        self._use_bn_inp = strtobool(_opt(options, pre + "_use_batchnorm_inp"))
        self._orthinit = strtobool(_opt(options, pre + "_orthinit"))
        self._act = str(_opt(options, pre + "_act")).split(",")
        self.bidir = strtobool(_opt(options, pre + "_bidir"))
        self.use_cuda = strtobool(_opt(options, "use_cuda"))
        self.to_do = _opt(options, "to_do")
        self.test_flag = self.to_do != "train"
        for a in self._act:
            if a not in F_.ACT:
                raise PkError("%s: activation %r is not supported inside the recurrence" % (self.KIND, a))
        # registration order = the reference's (w, u pairs gate by gate, then ln, bn, act): parameters() order is
        # what indexes torch optimizer state, so checkpoints' optimizer_par stay interchangeable
        for name in [n for pair in zip(w_order, u_order) for n in pair] + ["ln"] + bn_order + ["act"]:
            setattr(self, name, nn.ModuleList([]))
        if self._use_ln_inp:
            self.ln0 = LayerNorm(self.input_dim)
        if self._use_bn_inp:
            self.bn0 = nn.BatchNorm1d(self.input_dim, momentum=0.05)
        self._n_lay = len(lay)
        cur = self.input_dim
        for i in range(self._n_lay):
            self.act.append(act_fun(self._act[i]))
            add_bias = not (self._use_ln[i] or self._use_bn[i])
            for name in w_order:  # feed-forward connections, reference order
                getattr(self, name).append(nn.Linear(cur, lay[i], bias=add_bias))
            for name in u_order:  # recurrent connections
                getattr(self, name).append(nn.Linear(lay[i], lay[i], bias=False))
            if self._orthinit:
                for name in u_order:
                    nn.init.orthogonal_(getattr(self, name)[i].weight)
            for name in bn_order:
                getattr(self, name).append(nn.BatchNorm1d(lay[i], momentum=0.05))
            self.ln.append(LayerNorm(lay[i]))
            cur = 2 * lay[i] if self.bidir else lay[i]
        self.out_dim = lay[-1] + self.bidir * lay[-1]

    def pk_unused_parameters(self):
        """ln / bn_* of a layer whose flag is off (18 of 62 tensors in the shipped Li-GRU recipe)."""
        out = []
        for i in range(self._n_lay):
            if not self._use_ln[i]:
                out += list(self.ln[i].parameters())
            if not self._use_bn[i]:
                for (_, _, b) in self._gates:
                    out += list(getattr(self, b)[i].parameters())
        return out

    def _drop_mask(self, i, batch, device):
        """Bernoulli(1-p) mask (rows, H) of layer i, unscaled and constant over time (:1102-1107), or
        (None, 1-p) in test mode, in layer order like the reference draws them.
        settings.mask_rng = "device" (default): the GPU RNG; "reference": the reference's stream - what its
        own call on the CPU generator would give for the same seed - from the device mirror of that generator;
        "reference_host": the reference's own call on the host, a forward call ahead, then copied over."""
        p = self._drop[i]
        if self.test_flag:
            return None, 1.0 - p
        rows = 2 * batch if self.bidir else batch
        if F_.settings.mask_rng == "device":
            # layers of one width and one dropout rate (every shipped recipe): the masks of all layers are ONE draw at
            # layer 0 (a launch per layer and step otherwise); 16-byte aligned slices
            H = self._lay[i]
            if (self._n_lay > 1 and len(set(self._lay)) == 1 and len(set(self._drop)) == 1 and (rows * H) % 4 == 0
                    and F_._lib.experiment("mask_one_draw", "1") != "0"):
                if i == 0:
                    self._mask_all = torch.empty(self._n_lay, rows, H, device=device).bernoulli_(1 - p)
                m_all = getattr(self, "_mask_all", None)
                if m_all is not None and m_all.shape[1] == rows and m_all.device == device:
                    if i == self._n_lay - 1:
                        self._mask_all = None
                    return m_all[i], 1.0
            return torch.empty(rows, H, device=device).bernoulli_(1 - p), 1.0
        if F_.settings.mask_rng == "reference":
            # the reference's stream from the device mirror of its generator: bit-identical masks, no host work.  The
            # masks of the whole stack are drawn at layer 0, in line
            if i == 0:
                self._ref_masks = [(rows, device)] + F_._RefRng.masks(
                    [(rows, self._lay[j], self._drop[j]) for j in range(self._n_lay)], device)
            pend = getattr(self, "_ref_masks", None)
            if pend is None or pend[0] != (rows, device) or pend[1 + i] is None:
                # (a layer asked on its own: drawn on the spot - the stream still follows the order of the CALLS)
                return F_.ref_rng_mask(rows, self._lay[i], p, device), 1.0
            entry, pend[1 + i] = pend[1 + i], None
            if i == self._n_lay - 1:
                self._ref_masks = None
            return entry, 1.0
        if getattr(self, "_prefetch", None) is None or self._prefetch not in _MaskPrefetcher.live:
            self._prefetch = _MaskPrefetcher()
        m = self._prefetch.get(i, self._n_lay, rows, self._lay[i], p)  # the reference's own call, a forward call ahead
        return self._to_device_async(i, m, device), 1.0

    def _to_device_async(self, i, m, device):
        """H2D copy of a freshly drawn mask that does not drain the GPU queue: a pageable .to(device) waits for
        everything already enqueued.  Two pinned staging buffers per layer, each guarded by the event of its
        last copy (a buffer is reused two forward calls later)."""
        if device.type != "cuda":
            return m.to(device)
        if not hasattr(self, "_pin"):
            self._pin, self._pin_ev, self._pin_k = {}, {}, {}
        k = self._pin_k.get(i, 0)
        key = (i, k, tuple(m.shape))
        if key not in self._pin:
            self._pin[key] = torch.empty(m.shape, dtype=m.dtype).pin_memory()
            self._pin_ev[key] = None
        if self._pin_ev[key] is not None:
            self._pin_ev[key].synchronize()
        self._pin[key].copy_(m)
        d = self._pin[key].to(device, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._pin_ev[key] = ev
        self._pin_k[i] = 1 - k
        return d

    def pk_flat_groups(self):
        """optim.FlatParams layout: layer by layer, and inside a layer the gates of the input weights (of the recurrent
        weights, of the BatchNorm scales / shifts, of the biases) back to back in the order forward concatenates them -
        the concatenation is then a view of the flat buffer (functional.adjacent_view), and the weight-gradient GEMM
        accumulates into the matching view of the flat gradient."""
        out = []
        if self._use_ln_inp:
            out.append(list(self.ln0.parameters()))
        if self._use_bn_inp:
            out.append(list(self.bn0.parameters()))
        for i in range(self._n_lay):
            Ws = [getattr(self, w)[i] for (w, _, _) in self._gates]
            Us = [getattr(self, u)[i] for (_, u, _) in self._gates]
            bns = [getattr(self, b)[i] for (_, _, b) in self._gates]
            out.append([m.weight for m in Ws])
            out.append([m.weight for m in Us])
            out.append([m.bias for m in Ws if m.bias is not None])
            out.append([b.weight for b in bns])
            out.append([b.bias for b in bns])
            out.append(list(self.ln[i].parameters()))
        return out

    def forward(self, x, drop_masks=None):
        if not x.is_cuda:
            raise PkError("pytorch-kaldi_amd.nn.%s runs on the GPU only: set use_cuda=True" % self.KIND)
        if self._use_ln_inp:
            x = self.ln0(x)
        if self._use_bn_inp:
            T, B, D = x.shape
            x = F_.norm_act_drop(x.reshape(T * B, D), self.bn0, True, self.training, "linear").view(T, B, D)
        masks = scalars = None
        if drop_masks is not None:  # injected (parity tests): tensors (rows,H) in train mode, 1-element tensors in test mode
            masks = [m if m.numel() > 1 else None for m in drop_masks]
            scalars = [1.0 if m.numel() > 1 else float(m) for m in drop_masks]
            masks = [m.to(x.device).float().contiguous() if m is not None else None for m in masks]
        batch = x.shape[1]
        xb = xseg = None  # perf mode: bf16 copy of the running activation, handed from layer to layer
        for i in range(self._n_lay):
            mask_i, scalar_i = (masks[i], scalars[i]) if masks is not None else self._drop_mask(i, batch, x.device)
            H = self._lay[i]
            Ws = [getattr(self, w)[i] for (w, _, _) in self._gates]
            Us = [getattr(self, u)[i] for (_, u, _) in self._gates]
            Wcat = Ucat = None  # (concatenated below; on the perf path possibly as views of the flat parameter buffer)
            bcat = None
            if Ws[0].bias is not None:
                bcat = torch.cat([m.bias for m in Ws], 0) if len(Ws) > 1 else Ws[0].bias
            gamma = beta = rmean = rvar = None
            use_bn = bool(self._use_bn[i])
            bns = [getattr(self, b)[i] for (_, _, b) in self._gates]
            affine = edge_t = None
            if use_bn:
                gps, bps = [b.weight for b in bns], [b.bias for b in bns]
                if (self.training and F_.direct_affine_ok(gps + bps)
                        and F_.perf_path_ok(self.KIND, H, bool(self._use_ln[i]), use_bn, self.training)):
                    # the engine's own training step: scales / shifts as views of the flat buffer (no concatenation
                    # launches), their gradients added to the flat .grad by the BatchNorm backward itself
                    gamma, beta = F_.adjacent_view([q.detach() for q in gps]), F_.adjacent_view([q.detach() for q in bps])
                    if (gamma is not None and beta is not None and F_.adjacent_view([q.grad for q in gps]) is not None
                            and F_.adjacent_view([q.grad for q in bps]) is not None):
                        affine, edge_t = (gps, bps), gps[0]  # (the first scale carries the autograd edge, gradient None)
                if affine is None:
                    gamma = torch.cat(gps, 0)
                    beta = torch.cat(bps, 0)
                if not self.training:
                    rmean = torch.cat([b.running_mean for b in bns], 0)
                    rvar = torch.cat([b.running_var for b in bns], 0)
            if use_bn and self.training and x.shape[0] * x.shape[1] * (2 if self.bidir else 1) <= 1:
                # what torch's BatchNorm1d raises for the reference in the same situation (one row, training mode)
                raise ValueError("Expected more than 1 value per channel when training, got input size %s"
                                 % (tuple(x.shape),))
            cfg = (self.KIND, self._act[i], H, bool(self.bidir), use_bn, self.training, 1e-5, 0.05, scalar_i)
            if F_.perf_path_ok(self.KIND, H, bool(self._use_ln[i]), use_bn, self.training):
                wps, ups = [m.weight for m in Ws], [m.weight for m in Us]
                # weight gradients that go to the flat .grad buffer from the side stream do not pass through autograd:
                # the concatenated weights are handed over detached (otherwise cat's backward materialises a zero
                # gradient per gate and adds it to .grad on the main stream - launches, and a write that would race
                # a data-parallel bucket already being reduced)
                # (some other input must still carry the autograd edge that makes backward run at all: the BatchNorm
                # affine or the bias - one of the two always exists, neural_networks.py:1052-1055 - or x itself)
                edge = torch.is_grad_enabled() and (use_bn or bcat is not None or x.requires_grad)
                side_w = edge and F_.side_targets_ok(wps)
                side_u = edge and F_.side_targets_ok(ups)
                # detached weights: the concatenation is a view when optim.FlatParams packed the gates back to back
                if side_w or not torch.is_grad_enabled():
                    Wcat = F_.adjacent_view([w.detach() for w in wps])
                if side_u or not torch.is_grad_enabled():
                    Ucat = F_.adjacent_view([u.detach() for u in ups])
                if Wcat is None:
                    Wcat = torch.cat(wps, 0) if len(wps) > 1 else wps[0]
                if Ucat is None:
                    Ucat = torch.cat(ups, 0) if len(ups) > 1 else ups[0]
                # training-mode BatchNorm: the running statistics of every gate's module are updated by the launch that
                # turns the batch statistics into scale / shift (pk_bn_finalize_gates)
                # forward-only chunks: an inner layer's fp32 output is read by nobody (the next layer takes the bf16 copy)
                keep_y = torch.is_grad_enabled() or i == self._n_lay - 1
                stats_in_kernel = use_bn and self.training
                bn_bufs = ([b.running_mean for b in bns], [b.running_var for b in bns],
                           [b.num_batches_tracked for b in bns]) if stats_in_kernel else None
                y, bmean, bvar, xb = F_.RecLayerPerfFn.apply(x, xb, Wcat.detach() if side_w else Wcat, bcat,
                                                           Ucat.detach() if side_u else Ucat, gamma, beta, rmean, rvar,
                                                           mask_i, cfg + (xseg, wps, ups, side_w, side_u, bn_bufs, affine,
                                                                          keep_y), edge_t)
                xseg = (2 if self.bidir else 1, H, (H + 7) // 8 * 8)
                if stats_in_kernel:
                    x = y
                    continue
            else:
                wps, ups = [m.weight for m in Ws], [m.weight for m in Us]
                # exact-fp32 mode with flat-bucket parameters (round 6): the weight-gradient GEMMs go to the side stream and add
                # into the flat .grad, so the concatenated weights are handed over detached, as on the perf path above (some
                # other input carries the autograd edge: the BatchNorm affine, the bias or x itself)
                edge = (not F_.bf16_mode()) and torch.is_grad_enabled() and (use_bn or bcat is not None or x.requires_grad)
                side_w = edge and F_.side_targets_any_ok(wps)
                side_u = edge and F_.side_targets_any_ok(ups)
                if side_w:
                    Wcat = F_.adjacent_view([w.detach() for w in wps])
                    if Wcat is None:
                        Wcat = torch.cat([w.detach() for w in wps], 0)
                else:
                    Wcat = torch.cat(wps, 0) if len(wps) > 1 else wps[0]
                if side_u:
                    Ucat = F_.adjacent_view([u.detach() for u in ups])
                    if Ucat is None:
                        Ucat = torch.cat([u.detach() for u in ups], 0)
                else:
                    Ucat = torch.cat(ups, 0) if len(ups) > 1 else ups[0]
                lng = self.ln[i].gamma if self._use_ln[i] else None
                lnb = self.ln[i].beta if self._use_ln[i] else None
                y, bmean, bvar = F_.RecLayerFn.apply(x, Wcat, bcat, Ucat, gamma, beta, rmean, rvar, mask_i, lng, lnb,
                                                     cfg + (wps, ups, side_w, side_u))
                xb = xseg = None
            if use_bn and self.training:
                n = x.shape[0] * x.shape[1] * (2 if self.bidir else 1)
                with torch.no_grad():  # BatchNorm1d(momentum=0.05) running statistics, per gate: four multi-tensor launches
                    means, vars_ = [b.running_mean for b in bns], [b.running_var for b in bns]
                    torch._foreach_mul_(means + vars_, 0.95)
                    torch._foreach_add_(means, [bmean[k * H:(k + 1) * H] for k in range(len(bns))], alpha=0.05)
                    torch._foreach_add_(vars_, [bvar[k * H:(k + 1) * H] for k in range(len(bns))], alpha=0.05 * n / (n - 1))
                    torch._foreach_add_([b.num_batches_tracked for b in bns], 1)
            x = y
        if xb is not None:  # the bf16 copy the last layer published: a perf-mode Linear behind it uses it as its operand
            x._pk_twin = (xb, xseg, x._version)
        return x


class liGRU(_Recurrent):
    """neural_networks.py:997-1155."""
    KIND = "liGRU"


class minimalGRU(_Recurrent):
    """neural_networks.py:1158-1316."""
    KIND = "minimalGRU"


class GRU(_Recurrent):
    """neural_networks.py:486-655."""
    KIND = "GRU"


class LSTM(_Recurrent):
    """neural_networks.py:300-483."""
    KIND = "LSTM"


class RNN(_Recurrent):
    """neural_networks.py:1319-1461."""
    KIND = "RNN"


class SincConv(nn.Module):
    """Sinc band-pass bank (neural_networks.py:1668-1813).  The 2x128 parameters ->
    128x129 filter synthesis stays in torch autograd (tiny); the
    convolution itself runs in the fused HIP conv+pool kernel of the caller."""

    @staticmethod
    def to_mel(hz):
        return 2595 * np.log10(1 + hz / 700)

    @staticmethod
    def to_hz(mel):
        return 700 * (10 ** (mel / 2595) - 1)

    def __init__(self, in_channels, out_channels, kernel_size, sample_rate=16000, min_low_hz=50, min_band_hz=50):
        super().__init__()
        if in_channels != 1:
            raise ValueError("SincConv only support one input channel (here, in_channels = {%i})" % (in_channels))
        self.out_channels = out_channels
        self.kernel_size = kernel_size + 1 if kernel_size % 2 == 0 else kernel_size
        self.sample_rate, self.min_low_hz, self.min_band_hz = sample_rate, min_low_hz, min_band_hz
        low_hz = 30
        high_hz = self.sample_rate / 2 - (self.min_low_hz + self.min_band_hz)
        mel = np.linspace(self.to_mel(low_hz), self.to_mel(high_hz), self.out_channels + 1)
        hz = self.to_hz(mel) / self.sample_rate
        self.low_hz_ = nn.Parameter(torch.Tensor(hz[:-1]).view(-1, 1))
        self.band_hz_ = nn.Parameter(torch.Tensor(np.diff(hz)).view(-1, 1))
        n_lin = torch.linspace(0, self.kernel_size, steps=self.kernel_size)
        self.window_ = 0.54 - 0.46 * torch.cos(2 * math.pi * n_lin / self.kernel_size)
        n = (self.kernel_size - 1) / 2
        self.n_ = torch.arange(-n, n + 1).view(1, -1) / self.sample_rate

    def _sinc(self, v):
        left = v[:, 0:int((v.shape[1] - 1) / 2)]
        yl = torch.sin(left) / left
        return torch.cat([yl, torch.ones([v.shape[0], 1], device=v.device), torch.flip(yl, dims=[1])], dim=1)

    def filters(self):
        dev = self.low_hz_.device
        self.n_ = self.n_.to(dev)
        self.window_ = self.window_.to(dev)
        if self.low_hz_.is_cuda:  # the synthesis and its gradient as one launch each (pk_sinc.hip)
            return F_.sinc_bank(self.low_hz_, self.band_hz_, self.n_, self.window_, self.sample_rate,
                                self.min_low_hz / self.sample_rate, self.min_band_hz / self.sample_rate)
        # (CPU: only the parameter-free inspection of a module that has not been moved to the GPU; forward raises there)
        low = self.min_low_hz / self.sample_rate + torch.abs(self.low_hz_)
        high = low + self.min_band_hz / self.sample_rate + torch.abs(self.band_hz_)
        lp1 = 2 * low * self._sinc(2 * math.pi * (low * self.n_) * self.sample_rate)  # (N,1) x (1,K) outer product: a broadcast, no BLAS launch
        lp2 = 2 * high * self._sinc(2 * math.pi * (high * self.n_) * self.sample_rate)
        bp = lp2 - lp1
        mx, _ = torch.max(bp, dim=1, keepdim=True)
        bp = bp / mx
        return (bp * self.window_).view(self.out_channels, 1, self.kernel_size)


class _ConvStack(nn.Module):
    """Shared body of CNN (:1464-1556) and SincNet (:1559-1665)."""

    PRE = None

    def __init__(self, options, inp_dim):
        super().__init__()
        p = self.PRE
        self.input_dim = inp_dim
        self._n_filt = _ints(_opt(options, p + "_N_filt"))
        self._len_filt = _ints(_opt(options, p + "_len_filt"))
        self._pool = _ints(_opt(options, p + "_max_pool_len"))
        self._act = str(_opt(options, p + "_act")).split(",")
        self._drop = _floats(_opt(options, p + "_drop"))
        self._use_ln = _bools(_opt(options, p + "_use_laynorm"))
        self._use_bn = _bools(_opt(options, p + "_use_batchnorm"))
        self._use_ln_inp = strtobool(_opt(options, p + "_use_laynorm_inp"))
        self._use_bn_inp = strtobool(_opt(options, p + "_use_batchnorm_inp"))
        self._n_lay = len(self._n_filt)
        if p == "sinc":
            self.sinc_sample_rate = int(_opt(options, "sinc_sample_rate"))
            self.sinc_min_low_hz = int(_opt(options, "sinc_min_low_hz"))
            self.sinc_min_band_hz = int(_opt(options, "sinc_min_band_hz"))
        self.conv = nn.ModuleList([])
        self.bn = nn.ModuleList([])
        self.ln = nn.ModuleList([])
        self.act = nn.ModuleList([])
        self.drop = nn.ModuleList([])
        if self._use_ln_inp:
            self.ln0 = LayerNorm(self.input_dim)
        if self._use_bn_inp:
            self.bn0 = nn.BatchNorm1d([self.input_dim], momentum=0.05)
        cur = self.input_dim
        for i in range(self._n_lay):
            n_filt, len_filt = self._n_filt[i], self._len_filt[i]
            self.drop.append(nn.Dropout(p=self._drop[i]))
            self.act.append(act_fun(self._act[i]))
            pooled = int((cur - len_filt + 1) / self._pool[i])
            self.ln.append(LayerNorm([n_filt, pooled]))
            # the reference passes the pooled length positionally, i.e. as eps (:1515-1517 / :1615-1617)
            self.bn.append(nn.BatchNorm1d(n_filt, pooled, momentum=0.05))
            if i == 0:
                if p == "sinc":
                    self.conv.append(SincConv(1, n_filt, len_filt, sample_rate=self.sinc_sample_rate,
                                              min_low_hz=self.sinc_min_low_hz, min_band_hz=self.sinc_min_band_hz))
                else:
                    self.conv.append(nn.Conv1d(1, n_filt, len_filt))
            else:
                self.conv.append(nn.Conv1d(self._n_filt[i - 1], n_filt, len_filt))
            cur = pooled
        self.out_dim = cur * self._n_filt[-1]

    def pk_unused_parameters(self):
        out = []
        for i in range(self._n_lay):
            if not self._use_ln[i]:
                out += list(self.ln[i].parameters())
            if not self._use_bn[i]:
                out += list(self.bn[i].parameters())
        return out

    def _conv_pool(self, i, x):
        c = self.conv[i]
        if isinstance(c, SincConv):
            return F_.conv1d_pool(x, c.filters(), None, self._pool[i])
        return F_.conv1d_pool(x, c.weight, c.bias, self._pool[i])

    def _post(self, i, z, use_bn):
        B, C, L = z.shape
        mask = None
        if self.training and self._drop[i] > 0.0:
            mask = F_.dropout_mask(z, self._drop[i])
        act = self._act[i]
        if act == "softmax":
            raise PkError("softmax activation inside a conv stack is not supported")
        if use_bn:
            # BatchNorm1d over (B, L) per channel: put channels last for the column kernels
            zt = z.permute(0, 2, 1).reshape(B * L, C)
            mt = None if mask is None else mask.permute(0, 2, 1).reshape(B * L, C)
            y = F_.norm_act_drop(zt, self.bn[i], True, self.training, act, mt)
            return y.view(B, L, C).permute(0, 2, 1).contiguous()
        m2 = None if mask is None else mask.reshape(B * C, L)
        return F_.norm_act_drop(z.reshape(B * C, L), None, False, self.training, act, m2).view(B, C, L)

    def forward(self, x):
        if not x.is_cuda:
            raise PkError("pytorch-kaldi_amd.nn conv stacks run on the GPU only: set use_cuda=True")
        batch, seq_len = x.shape[0], x.shape[1]
        if self._use_ln_inp:
            x = self.ln0(x)
        if self._use_bn_inp:
            x = F_.norm_act_drop(x, self.bn0, True, self.training, "linear")
        x = x.reshape(batch, 1, seq_len)
        for i in range(self._n_lay):
            x_in = x
            if self._use_ln[i]:
                ln = self.ln[i]
                z = self._conv_pool(i, x_in)
                if ln.gamma.dim() == 2 and self._act[i] != "softmax" and tuple(ln.gamma.shape) == tuple(z.shape[1:]):
                    # the layer's tail - LayerNorm, activation, dropout - as one launch each way
                    mask = F_.dropout_mask(z, self._drop[i]) if (self.training and self._drop[i] > 0.0) else None
                    x = F_.ln_last_act_drop(z, ln.gamma, ln.beta, ln.eps, self._act[i], mask)
                else:
                    x = self._post(i, ln(z), False)
            if self._use_bn[i]:
                x = self._post(i, self._conv_pool(i, x), True)
            if not self._use_bn[i] and not self._use_ln[i]:
                x = self._post(i, self._conv_pool(i, x_in), False)
        return x.reshape(batch, -1)


class CNN(_ConvStack):
    """neural_networks.py:1464-1556."""
    PRE = "cnn"


class SincNet(_ConvStack):
    """neural_networks.py:1559-1665."""
    PRE = "sinc"
