"""CPU oracle for the PyTorch-Kaldi `neural_networks.py` hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pytorch-kaldi_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg
of ``bench.py`` use it, and there only as the checker / the reported CPU baseline.

It is a plain torch-CPU restatement (fp32 or fp64, autograd for gradients) of the
reference algorithm, written from the reference's behaviour and citing the lines
it follows (paths relative to the reference checkout):

    LayerNorm                neural_networks.py:23-33
    act_fun                  neural_networks.py:36-57
    MLP.forward              neural_networks.py:126-150
    LSTM.forward             neural_networks.py:402-483
    GRU.forward              neural_networks.py:579-655
    liGRU.forward            neural_networks.py:1082-1155
    minimalGRU.forward       neural_networks.py:1243-1316
    RNN.forward              neural_networks.py:1396-1461
    CNN.forward              neural_networks.py:1530-1556
    SincNet.forward          neural_networks.py:1639-1665
    SincConv.forward/sinc    neural_networks.py:1763-1813
    flip                     neural_networks.py:1962-1970
    forward_model (glue)     utils.py:2296-2420

Parity pin: the reference ships no tests or golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself, generated in the build container by ``oracle/make_golden.py`` (which
imports ``/root/reference/neural_networks.py`` unmodified) and committed under
``tests/golden/``.  ``tests/test_oracle_golden.py`` checks every fixture.

Parameters are passed as a ``dict`` that uses the reference's ``state_dict`` key
names (``wh.0.weight``, ``bn_wz.3.running_var``, ``ln.0.gamma`` ...), so a
reference checkpoint can be fed to the oracle unchanged.
"""

import math

import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# option parsing (the reference reads everything from strings)
# ----------------------------------------------------------------------------
def _tobool(s):
    s = str(s).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return 1
    if s in ("n", "no", "f", "false", "off", "0"):
        return 0
    raise ValueError("invalid truth value %r" % (s,))


def _ints(s):
    return [int(v) for v in str(s).split(",")]


def _floats(s):
    return [float(v) for v in str(s).split(",")]


def _bools(s):
    return [_tobool(v) for v in str(s).split(",")]


def _strs(s):
    return str(s).split(",")


def _opt(options, key):
    # configparser lower-cases option names; accept either spelling
    if key in options:
        return options[key]
    return options[key.lower()]


# ----------------------------------------------------------------------------
# primitives
# ----------------------------------------------------------------------------
def layer_norm(x, gamma, beta, eps=1e-6):
    """neural_networks.py:30-33 - unbiased std, eps added to the std."""
    mean = x.mean(-1, keepdim=True)
    std = x.std(-1, keepdim=True)
    return gamma * (x - mean) / (std + eps) + beta


def activation(name, x):
    """neural_networks.py:36-57.  'softmax' is LogSoftmax(dim=1); 'linear' is
    LeakyReLU(1) i.e. identity."""
    if name == "relu":
        return torch.relu(x)
    if name == "tanh":
        return torch.tanh(x)
    if name == "sigmoid":
        return torch.sigmoid(x)
    if name == "leaky_relu":
        return F.leaky_relu(x, 0.2)
    if name == "elu":
        return F.elu(x)
    if name == "softmax":
        return F.log_softmax(x, dim=1)
    if name == "linear":
        return F.leaky_relu(x, 1.0)
    raise ValueError("unknown activation " + name)


def batch_norm(x, sd, prefix, training, momentum=0.05, eps=1e-5):
    """nn.BatchNorm1d(momentum=0.05) semantics on (N, C) or (N, C, L) input;
    running buffers in ``sd`` are updated in place in training mode."""
    w = sd[prefix + ".weight"]
    b = sd[prefix + ".bias"]
    rm = sd.get(prefix + ".running_mean")
    rv = sd.get(prefix + ".running_var")
    out = F.batch_norm(x, rm, rv, w, b, training, momentum, eps)
    key = prefix + ".num_batches_tracked"
    if training and key in sd:
        sd[key] += 1
    return out


def flip_time(x):
    """neural_networks.py:1962-1970 with dim=0."""
    return torch.flip(x, dims=[0])


# ----------------------------------------------------------------------------
# bf16-operand model of the engine's perf mode (test infrastructure, like the rest of this file).
# The HIP engine's "bf16" mode computes the SAME algorithm with the operands of every GEMM - projections, the
# recurrent h_{t-1}.U^T, Linear layers, and in backward dY.W, dY^T.x, dgates.U, dgates^T.h - rounded to bf16
# (round-to-nearest-even, v_cvt_pk_bf16_f32) and the products accumulated in fp32; everything element-wise (gates,
# BatchNorm, LayerNorm, softmax, state) stays fp32; with `bf16_operands(conv=True)` the convolutions of the SincNet / CNN
# stacks too (x, w and the un-pooled output gradient as bf16: the engine's opt-in pk_conv_bf16.hip).
# `with bf16_operands():` makes every matrix product of this oracle
# do exactly that, so a test can separate (i) "the engine implements the bf16-operand algorithm" (engine vs this model:
# tight) from (ii) "how far the bf16-operand algorithm is from the reference's fp32 results on this network" (this
# model vs the golden arrays: intrinsic, network-dependent).
# ----------------------------------------------------------------------------
_EMUL = {"bf16": False, "conv": False}


class bf16_operands:
    """conv=True: the convolutions of layers with at least 8 input channels take bf16 operands too (the engine with
    PK_CONV_BF16=1); conv="all": every convolution (PK_CONV_BF16=2)."""

    def __init__(self, conv=False):
        self.conv = conv

    def __enter__(self):
        self.prev = (_EMUL["bf16"], _EMUL["conv"])
        _EMUL["bf16"], _EMUL["conv"] = True, self.conv
        return self

    def __exit__(self, *exc):
        _EMUL["bf16"], _EMUL["conv"] = self.prev


def _rb(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _Bf16Linear(torch.autograd.Function):
    """y = round(x) . round(w)^T in fp32; backward dx = round(g) . round(w), dw = round(g)^T . round(x)."""

    @staticmethod
    def forward(ctx, x, w):
        xb, wb = _rb(x), _rb(w)
        ctx.save_for_backward(xb, wb)
        return F.linear(xb, wb)

    @staticmethod
    def backward(ctx, g):
        xb, wb = ctx.saved_tensors
        gb = _rb(g)
        dx = gb.matmul(wb)
        dw = gb.reshape(-1, gb.shape[-1]).t().matmul(xb.reshape(-1, xb.shape[-1]))
        return dx, dw


class _Bf16Conv(torch.autograd.Function):
    """conv1d(round(x), round(w)) in fp32; backward dx = conv_transpose(round(g), round(w)), dw = corr(round(g), round(x))
    - the engine's perf-mode convolutions (pk_conv_bf16.hip): x, w and the un-pooled output gradient enter as bf16."""

    @staticmethod
    def forward(ctx, x, w):
        xb, wb = _rb(x), _rb(w)
        ctx.save_for_backward(xb, wb)
        return F.conv1d(xb, wb)

    @staticmethod
    def backward(ctx, g):
        xb, wb = ctx.saved_tensors
        gb = _rb(g)
        dx = torch.nn.grad.conv1d_input(xb.shape, wb, gb) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv1d_weight(xb, wb.shape, gb) if ctx.needs_input_grad[1] else None
        return dx, dw


def _conv1d(x, w, b=None):
    """F.conv1d - the one place a convolution operand enters (bf16-operand model: see _Bf16Conv)."""
    if _EMUL["bf16"] and _EMUL["conv"] and (x.shape[1] >= 8 or _EMUL["conv"] == "all"):  # (the engine's rule: layers with >= 8 input channels)
        y = _Bf16Conv.apply(x, w)
        return y if b is None else y + b.view(1, -1, 1)
    return F.conv1d(x, w, b)


def _mm(x, w):
    """x . w^T - the one place a GEMM operand enters."""
    if _EMUL["bf16"]:
        return _Bf16Linear.apply(x, w)
    return F.linear(x, w)


def _linear(x, sd, prefix):
    b = sd.get(prefix + ".bias")
    if _EMUL["bf16"]:
        y = _mm(x, sd[prefix + ".weight"])
        return y if b is None else y + b
    return F.linear(x, sd[prefix + ".weight"], b)


def _dropout(x, p, training, mask):
    if mask is not None:
        return x * mask
    if training and p > 0.0:
        return F.dropout(x, p, True)
    return x


# ----------------------------------------------------------------------------
# MLP
# ----------------------------------------------------------------------------
def _act_maybe_forced(name, z, pattern):
    """activation(name, z), differentiated on another run's ReLU pattern when one is given (see _act_kink_forced)."""
    if pattern is not None and name == "relu":
        return _act_kink_forced(name, z, pattern.reshape(z.shape))
    return activation(name, z)


def mlp_forward(options, sd, x, training=True, drop_masks=None, kinks=None):
    """neural_networks.py:126-150.  ``drop_masks[i]`` (already scaled by
    1/(1-p)) replaces nn.Dropout of layer i when given; ``kinks[i]``: the ReLU pattern
    (z > 0) of another run of layer i (kink-forced derivative, test mode)."""
    lay = _ints(_opt(options, "dnn_lay"))
    drop = _floats(_opt(options, "dnn_drop"))
    use_bn = _bools(_opt(options, "dnn_use_batchnorm"))
    use_ln = _bools(_opt(options, "dnn_use_laynorm"))
    acts = _strs(_opt(options, "dnn_act"))
    if _tobool(_opt(options, "dnn_use_laynorm_inp")):
        x = layer_norm(x, sd["ln0.gamma"], sd["ln0.beta"])
    if _tobool(_opt(options, "dnn_use_batchnorm_inp")):
        x = batch_norm(x, sd, "bn0", training)
    for i in range(len(lay)):
        z = _linear(x, sd, "wx.%d" % i)
        if use_ln[i]:
            z = layer_norm(z, sd["ln.%d.gamma" % i], sd["ln.%d.beta" % i])
        if use_bn[i]:
            z = batch_norm(z, sd, "bn.%d" % i, training)
        z = _act_maybe_forced(acts[i], z, None if kinks is None else kinks[i])
        m = None if drop_masks is None else drop_masks[i]
        x = _dropout(z, drop[i], training, m)
    return x


# ----------------------------------------------------------------------------
# recurrent family
# ----------------------------------------------------------------------------
_REC = {
    # kind: (option prefix, [(input-proj name, recurrent name, bn name)...])
    "liGRU": ("ligru", [("wz", "uz", "bn_wz"), ("wh", "uh", "bn_wh")]),
    "minimalGRU": ("minimalgru", [("wz", "uz", "bn_wz"), ("wh", "uh", "bn_wh")]),
    "GRU": ("gru", [("wz", "uz", "bn_wz"), ("wr", "ur", "bn_wr"), ("wh", "uh", "bn_wh")]),
    "LSTM": ("lstm", [("wfx", "ufh", "bn_wfx"), ("wix", "uih", "bn_wix"),
                      ("wox", "uoh", "bn_wox"), ("wcx", "uch", "bn_wcx")]),
    "RNN": ("rnn", [("wh", "uh", "bn_wh")]),
}


def make_drop_masks(kind, options, batch, to_do="train", generator=None):
    """The reference samples one Bernoulli(1-p) mask of shape (rows, H) per layer
    per forward call on the CPU RNG, unscaled, constant over time; in test mode
    it uses the scalar (1-p) (e.g. neural_networks.py:1102-1107)."""
    pre = _REC[kind][0]
    lay = _ints(_opt(options, pre + "_lay"))
    drop = _floats(_opt(options, pre + "_drop"))
    bidir = _tobool(_opt(options, pre + "_bidir"))
    rows = 2 * batch if bidir else batch
    masks = []
    for i, h in enumerate(lay):
        if to_do == "train":
            masks.append(torch.bernoulli(torch.Tensor(rows, h).fill_(1 - drop[i]), generator=generator))
        else:
            masks.append(torch.FloatTensor([1 - drop[i]]))
    return masks


def _act_kink_forced(name, a, pattern):
    """Value act(a); derivative taken on the linear piece `pattern` (a_t > 0 as ANOTHER run saw it) instead of this
    run's own sign of a (two runs that differ by rounding noise then differentiate the same
    piecewise-linear function).  ReLU only - the smooth activations have no kinks."""
    if name != "relu":
        raise ValueError("kink-forced derivative is defined for relu")
    lin = a * pattern.to(a.dtype)
    return lin + (torch.relu(a) - lin).detach()


def recurrent_forward(kind, options, sd, x, training=True, to_do="train", drop_masks=None,
                      index_like_reference=False, return_all=False, kinks=None, kink_log=None):
    """Forward of LSTM/GRU/liGRU/minimalGRU/RNN exactly as the reference orders
    it: pack bidirectional on the batch axis, per-gate Linear over all steps,
    per-gate BatchNorm over the T*rows rows, python time loop from h=0, stack,
    unpack (neural_networks.py:402-483, 579-655, 1082-1155, 1243-1316, 1396-1461).

    ``kinks``: optional list (one per layer) of (T, rows, H) bool tensors = the pattern (a_t > 0) of another run; the
    candidate's ReLU then takes its derivative from that pattern (test mode for long sequences, see
    _act_kink_forced).

    ``kink_log``: optional list; receives one (T, rows, H) bool tensor per layer, this run's own pattern (a_t > 0) -
    what a second implementation is given as ``kinks`` to differentiate the same linear pieces.

    ``index_like_reference=True`` indexes the projections with ``w_out[k]`` inside
    the loop as the reference does (this is what makes its backward O(T^2));
    the default unbinds once, which is arithmetically identical and faster.
    """
    pre, gates = _REC[kind]
    lay = _ints(_opt(options, pre + "_lay"))
    use_bn = _bools(_opt(options, pre + "_use_batchnorm"))
    use_ln = _bools(_opt(options, pre + "_use_laynorm"))
    acts = _strs(_opt(options, pre + "_act"))
    bidir = _tobool(_opt(options, pre + "_bidir"))
    if drop_masks is None:
        drop_masks = make_drop_masks(kind, options, x.shape[1], to_do)

    if _tobool(_opt(options, pre + "_use_laynorm_inp")):
        x = layer_norm(x, sd["ln0.gamma"], sd["ln0.beta"])
    if _tobool(_opt(options, pre + "_use_batchnorm_inp")):
        xb = batch_norm(x.reshape(x.shape[0] * x.shape[1], x.shape[2]), sd, "bn0", training)
        x = xb.view(x.shape[0], x.shape[1], x.shape[2])

    per_layer = []
    for i, H in enumerate(lay):
        if bidir:
            x = torch.cat([x, flip_time(x)], 1)
        T, R = x.shape[0], x.shape[1]
        h0 = torch.zeros(R, H, dtype=x.dtype)
        mask = drop_masks[i].to(x.dtype)

        proj = {}
        for (wn, un, bnn) in gates:
            p = _linear(x, sd, "%s.%d" % (wn, i))
            if use_bn[i]:
                pb = batch_norm(p.reshape(T * R, H), sd, "%s.%d" % (bnn, i), training)
                p = pb.view(T, R, H)
            proj[wn] = p
        if not index_like_reference:
            proj = {k: v.unbind(0) for k, v in proj.items()}

        def U(name, h):
            return _mm(h, sd["%s.%d.weight" % (name, i)])

        seen = []

        def cand(at, k):
            if kink_log is not None:
                seen.append(at.detach() > 0)
            if kinks is not None:
                return _act_kink_forced(acts[i], at, kinks[i][k])
            return activation(acts[i], at)

        hs = []
        ht = h0
        ct = h0
        for k in range(T):
            if kind == "liGRU":  # :1133-1136
                zt = torch.sigmoid(proj["wz"][k] + U("uz", ht))
                at = proj["wh"][k] + U("uh", ht)
                hcand = cand(at, k) * mask
                ht = zt * ht + (1 - zt) * hcand
            elif kind == "minimalGRU":  # :1294-1297
                zt = torch.sigmoid(proj["wz"][k] + U("uz", ht))
                at = proj["wh"][k] + U("uh", zt * ht)
                hcand = cand(at, k) * mask
                ht = zt * ht + (1 - zt) * hcand
            elif kind == "GRU":  # :632-636
                zt = torch.sigmoid(proj["wz"][k] + U("uz", ht))
                rt = torch.sigmoid(proj["wr"][k] + U("ur", ht))
                at = proj["wh"][k] + U("uh", rt * ht)
                hcand = cand(at, k) * mask
                ht = zt * ht + (1 - zt) * hcand
            elif kind == "LSTM":  # :460-464
                ft = torch.sigmoid(proj["wfx"][k] + U("ufh", ht))
                it = torch.sigmoid(proj["wix"][k] + U("uih", ht))
                ot = torch.sigmoid(proj["wox"][k] + U("uoh", ht))
                ct = it * activation(acts[i], proj["wcx"][k] + U("uch", ht)) * mask + ft * ct
                ht = ot * activation(acts[i], ct)
            elif kind == "RNN":  # :1441-1442
                at = proj["wh"][k] + U("uh", ht)
                ht = cand(at, k) * mask
            else:
                raise ValueError(kind)
            if use_ln[i]:
                ht = layer_norm(ht, sd["ln.%d.gamma" % i], sd["ln.%d.beta" % i])
            hs.append(ht)
        h = torch.stack(hs)
        if kink_log is not None:
            kink_log.append(torch.stack(seen) if seen else None)
        if bidir:
            B = R // 2
            h = torch.cat([h[:, :B], flip_time(h[:, B:].contiguous())], 2)
        x = h
        per_layer.append(h)
    if return_all:
        return x, per_layer
    return x


# ----------------------------------------------------------------------------
# CNN / SincNet
# ----------------------------------------------------------------------------
def sinc_filters(low_hz_, band_hz_, kernel_size, sample_rate, min_low_hz, min_band_hz):
    """SincConv bank synthesis, neural_networks.py:1754-1803 (window grid is
    linspace(0, K, K), not the textbook Hamming grid; division by the row max
    is differentiated through)."""
    K = kernel_size + 1 if kernel_size % 2 == 0 else kernel_size
    dt = low_hz_.dtype
    n_lin = torch.linspace(0, K, steps=K).to(dt)
    window = 0.54 - 0.46 * torch.cos(2 * math.pi * n_lin / K)
    n = (K - 1) / 2
    n_ = (torch.arange(-n, n + 1).view(1, -1) / sample_rate).to(dt)

    def sinc(v):
        left = v[:, 0:int((v.shape[1] - 1) / 2)]
        yl = torch.sin(left) / left
        return torch.cat([yl, torch.ones([v.shape[0], 1], dtype=dt), torch.flip(yl, dims=[1])], dim=1)

    low = min_low_hz / sample_rate + torch.abs(low_hz_)
    high = low + min_band_hz / sample_rate + torch.abs(band_hz_)
    lp1 = 2 * low * sinc(2 * math.pi * torch.matmul(low, n_) * sample_rate)
    lp2 = 2 * high * sinc(2 * math.pi * torch.matmul(high, n_) * sample_rate)
    bp = lp2 - lp1
    mx, _ = torch.max(bp, dim=1, keepdim=True)
    bp = bp / mx
    return (bp * window).view(low_hz_.shape[0], 1, K)


def _max_pool(v, pool, forced_idx):
    """F.max_pool1d(v, pool); with ``forced_idx`` (another run's arg-max positions, absolute indices into v's last
    axis) the pooled value is gathered from THOSE positions, so that both runs route the gradient identically."""
    if forced_idx is None:
        return F.max_pool1d(v, pool)
    return v.gather(2, forced_idx.to(torch.int64))


def conv_stack_forward(kind, options, sd, x, training=True, drop_masks=None, kinks=None, pool_idx=None):
    """CNN.forward (:1530-1556) / SincNet.forward (:1639-1665).  Note the
    reference constructs its BatchNorm1d with eps = pooled length (positional
    slip at :1515-1517 / :1615-1617) and runs the block twice when both
    laynorm and batchnorm are set; both quirks are kept."""
    pre = "cnn" if kind == "CNN" else "sinc"
    n_filt = _ints(_opt(options, pre + "_N_filt"))
    len_filt = _ints(_opt(options, pre + "_len_filt"))
    pool = _ints(_opt(options, pre + "_max_pool_len"))
    acts = _strs(_opt(options, pre + "_act"))
    drop = _floats(_opt(options, pre + "_drop"))
    use_ln = _bools(_opt(options, pre + "_use_laynorm"))
    use_bn = _bools(_opt(options, pre + "_use_batchnorm"))
    batch, seq_len = x.shape[0], x.shape[1]
    if _tobool(_opt(options, pre + "_use_laynorm_inp")):
        x = layer_norm(x, sd["ln0.gamma"], sd["ln0.beta"])
    if _tobool(_opt(options, pre + "_use_batchnorm_inp")):
        x = batch_norm(x, sd, "bn0", training)
    x = x.view(batch, 1, seq_len)
    cur = seq_len
    for i in range(len(n_filt)):
        pooled = int((cur - len_filt[i] + 1) / pool[i])

        def conv(v):
            if kind == "SincNet" and i == 0:
                w = sinc_filters(sd["conv.0.low_hz_"], sd["conv.0.band_hz_"], len_filt[0],
                                 int(_opt(options, "sinc_sample_rate")),
                                 int(_opt(options, "sinc_min_low_hz")),
                                 int(_opt(options, "sinc_min_band_hz")))
                return _conv1d(v, w)
            return _conv1d(v, sd["conv.%d.weight" % i], sd["conv.%d.bias" % i])

        m = None if drop_masks is None else drop_masks[i]
        kp = None if kinks is None else kinks[i]
        pi = None if pool_idx is None else pool_idx[i]
        if (kp is not None or pi is not None) and use_ln[i] and use_bn[i]:
            raise ValueError("forced patterns: one pooling / activation call per layer (laynorm XOR batchnorm)")
        x_in = x
        if use_ln[i]:
            z = _max_pool(conv(x_in), pool[i], pi)
            z = layer_norm(z, sd["ln.%d.gamma" % i], sd["ln.%d.beta" % i])
            x = _dropout(_act_maybe_forced(acts[i], z, kp), drop[i], training, m)
        if use_bn[i]:
            z = _max_pool(conv(x), pool[i], pi)
            z = batch_norm(z, sd, "bn.%d" % i, training, eps=float(pooled))
            x = _dropout(_act_maybe_forced(acts[i], z, kp), drop[i], training, m)
        if not use_bn[i] and not use_ln[i]:
            z = _max_pool(conv(x_in), pool[i], pi)
            x = _dropout(_act_maybe_forced(acts[i], z, kp), drop[i], training, m)
        cur = pooled
    return x.view(batch, -1)


# ----------------------------------------------------------------------------
# dispatcher + forward_model glue
# ----------------------------------------------------------------------------
def arch_forward(arch_class, options, sd, x, training=True, to_do="train", drop_masks=None,
                 index_like_reference=False, kinks=None, pool_idx=None):
    if arch_class == "MLP":
        return mlp_forward(options, sd, x, training, drop_masks, kinks)
    if arch_class in _REC:
        return recurrent_forward(arch_class, options, sd, x, training, to_do, drop_masks,
                                 index_like_reference, kinks=kinks)
    if arch_class in ("CNN", "SincNet"):
        return conv_stack_forward(arch_class, options, sd, x, training, drop_masks, kinks, pool_idx)
    raise ValueError("oracle does not cover arch_class " + arch_class)


def two_head_loss(rec_out, sd_cd, opt_cd, sd_mono, opt_mono, lab_cd, lab_mono, mono_weight=1.0):
    """The [model] section of cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg:189-197 as
    forward_model evaluates it (utils.py:2333-2339, 2344-2361, 2379-2381, 2395-2401):
    out_dnn2/out_dnn3 = MLP heads on the (T*B, D) view, NLLLoss (mean) on each,
    loss_final = loss_cd + w * loss_mono, err_final on the cd head."""
    flat = rec_out.reshape(rec_out.shape[0] * rec_out.shape[1], -1) if rec_out.dim() == 3 else rec_out
    out_cd = mlp_forward(opt_cd, sd_cd, flat)
    out_mono = mlp_forward(opt_mono, sd_mono, flat)
    loss_cd = F.nll_loss(out_cd, lab_cd)
    loss_mono = F.nll_loss(out_mono, lab_mono)
    loss = loss_cd + loss_mono * mono_weight
    err = torch.mean((torch.max(out_cd, dim=1)[1] != lab_cd).float())
    return loss, err, out_cd, out_mono


def recipe_forward(model, options, arch_dict, sds, inp, fea_dict, lab_dict, rec_masks=None, drop_masks=None,
                   kinks=None, training=True, to_do="train", relu_patterns=None, pool_idx=None):
    """A whole [model] section on one batch, as utils.forward_model evaluates it (utils.py:2296-2420), over this
    oracle's arch functions - the generic form of two_head_loss, for the shipped recipes of the other BASELINE
    configurations (LSTM / GRU / SincNet + MLP / MLP).

    model      the [model] lines; options[section] the cfg sections (strings); arch_dict[name] = [section, name, seq]
    sds        {arch name: parameter dict}
    inp        (T, B, feat + labels) or (N, feat + labels); label columns are float, cast with .long() (:2352)
    rec_masks  recurrent drop masks, consumed layer by layer in call order (torch.bernoulli tap of the reference run)
    drop_masks {"<arch>/drop.<i>": 0/1 mask} of the nn.Dropout modules (unscaled; scaled by 1/(1-p) here)
    kinks      per-layer ReLU patterns of the FIRST recurrent architecture (kink-forced mode)
    relu_patterns {"<arch>/act.<i>": bool tensor} ReLU patterns of feed-forward layers (MLP / CNN / SincNet), and
    pool_idx   {"<arch>/conv.<i>": arg-max positions} of the max-pools of conv stacks - another run's discrete
               decisions, so that both runs differentiate the same piecewise-linear function (test mode)
    -> dict of every named result (out_*, loss_*, err_*)
    """
    import re

    outs = {}
    rec_masks = list(rec_masks) if rec_masks is not None else None
    T = inp.shape[0] if inp.dim() == 3 else None
    for fea, spec in fea_dict.items():
        outs[fea] = inp[..., spec[5]:spec[6]]

    def labels(name):
        return inp[..., lab_dict[name][3]].reshape(-1).long()

    def flat(t):
        return t.reshape(-1, t.shape[-1]) if t.dim() == 3 else t

    for line in model:
        out_name, op, a, b = [s.strip() for s in re.findall(r"(.*)=(.*)\((.*),(.*)\)", line)[0]]
        if op == "compute":
            sec, name, seq = arch_dict[a]
            o = options[sec]
            cls = o["arch_class"]
            x = outs[b]
            if not seq and x.dim() == 3:
                x = x.reshape(-1, x.shape[-1])
            if seq and x.dim() == 2:
                x = x.reshape(T, -1, x.shape[-1])
            if cls in _REC:
                n_lay = len(_ints(_opt(o, _REC[cls][0] + "_lay")))
                m = None
                if rec_masks is not None:
                    m, rec_masks = rec_masks[:n_lay], rec_masks[n_lay:]
                outs[out_name] = recurrent_forward(cls, o, sds[name], x, training, to_do, m, kinks=kinks)
                kinks = None
            else:
                pre = {"MLP": "dnn", "CNN": "cnn", "SincNet": "sinc"}[cls]
                drops = _floats(_opt(o, pre + "_drop"))
                dm = None
                if drop_masks is not None and training and any(p > 0 for p in drops):
                    dm = []
                    for i, p in enumerate(drops):
                        mk = drop_masks.get("%s/drop.%d" % (name, i))
                        dm.append(None if mk is None else mk.to(x.dtype) / (1.0 - p))
                n_l = len(drops)
                kp = pi = None
                if relu_patterns is not None and any(("%s/act.%d" % (name, i)) in relu_patterns for i in range(n_l)):
                    kp = [relu_patterns.get("%s/act.%d" % (name, i)) for i in range(n_l)]
                if pool_idx is not None and cls != "MLP" and any(("%s/conv.%d" % (name, i)) in pool_idx for i in range(n_l)):
                    pi = [pool_idx.get("%s/conv.%d" % (name, i)) for i in range(n_l)]
                outs[out_name] = arch_forward(cls, o, sds[name], x, training, to_do, dm, kinks=kp, pool_idx=pi)
        elif op == "cost_nll":
            outs[out_name] = F.nll_loss(flat(outs[a]), labels(b))
        elif op == "cost_err":
            outs[out_name] = torch.mean((torch.max(flat(outs[a]), dim=1)[1] != labels(b)).float())
        elif op == "concatenate":
            outs[out_name] = torch.cat((outs[a], outs[b]), outs[a].dim() - 1)
        elif op == "mult":
            outs[out_name] = outs[a] * outs[b]
        elif op == "sum":
            outs[out_name] = outs[a] + outs[b]
        elif op == "mult_constant":
            outs[out_name] = outs[a] * float(b)
        elif op == "sum_constant":
            outs[out_name] = outs[a] + float(b)
        elif op == "avg":
            outs[out_name] = (outs[a] + outs[b]) / 2
        elif op == "mse":
            outs[out_name] = torch.mean((outs[a] - outs[b]) ** 2)
        else:
            raise ValueError("unknown [model] operation " + op)
    return outs


# ----------------------------------------------------------------------------------------------------------------------
# The reference's drop-mask STREAM at the level of the generator (what PK_MASK_RNG=reference must reproduce on the device).
#
# neural_networks.py:1102-1107 (and :430-441, :604-615, :1263-1268, :1416-1421) draws a layer's mask with
# torch.bernoulli(torch.Tensor(rows, H).fill_(1 - p)) on torch's global CPU generator.  That algorithm is not in the
# reference checkout: it lives in its dependency PyTorch (here 2.10.0: at::CPUGeneratorImpl over at::mt19937, the
# published MT19937 of Matsumoto & Nishimura with a 624-word state refilled block by block; bernoulli_(Tensor p) on the
# CPU is a serial loop taking one 32-bit output y per element, u = (y & (2^24 - 1)) * 2^-24, mask = u < p).  Restated
# below in numpy; PINNED in tests/test_ref_mask_stream_host.py against torch itself run in the test (masks AND the
# generator state afterwards), so a torch build with another layout or sampling rule fails there, not silently.
def mt19937_refill_np(words):
    """One block refill of at::mt19937 (next_state): 624 new words from 624 old ones."""
    import numpy as np
    N, M = 624, 397
    p = np.asarray(words, dtype=np.uint32)
    q = np.empty(N, dtype=np.uint32)

    def twist(u, v):
        y = (u & np.uint32(0x80000000)) | (v & np.uint32(0x7FFFFFFF))
        return (y >> np.uint32(1)) ^ np.where(v & np.uint32(1), np.uint32(0x9908B0DF), np.uint32(0))
    q[:N - M] = p[M:] ^ twist(p[:N - M], p[1:N - M + 1])                       # j <  227: uses old words only
    for lo in range(N - M, N - 1, N - M):                                      # then 227 at a time on the new ones
        hi = min(lo + (N - M), N - 1)
        q[lo:hi] = q[lo - (N - M):hi - (N - M)] ^ twist(p[lo:hi], p[lo + 1:hi + 1])
    q[N - 1] = q[M - 1] ^ twist(p[N - 1:N], q[0:1])[0]
    return q


def mt19937_bernoulli_np(state626, n, keep):
    """state626: uint32 [626] = the engine's 624 words, `left`, `next` (the fields of at::mt19937's data).  Returns
    (mask float32 [n], new state626): element i is 1 if the i-th draw's uniform < keep (a float32) else 0."""
    import numpy as np
    st = np.asarray(state626, dtype=np.uint32).copy()
    words, left, nxt = st[:624].copy(), int(st[624]), int(st[625])
    keep = np.float32(keep)
    out = np.empty(n, dtype=np.float32)
    i = 0
    while i < n:
        left -= 1
        if left == 0:
            words = mt19937_refill_np(words)
            left, nxt = 624, 0
        take = min(n - i, left)
        y = words[nxt:nxt + take].copy()
        y ^= y >> np.uint32(11)
        y ^= (y << np.uint32(7)) & np.uint32(0x9D2C5680)
        y ^= (y << np.uint32(15)) & np.uint32(0xEFC60000)
        y ^= y >> np.uint32(18)
        u = (y & np.uint32(0xFFFFFF)).astype(np.float32) * np.float32(2.0 ** -24)
        out[i:i + take] = (u < keep).astype(np.float32)
        i += take
        nxt += take
        left -= take - 1
    return out, np.concatenate([words, np.array([left, nxt], dtype=np.uint32)])
