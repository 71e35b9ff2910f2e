"""CPU-side checks of the round-3 host logic: coverage predicates and buffer sizes of the new entry points (pure host
functions of the library), the algorithm choice for layers that normalise h_t, the grouped dropout draw, and the oracle's
bf16-operand convolution (the checker of the opt-in bf16 convolutions) against plain autograd on rounded operands."""
import importlib
import os
import sys

import pytest
import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
_lib = importlib.import_module("pytorch-kaldi_amd._lib")
F_ = importlib.import_module("pytorch-kaldi_amd.functional")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        importlib.import_module("pytorch-kaldi_amd.build").build()
    return _lib.load()


def test_layernorm_saved_tensor_holds_both_layouts(lib):
    """pk_rec_ln_saved_floats covers the step-wise layout ([mean, rinv, pre-LN h] rows) and the persistent one (pre-LN h in
    Y's layout, 16-byte aligned statistics behind it, slack for the row quads of the last cluster)."""
    for T, B, bidir, H in ((1, 1, 0, 2), (7, 3, 1, 13), (500, 128, 1, 550), (12, 24, 1, 550)):
        R = B * (1 + bidir)
        n = lib.pk_rec_ln_saved_floats(T, B, bidir, H)
        assert n >= T * R * (H + 2)
        assert n >= (T * R * H + 3) // 4 * 4 + T * R * 2 + 64


def test_small_batch_and_conv_coverage_predicates(lib):
    assert lib.pk_linear_bn_act_bf16_covers(128, 1024, 440) == 1
    assert lib.pk_linear_bn_act_bf16_covers(129, 1024, 440) == 0      # more rows than one row tile
    assert lib.pk_linear_bn_act_bf16_covers(128, 1023, 440) == 0      # the layer's bf16 copy is written 8 bytes at a time
    assert lib.pk_linear_bn_act_bf16_covers(1, 1024, 440) == 0        # BatchNorm over one row: torch raises, so do we
    assert lib.pk_bn_act_bwd_small_covers(128, 1024) == 1 and lib.pk_bn_act_bwd_small_covers(129, 8) == 0
    # SincNet's four layers and the CNN recipe's three (cfg/TIMIT_baselines/TIMIT_{SincNet_raw,CNN_fbank}.cfg)
    for cin, cout, k, pool in ((1, 128, 129, 3), (128, 60, 5, 3), (60, 60, 5, 3), (60, 60, 3, 2), (40, 80, 10, 3),
                               (80, 60, 3, 2), (60, 60, 3, 1)):
        assert lib.pk_conv_bf16_covers(cin, cout, k, pool) == 1, (cin, cout, k, pool)
        fwd = lib.pk_conv_bf16_work_floats(128, cin, 3200, cout, k, pool, 0)
        bwd = lib.pk_conv_bf16_work_floats(128, cin, 3200, cout, k, pool, 1)
        assert 0 < fwd < bwd and bwd - fwd == 512 * cout * cin * k   # the filter gradient's per-workgroup partial sums
    assert lib.pk_conv_bf16_covers(2, 3, 4, 5) == 0      # pool width that does not divide a wave's 48 positions
    assert lib.pk_conv_bf16_covers(5, 20, 251, 3) == 0   # filter longer than nine tap blocks
    assert lib.pk_conv_bf16_covers(8, 129, 5, 3) == 0    # more than 128 output channels


def test_algorithm_choice_for_layers_that_normalise_h(monkeypatch):
    """functional.choose_rec_algo: per-step LayerNorm inside the persistent loop where the kernels have it, step-wise
    elsewhere; PK_EXPERIMENT rec_ln_persist=0 and the forced algorithms are honoured."""
    monkeypatch.delenv("PK_EXPERIMENT", raising=False)
    old_prec, old_algo = F_.settings.precision, F_.settings.rec_algo
    try:
        F_.set_rec_algo("auto")
        F_.set_precision("bf16")
        for cell in ("liGRU", "RNN", "LSTM", "GRU", "minimalGRU"):
            assert F_.choose_rec_algo(cell, 550, True) == F_.REC_PERSISTENT, cell
        assert F_.choose_rec_algo("GRU", 550, False) == F_.REC_STEPWISE  # (without LayerNorm GRU takes the perf pipeline, not this path)
        assert F_.choose_rec_algo("liGRU", 577, True) == F_.REC_STEPWISE
        assert F_.choose_rec_algo("liGRU", 1, True) == F_.REC_STEPWISE   # the unbiased std of one unit does not exist
        F_.set_precision("fp32")
        assert F_.choose_rec_algo("liGRU", 550, True) == F_.REC_PERSISTENT
        assert F_.choose_rec_algo("RNN", 14, True) == F_.REC_PERSISTENT
        # (round 6: the fourth-generation fp32 kernels normalise h_t too; without them - rec_f32_gen4=0 - these cells go back
        # to the step-wise algorithm and a forced persistent run says that it cannot)
        for cell in ("LSTM", "GRU", "minimalGRU"):
            assert F_.choose_rec_algo(cell, 550, True) == F_.REC_PERSISTENT, cell
        monkeypatch.setenv("PK_EXPERIMENT", "rec_f32_gen4=0")
        for cell in ("LSTM", "GRU", "minimalGRU"):
            assert F_.choose_rec_algo(cell, 550, True) == F_.REC_STEPWISE, cell
        F_.set_rec_algo("persistent")
        with pytest.raises(_lib.PkError):
            F_.choose_rec_algo("LSTM", 550, True)
        monkeypatch.delenv("PK_EXPERIMENT")
        assert F_.choose_rec_algo("LSTM", 550, True) == F_.REC_PERSISTENT
        F_.set_rec_algo("auto")
        monkeypatch.setenv("PK_EXPERIMENT", "rec_ln_persist=0")
        assert F_.choose_rec_algo("GRU", 550, True) == F_.REC_STEPWISE
        assert F_.choose_rec_algo("liGRU", 550, True) == F_.REC_STEPWISE
        monkeypatch.setenv("PK_EXPERIMENT", "rec_ln_persist=1,rec_f32_gen=1")
        assert F_.choose_rec_algo("liGRU", 550, True) == F_.REC_STEPWISE  # the first-generation fp32 kernels do not normalise
    finally:
        F_.set_precision(old_prec)
        F_.set_rec_algo(old_algo)


def test_grouped_dropout_draw_on_the_host():
    """masks_ahead / dropout_mask: one flat draw per distinct p, 16-byte aligned slices handed out in call order, a fresh
    draw (and an emptied queue) when the forward takes another route, nothing queued under forced masks."""
    dev = torch.device("cpu")
    specs = [(8, 12, 0.25), (8, 6, 0.5), (8, 12, 0.25)]
    torch.manual_seed(1)
    F_.masks_ahead(specs, dev)
    assert len(F_._Ahead.queue) == 3
    a = F_.dropout_mask(torch.empty(8, 12), 0.25)
    b = F_.dropout_mask(torch.empty(8, 6), 0.5)
    c = F_.dropout_mask(torch.empty(8, 12), 0.25)
    assert F_._Ahead.queue == []
    assert a.untyped_storage().data_ptr() == c.untyped_storage().data_ptr() != b.untyped_storage().data_ptr()
    assert (c.storage_offset() - a.storage_offset()) % 4 == 0
    for m, p in ((a, 0.25), (b, 0.5), (c, 0.25)):
        assert all(v == 0.0 or abs(v - 1.0 / (1.0 - p)) < 1e-6 for v in torch.unique(m).tolist())
    F_.masks_ahead(specs, dev)
    m = F_.dropout_mask(torch.empty(3, 3), 0.25)
    assert tuple(m.shape) == (3, 3) and F_._Ahead.queue == []
    F_.set_forced_dropout([torch.ones(8, 12)])
    try:
        F_.masks_ahead(specs, dev)
        assert F_._Ahead.queue == []
        assert torch.equal(F_.dropout_mask(torch.empty(8, 12), 0.5), torch.ones(8, 12) / 0.5)
    finally:
        F_.set_forced_dropout(None)
    F_.masks_ahead(specs[:1], dev)   # a single call: nothing to group
    assert F_._Ahead.queue == []


def test_oracle_bf16_convolution_model():
    """The checker of the opt-in bf16 convolutions: conv1d on bf16-rounded x / w, gradients from the bf16-rounded output
    gradient - equal to plain autograd on the rounded operands when the output gradient is bf16-exact, and switched on
    only by bf16_operands(conv=True)."""
    import pk_oracle as O

    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 4, 50, generator=g, dtype=torch.float64).requires_grad_(True)
    w = (torch.randn(6, 4, 7, generator=g, dtype=torch.float64) / 5).requires_grad_(True)
    b = torch.randn(6, generator=g, dtype=torch.float64).requires_grad_(True)
    cot = torch.randn(3, 6, 44, generator=g, dtype=torch.float64).to(torch.bfloat16).double()  # bf16-exact cotangent
    with O.bf16_operands(conv="all"):  # (4 input channels: below the engine's default threshold of 8)
        y = O._conv1d(x, w, b)
        (y * cot).sum().backward()
    xb = x.detach().to(torch.bfloat16).double().requires_grad_(True)
    wb = w.detach().to(torch.bfloat16).double().requires_grad_(True)
    yr = TF.conv1d(xb, wb, b.detach())
    (yr * cot).sum().backward()
    assert torch.allclose(y, yr, rtol=0, atol=1e-12)
    assert torch.allclose(x.grad, xb.grad, rtol=0, atol=1e-12) and torch.allclose(w.grad, wb.grad, rtol=0, atol=1e-12)
    assert torch.allclose(b.grad, cot.sum((0, 2)), rtol=0, atol=1e-12)
    with O.bf16_operands():  # without conv= the model leaves convolutions in full precision (the engine's default)
        y2 = O._conv1d(x.detach(), w.detach(), b.detach())
    assert torch.allclose(y2, TF.conv1d(x.detach(), w.detach(), b.detach()), rtol=0, atol=0)
    with O.bf16_operands(conv=True):  # the default rule leaves a 4-channel layer exact as well
        y3 = O._conv1d(x.detach(), w.detach(), b.detach())
    assert torch.equal(y3, y2)


def test_direct_gradients_only_without_a_reducer_and_inside_backward(monkeypatch):
    """functional.direct_grads_ok: kernels may accumulate into the flat .grad themselves only when the parameters are
    flat-bucket views with a pre-allocated gradient, the caller has declared an accumulating backward pass
    (functional.accumulating_backward: run_nn's step does; torch.autograd.grad() callers do not), autograd is not
    recording, no data-parallel reducer listens for gradient hooks, and PK_EXPERIMENT direct_grads is not 0."""
    monkeypatch.delenv("PK_EXPERIMENT", raising=False)
    old = F_.settings.precision
    q = torch.nn.Parameter(torch.zeros(4, 3))
    q.grad = torch.zeros(4, 3)
    q._pk_flat = True
    plain = torch.nn.Parameter(torch.zeros(4, 3))
    plain.grad = torch.zeros(4, 3)
    try:
        F_.set_precision("bf16")
        monkeypatch.setattr(F_._Side, "listener", None)
        assert not F_.direct_grads_ok([q])            # autograd is recording: not inside a backward pass
        with torch.no_grad():
            assert not F_.direct_grads_ok([q])        # nobody said this backward pass accumulates into .grad
        with torch.no_grad(), F_.accumulating_backward():
            assert F_.direct_grads_ok([q])
            with F_.accumulating_backward():
                assert F_.direct_grads_ok([q])
            assert F_.direct_grads_ok([q])            # (nests)
            assert not F_.direct_grads_ok([plain])    # not a flat-bucket parameter
            assert not F_.direct_grads_ok([q, plain])
            monkeypatch.setattr(F_._Side, "listener", lambda params: None)
            assert not F_.direct_grads_ok([q])        # a reducer counts gradient hooks: everything goes through autograd
            monkeypatch.setattr(F_._Side, "listener", None)
            monkeypatch.setenv("PK_EXPERIMENT", "direct_grads=0")
            assert not F_.direct_grads_ok([q])
            monkeypatch.delenv("PK_EXPERIMENT")
            F_.set_precision("fp32")
            assert not F_.direct_grads_ok([q])        # parity mode: node by node
    finally:
        F_.set_precision(old)
