"""Round-6 GPU tests of the host-side plumbing around the kernels (each against the engine's own unshared / eager path or
a torch evaluation of the same op; the model-level parity lives in test_gpu_parity.py / test_gpu_reference_pins.py)."""
import importlib

import pytest
import torch

from golden_util import rel_err

pytestmark = pytest.mark.gpu
F_ = importlib.import_module("pytorch-kaldi_amd.functional")


@pytest.mark.parametrize("second_use", ["posteriors", "posteriors_and_input", "none"])
def test_concatenated_head_gradient_with_other_contributors(second_use):
    """Two fused-cost heads on one activation at >= 4096 rows take the ONE-GEMM input gradient (functional._cat_finish,
    neural_networks.py:139-148 twice + utils.py:2361).  A head whose log-posteriors ALSO feed a second cost term sends a
    second gradient into the heads' private alias through its own node (LinearLogSoftmaxFn.backward), and a direct
    consumer of the activation adds to the activation's own buffer: the shared product must be added to those, never
    matched by buffer identity (advisor, round 5: an uninitialised buffer used to enter autograd's accumulation)."""
    g = torch.Generator().manual_seed(11)
    rows, K = 4224, 160
    x0 = torch.randn(rows, K, generator=g)
    w1, w2 = torch.randn(45, K, generator=g) / 12, torch.randn(13, K, generator=g) / 12
    l1, l2 = torch.randint(0, 45, (rows,), generator=g).cuda(), torch.randint(0, 13, (rows,), generator=g).cuda()
    F_.set_precision("bf16")
    res = {}
    try:
        for share in (True, False):
            F_._DxShare.on = share
            x = x0.clone().cuda().requires_grad_(True)
            h = x * 1.0
            a, b = w1.clone().cuda().requires_grad_(True), w2.clone().cuda().requires_grad_(True)
            y1, y2 = F_.linear_log_softmax(h, a, None), F_.linear_log_softmax(h, b, None)
            loss = F_.head_nll(y1, l1)[0] + 0.5 * F_.head_nll(y2, l2)[0]
            if second_use != "none":
                loss = loss + 0.3 * (y1 * y1).mean()  # e.g. a confidence penalty on the senone posteriors
            if second_use == "posteriors_and_input":
                loss = loss + 2.0 * (h * h).mean()
            loss.backward()
            torch.cuda.synchronize()
            assert torch.isfinite(x.grad).all()
            res[share] = (x.grad.clone(), a.grad.clone(), b.grad.clone())
            assert not F_._DxShare.cat or not share
    finally:
        F_._DxShare.on = True
        F_.set_precision("fp32")
    for got, ref in zip(res[True], res[False]):
        assert rel_err(got, ref) < 2e-6


@pytest.mark.parametrize("kind,pre,act", [("liGRU", "ligru", "relu"), ("RNN", "rnn", "tanh"), ("LSTM", "lstm", "tanh"), ("GRU", "gru", "tanh"),
                                           ("minimalGRU", "minimalgru", "relu")])
def test_forward_only_chunks_save_nothing_and_change_nothing(kind, pre, act):
    """Validation / forward chunks (core.py:644-671: module.eval(), torch.no_grad()): the perf-mode recurrences save nothing
    for a backward pass and the inner layers of a stack write no fp32 output (functional.RecLayerPerfFn, cfg[16]) - the
    stack's output must be BIT-identical to the same eval-mode forward with autograd on, which takes the training kernels."""
    nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
    H, T, B, D = 550, 23, 37, 40
    j = lambda v: ",".join([str(v)] * 3)  # noqa: E731
    opts = {pre + "_lay": j(H), pre + "_drop": j(0.2), pre + "_use_laynorm_inp": "False", pre + "_use_batchnorm_inp": "False",
            pre + "_use_laynorm": j(False), pre + "_use_batchnorm": j(False), pre + "_bidir": "True", pre + "_act": j(act),
            pre + "_orthinit": "True", "use_cuda": "True", "to_do": "valid"}
    torch.manual_seed(3)
    net = getattr(nn_amd, kind)(opts, D).cuda().eval()
    x = torch.randn(T, B, D, generator=torch.Generator().manual_seed(4)).cuda()
    F_.set_precision("bf16")
    try:
        y_grad = net(x.clone().requires_grad_(True))     # autograd on: S and every layer's Y are written
        with torch.no_grad():
            y_fwd = net(x)
        torch.cuda.synchronize()
    finally:
        F_.set_precision("fp32")
    assert torch.equal(y_fwd, y_grad.detach())
    tw_a, tw_b = getattr(y_fwd, "_pk_twin", None), getattr(y_grad, "_pk_twin", None)
    assert tw_a is not None and tw_b is not None
    used = 2 * ((H + 7) // 8 * 8)  # (columns beyond ndir * Hp of the pitch are left undefined, include/pk_amd.h)
    assert torch.equal(tw_a[0][:, :used], tw_b[0][:, :used])   # the bf16 copy the heads read


@pytest.mark.parametrize("kind,pre,act,bn", [("liGRU", "ligru", "relu", True), ("LSTM", "lstm", "tanh", False), ("GRU", "gru", "tanh", True),
                                              ("minimalGRU", "minimalgru", "relu", False), ("RNN", "rnn", "tanh", False)])
def test_fp32_weight_gradients_on_the_side_stream_equal_autograds(kind, pre, act, bn, monkeypatch):
    """Exact-fp32 mode with flat-bucket parameters (round 6): the dW / dU GEMMs of a recurrent layer run on the side stream
    and accumulate into the flat .grad (functional.RecLayerFn, _deferred_dU_f32) instead of being returned to autograd
    (neural_networks.py:351-379 et al.: every gate is an nn.Linear whose weight autograd fills).  Same kernels, same
    operands, same split of every reduction: BIT-identical to the autograd route (PK_EXPERIMENT f32_wgrad_side=0) for the
    one backward pass per zero_grad() a training step makes - the 30-step RMSprop fixture amplifies a different summation
    order of dU into 1e-3 of the loss - and equal within fp32 rounding when a second pass adds to the first (gradient
    accumulation) and to the plain-parameter route."""
    nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
    optim = importlib.import_module("pytorch-kaldi_amd.optim")
    H, T, B, D = 96, 21, 19, 40
    j = lambda v: ",".join([str(v)] * 2)  # noqa: E731
    opts = {pre + "_lay": j(H), pre + "_drop": j(0.0), pre + "_use_laynorm_inp": "False", pre + "_use_batchnorm_inp": "False",
            pre + "_use_laynorm": j(False), pre + "_use_batchnorm": j(bn), pre + "_bidir": "True", pre + "_act": j(act),
            pre + "_orthinit": "True", "use_cuda": "True", "to_do": "train"}
    x = torch.randn(T, B, D, generator=torch.Generator().manual_seed(4)).cuda()
    tgt = torch.randn(T, B, 2 * H, generator=torch.Generator().manual_seed(5)).cuda()
    assert not F_.bf16_mode()
    grads = {}
    launches = []
    real = F_.side_launch
    monkeypatch.setattr(F_, "side_launch", lambda *a, **k: (launches.append(1), real(*a, **k))[1])
    for route in ("plain", "flat_autograd", "flat_side"):
        monkeypatch.setenv("PK_EXPERIMENT", "f32_wgrad_side=%d" % (route == "flat_side"))
        torch.manual_seed(3)
        net = getattr(nn_amd, kind)(opts, D).cuda().train()
        fp = optim.FlatParams(net) if route != "plain" else None
        n0 = len(launches)
        for k in (1, 2):
            y = net(x)
            ((y - tgt) ** 2).mean().backward()
            F_.join_side()
            torch.cuda.synchronize()
            grads[route, k] = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
        assert (len(launches) - n0 == 2 * 2 * 2) == (route == "flat_side")  # (dW and dU) x layers x passes
        del fp
    for n, g in grads["flat_autograd", 1].items():
        assert torch.equal(grads["flat_side", 1][n], g), n
    # the second pass adds (grad + d0) + d1 where autograd adds grad + (d0 + d1): fp32 rounding of the association
    for k in (1, 2):
        for other in ("flat_autograd", "plain"):
            worst = max((rel_err(grads["flat_side", k][n], g), n) for n, g in grads[other, k].items())
            assert worst[0] < 2e-6, (k, other, worst)
