"""The eight-wave LSTM time loops (pk_rec_persist2_lstm.hip, the default) against the four-wave ones
(pk_rec_persist2.hip, PK_EXPERIMENT lstm_waves=4 / pk_persist2_set_lstm_waves(4)) on identical inputs.

Forward: the gate split keeps the per-gate MFMA accumulation order, so the two kernels differ only where an fp32
expression contracts differently and a bf16 rounding of h_t flips (measured 1e-7 .. 2e-4 norm-relative).  Backward: the
K split reorders the fp32 sum over the gates (measured up to 2.7e-3 on the bf16-rounded operands).  Parity of either
kernel with the CPU oracle is test_gpu_parity.py's job (it runs with the default, i.e. the eight-wave kernels).
Reference loop: neural_networks.py:457-469.
"""
import importlib

import pytest
import torch

from golden_util import rel_err

_lib_mod = importlib.import_module("pytorch-kaldi_amd._lib")

pytestmark = pytest.mark.gpu


def _opts(lay, act, bidir):
    n = len(lay)
    j = lambda v: ",".join([str(v)] * n)  # noqa: E731
    return {"lstm_lay": ",".join(map(str, lay)), "lstm_drop": j(0.2), "lstm_use_laynorm_inp": "False",
            "lstm_use_batchnorm_inp": "False", "lstm_use_laynorm": j(False), "lstm_use_batchnorm": j(True),
            "lstm_bidir": str(bidir), "lstm_act": j(act), "lstm_orthinit": "True", "use_cuda": "True", "to_do": "train"}


def _both(net, x, cot, masks, lib):
    res = {}
    for w in (4, 8):
        lib.pk_persist2_set_lstm_waves(w)
        assert lib.pk_persist2_get_lstm_waves() == w
        lib.pk_persist2_error_reset()
        net.zero_grad()
        xe = x.clone().requires_grad_(True)
        y = net(xe, drop_masks=masks)
        (y * cot).sum().backward()
        torch.cuda.synchronize()
        assert lib.pk_persist2_error_count() == 0
        res[w] = (y.detach().cpu(), xe.grad.cpu(), {k: p.grad.detach().cpu().clone() for k, p in net.named_parameters()
                                                     if p.grad is not None})
    return res


@pytest.fixture
def engine():
    from engine_util import F_amd, nn_amd

    lib = importlib.import_module("pytorch-kaldi_amd._lib").load()
    old = (F_amd.settings.precision, F_amd.settings.rec_algo, lib.pk_persist2_get_lstm_waves())
    F_amd.set_precision("bf16")
    F_amd.set_rec_algo("persistent")
    yield F_amd, nn_amd, lib
    F_amd.set_precision(old[0])
    F_amd.set_rec_algo(old[1])
    lib.pk_persist2_set_lstm_waves(old[2])
    lib.pk_persist2_set_mode(0)


def test_default_is_eight_waves(engine):
    _, _, lib = engine
    assert lib.pk_persist2_get_lstm_waves() == (4 if _lib_mod.experiment("lstm_waves") == "4" else 8)


@pytest.mark.parametrize("H,T,B,bidir,act,safe", [
    (550, 12, 5, True, "tanh", 0),     # the recipes' width: the last wave straddles H
    (40, 9, 3, True, "tanh", 0),
    (20, 7, 4, False, "tanh", 1),      # one direction, placement-independent exchange
    (14, 5, 33, True, "relu", 0),      # three clusters, the last one ragged
    (576, 3, 2, True, "tanh", 0),      # widest layer the register-resident U covers
    (24, 4, 300, True, "elu", 0),      # 600 rows: more than one launch; run-time activation switch
    (550, 40, 128, True, "tanh", 0),   # full clusters of 16 rows, both directions
    (8, 1, 2, False, "tanh", 1),       # one step
])
def test_training_path(engine, H, T, B, bidir, act, safe):
    _, nn_amd, lib = engine
    torch.manual_seed(21)
    net = nn_amd.LSTM(_opts([H, H], act, bidir), 23).cuda().train()
    g = torch.Generator().manual_seed(13)
    x = torch.randn(T, B, 23, generator=g).cuda()
    cot = torch.randn(T, B, net.out_dim, generator=g).cuda()
    masks = [torch.bernoulli(torch.full((B * (2 if bidir else 1), H), 0.8), generator=g) for _ in range(2)]
    lib.pk_persist2_set_mode(safe)
    res = _both(net, x, cot, masks, lib)
    assert rel_err(res[8][0], res[4][0]) < 1e-3
    assert rel_err(res[8][1], res[4][1]) < 1e-2
    for k, v in res[4][2].items():
        assert rel_err(res[8][2][k], v) < 1e-2, k


@pytest.mark.parametrize("H,T,B", [(550, 10, 6), (33, 6, 20)])
def test_frozen_batchnorm_path(engine, H, T, B):
    """Eval-mode module with autograd on: BatchNorm backward through the running statistics stays on the general path,
    which asks the time loop for the fp32 gate gradients (dP2) as well - the second wave of a pair writes half of them."""
    _, nn_amd, lib = engine
    torch.manual_seed(5)
    net = nn_amd.LSTM(_opts([H], "tanh", True), 19).cuda().eval()
    g = torch.Generator().manual_seed(7)
    x = torch.randn(T, B, 19, generator=g).cuda()
    cot = torch.randn(T, B, net.out_dim, generator=g).cuda()
    masks = [torch.bernoulli(torch.full((2 * B, H), 0.8), generator=g)]
    res = _both(net, x, cot, masks, lib)
    assert rel_err(res[8][0], res[4][0]) < 1e-3
    assert rel_err(res[8][1], res[4][1]) < 1e-2
    for k, v in res[4][2].items():
        assert rel_err(res[8][2][k], v) < 1e-2, k
