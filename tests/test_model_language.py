"""pytorch-kaldi_amd/utils.py::model_init / forward_model (the host-side mirror of utils.py:2031-2103, 2296-2420) on the
[model] graph of tests/golden/e2e_model_language.npz, which the reference's own model_init / forward_model evaluated
(oracle/make_golden.py::model_lang_case): two input streams, concatenate, avg, mult, sum, sum_constant, mult_constant,
mse, the (T*B, .) views of a sequence batch fed to non-sequence networks, and the forward pass that stops at
forward_outs[-1].  The networks are CPU stand-ins built on the oracle (tests/cpu_arch.py): this runs without a GPU."""
import configparser
import importlib

import torch

from golden_util import Golden, rel_err

U = importlib.import_module("pytorch-kaldi_amd.utils")


def _setup(to_do):
    g = Golden("e2e_model_language")
    m = g.meta
    cfg = configparser.ConfigParser()
    cfg["exp"] = {"to_do": to_do, "use_cuda": "False"}
    for sec, opts in m["options"].items():
        cfg[sec] = dict(opts, arch_library="cpu_arch")
    inp_out_dict = {k: list(v) for k, v in m["fea_dict"].items()}
    nns, costs = U.model_init(inp_out_dict, m["model"], cfg, m["arch_dict"], False, False, to_do)
    for name, net in nns.items():
        sd = g.group("sd/%s/" % name)
        for v in sd.values():
            if v.is_floating_point():
                v.requires_grad_(True)
        net.sd = sd
    return g, m, inp_out_dict, nns, costs


def test_model_init_records_dimensions_like_the_reference():
    g, m, inp_out_dict, nns, costs = _setup("train")
    assert inp_out_dict == m["inp_out_dict"]          # every intermediate's dimension, by name
    assert sorted(nns) == sorted(m["arch_dict"]) and sorted(costs) == ["loss_cd", "loss_mono"]
    assert all(net.training for net in nns.values())


def test_training_step_matches_reference():
    g, m, inp_out_dict, nns, costs = _setup("train")
    outs = U.forward_model(m["fea_dict"], m["lab_dict"], m["arch_dict"], m["model"], nns, costs, g.t("inp"), inp_out_dict,
                           m["T"], m["B"], "train", [])
    assert sorted(outs) == m["train_keys"]
    for k in m["train_keys"]:
        ref = g.t("train/" + k)
        assert outs[k].shape == ref.shape, k
        if ref.dim() == 0:
            assert abs(float(outs[k].detach()) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), k
        else:
            assert rel_err(outs[k], ref) < 2e-6, k
    outs["loss_final"].backward()
    for name, net in nns.items():
        for k, ref in g.group("grad/%s/" % name).items():
            assert rel_err(net.sd[k].grad, ref) < 5e-5, (name, k)


def test_forward_pass_stops_at_the_last_requested_output():
    g, m, inp_out_dict, nns, costs = _setup("forward")
    assert not any(net.training for net in nns.values())
    with torch.no_grad():
        outs = U.forward_model(m["fea_dict"], m["lab_dict"], m["arch_dict"], m["model"], nns, costs, g.t("inp")[:, 0, :],
                               inp_out_dict, m["T"], 1, "forward", ["out_dnn3"])
    assert sorted(outs) == m["forward_keys"]           # nothing after out_dnn3, no costs
    for k in m["forward_keys"]:
        assert rel_err(outs[k], g.t("forward/" + k)) < 2e-6, k


def test_cost_lines_on_a_fused_head_take_the_joined_path(monkeypatch):
    """Host logic of the perf-mode fusion across the plug-in boundary (no GPU: functional.head_nll is replaced by a torch
    stand-in): a cost_nll line on an output that carries `_pk_head` goes through head_nll, the cost_err line on the same
    (output, label) pair reads the error rate the same pass produced, a cost on an ordinary output and a cost object that is
    not a plain NLLLoss() take torch's path - and every value still equals the reference's."""
    g, m, inp_out_dict, nns, costs = _setup("train")
    calls = []

    def fake_head_nll(y, lab, ignore_index=-100):
        loss = torch.nn.functional.nll_loss(y, lab, ignore_index=ignore_index)
        err = (y.detach().argmax(1) != lab).float().mean()
        calls.append((tuple(y.shape), int(ignore_index)))
        return loss, torch.stack([loss.detach(), err, torch.tensor(float(lab.numel())), torch.tensor(0.0)])

    noted = []
    monkeypatch.setattr(U.F_, "head_nll", fake_head_nll)
    monkeypatch.setattr(U.F_, "note_label_check", lambda stats: noted.append(stats))
    head = nns["head_cd"]
    plain_forward = head.forward

    def tagged(x):  # what nn.MLP does for a perf-mode output layer: the output carries the head's inputs
        y = plain_forward(x)
        y._pk_head = ("stand-in",)
        return y

    monkeypatch.setattr(head, "forward", tagged)
    outs = U.forward_model(m["fea_dict"], m["lab_dict"], m["arch_dict"], m["model"], nns, costs, g.t("inp"), inp_out_dict,
                           m["T"], m["B"], "train", [])
    assert len(calls) == 1 and calls[0][1] == -100 and len(noted) == 1   # loss_cd only: out_dnn4 is an ordinary output
    for k in ("loss_cd", "loss_mono", "loss_final", "err_final"):
        ref = g.t("train/" + k)
        assert abs(float(outs[k].detach()) - float(ref)) <= 2e-6 * max(1.0, abs(float(ref))), k
    assert outs["err_final"] is noted[0][1] or float(outs["err_final"]) == float(noted[0][1])
    # a weighted NLLLoss is not what head_nll implements: torch's path, no call
    calls.clear()
    costs["loss_cd"] = torch.nn.NLLLoss(weight=torch.ones(outs["out_dnn3"].shape[1]))
    U.forward_model(m["fea_dict"], m["lab_dict"], m["arch_dict"], m["model"], nns, costs, g.t("inp"), inp_out_dict,
                    m["T"], m["B"], "train", [])
    assert calls == []


def test_the_published_twin_follows_the_flattening_reshape(monkeypatch):
    """A sequence network's output that carries `_pk_twin` (the bf16 copy a perf-mode recurrent stack published) keeps it
    across the (T, B, F) -> (T*B, F) reshape forward_model does in front of a non-sequence network."""
    seen = {}

    class Seq(torch.nn.Module):
        def forward(self, x):
            y = x * 2.0
            y._pk_twin = ("xb", (2, 3, 8), y._version)
            return y

    class Flat(torch.nn.Module):
        def forward(self, x):
            seen["twin"] = getattr(x, "_pk_twin", None)
            seen["shape"] = tuple(x.shape)
            return x.sum(1, keepdim=True)

    T, B, F = 5, 3, 6
    inp = torch.randn(T, B, F + 1)
    fea_dict = {"fea": ["fea", "lst", "opts", "cw_l", "cw_r", 0, F]}
    lab_dict = {"lab": ["lab", "folder", "opts", F]}
    arch_dict = {"rec": ["architecture1", "rec", True], "head": ["architecture2", "head", False]}
    model = ["out1=compute(rec,fea)", "out2=compute(head,out1)"]
    inp_out_dict = {"fea": ["a", "b", "c", "d", "e", 0, F, F], "out1": [F], "out2": [1]}
    outs = U.forward_model(fea_dict, lab_dict, arch_dict, model, {"rec": Seq(), "head": Flat()}, {}, inp, inp_out_dict,
                           T, B, "train", [])
    assert seen["shape"] == (T * B, F)
    assert seen["twin"] is not None and seen["twin"][0] == "xb" and seen["twin"][1] == (2, 3, 8)
    assert outs["out2"].shape == (T * B, 1)
