"""The timed (bf16) path and the training trajectory pinned DIRECTLY to outputs of the reference.

Round-1 compared the bf16 pipeline only with the engine's own fp32 mode.  Here every comparison is against arrays the
reference itself produced (oracle/make_golden.py imports /root/reference unmodified):

  * test_golden_module_bf16        every recurrent / MLP module fixture with set_precision("bf16");
  * test_golden_e2e_bf16           the two-head recipe one level up, bf16;
  * test_ce_loss_trajectory        30 optimizer steps of the (scaled) shipped Li-GRU recipe: loss_final / err_final of
                                   EVERY step and the final parameters against the reference's own run of
                                   core.py:616-642 (forward_model, zero_grad, backward, RMSprop.step);
  * test_config_scale_golden       the UNSCALED recipe (liGRU 5 x 550 bidirectional + 1938 / 48 heads) at T = 500:
                                   posteriors / loss at 1e-4, gradients at 1e-4 with the kink-forced backward
                                   (functional.set_forced_kinks).

bf16 mode is graded in two steps:
  (A) the engine implements the bf16-OPERAND form of the reference algorithm: every GEMM operand (projections,
      h_{t-1}.U^T, Linear layers; in backward dY.W, dY^T.x, dgates.U, dgates^T.h) rounded to bf16 (unit round-off
      u = 2^-9), products accumulated in fp32, everything element-wise in fp32.  oracle/pk_oracle.py models exactly that
      (`with O.bf16_operands():` - the fp32 oracle, itself pinned to the reference at 2e-6, with operand rounding), so
      engine vs model must be TIGHT: 5e-3 on outputs, 2e-2 on gradients (what remains is fp32 summation order moving a
      few values across a bf16 rounding boundary or a ReLU kink);
  (B) engine vs the reference's own fp32 arrays: bounded by how far the bf16-operand algorithm ITSELF is from fp32 on that
      network - 1.5 x the model's deviation + 4u.  Analytically one GEMM stage adds u*sqrt(2/3)*sqrt(2) = 2.3e-3
      norm-relative (independent of K), stages add in quadrature (outputs land at 2-5e-3 for the 2-3 layer fixtures,
      5e-2 with an input LayerNorm on a 6-dim input); gradients of ReLU networks are dominated by kink flips instead
      (a pre-activation within a bf16 rounding error of 0 changes the linear piece): 5-13 % on the tiny fixtures,
      where one flip is a visible fraction of a 9-step, 3-sequence gradient - both numbers are printed by the test.
"""
import configparser
import importlib
import os
import math

import numpy as np
import pytest
import torch

from golden_util import conv_bf16_on, Golden, analytically_zero, grad_err, list_cases, rel_err

pytestmark = pytest.mark.gpu
U_BF16 = 2.0 ** -9
REC = ("liGRU", "LSTM", "GRU", "minimalGRU", "RNN")
BF16_CASES = [c for c in list_cases() if not c.startswith(("e2e_", "chunk_", "io_", "train_", "scale_"))
              and Golden(c).meta["arch_class"] in REC + ("MLP",)]


TIGHT_OUT, TIGHT_GRAD = 5e-3, 2e-2


def model_bound(e_model):
    """(B): engine-vs-reference bound from the bf16-operand model's own deviation from the reference."""
    return 1.5 * e_model + 4 * U_BF16


def emulate_module(g):
    """The bf16-operand model on a module fixture: y, dx, grads (CPU)."""
    import pk_oracle as O

    m = g.meta
    sd = g.group("sd/")
    for k in sd:
        if sd[k].is_floating_point() and "running" not in k:
            sd[k].requires_grad_(True)
    x = g.t("x").clone().requires_grad_(True)
    with O.bf16_operands(conv=conv_bf16_on()):
        y = O.arch_forward(m["arch_class"], m["options"], sd, x, training=m["training"], to_do=m["to_do"],
                           drop_masks=g.masks() or None)
        dx = grads = None
        if "dx" in g.arrays:
            (y * g.t("cot")).sum().backward()
            dx = x.grad
            grads = {k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}
    return y.detach(), dx, grads


@pytest.fixture(autouse=True)
def _restore_mode():
    F_ = importlib.import_module("pytorch-kaldi_amd.functional")
    yield
    F_.set_precision("fp32")
    F_.set_rec_algo("auto")
    F_.set_forced_kinks(None)
    F_.set_forced_dropout(None)
    F_.set_forced_decisions(None, None)


def _two_step(name, got, model, ref, tight, total=None):
    """(A) engine vs bf16-operand model, (B) engine vs reference within the model's own deviation."""
    if total is None:
        e_a, e_m, e_b = rel_err(got, model), rel_err(model, ref), rel_err(got, ref)
    else:
        e_a, e_m, e_b = grad_err(got, model, total), grad_err(model, ref, total), grad_err(got, ref, total)
    assert e_a < tight, (name, "engine vs bf16-operand model", e_a)
    assert e_b < model_bound(e_m), (name, "engine vs reference", e_b, "model vs reference", e_m)
    return e_a, e_m, e_b


@pytest.mark.parametrize("case", BF16_CASES)
def test_golden_module_bf16(case):
    """bf16 perf mode against the reference's own arrays (not against the fp32 engine)."""
    from engine_util import F_amd, build_engine, run_engine

    g = Golden(case)
    m = g.meta
    F_amd.set_precision("bf16")
    F_amd.set_rec_algo("auto")
    net = build_engine(m, g.group("sd/"))
    has_bwd = "dx" in g.arrays
    y, dx, grads = run_engine(net, m, g.t("x"), g.masks(), g.t("cot") if has_bwd else None)
    ym, dxm, gm = emulate_module(g)
    rep = {"y": _two_step("y", y, ym, g.t("y"), TIGHT_OUT)}
    if has_bwd:
        rep["dx"] = _two_step("dx", dx, dxm, g.t("dx"), TIGHT_GRAD)
        ref = g.group("grad/")
        total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ref.values())))
        worst = (0.0, 0.0, 0.0)
        for k, r in ref.items():
            if analytically_zero(m, k):
                assert float(grads[k].double().norm()) <= TIGHT_GRAD * total, k
                continue
            worst = max(worst, _two_step(k, grads[k], gm[k], r, TIGHT_GRAD, total), key=lambda t: t[2])
        rep["worst grad"] = worst
    print("\n%s bf16: " % case + "; ".join("%s engine-vs-model %.1e, model-vs-ref %.1e, engine-vs-ref %.1e" % ((k,) + v)
                                           for k, v in rep.items()))
    sd = {k: v.detach().cpu() for k, v in net.state_dict().items()}
    for k, ref in g.group("sd_after/").items():  # running statistics of BatchNorm: one projection stage
        if ref.is_floating_point():
            assert rel_err(sd[k], ref) < 4 * U_BF16, k
        else:
            assert int(sd[k]) == int(ref), k


def _recipe_engine(meta, sds, to_do="train"):
    """Engine networks of a recipe fixture through the model_init mirror (arch_library switched - the one-line
    change a user makes), parameters loaded from the fixture."""
    U = importlib.import_module("pytorch-kaldi_amd.utils")
    cfg = configparser.ConfigParser()
    cfg["exp"] = {"to_do": to_do, "use_cuda": "True"}
    for sec, opts in meta["options"].items():
        cfg[sec] = {k: v.replace("%", "%%") for k, v in opts.items()}
        cfg[sec]["arch_library"] = "pytorch-kaldi_amd.nn"
    nfea = meta["nfea"]
    fea_dict = {"fmllr": ["fmllr", "lst", "opts", "0", "0", 0, nfea, nfea]}
    lab_dict = {"lab_cd": ["lab_cd", "f", "o", nfea], "lab_mono": ["lab_mono", "f", "o", nfea + 1]}
    arch_dict = {"liGRU_layers": ["architecture1", "liGRU_layers", True],
                 "MLP_layers": ["architecture2", "MLP_layers", False],
                 "MLP_layers2": ["architecture3", "MLP_layers2", False]}
    iod = {"fmllr": fea_dict["fmllr"][5:]}
    nns, costs = U.model_init(iod, meta["model"], cfg, arch_dict, True, False, to_do)
    if sds is not None:
        for name, net in nns.items():
            net.load_state_dict(sds[name])
            net.cuda()
    return U, cfg, fea_dict, lab_dict, arch_dict, iod, nns, costs


def _with_masks(rec, masks):
    orig = rec.forward

    def fwd(x):
        return orig(x, drop_masks=masks)

    return orig, fwd


def _oracle_recipe(g, sds_prefix="sd/"):
    """Oracle-side parameter dicts of a recipe fixture (leaf tensors with requires_grad)."""
    sds = {}
    for arch in ("liGRU_layers", "MLP_layers", "MLP_layers2"):
        sd = g.group("%s%s/" % (sds_prefix, arch))
        for k in sd:
            if sd[k].is_floating_point() and "running" not in k:
                sd[k].requires_grad_(True)
        sds[arch] = sd
    return sds


def _oracle_step(O, m, sds, inp, masks, emulate, kinks=None):
    import contextlib

    nfea = m["nfea"]
    with (O.bf16_operands() if emulate else contextlib.nullcontext()):
        out1 = O.recurrent_forward("liGRU", m["options"]["architecture1"], sds["liGRU_layers"], inp[:, :, :nfea],
                                   drop_masks=masks, kinks=kinks)
        loss, err, out2, out3 = O.two_head_loss(out1, sds["MLP_layers"], m["options"]["architecture2"], sds["MLP_layers2"],
                                                m["options"]["architecture3"], inp[:, :, nfea].reshape(-1).long(),
                                                inp[:, :, nfea + 1].reshape(-1).long())
    return out1, out2, out3, loss, err


def test_golden_e2e_bf16():
    import pk_oracle as O
    from engine_util import F_amd

    g = Golden("e2e_ligru_two_heads")
    m = g.meta
    F_amd.set_precision("bf16")
    sds = {n: g.group("sd/%s/" % n) for n in ("liGRU_layers", "MLP_layers", "MLP_layers2")}
    U, cfg, fea_dict, lab_dict, arch_dict, iod, nns, costs = _recipe_engine(m, sds)
    rec = nns["liGRU_layers"]
    orig, rec.forward = _with_masks(rec, g.masks())
    outs = U.forward_model(fea_dict, lab_dict, arch_dict, m["model"], nns, costs, g.t("inp").cuda(), iod, m["T"], m["B"],
                           "train", [])
    outs["loss_final"].backward()
    torch.cuda.synchronize()
    rec.forward = orig
    osd = _oracle_recipe(g)
    o1, o2, o3, oloss, _ = _oracle_step(O, m, osd, g.t("inp"), g.masks(), True)
    oloss.backward()
    for k, om in (("out_dnn1", o1), ("out_dnn2", o2), ("out_dnn3", o3)):
        _two_step(k, outs[k].reshape(g.t(k).shape), om.reshape(g.t(k).shape), g.t(k), TIGHT_OUT)
    lref = float(g.t("loss_final"))
    assert abs(float(outs["loss_final"]) - float(oloss)) < TIGHT_OUT * lref
    assert abs(float(outs["loss_final"]) - lref) < model_bound(abs(float(oloss) - lref) / lref) * lref
    for name, net in nns.items():
        ref = g.group("grad/%s/" % name)
        total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ref.values())))
        for k, p_ in net.named_parameters():
            if k in ref and float(ref[k].norm()) > 1e-6 * total:
                _two_step((name, k), p_.grad, osd[name][k].grad, ref[k], TIGHT_GRAD, total)


# --------------------------------------------------------------------------------------------------------------------
# CE-loss match (BASELINE.json "frames/sec ...; CE-loss match"): the training trajectory
# --------------------------------------------------------------------------------------------------------------------
def _model_trajectory(g, n_steps):
    """The bf16-operand model trained like the fixture (CPU): loss per step."""
    import pk_oracle as O

    m = g.meta
    sds = _oracle_recipe(g)
    optims = []
    for arch, sec in (("liGRU_layers", "architecture1"), ("MLP_layers", "architecture2"), ("MLP_layers2", "architecture3")):
        o = m["options"][sec]
        optims.append(torch.optim.RMSprop([v for v in sds[arch].values() if v.requires_grad], lr=float(o["arch_lr"]),
                                          alpha=float(o["opt_alpha"]), eps=float(o["opt_eps"])))
    batches, masks, n_lay = g.t("batches"), g.masks(), m["n_lay"]
    losses = []
    for step in range(n_steps):
        _, _, _, loss, _ = _oracle_step(O, m, sds, batches[step % m["n_batches"]], masks[step * n_lay:(step + 1) * n_lay], True)
        for o in optims:
            o.zero_grad()
        loss.backward()
        for o in optims:
            o.step()
        losses.append(float(loss.detach()))
    return np.array(losses)


@pytest.mark.parametrize("prec,mode", [("fp32", "torch-optim"), ("fp32", "fused"), ("bf16", "fused")])
def test_ce_loss_trajectory(prec, mode):
    """30 optimizer steps of the (scaled) shipped Li-GRU recipe, reference drop masks, loss_final / err_final of
    every step against the reference's own training run.  "fused": FlatParams + the fused RMSprop kernel + side-stream
    weight gradients (what bench.py / run_nn_dp use); "torch-optim": the reference's optimizer_init on the engine
    classes (what a user who only switches arch_library gets).
    fp32: 1e-4 at EVERY step is tighter than the reference is to itself under round-off - one ulp on the inputs moves its
    own run 9.4e-4 by step 24 and its final parameters up to 8.5e-2 (tools/diag_fp32_trajectory_floor.py,
    tests/golden/fp32_trajectory_floor.json).  The engine holds it because its exact-fp32 GEMMs sum every output element
    as one fmaf chain over ascending k (pk_gemm.hip, both forms), which is what the reference's CPU library does on
    products of this size: the same discrete events (ReLU kinks, RMSprop's first-step signs) happen on both sides."""
    from engine_util import F_amd

    optim_ = importlib.import_module("pytorch-kaldi_amd.optim")
    g = Golden("train_ligru_30steps")
    m = g.meta
    F_amd.set_precision(prec)
    sds = {n: g.group("sd/%s/" % n) for n in ("liGRU_layers", "MLP_layers", "MLP_layers2")}
    U, cfg, fea_dict, lab_dict, arch_dict, iod, nns, costs = _recipe_engine(m, sds)
    opts = optim_.fused_optimizer_init(nns, cfg, arch_dict) if mode == "fused" else U.optimizer_init(nns, cfg, arch_dict)
    batches = g.t("batches").cuda()
    masks = [mk.cuda() for mk in g.masks()]
    rec = nns["liGRU_layers"]
    n_lay = m["n_lay"]
    losses, errs = [], []
    for step in range(m["n_steps"]):
        orig, rec.forward = _with_masks(rec, masks[step * n_lay:(step + 1) * n_lay])
        outs = U.forward_model(fea_dict, lab_dict, arch_dict, m["model"], nns, costs, batches[step % m["n_batches"]], iod,
                               m["T"], m["B"], "train", [])
        rec.forward = orig
        for o in opts.values():
            o.zero_grad()
        outs["loss_final"].backward()
        for o in opts.values():
            o.step()
        losses.append(outs["loss_final"].detach())
        errs.append(outs["err_final"].detach())
    torch.cuda.synchronize()
    loss = torch.stack(losses).double().cpu().numpy()
    err = torch.stack(errs).double().cpu().numpy()
    ref_loss, ref_err = g.arrays["loss"], g.arrays["err"]
    assert ref_loss[-1] < ref_loss[0] - 0.2  # the fixture really trains (5.08 -> 4.75)
    rel = np.abs(loss - ref_loss) / np.abs(ref_loss)
    flips = np.abs(err - ref_err) * m["T"] * m["B"]  # frame error rate: a count over T*B = 80 frames
    if prec == "fp32":
        # north_star's 1e-4 at EVERY step, the argmax of every frame, and the final parameters
        assert rel.max() < 1e-4, (mode, float(rel.max()), int(rel.argmax()))
        assert flips.max() < 0.5, flips.max()
        for name, net in nns.items():
            ref = g.group("sd_final/%s/" % name)
            for k, v in net.state_dict().items():
                if v.is_floating_point() and float(ref[k].norm()) > 0:
                    assert rel_err(v, ref[k]) < 1e-4, (name, k, rel_err(v, ref[k]))
                elif not v.is_floating_point():
                    assert int(v) == int(ref[k]), (name, k)
        return
    # bf16: (A) the engine follows the bf16-operand model step for step while the two are still the same trajectory
    # (a training run amplifies any difference - also the model's and the engine's different fp32 summation orders -
    # so the tight comparison is made over the first steps), (B) over all 30 steps it stays within the model's own
    # distance from the reference's fp32 run
    model = _model_trajectory(g, m["n_steps"])
    rel_model = np.abs(model - ref_loss) / ref_loss
    rel_a = np.abs(loss - model) / model
    print("\nCE trajectory bf16: engine-vs-model first 5 steps %.1e, all %.1e; model-vs-reference %.1e; "
          "engine-vs-reference %.1e" % (rel_a[:5].max(), rel_a.max(), rel_model.max(), rel.max()))
    assert rel_a[:5].max() < 2e-4, rel_a[:5]
    assert rel.max() < model_bound(rel_model.max()), (float(rel.max()), float(rel_model.max()))
    assert flips.max() < 4.5, flips.max()
    assert loss[-1] < loss[0] - 0.2  # and it trains


# --------------------------------------------------------------------------------------------------------------------
# config-scale golden: liGRU 5 x 550 + heads at T = 500 (the metric's network and sequence length)
# --------------------------------------------------------------------------------------------------------------------
def _ck(t, seed, k=4):
    v = t.detach().double().cpu().reshape(-1).numpy()
    rs = np.random.RandomState(seed)
    return np.concatenate(([float(np.linalg.norm(v))],
                           [float(np.dot(v, rs.randint(0, 2, v.size) * 2.0 - 1.0)) for _ in range(k)]))


def _rows(t, stride):
    t = t.detach().cpu()
    if t.dim() <= 1:
        return t.reshape(-1)[::stride]
    return t.reshape(t.shape[0], -1)[::stride]


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["scale_ligru_T500", "scale_ligru_T500_B32"])
def test_config_scale_golden(case, prec):
    """fp32: posteriors / loss / kink-forced gradients at 1e-4 against the reference's own run.  bf16 (the mode bench.py
    times, same network, same T): the two-step grading against the bf16-operand model run here on the host CPU from the
    same seed-derived parameters.  Two reference runs: B = 4 (one partial cluster of the persistent launch) and B = 32
    (round 6: 64 rows = four full 16-row clusters x 500 steps - the reference's own arrays above toy batch; its O(T^2)
    autograd takes 95 s there)."""
    from engine_util import F_amd

    g = Golden(case)
    m = g.meta
    T, B, H, L = m["T"], m["B"], m["H"], m["n_lay"]
    F_amd.set_precision(prec)
    torch.manual_seed(m["seed"])  # the parameters are what this seed gives the reference (checked below)
    U, cfg, fea_dict, lab_dict, arch_dict, iod, nns, costs = _recipe_engine(m, None)
    for name, net in nns.items():  # the initialisation itself is part of the drop-in contract
        for k, v in net.state_dict().items():
            if v.is_floating_point():
                ref = g.arrays["init_ck/%s/%s" % (name, k)]
                got = _ck(v, 7)
                assert np.abs(got - ref).max() <= 1e-5 * max(1.0, ref[0]), ("initialisation differs", name, k)
    init = {n: {k: v.detach().cpu().clone() for k, v in net.state_dict().items()} for n, net in nns.items()}
    masks = [g.t("mask/%d" % i).float().cuda() for i in range(m["n_masks"])]
    kinks = [torch.from_numpy(np.unpackbits(g.arrays["kink/%d" % i])[:T * 2 * B * H].reshape(T, 2 * B, H).astype(bool))
             for i in range(L)]
    # both precisions differentiate the reference's own piecewise-linear function (kink-forced): at T = 500 ANY
    # perturbation - 1e-7 of fp32 summation order, 4e-3 of bf16 rounding - otherwise moves thousands of ReLU
    # pre-activations across 0 and the gradient comparison measures that chaos (3e-3 between two fp32 runs of the
    # reference itself, Appendix B; 8-10 % for bf16 operands) instead of the arithmetic
    report = F_amd.set_forced_kinks(kinks)
    rec = nns["liGRU_layers"]
    orig, rec.forward = _with_masks(rec, masks)
    outs = U.forward_model(fea_dict, lab_dict, arch_dict, m["model"], nns, costs, g.t("inp").cuda(), iod, T, B, "train", [])
    outs["loss_final"].backward()
    torch.cuda.synchronize()
    rec.forward = orig
    st = m["strides"]
    lref, eref = float(g.t("loss_final")), float(g.t("err_final"))
    gtotal = math.sqrt(sum(float(g.arrays[k][0]) ** 2 for k in g.arrays if k.startswith("grad/") and k.endswith("/ck")))

    def grad_items():
        for name, net in nns.items():
            for k, p in net.named_parameters():
                key = "grad/%s/%s" % (name, k)
                if key + "/rows" not in g.arrays:
                    assert p.grad is None or float(p.grad.abs().max()) == 0.0, (name, k)
                    continue
                ref_ck = g.arrays[key + "/ck"]
                if ref_ck[0] < 1e-6 * gtotal:  # analytically-zero gradients (rounding noise in the reference)
                    assert float(p.grad.norm()) < 1e-4 * gtotal
                    continue
                yield name, k, key, p, g.t(key + "/rows"), ref_ck

    if prec == "fp32":
        for k in ("out_dnn1", "out_dnn2", "out_dnn3"):
            o = outs[k].reshape(T * B, -1)
            e = rel_err(_rows(o, st["out/%s/stride" % k]), g.t("out/%s/rows" % k))
            assert e < 1e-4, (k, e)
            ck, ref = _ck(o, 11), g.arrays["out/%s/ck" % k]
            assert abs(ck[0] - ref[0]) < 1e-4 * ref[0], k                      # whole-tensor norm
            assert np.abs(ck[1:] - ref[1:]).max() < 4e-4 * ref[0], k          # +-1 projections: |<d, r>| ~ ||d||
        assert abs(float(outs["loss_final"]) - lref) < 1e-4 * abs(lref)
        assert abs(float(outs["err_final"]) - eref) * T * B < 0.5
        # the reference's kink pattern differs from the engine's own only where a_t is rounding noise
        assert len(report) == L
        for flipped, total, worst in report:
            assert flipped < 2e-4 * total and worst < 1e-4, report
        worst = 0.0
        for name, k, key, p, ref_rows, ref_ck in grad_items():
            got_rows = _rows(p.grad, st[key + "/stride"])
            frac = float(ref_rows.double().norm()) / ref_ck[0]  # share of the tensor the sample holds
            e = grad_err(got_rows, ref_rows, gtotal * frac)
            worst = max(worst, e)
            assert e < 1e-4, (name, k, e)
            ck = _ck(p.grad, 13)
            assert abs(ck[0] - ref_ck[0]) < 1e-4 * max(ref_ck[0], 1e-3 * gtotal), (name, k)
            assert np.abs(ck[1:] - ref_ck[1:]).max() < 4e-4 * max(ref_ck[0], 1e-3 * gtotal), (name, k)
        print("\nconfig-scale golden %s [fp32]: worst gradient row-sample error %.2e; kink report %s" % (case, worst, report))
        return
    # ---- bf16: the bf16-operand model on the host CPU, same parameters / input / masks (a few seconds)
    import pk_oracle as O

    osd = {}
    for n in init:
        osd[n] = {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in init[n].items()}
    o1, o2, o3, oloss, oerr = _oracle_step(O, m, osd, g.t("inp"), [mk.cpu() for mk in masks], True, kinks=kinks)
    oloss.backward()
    rep = {}
    for k, om in (("out_dnn1", o1), ("out_dnn2", o2), ("out_dnn3", o3)):
        s_ = st["out/%s/stride" % k]
        rep[k] = _two_step(k, _rows(outs[k].reshape(T * B, -1), s_), _rows(om.detach().reshape(T * B, -1), s_),
                           g.t("out/%s/rows" % k), TIGHT_OUT)
    assert abs(float(outs["loss_final"]) - float(oloss)) < TIGHT_OUT * lref
    assert abs(float(outs["loss_final"]) - lref) < model_bound(abs(float(oloss) - lref) / lref) * lref
    assert abs(float(outs["err_final"]) - eref) * T * B < 8.5
    worst = (0.0, 0.0, 0.0)
    for name, k, key, p, ref_rows, ref_ck in grad_items():
        s_ = st[key + "/stride"]
        frac = float(ref_rows.double().norm()) / ref_ck[0]
        worst = max(worst, _two_step((name, k), _rows(p.grad, s_), _rows(osd[name][k].grad, s_), ref_rows, TIGHT_GRAD,
                                     gtotal * frac), key=lambda t: t[2])
    flipped = sum(r[0] for r in report) / float(sum(r[1] for r in report))
    print("\nconfig-scale golden " + case + " [bf16]: outputs (engine-vs-model, model-vs-ref, engine-vs-ref) %s; worst kink-forced "
          "gradient %s; share of ReLU pre-activations whose sign bf16 rounding changed: %.2e"
          % ({k: tuple("%.1e" % x for x in v) for k, v in rep.items()}, tuple("%.1e" % x for x in worst), flipped))


# --------------------------------------------------------------------------------------------------------------------
# config-scale goldens of the other four BASELINE configurations (shipped cfg files, unscaled, run by the reference)
# --------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["scale_lstm_T500", "scale_gru_libri_T500", "scale_sincnet_3200", "scale_mlp_440"])
def test_recipe_scale_golden(case, prec):
    """LSTM 4 x 550 (cfg/TIMIT_baselines/TIMIT_LSTM_fmllr.cfg:131-217) and GRU 5 x 550 + 3400-way head
    (cfg/Librispeech_baselines/libri_GRU_fmllr.cfg:76-146) at T = 500, SincNet [128,60,60,60] / [129,5,5,3] on 3200
    samples + MLP + heads (TIMIT_SincNet_raw.cfg:87-211), MLP 440 -> 1024 x 5 + heads at batch 128
    (TIMIT_MLP_fmllr.cfg:131-214): parameters from the seed (initialisation checked against the reference's checksums),
    the reference run's recurrent drop masks and nn.Dropout masks injected.
    fp32: posteriors, loss, every parameter gradient at 1e-4 against the reference's own run (tanh cells: nothing is
    kink-forced).  bf16: the two-step grading against the bf16-operand model run here on the host CPU."""
    import pk_oracle as O
    import scale_util as SU
    from engine_util import F_amd

    g = Golden(case)
    m = g.meta
    F_amd.set_precision(prec)
    U, cfg, iod, nns, costs = SU.build(g, True)
    init = SU.oracle_params(nns) if prec == "bf16" else None
    rmasks = [mk.cuda() for mk in SU.rec_masks(g)]
    restore = []
    for name, net in nns.items():  # the recurrent architecture takes the reference run's masks, layer by layer
        if type(net).__name__ in REC:
            n = len(net._lay)
            orig, net.forward = _with_masks(net, rmasks[:n])
            rmasks = rmasks[n:]
            restore.append((net, orig))
    F_amd.set_forced_dropout([mk for _, mk in SU.dropout_masks(g)])
    # the feed-forward stacks differentiate the reference run's own piecewise-linear function: its ReLU patterns and the
    # arg-max positions of its max-pools (a conv stack's 4 M pooling windows hold a handful of ties at fp32 resolution,
    # and ONE re-routed element of a 4 M-element layer moves that layer's parameter gradient by 5e-4 of its norm; the
    # reference's own fp32 gradients sit 1.4e-2 from an fp64 evaluation for the same reason - both measured)
    relus, pools = SU.relu_patterns(g), SU.pool_offsets(g)
    conv_bf16 = prec == "bf16" and bool(pools) and conv_bf16_on()
    if conv_bf16:
        # bf16 convolutions: the conv outputs sit ~1e-2 from the reference's, so the reference's arg-max positions / ReLU
        # patterns are no longer (near-)decisions of THIS run and forcing them would change VALUES by that noise: the
        # engine and the bf16-operand model both take their own (they run the same function)
        relus, pools = [], []
    decisions = F_amd.set_forced_decisions([pt for _, pt in relus] if relus else None,
                                           [(pool, off) for _, pool, off in pools] if pools else None)
    inp = g.t("inp").cuda()
    Tm, Bm = (m["T"], m["B"]) if m["seq"] else (m["B"], 1)
    try:
        outs = U.forward_model(m["fea_dict"], m["lab_dict"], m["arch_dict"], m["model"], nns, costs, inp, iod, Tm, Bm,
                               "train", [])
        outs["loss_final"].backward()
        torch.cuda.synchronize()
    finally:
        F_amd.set_forced_dropout(None)
        F_amd.set_forced_decisions(None, None)
        for net, orig in restore:
            net.forward = orig
    grads_of = lambda name: [(k, q.grad) for k, q in nns[name].named_parameters()]  # noqa: E731
    flips = {}
    for kind, diff, total in decisions:
        d0, t0 = flips.get(kind, (0, 0))
        flips[kind] = (d0 + diff, t0 + total)
    if prec == "fp32":
        worst = SU.check_fp32(g, outs, grads_of, tol=1e-4, tol_grad=1e-4)
        print("\n%s [fp32]: worst gradient row-sample error %.2e (%s); decisions that differed from the reference's "
              "(differing, of): %s" % (case, worst[0], worst[1], flips))
        for kind, (diff, total) in flips.items():  # the engine's own decisions differ only where values tie at fp32 resolution
            assert diff <= 2e-5 * total + 2, (kind, diff, total)
        return
    oouts = SU.oracle_run(O, g, init, emulate=True, forced=not conv_bf16, force_pool=not conv_bf16, conv_bf16=conv_bf16)
    st = m["strides"]
    # With bf16 convolutions the model's OWN noise floor on this network is above the engine-vs-model limits: a 1e-6
    # relative perturbation of the waveform (what another fp32 summation order amounts to) moves the trunk's output
    # by 8e-3 (bf16 rounding flips through four conv + LayerNorm layers and the MLP; measured on the host).  The limits
    # are therefore taken as max(fixed limit, 2 x that floor), per tensor, from a second model run.
    noisy_out, noisy = {}, None
    if conv_bf16:
        noisy = SU.oracle_params(nns)
        nouts = SU.oracle_run(O, g, noisy, emulate=True, forced=False, force_pool=False, conv_bf16=conv_bf16, inp_noise=1e-6)
        noisy_out = {k: nouts[k].detach() for k in m["out_keys"]}
    rep = {}
    for k in m["out_keys"]:
        s_ = st["out/%s/stride" % k]
        mrows = SU.rows(oouts[k].detach().reshape(-1, oouts[k].shape[-1]), s_)
        tight = TIGHT_OUT
        if conv_bf16:
            tight = max(tight, 2.0 * rel_err(SU.rows(noisy_out[k].reshape(-1, noisy_out[k].shape[-1]), s_), mrows))
        rep[k] = _two_step(k, SU.rows(outs[k].reshape(-1, outs[k].shape[-1]), s_), mrows, g.t("out/%s/rows" % k), tight)
    lref, lo = float(g.t("loss_final")), float(oouts["loss_final"].detach())
    assert abs(float(outs["loss_final"]) - lo) < TIGHT_OUT * lref
    assert abs(float(outs["loss_final"]) - lref) < model_bound(abs(lo - lref) / lref) * lref
    gtot = SU.grad_total(g)
    worst = (0.0, 0.0, 0.0)
    for name, k, key, gr, ref_rows, ref_ck in SU.grad_items(g, grads_of):
        s_ = st[key + "/stride"]
        frac = float(ref_rows.double().norm()) / ref_ck[0]
        tight = TIGHT_GRAD
        if conv_bf16:
            # (A) on WHOLE tensors.  Round 5 (tools/diag_conv_bf16_floor.py, profiles/r05_conv_bf16_floor.json): the miss of
            # rounds 3 / 4 ("MLP_layers wx.4.weight 3.2e-2 against 2e-2") was the limit - the floor was estimated on the
            # fixture's handful of sampled rows (0.8e-2 for that tensor) while the model's distance from itself over the
            # whole tensor is 8e-2, seed after seed (the flips of a run concentrate in few rows of a weight gradient)
            whole = float(init[name][k].grad.double().norm())
            tight = max(tight, 2.0 * grad_err(noisy[name][k].grad, init[name][k].grad, whole))
            e_whole = grad_err(gr, init[name][k].grad, whole)
            assert e_whole < tight, ((name, k), "engine vs bf16-operand model, whole tensor", e_whole, "limit", tight)
            tight = float("inf")  # (the row sample below keeps step (B): engine vs reference within the model's deviation)
        worst = max(worst, _two_step((name, k), SU.rows(gr, s_), SU.rows(init[name][k].grad, s_), ref_rows, tight,
                                     gtot * frac), key=lambda t: t[2])
    print("\n%s [bf16]: outputs (engine-vs-model, model-vs-ref, engine-vs-ref) %s; worst gradient %s; decisions that bf16 "
          "rounding changed (differing, of): %s"
          % (case, {k: tuple("%.1e" % x for x in v) for k, v in rep.items()}, tuple("%.1e" % x for x in worst), flips))
