"""Shared by the CPU (oracle) and GPU (engine) tests of the config-scale goldens tests/golden/scale_*.npz
(oracle/make_golden.py::recipe_scale_case: a SHIPPED cfg file, unscaled, run by the reference itself).

The fixtures do not store parameters: they are what torch.manual_seed(seed) + model_init gives (this package's classes
reproduce the reference's initialisation; per-tensor checksums are in the fixture), so every helper here starts from the
seed."""
import configparser
import importlib
import math

import numpy as np
import torch

from golden_util import Golden, grad_err, rel_err

RECIPE_CASES = ["scale_lstm_T500", "scale_gru_libri_T500", "scale_sincnet_3200", "scale_mlp_440"]


def ck(t, seed, k=4):
    """norm + k Rademacher projections (numpy RandomState: the same directions the generator used)."""
    v = t.detach().double().cpu().reshape(-1).numpy()
    rs = np.random.RandomState(seed)
    return np.concatenate(([float(np.linalg.norm(v))], [float(np.dot(v, rs.randint(0, 2, v.size) * 2.0 - 1.0))
                                                        for _ in range(k)]))


def rows(t, stride):
    t = t.detach().cpu()
    if t.dim() <= 1:
        return t.reshape(-1)[::stride]
    return t.reshape(t.shape[0], -1)[::stride]


def build(g, use_cuda):
    """Engine classes of the fixture's recipe through the model_init mirror with arch_library switched (the one-line
    change a user makes), initialised from the fixture's seed and checked against the reference's initialisation."""
    m = g.meta
    U = importlib.import_module("pytorch-kaldi_amd.utils")
    cfg = configparser.ConfigParser()
    cfg["exp"] = {"to_do": "train", "use_cuda": str(bool(use_cuda))}
    for sec, opts in m["options"].items():
        cfg[sec] = {k: v.replace("%", "%%") for k, v in opts.items()}
        cfg[sec]["arch_library"] = "pytorch-kaldi_amd.nn"
    fea = next(iter(m["fea_dict"]))
    iod = {fea: list(m["fea_dict"][fea][5:])}
    torch.manual_seed(m["seed"])
    nns, costs = U.model_init(iod, m["model"], cfg, m["arch_dict"], use_cuda, False, "train")
    for name, net in nns.items():
        for k, v in net.state_dict().items():
            if v.is_floating_point():
                ref = g.arrays["init_ck/%s/%s" % (name, k)]
                assert np.abs(ck(v, 7) - ref).max() <= 1e-5 * max(1.0, ref[0]), ("initialisation differs", name, k)
    return U, cfg, iod, nns, costs


def rec_masks(g):
    return [g.t("mask/%d" % i).float() for i in range(g.meta["n_masks"])]


def dropout_masks(g):
    """[(tag, 0/1 float tensor)] in the reference's call order."""
    out = []
    for i, (tag, shape) in enumerate(g.meta["dmasks"]):
        n = int(np.prod(shape))
        out.append((tag, torch.from_numpy(np.unpackbits(g.arrays["dmask/%d" % i])[:n].reshape(shape).astype(np.float32))))
    return out


def relu_patterns(g):
    """[(tag "<arch>/act.<i>", bool tensor)] - the reference run's ReLU patterns of the feed-forward stacks, call order."""
    out = []
    for i, (tag, shape) in enumerate(g.meta.get("relus", [])):
        n = int(np.prod(shape))
        out.append((tag, torch.from_numpy(np.unpackbits(g.arrays["relu/%d" % i])[:n].reshape(shape).astype(bool))))
    return out


def pool_offsets(g):
    """[(tag "<arch>/conv.<i>", pool length, uint8 offsets of the arg-max inside its pooling window)], call order."""
    return [(tag, pool, torch.from_numpy(np.array(g.arrays["pool/%d" % i]).reshape(shape)))
            for i, (tag, pool, shape) in enumerate(g.meta.get("pools", []))]


def oracle_params(nns):
    return {n: {k: v.detach().cpu().clone().requires_grad_(v.is_floating_point() and "running" not in k)
                for k, v in net.state_dict().items()} for n, net in nns.items()}


def oracle_run(O, g, sds, emulate=False, forced=False, force_pool=True, conv_bf16=False, inp_noise=0.0, noise_seed=99):
    """The oracle on the fixture's batch (CPU): outs dict, gradients left in sds.  forced: differentiate on the
    reference run's discrete decisions (ReLU patterns, pooling arg-max when force_pool) instead of this run's own."""
    import contextlib

    m = g.meta
    dm = {tag: mk for tag, mk in dropout_masks(g)}
    rp = pi = None
    if forced:
        rp = {tag: pat for tag, pat in relu_patterns(g)}
        pi = {}
        for tag, pool, off in (pool_offsets(g) if force_pool else []):
            pi[tag] = (torch.arange(off.shape[-1]) * pool)[None, None, :] + off.long()
    inp = g.t("inp")
    if inp_noise:  # the features (not the label columns) perturbed at fp32-rounding level: the model's own noise floor
        nfea = inp.shape[-1] - len(m["lab_dict"])
        inp = inp.clone()
        gen = torch.Generator().manual_seed(noise_seed)
        inp[..., :nfea] *= 1.0 + inp_noise * torch.randn(inp[..., :nfea].shape, generator=gen)
    with (O.bf16_operands(conv=conv_bf16) if emulate else contextlib.nullcontext()):
        outs = O.recipe_forward(m["model"], m["options"], m["arch_dict"], sds, inp, m["fea_dict"], m["lab_dict"],
                                rec_masks=rec_masks(g), drop_masks=dm, relu_patterns=rp or None, pool_idx=pi or None)
        outs["loss_final"].backward()
    return outs


def grad_total(g):
    return math.sqrt(sum(float(g.arrays[k][0]) ** 2 for k in g.arrays if k.startswith("grad/") and k.endswith("/ck")))


def grad_items(g, grads_of):
    """Yields (arch, key, fixture key, gradient tensor or None, reference row sample, reference checksum) for every
    parameter of every architecture; grads_of(arch) -> [(name, grad)]."""
    gtot = grad_total(g)
    for name in g.meta["arch_dict"]:
        for k, gr in grads_of(name):
            key = "grad/%s/%s" % (name, k)
            if key + "/rows" not in g.arrays:
                assert gr is None or float(gr.abs().max()) == 0.0, (name, k)
                continue
            ref_ck = g.arrays[key + "/ck"]
            if ref_ck[0] < 1e-6 * gtot:  # analytically-zero gradients (rounding noise in the reference)
                assert gr is None or float(gr.norm()) < 1e-4 * gtot, (name, k)
                continue
            yield name, k, key, gr, g.t(key + "/rows"), ref_ck


def check_fp32(g, outs, grads_of, tol=1e-4, tol_grad=1e-4):
    """Outputs / loss / gradients against the reference's run: row samples at `tol`, whole-tensor norm at `tol`, the
    +-1 projections at 4 tol (|<d, r>| ~ ||d|| for a random sign vector r).  -> worst gradient error"""
    m = g.meta
    st = m["strides"]
    for k in m["out_keys"]:
        o = outs[k].reshape(-1, outs[k].shape[-1])
        e = rel_err(rows(o, st["out/%s/stride" % k]), g.t("out/%s/rows" % k))
        assert e < tol, (k, e)
        c, ref = ck(o, 11), g.arrays["out/%s/ck" % k]
        assert abs(c[0] - ref[0]) < tol * ref[0], k
        assert np.abs(c[1:] - ref[1:]).max() < 4 * tol * ref[0], k
    lref = float(g.t("loss_final"))
    assert abs(float(outs["loss_final"].detach()) - lref) < tol * abs(lref)
    n_rows = outs[m["out_keys"][-1]].reshape(-1, outs[m["out_keys"][-1]].shape[-1]).shape[0]
    assert abs(float(outs["err_final"].detach()) - float(g.t("err_final"))) * n_rows < 0.5
    gtot = grad_total(g)
    worst = (0.0, ())
    for name, k, key, gr, ref_rows, ref_ck in grad_items(g, grads_of):
        got_rows = rows(gr, st[key + "/stride"])
        frac = float(ref_rows.double().norm()) / ref_ck[0]  # share of the tensor the sample holds
        e = grad_err(got_rows, ref_rows, gtot * frac)
        worst = max(worst, (e, (name, k)))
        assert e < tol_grad, (name, k, e)
        c = ck(gr, 13)
        assert abs(c[0] - ref_ck[0]) < tol_grad * max(ref_ck[0], 1e-3 * gtot), (name, k)
        assert np.abs(c[1:] - ref_ck[1:]).max() < 4 * tol_grad * max(ref_ck[0], 1e-3 * gtot), (name, k)
    return worst


__all__ = ["Golden", "RECIPE_CASES", "build", "check_fp32", "ck", "rows", "rec_masks", "dropout_masks", "relu_patterns",
           "pool_offsets", "oracle_params",
           "oracle_run", "grad_items", "grad_total"]
