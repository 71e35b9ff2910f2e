"""Helpers shared by the parity tests: load a tests/golden/*.npz fixture."""
import glob
import json
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def list_cases(prefix=""):
    return sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, prefix + "*.npz")))


class Golden:
    def __init__(self, name):
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.name = name
        self.meta = json.loads(bytes(z["meta"]).decode())
        self.arrays = {k: z[k] for k in z.files if k != "meta"}

    def t(self, key, dtype=None):
        v = torch.from_numpy(np.array(self.arrays[key]))
        return v.to(dtype) if dtype is not None and v.is_floating_point() else v

    def group(self, prefix, dtype=None):
        n = len(prefix)
        return {k[n:]: self.t(k, dtype) for k in self.arrays if k.startswith(prefix)}

    def masks(self, dtype=None):
        return [self.t("mask/%d" % i, dtype) for i in range(self.meta.get("n_masks", 0))]


def rel_err(a, b):
    """Norm-relative error ||a-b|| / ||b|| (SURVEY.md Appendix B protocol)."""
    a = a.detach().double().cpu().reshape(-1)
    b = b.detach().double().cpu().reshape(-1)
    den = float(b.norm())
    if den == 0.0:
        return float((a - b).norm())
    return float((a - b).norm()) / den


def grad_err(got, ref, total_norm, floor=1e-3):
    """Gradient check of SURVEY.md Appendix B: norm-relative, except that tensors
    whose gradient is analytically zero (a bias feeding BatchNorm, a conv bias
    feeding max-pool -> LayerNorm) hold only rounding noise in the reference, so
    the denominator is floored at `floor` x the norm of the whole gradient."""
    a = got.detach().double().cpu().reshape(-1)
    b = ref.detach().double().cpu().reshape(-1)
    den = max(float(b.norm()), floor * float(total_norm))
    return float((a - b).norm()) / den


def analytically_zero(meta, name):
    """Parameters whose gradient is mathematically zero in the reference graph: a
    bias that feeds BatchNorm directly (MLP.wx.i.bias is always created,
    neural_networks.py:120) or through max-pool (CNN/SincNet conv.k.bias): the
    normalisation removes any per-channel constant.  The reference returns
    rounding noise there (SURVEY.md Appendix B note)."""
    o = {k.lower(): v for k, v in meta["options"].items()}
    cls = meta["arch_class"]
    parts = name.split(".")
    if cls == "MLP" and parts[0] == "wx" and parts[2] == "bias":
        i = int(parts[1])
        tb = lambda s: s.strip().lower() == "true"  # noqa: E731
        # LayerNorm is over features, so it does not remove a per-feature bias
        return tb(o["dnn_use_batchnorm"].split(",")[i]) and not tb(o["dnn_use_laynorm"].split(",")[i])
    if cls in ("CNN", "SincNet") and parts[0] == "conv" and parts[2] == "bias":
        pre = "cnn" if cls == "CNN" else "sinc"
        i = int(parts[1])
        tb = lambda s: s.strip().lower() == "true"  # noqa: E731
        return tb(o[pre + "_use_batchnorm"].split(",")[i]) or tb(o[pre + "_use_laynorm"].split(",")[i])
    return False


def check_grads(got_by_name, ref_by_name, meta, tol, zero_tol=1e-4):
    """Assert per-tensor gradient parity; returns the worst relative error."""
    import torch

    total = float(torch.sqrt(sum((v.double() ** 2).sum() for v in ref_by_name.values())))
    worst = 0.0
    for k, ref in ref_by_name.items():
        got = got_by_name[k]
        assert got is not None, "no gradient for " + k
        if meta is not None and analytically_zero(meta, k):
            assert float(got.double().norm()) <= zero_tol * total + 1e-12, (k, float(got.norm()), total)
            continue
        e = grad_err(got, ref, total)
        worst = max(worst, e)
        assert e < tol, (k, e)
    return worst


def conv_bf16_on():
    """Which of the engine's perf-mode convolutions take bf16 operands (pk_conv_bf16.hip): False = none (default),
    True = layers with at least 8 input channels (PK_CONV_BF16=1), "all" (PK_CONV_BF16=2).  The bf16-operand model of the oracle is
    switched the same way: O.bf16_operands(conv=conv_bf16_on())."""
    import importlib

    mode = importlib.import_module("pytorch-kaldi_amd.functional").conv_bf16_mode()
    return True if mode == "1" else ("all" if mode == "2" else False)
