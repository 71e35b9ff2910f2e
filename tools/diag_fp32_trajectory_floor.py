#!/usr/bin/env python
"""HOST ONLY: how far the 30-step CE-loss trajectory of tests/golden/train_ligru_30steps.npz moves under fp32 round-off
alone.  The exact-fp32 oracle (pinned to the reference at 2e-6 per step) is trained like the fixture (a) as it is, (b) with
the input batches scaled by (1 + 1e-7) - one fp32 ulp, what a different summation order does to a GEMM output - and
(c) with one MKL thread instead of all.  Written: per step, the largest relative distance of the (b) runs' loss from (a),
their frame-error flips, and the distance of the final parameters per tensor - the envelope tests/test_gpu_reference_pins.py::
test_ce_loss_trajectory holds the exact-fp32 engine to beyond the first steps (tests/golden/fp32_trajectory_floor.json).
    python tools/diag_fp32_trajectory_floor.py [out.json]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pk_oracle as O  # noqa: E402
from golden_util import Golden  # noqa: E402


def recipe(g):
    sds = {}
    for arch in ("liGRU_layers", "MLP_layers", "MLP_layers2"):
        sd = g.group("sd/%s/" % arch)
        for k in sd:
            if sd[k].is_floating_point() and "running" not in k:
                sd[k].requires_grad_(True)
        sds[arch] = sd
    return sds


def step(m, sds, inp, masks):
    nfea = m["nfea"]
    out1 = O.recurrent_forward("liGRU", m["options"]["architecture1"], sds["liGRU_layers"], inp[:, :, :nfea], drop_masks=masks)
    loss, err, _, _ = O.two_head_loss(out1, sds["MLP_layers"], m["options"]["architecture2"], sds["MLP_layers2"],
                                      m["options"]["architecture3"], inp[:, :, nfea].reshape(-1).long(),
                                      inp[:, :, nfea + 1].reshape(-1).long())
    return loss, err


def trajectory(g, scale=1.0, threads=None):
    if threads is not None:
        torch.set_num_threads(threads)
    m = g.meta
    sds = recipe(g)
    optims = []
    for arch, sec in (("liGRU_layers", "architecture1"), ("MLP_layers", "architecture2"), ("MLP_layers2", "architecture3")):
        o = m["options"][sec]
        optims.append(torch.optim.RMSprop([v for v in sds[arch].values() if v.requires_grad], lr=float(o["arch_lr"]),
                                          alpha=float(o["opt_alpha"]), eps=float(o["opt_eps"])))
    batches, masks, n_lay = g.t("batches").clone(), g.masks(), m["n_lay"]
    batches[..., :m["nfea"]] *= scale
    losses, errs = [], []
    for s in range(m["n_steps"]):
        loss, err = step(m, sds, batches[s % m["n_batches"]], masks[s * n_lay:(s + 1) * n_lay])
        for o in optims:
            o.zero_grad()
        loss.backward()
        for o in optims:
            o.step()
        losses.append(float(loss.detach()))
        errs.append(float(err))
    final = {"%s/%s" % (a, k): v.detach().clone() for a, sd in sds.items() for k, v in sd.items() if v.is_floating_point()}
    return np.array(losses), np.array(errs), final


g = Golden("train_ligru_30steps")
m = g.meta
ref = g.arrays["loss"].astype(np.float64)
n0 = torch.get_num_threads()
a, ea, fa = trajectory(g)
c, _, _ = trajectory(g, threads=1)
torch.set_num_threads(n0)
rel = lambda x, y: np.abs(x - y) / np.abs(y)
SCALES = [1.0 + 1e-7, 1.0 - 1e-7, 1.0 + 2e-7, 1.0 - 2e-7]
runs = [trajectory(g, scale=s_) for s_ in SCALES]
loss_d = np.max([rel(b, a) for b, _, _ in runs], axis=0)
flips = np.max([np.abs(eb - ea) * m["T"] * m["B"] for _, eb, _ in runs], axis=0)


def tdist(x, y):
    den = float(y.double().norm())
    return float((x.double() - y.double()).norm()) / den if den > 0 else 0.0


param_d = {k: max(tdist(fb[k], fa[k]) for _, _, fb in runs) for k in fa}
out = {"what": __doc__.split("\n\n")[0].replace("\n", " "), "fixture": "train_ligru_30steps", "input_scales": SCALES,
       "oracle_vs_reference_loss": rel(a, ref).tolist(), "one_thread_vs_oracle_loss": rel(c, a).tolist(),
       "loss_rel_envelope": loss_d.tolist(), "frame_error_flips_envelope": flips.tolist(), "final_param_rel_envelope": param_d}
print("oracle vs reference: max %.2e; one MKL thread vs all: max %.2e" % (rel(a, ref).max(), rel(c, a).max()))
print("one-ulp input scalings vs oracle, loss per step (max over %d runs):" % len(SCALES))
print(np.array2string(loss_d, precision=1, max_line_width=160))
print("frame-error flips per step:", flips.astype(int).tolist())
print("final parameters: worst tensor %.2e (%s), median %.2e" % (max(param_d.values()), max(param_d, key=param_d.get),
                                                                 float(np.median(list(param_d.values())))))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
