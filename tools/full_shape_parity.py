#!/usr/bin/env python
"""ONE training step of the headline network at the metric's FULL shape (Li-GRU 5 x 550 bidirectional + 1938 / 48 heads,
T = 500, B = 128: 16 clusters x 500 steps per launch), value for value: the engine's bf16 mode against the oracle's
bf16-operand model (oracle/pk_oracle.py: the reference algorithm with the operands of every GEMM rounded to bf16 - itself
pinned to the reference at 2e-6 / 5e-5 by tests/test_oracle_golden.py) run on the GPU box's host cores from the same
seed-derived parameters, batch and drop masks; the engine differentiates the model run's own ReLU kink pattern
(functional.set_forced_kinks - DESIGN.md section 2).  Minutes of CPU time: run once per round
(tools/gpu_evidence.sh; since round 5 also tests/test_gpu_full_shape.py, i.e. the driver's `pytest -m gpu`).

    python tools/full_shape_parity.py [--T 500 --B 128] --out gpurun_out/x/r04_full_shape_parity.json

TEST INFRASTRUCTURE (imports oracle/): not part of the product path."""
import argparse
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

ap = argparse.ArgumentParser()
ap.add_argument("--T", type=int, default=500)
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--out", default=None)
ap.add_argument("--precision", default="bf16", choices=("bf16", "fp32"),
                help="bf16: the timed mode against the oracle's bf16-operand model; fp32: the parity mode against the exact "
                     "fp32 oracle at north_star's 1e-4 (round-5 review, row n1)")
args = ap.parse_args()
FP32 = args.precision == "fp32"

import pk_oracle as O  # noqa: E402
from golden_util import Golden, rel_err  # noqa: E402
import test_gpu_reference_pins as P  # noqa: E402  (the recipe plumbing of the config-scale golden test)

F_amd = importlib.import_module("pytorch-kaldi_amd.functional")
g = Golden("scale_ligru_T500")  # (only its meta: the UNSCALED shipped recipe; batch, masks and kinks are made here)
m = g.meta
T, B, H, L, nfea = args.T, args.B, m["H"], m["n_lay"], m["nfea"]
F_amd.set_precision(args.precision)
torch.manual_seed(m["seed"])
U, cfg, fea_dict, lab_dict, arch_dict, iod, nns, costs = P._recipe_engine(m, None)
init = {n: {k: v.detach().cpu().clone() for k, v in net.state_dict().items()} for n, net in nns.items()}
gen = torch.Generator().manual_seed(20260922)
inp = torch.randn(T, B, nfea + 2, generator=gen)
inp[:, :, nfea] = torch.randint(0, 1938, (T, B), generator=gen).float()
inp[:, :, nfea + 1] = torch.randint(0, 48, (T, B), generator=gen).float()
opts1 = m["options"]["architecture1"]
masks = O.make_drop_masks("liGRU", opts1, B, "train", generator=gen)

# ---- the bf16-operand model on the host (the slow part)
# (16 threads: with one thread per logical core of a 256-thread host the step's many small ops spend their time in the
# thread pool - the first attempt of round 4 did not finish in 15 minutes; 8 cores of the build container take 73 s)
cores = min(len(os.sched_getaffinity(0)), 16)
torch.set_num_threads(cores)
osd = {n: {k: v.clone().requires_grad_(v.is_floating_point() and "running" not in k) for k, v in init[n].items()} for n in init}
t0 = time.time()
log = []
with (torch.enable_grad() if FP32 else O.bf16_operands()):
    o1 = O.recurrent_forward("liGRU", opts1, osd["liGRU_layers"], inp[:, :, :nfea], drop_masks=masks, kink_log=log)
    oloss, oerr, o2, o3 = O.two_head_loss(o1, osd["MLP_layers"], m["options"]["architecture2"], osd["MLP_layers2"],
                                          m["options"]["architecture3"], inp[:, :, nfea].reshape(-1).long(),
                                          inp[:, :, nfea + 1].reshape(-1).long())
oloss.backward()
cpu_s = time.time() - t0
print("model step on %d host cores: %.0f s, loss %.6f" % (cores, cpu_s, float(oloss)), flush=True)

# ---- the engine, on the model run's kink pattern
for net in nns.values():
    net.cuda()
report = F_amd.set_forced_kinks(log)
rec = nns["liGRU_layers"]
orig, rec.forward = P._with_masks(rec, [mk.cuda() for mk in masks])
try:
    outs = U.forward_model(fea_dict, lab_dict, arch_dict, m["model"], nns, costs, inp.cuda(), iod, T, B, "train", [])
    outs["loss_final"].backward()
    torch.cuda.synchronize()
finally:
    rec.forward = orig
    F_amd.set_forced_kinks(None)
_lib = importlib.import_module("pytorch-kaldi_amd._lib")
_lib.raise_if_persist_failed()

res = {"what": ("engine fp32 (parity mode) step vs the exact fp32 oracle" if FP32 else
                "engine bf16 step vs the oracle's bf16-operand model") + ", full headline shape, kink-forced",
       "precision": args.precision, "T": T, "B": B, "H": H,
       "layers": L, "host_cores": cores, "model_step_seconds": round(cpu_s, 1),
       "loss_engine": float(outs["loss_final"]), "loss_model": float(oloss),
       "loss_rel_diff": abs(float(outs["loss_final"]) - float(oloss)) / abs(float(oloss)),
       "err_engine": float(outs["err_final"]), "err_model": float(oerr)}
stride = 997
for k, om in (("out_dnn1", o1), ("out_dnn2", o2), ("out_dnn3", o3)):
    a, b = outs[k].reshape(T * B, -1)[::stride].detach().cpu(), om.detach().reshape(T * B, -1)[::stride]
    res["out_rel_err/" + k] = rel_err(a, b)
    res["out_norm_rel_diff/" + k] = abs(float(outs[k].double().norm()) - float(om.double().norm())) / float(om.double().norm())
gtotal = float(torch.sqrt(sum((v.grad.double() ** 2).sum() for sd in osd.values() for v in sd.values() if v.requires_grad and v.grad is not None)))
worst = ("", 0.0)
gerr = {}
for name, net in nns.items():
    for k, p in net.named_parameters():
        ref = osd[name][k].grad
        if ref is None or p.grad is None or float(ref.norm()) < 1e-6 * gtotal:
            continue
        e = float((p.grad.detach().cpu().double() - ref.double()).norm()) / max(float(ref.double().norm()), 1e-3 * gtotal)
        gerr["%s/%s" % (name, k)] = e
        if e > worst[1]:
            worst = ("%s/%s" % (name, k), e)
res["grad_rel_err_worst"] = {"tensor": worst[0], "err": worst[1]}
# error growth over the stack: the input weights of the update gate, layer by layer (the worst family in round 4)
res["grad_rel_err_by_layer"] = {fam: [round(gerr.get("liGRU_layers/%s.%d.weight" % (fam, i), float("nan")), 6) for i in range(L)]
                                for fam in ("wz", "wh", "uz", "uh")}
res["grad_rel_err"] = {k: round(v, 6) for k, v in sorted(gerr.items(), key=lambda kv: -kv[1])[:12]}
res["kink_report_flipped_total_worst_a"] = [[int(a), int(b), float(c)] for a, b, c in report]
if FP32:
    LIM_OUT, LIM_GRAD = 1e-4, 1e-4
    res["limits"] = {"outputs": LIM_OUT, "gradients": LIM_GRAD, "loss": 1e-4,
                     "note": "north_star: posteriors, CE loss, gradients within 1e-4 relative fp32 (gradients kink-forced, SURVEY.md Appendix B 3b)"}
else:
    LIM_OUT, LIM_GRAD = 5e-3, 2e-2
    res["limits"] = {"outputs": LIM_OUT, "gradients": LIM_GRAD,
                     "note": "the limits of tests/test_gpu_reference_pins.py step (A): engine vs the bf16-operand model; "
                             "tests/test_gpu_full_shape.py additionally holds every family / layer to the model's own "
                             "distance from itself under fp32 rounding noise (tests/golden/bf16_model_floor_full_shape.json)"}
res["pass"] = bool(res["loss_rel_diff"] < LIM_OUT and all(res["out_rel_err/" + k] < LIM_OUT for k in ("out_dnn1", "out_dnn2", "out_dnn3"))
                   and worst[1] < LIM_GRAD)
if not FP32:
    res["worst_gradient_under_1p5e-2"] = bool(worst[1] < 1.5e-2)
print(json.dumps(res, indent=1))
if args.out:
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
