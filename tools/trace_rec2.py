#!/usr/bin/env python
"""Phase timing of the persistent bf16 recurrences (diagnostics): runs one Li-GRU layer fwd+bwd at
the BASELINE geometry with pk_persist2_set_trace and prints the mean shader-clock cycles per phase."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F_ = importlib.import_module("pytorch-kaldi_amd.functional")
_lib = importlib.import_module("pytorch-kaldi_amd._lib")
nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")

T, B, H = int(os.environ.get("T", 500)), int(os.environ.get("B", 128)), 550
kind = os.environ.get("KIND", "liGRU")
pre = {"liGRU": "ligru", "LSTM": "lstm", "RNN": "rnn"}[kind]
opts = {pre + "_lay": str(H), pre + "_drop": "0.2", pre + "_use_laynorm_inp": "False", pre + "_use_batchnorm_inp": "False",
        pre + "_use_laynorm": "False", pre + "_use_batchnorm": "True", pre + "_bidir": "True",
        pre + "_act": "relu" if kind != "LSTM" else "tanh", pre + "_orthinit": "True", "use_cuda": "True", "to_do": "train"}
F_.set_precision("bf16")
net = getattr(nn_amd, kind)(opts, 40).cuda().train()
x = torch.randn(T, B, 40, device="cuda", requires_grad=True)
lib = _lib.load()
lib.pk_persist2_set_mode(int(os.environ.get("SAFE", "0")))
lib.pk_persist2_set_poll_delay(int(os.environ.get("DELAY", "-1")))  # -1: the library defaults
EMPTY = int(os.environ.get("EMPTY", "0"))  # 1: steps without MFMA block / gate math = the hand-off floor of a step
lib.pk_persist2_set_empty_step(EMPTY)
names = ["poll", "prefetch-issue+barrier", "mfma", "gate math", "publish", "loop tail"]
prof = _lib.Profiler()
for rep in range(3):
    if rep == 2:
        prof.__enter__()  # HIP events around every C-ABI call of the last repetition
    tr_f = torch.zeros(T, 8, dtype=torch.int64, device="cuda")
    lib.pk_persist2_set_trace(tr_f.data_ptr())
    y = net(x)
    torch.cuda.synchronize()
    tr_b = torch.zeros(T, 8, dtype=torch.int64, device="cuda")
    lib.pk_persist2_set_trace(tr_b.data_ptr())
    y.sum().backward()
    torch.cuda.synchronize()
    lib.pk_persist2_set_trace(None)
prof.__exit__(None, None, None)
launch_ms = {k: v["avg_ms"] for k, v in prof.summary(1).items() if k.startswith("pk_rec")}
import json  # noqa: E402

summary = {}
for tag, tr in (("fwd", tr_f.cpu()), ("bwd", tr_b.cpu())):
    tr = tr[5:-5].double()
    d = tr[:, 1:6] - tr[:, 0:5]
    step = tr[1:, 0] - tr[:-1, 0]
    print("%s: cycles/step mean %.0f median %.0f p90 %.0f (s_memtime ticks)" % (tag, step.mean(), step.median(), step.quantile(0.9)))
    for i, n in enumerate(names[:5]):
        print("   %-24s mean %8.0f  median %8.0f" % (n, d[:, i].mean(), d[:, i].median()))
    print("   poll retries per step: mean %.2f  max %d" % (tr[:, 6].mean(), int(tr[:, 6].max())))
    if float(tr[:, 7].abs().sum()) > 0:  # role-split kernels: slot 7 = a helper wave's time stamp of the step
        lead = tr[:, 1] - tr[:, 7]       # barrier release seen by compute wave 0 minus the helper's stamp
        print("   helper stamp -> barrier release: mean %8.0f  median %8.0f  p10 %8.0f" % (lead.mean(), lead.median(), lead.quantile(0.1)))
        summary.setdefault(tag + "_helper_lead", {"mean": float(lead.mean()), "median": float(lead.median())})
    tail = tr[1:, 0] - tr[:-1, 5]
    print("   %-24s mean %8.0f  median %8.0f" % (names[5], tail.mean(), tail.median()))
    summary[tag] = {"cycles_per_step_mean": float(step.mean()), "cycles_per_step_median": float(step.median()),
                    "phases_mean": {n: float(d[:, i].mean()) for i, n in enumerate(names[:5])},
                    "poll_retries_per_step": float(tr[:, 6].mean())}
lib.pk_persist2_set_empty_step(0)
if os.environ.get("JSON_OUT"):
    with open(os.environ["JSON_OUT"], "w") as f:
        json.dump({"kind": kind, "T": T, "B": B, "H": H, "empty_step": EMPTY, "unit": "s_memtime ticks",
                   "launch_ms_hip_events": launch_ms,
                   "us_per_step_hip_events": {k: v * 1e3 / T for k, v in launch_ms.items()}, **summary}, f, indent=1)
print("launch ms (HIP events):", launch_ms)
