#!/usr/bin/env python
"""Forward + backward time of one recurrent layer stack with `*_use_laynorm=True` at the BASELINE geometry (B = 128
bidirectional, H = 550): per-step LayerNorm inside the persistent time loop against the step-wise algorithm
(the semantics of PK_EXPERIMENT rec_ln_persist=0, selected with set_rec_algo).  Prints one JSON object."""
import importlib
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F_ = importlib.import_module("pytorch-kaldi_amd.functional")
nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")


def opts(pre, act, n):
    j = lambda v: ",".join([str(v)] * n)  # noqa: E731
    return {pre + "_lay": j(550), pre + "_drop": j(0.2), pre + "_use_laynorm_inp": "False", pre + "_use_batchnorm_inp": "False",
            pre + "_use_laynorm": j(True), pre + "_use_batchnorm": j(False), pre + "_bidir": "True", pre + "_act": j(act),
            pre + "_orthinit": "True", "use_cuda": "True", "to_do": "train"}


def run(kind, pre, act, prec, algo, T=200, B=128, D=440, layers=2, reps=5):
    F_.set_precision(prec)
    F_.set_rec_algo(algo)
    torch.manual_seed(1)
    net = getattr(nn_amd, kind)(opts(pre, act, layers), D).cuda().train()
    x = torch.randn(T, B, D, device="cuda")
    ts = []
    for i in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net.zero_grad(set_to_none=True)
        net(x).square().mean().backward()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return round(sorted(ts[2:])[len(ts[2:]) // 2], 3)


out = {"geometry": "T=200, B=128 bidirectional (256 rows), H=550, 2 layers, fwd+bwd, median of 5", "ms": {}}
for kind, pre, act, precs in (("liGRU", "ligru", "relu", ("bf16", "fp32")), ("LSTM", "lstm", "tanh", ("bf16",)),
                              ("GRU", "gru", "tanh", ("bf16",))):
    for prec in precs:
        for algo in ("persistent", "stepwise"):
            out["ms"]["%s %s %s" % (kind, prec, algo)] = run(kind, pre, act, prec, algo)
print(json.dumps(out))
