#!/usr/bin/env python
"""Phase stamps of the fourth-generation fp32 recurrences (pk_rec_persist4_f32.hip): one layer at the BASELINE geometry,
mean shader clocks between the stamps of (workgroup 0, thread 0).  KIND=LSTM|GRU|minimalGRU, T, B from the environment."""
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F_ = importlib.import_module("pytorch-kaldi_amd.functional")
_lib = importlib.import_module("pytorch-kaldi_amd._lib")
nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
T, B = int(os.environ.get("T", 500)), int(os.environ.get("B", 128))
out = {}
for kind in os.environ.get("KIND", "LSTM,GRU").split(","):
    pre, act = {"LSTM": ("lstm", "tanh"), "GRU": ("gru", "tanh"), "minimalGRU": ("minimalgru", "relu")}[kind]
    opts = {pre + "_lay": "550", pre + "_drop": "0.2", pre + "_use_laynorm_inp": "False", pre + "_use_batchnorm_inp": "False",
            pre + "_use_laynorm": "False", pre + "_use_batchnorm": "True", pre + "_bidir": "True", pre + "_act": act,
            pre + "_orthinit": "True", "use_cuda": "True", "to_do": "train"}
    F_.set_precision("fp32")
    torch.manual_seed(1)
    net = getattr(nn_amd, kind)(opts, 40).cuda().train()
    x = torch.randn(T, B, 40, device="cuda", requires_grad=True)
    lib = _lib.load()
    for rep in range(2):
        tr_f = torch.zeros(T, 8, dtype=torch.int64, device="cuda")
        lib.pk_persist2_set_trace(tr_f.data_ptr())
        y = net(x)
        torch.cuda.synchronize()
        tr_b = torch.zeros(T, 8, dtype=torch.int64, device="cuda")
        lib.pk_persist2_set_trace(tr_b.data_ptr())
        y.sum().backward()
        torch.cuda.synchronize()
        lib.pk_persist2_set_trace(None)
    fw = ["poll", "flush+prefetch+mfma", "xsum write+barrier", "reduce+gate math (+phase 2)", "patches+publish"]
    bw = ["carry polls+mfma", "xsum+barrier", "reduce+phase A+publish da", "poll da+mfma", "xsum+barrier", "phase B math", "patches+publish"]
    for tag, tr, names in (("fwd", tr_f.cpu(), fw), ("bwd", tr_b.cpu(), bw)):
        tr = tr[5:-5].double()
        step = tr[1:, 0] - tr[:-1, 0]
        rec = {"cycles_per_step_mean": float(step.mean()), "median": float(step.median())}
        used = [i for i in range(8) if float(tr[:, i].abs().sum()) > 0]
        for a_, b_ in zip(used[:-1], used[1:]):
            rec["%d->%d %s" % (a_, b_, names[a_] if a_ < len(names) else "")] = round(float((tr[:, b_] - tr[:, a_]).mean()), 0)
        rec["%d->0 loop tail" % used[-1]] = round(float((tr[1:, 0] - tr[:-1, used[-1]]).mean()), 0)
        out["%s %s" % (kind, tag)] = rec
print(json.dumps(out, indent=1))
