#!/usr/bin/env python
"""Per-launch time of the exact-fp32 recurrences of ONE layer at the BASELINE geometry (T = 500, B = 128 bidirectional,
H = 550), and where a step's time goes: the same launches with the MFMAs and / or the polls switched off
(pk_persist2_set_empty_step: bit 0 / bit 1 - timing only).  Prints one JSON object.

    python tools/bench_rec4.py [--cells LSTM,GRU] [--T 500]"""
import argparse
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

ap = argparse.ArgumentParser()
ap.add_argument("--cells", default="LSTM,GRU,liGRU")
ap.add_argument("--T", type=int, default=500)
ap.add_argument("--B", type=int, default=128)
ap.add_argument("--flags", default="0,1,2,3")
args = ap.parse_args()
PRE = {"LSTM": ("lstm", "tanh"), "GRU": ("gru", "tanh"), "minimalGRU": ("minimalgru", "relu"), "liGRU": ("ligru", "relu")}


def opts(pre, act):
    return {pre + "_lay": "550", pre + "_drop": "0.2", pre + "_use_laynorm_inp": "False", pre + "_use_batchnorm_inp": "False",
            pre + "_use_laynorm": "False", pre + "_use_batchnorm": "True", pre + "_bidir": "True", pre + "_act": act,
            pre + "_orthinit": "True", "use_cuda": "True", "to_do": "train"}


F_.set_precision("fp32")
F_.set_rec_algo("auto")
lib = _lib.load()
out = {"geometry": "T=%d, B=%d bidirectional, H=550, one layer, exact fp32" % (args.T, args.B), "ms_per_launch_call": {}}
for kind in args.cells.split(","):
    pre, act = PRE[kind]
    torch.manual_seed(1)
    net = getattr(nn_amd, kind)(opts(pre, act), 40).cuda().train()
    x = torch.randn(args.T, args.B, 40, device="cuda")
    for flag in [int(v) for v in args.flags.split(",")]:
        lib.pk_persist2_set_empty_step(flag)
        prof = _lib.Profiler()
        for i in range(3):
            if i == 1:
                prof.__enter__()
            net.zero_grad(set_to_none=True)
            net(x).square().mean().backward()
        torch.cuda.synchronize()
        prof.__exit__(None, None, None)
        summ = prof.summary(2)
        out["ms_per_launch_call"]["%s flags=%d" % (kind, flag)] = {k: round(v["avg_ms"], 3) for k, v in summ.items() if k.startswith("pk_rec")}
        lib.pk_persist2_set_empty_step(0)
        _lib.load().pk_persist_error_reset()
        _lib.load().pk_persist2_error_reset()
print(json.dumps(out, indent=1))
