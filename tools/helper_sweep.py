#!/usr/bin/env python
"""L2 run-ahead helpers of the persistent recurrences (pk_rec_helper.hip): launch time of one layer's forward / backward
recurrence at the BASELINE geometry (T = 500, B = 128 bidirectional, H = 550) per helper mode, lead and helper count -
all in ONE process on ONE box, round-robin, HIP events around the C-ABI calls.  Also checks that the results do not
depend on the helpers (they only load): outputs and gradients bit-identical to mode 0.
    KIND=liGRU|LSTM|GRU python tools/helper_sweep.py [out.json]"""
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

T, B, H = int(os.environ.get("T", 500)), int(os.environ.get("B", 128)), 550
kind = os.environ.get("KIND", "liGRU")
pre = {"liGRU": "ligru", "LSTM": "lstm", "GRU": "gru"}[kind]
opts = {pre + "_lay": str(H), pre + "_drop": "0.2", pre + "_use_laynorm_inp": "False", pre + "_use_batchnorm_inp": "False",
        pre + "_use_laynorm": "False", pre + "_use_batchnorm": "True", pre + "_bidir": "True",
        pre + "_act": "relu" if kind == "liGRU" else "tanh", pre + "_orthinit": "True", "use_cuda": "True", "to_do": "train"}
F_.set_precision("bf16")
torch.manual_seed(5)
net = getattr(nn_amd, kind)(opts, 40).cuda().train()
x = torch.randn(T, B, 40, device="cuda", requires_grad=True)
masks = [(torch.rand(2 * B, H, device="cuda") > 0.2).float() / 0.8]
lib = _lib.load()


def pack(mode, lp=0, lo=0, lb=0, wgs=0, delay=0):
    """delay: units of 256 clocks between seeing a publish and touching"""
    return mode | (lp << 8) | (lo << 12) | (lb << 16) | (wgs << 20) | (delay << 24)


CONFIGS = [("off", pack(0)),
           ("P lead3 x4 d0", pack(1, 3, 1, 2, 4, 0)), ("P lead3 x4 d4", pack(1, 3, 1, 2, 4, 4)), ("P lead3 x4 d8", pack(1, 3, 1, 2, 4, 8)),
           ("P lead3 x4 d12", pack(1, 3, 1, 2, 4, 12)), ("P lead4 x2 d8", pack(1, 4, 1, 2, 2, 8)), ("P lead4 x6 d8", pack(1, 4, 1, 2, 6, 8)),
           ("P+out lead3/1 x4 d8", pack(3, 3, 1, 2, 4, 8)), ("P+out lead4/2 x4 d12", pack(3, 4, 2, 2, 4, 12)),
           ("bwd lead4 x4 d0", pack(4, 3, 1, 4, 4, 0)), ("bwd lead4 x4 d8", pack(4, 3, 1, 4, 4, 8)), ("bwd lead6 x6 d8", pack(4, 3, 1, 6, 6, 8)),
           ("all lead3/1/4 x4 d8", pack(7, 3, 1, 4, 4, 8))]
if os.environ.get("CONFIGS"):
    keep = set(os.environ["CONFIGS"].split(";"))
    CONFIGS = [c for c in CONFIGS if c[0] in keep or c[0] == "off"]


def run_once():
    x.grad = None
    net.zero_grad(set_to_none=True)
    orig = net.forward
    y = net(x)
    (y * y).sum().backward()
    return y.detach(), x.grad.detach().clone()


# masks: the layer draws its own with the device RNG - fix the generator so that every configuration sees the same step
def seeded():
    torch.manual_seed(11)
    torch.cuda.manual_seed(11)
    return run_once()


res = {}
ref = None
ROUNDS = int(os.environ.get("ROUNDS", 3))
for rnd in range(ROUNDS):
    for name, word in CONFIGS:
        lib.pk_rec_helper_set_mode(word)
        seeded()  # warm
        prof = _lib.Profiler()
        with prof:
            y, gx = seeded()
        s = prof.summary(1)
        torch.cuda.synchronize()
        if ref is None:
            ref = (y.clone(), gx.clone())
        same = bool(torch.equal(y, ref[0]) and torch.equal(gx, ref[1]))
        r = res.setdefault(name, {"fwd_ms": [], "bwd_ms": [], "bit_identical_to_off": True})
        for k, v in s.items():
            if "fwd" in k and k.startswith("pk_rec"):
                r["fwd_ms"].append(round(v["avg_ms"], 4))
            if "bwd" in k and k.startswith("pk_rec"):
                r["bwd_ms"].append(round(v["avg_ms"], 4))
        r["bit_identical_to_off"] = r["bit_identical_to_off"] and same
lib.pk_rec_helper_set_mode(-1)
_lib.raise_if_persist_failed()
for name, r in res.items():
    print("%-22s fwd %s  bwd %s  identical %s" % (name, r["fwd_ms"], r["bwd_ms"], r["bit_identical_to_off"]), flush=True)
if len(sys.argv) > 1:
    json.dump({"kind": kind, "T": T, "B": B, "H": H, "configs": res}, open(sys.argv[1], "w"), indent=1)
