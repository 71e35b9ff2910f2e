#!/usr/bin/env python
"""Micro-benchmark of pk_bn_bwd_bf16 (BatchNorm backward from the bf16 gate gradients) at the BASELINE shape:
T*B = 64000 rows, 2 gates x 550 units, two directions.  Knobs of the library: PK_EXPERIMENT keys bnb_rbr / bnb_rba
(row blocks of the reduction / apply pass)."""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_lib = importlib.import_module("pytorch-kaldi_amd._lib")
lib = _lib.load()
TB, G, H = 64000, int(os.environ.get("G", "2")), 550
Hp = (H + 7) // 8 * 8
Gp = (G * Hp + 63) // 64 * 64
GH = G * H
dGb = (torch.randn(2 * TB, Gp, device="cuda") * 0.1).to(torch.bfloat16)
P = torch.randn(TB, GH, device="cuda")
mean, var = P.mean(0), P.var(0, unbiased=False)
gamma = torch.rand(GH, device="cuda") + 0.5
part = torch.empty(2048 * GH * 2, device="cuda")
sum_g, sum_gx = torch.empty(GH, device="cuda"), torch.empty(GH, device="cuda")
dPb = torch.empty(TB, (GH + 63) // 64 * 64, device="cuda", dtype=torch.bfloat16)
p = lambda t: ctypes.c_void_p(t.data_ptr())
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
g1 = ctypes.c_void_p(dGb.data_ptr() + 2 * TB * Gp)


def run():
    _lib.check(lib.pk_bn_bwd_bf16(st, p(dGb), g1, Gp, G, H, p(P), GH, TB, p(mean), p(var), 1e-5, p(gamma), float(TB), p(part),
                                  p(sum_g), p(sum_gx), p(dPb), dPb.shape[1], None, None), "pk_bn_bwd_bf16")


big = torch.empty(256 << 20, device="cuda", dtype=torch.float32)  # 1 GB: flushes the Infinity Cache between launches
for _ in range(2):
    run()
torch.cuda.synchronize()
ts = []
for _ in range(8):
    if os.environ.get("FLUSH", "1") == "1":
        big.zero_()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ts.sort()
bytes_ = 2 * (dGb.numel() * 2 + P.numel() * 4) + dPb.numel() * 2
print("PK_EXPERIMENT=%s  median %.3f ms  min %.3f ms  %.2f TB/s (2 passes over dGb + P, one bf16 write; %s)" % (
    os.environ.get("PK_EXPERIMENT", "-"),
    ts[len(ts) // 2], ts[0], bytes_ / ts[len(ts) // 2] / 1e9, "checksum %.6e" % float(dPb.float().sum())))
