#!/usr/bin/env python
"""Does a weight-gradient GEMM on a second HIP stream run for free next to the persistent recurrent kernels?
Times the headline training step alone, a batch of dW/dU-shaped GEMMs alone, and both together."""
import argparse
import importlib
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

F_ = importlib.import_module("pytorch-kaldi_amd.functional")
args = argparse.Namespace(recipe="timit_ligru", T=500, B=128, prec="bf16", algo="auto", layers=None, mask_rng="device",
                          overlap=False, torch_optim=False)
tr = bench.Trainer(args, 0, 1)
for i in range(3):
    tr.step(i)
torch.cuda.synchronize()


def up(n, m):
    return (n + m - 1) // m * m


TB = 64000
shapes = [(1100, 1104, TB)] * 5 + [(550, 1100, TB - 128)] * 10 + [(1938, 1100, TB)]
ops = []
for M, N, K in shapes:
    A = torch.randn(K, up(M, 64), device="cuda").to(torch.bfloat16)
    Bm = torch.randn(K, up(N, 64), device="cuda").to(torch.bfloat16)
    C = torch.empty(M, N, device="cuda")
    ops.append((M, N, K, A, Bm, C, F_._splitk_bf(F_._tiles_bf(M, N), K)))
side = torch.cuda.Stream()


def gemms():
    with torch.cuda.stream(side):
        for M, N, K, A, Bm, C, sk in ops:
            F_.gemm_bf16(M, N, K, A, A.shape[1], 0, Bm, Bm.shape[1], 0, C, N, splitk=sk)


def wall(fn, n=5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


gemms()
torch.cuda.synchronize()
t_step = wall(lambda i: tr.step(i))
t_gemm = wall(lambda i: gemms())


def both(i):
    gemms()
    tr.step(i)


t_both = wall(both)
print("step alone %.2f ms   gemms alone %.2f ms   together %.2f ms   (serial sum %.2f)" % (t_step, t_gemm, t_both, t_step + t_gemm))
