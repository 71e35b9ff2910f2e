#!/usr/bin/env python
"""Micro-benchmark of the exact-fp32 GEMM (pk_gemm, PK_PREC_F32) on the shapes of the reference-precision mode:
the row-streaming products of a layer (T*B = 64000 rows, 2H = 1100), its weight gradient (split-K), and the per-step
products of the step-wise recurrences (256 rows).  PK_EXPERIMENT (f32_bk, gemm_f32_flat, f32_splitk_slots) selects the arm:
one process per arm.  SPLITS="a,b,.." overrides the split-K list tried for the per-step shapes."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F_ = importlib.import_module("pytorch-kaldi_amd.functional")

TB, H2 = 64000, 1100
REPS = int(os.environ.get("REPS", "8"))
SPLITS = [int(v) for v in os.environ.get("SPLITS", "1,3,6,8,12").split(",")]
dev = "cuda"


def run(name, M, N, K, a_kc, b_kc, splitk, reps):
    A = torch.randn((M, K) if a_kc else (K, M), device=dev)
    B = torch.randn((N, K) if b_kc else (K, N), device=dev)
    C = torch.empty(M, N, device=dev)
    a_rs, a_cs = (K, 1) if a_kc else (1, M)
    b_rs, b_cs = (1, K) if b_kc else (N, 1)
    for _ in range(2):
        F_.gemm(M, N, K, A, a_rs, a_cs, B, b_rs, b_cs, C, N, splitk=splitk, prec="fp32")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        F_.gemm(M, N, K, A, a_rs, a_cs, B, b_rs, b_cs, C, N, splitk=splitk, prec="fp32")
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    ref = (A if a_kc else A.t()).double() @ (B.t() if b_kc else B).double()
    err = float((C.double() - ref).abs().max() / ref.abs().max())
    print("%-28s %6d x %5d x %6d split %2d  %9.1f us  %6.1f TFLOP/s  err %.1e" % (
        name, M, N, K, splitk, us, 2.0 * M * N * K / us * 1e-6, err), flush=True)


print("arm:", os.environ.get("PK_EXPERIMENT", "(default)"))
run("projection x.W^T", TB, H2, H2, 1, 1, 1, REPS)
run("head 1938 x.W^T", TB, 1938, H2, 1, 1, 1, REPS)
run("dX dP.W", TB, H2, H2, 1, 0, 1, REPS)
run("dW dP^T.x", H2, H2, TB, 0, 0, F_._splitk(F_._tiles(H2, H2), TB), REPS)
run("dU dG^T.h", 550, 550, TB, 0, 0, F_._splitk(F_._tiles(550, 550), TB), REPS)
for sk in SPLITS:
    run("step U.h (GRU z,r + c)", 256, 1650, 550, 1, 1, sk, 200)
for sk in SPLITS:
    run("step U.h (z,r)", 256, 1100, 550, 1, 1, sk, 200)
for sk in SPLITS:
    run("step dh = dG.U", 256, 550, 1650, 1, 0, sk, 200)
