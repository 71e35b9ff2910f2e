#!/usr/bin/env python
"""Micro-benchmark of pk_gemm_bf16 on the BASELINE shapes (T*B = 64000 rows, H = 550):
forward projection (NT), dX (A k-contiguous, B k-major), dW / dU (both k-major, split-K)."""
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
F_ = importlib.import_module("pytorch-kaldi_amd.functional")

TB = 64000
SHAPES = [
    # name, M, N, K, a_kc, b_kc
    ("fwd proj  x.W^T", TB, 1100, 1104, 1, 1),
    ("head 1938 x.W^T", TB, 1938, 1100, 1, 1),
    ("dX        dP.W ", TB, 1100, 1100, 1, 0),
    ("dW     dP^T.x  ", 1100, 1104, TB, 0, 0),
    ("dU     dG^T.h  ", 550, 550, TB - 128, 0, 0),
    ("dWhead dy^T.x  ", 1938, 1100, TB, 0, 0),
]


if os.environ.get("SHAPES"):  # "M:N:K:a_kc:b_kc;..." replaces the list above
    SHAPES = [("%sx%sx%s %s%s" % tuple(f.split(":")),) + tuple(int(v) for v in f.split(":"))
              for f in os.environ["SHAPES"].split(";")]


def up(n, m):
    return (n + m - 1) // m * m


_lib = importlib.import_module("pytorch-kaldi_amd._lib")
TILES = [int(t) for t in os.environ.get("TILES", "128,256").split(",")]
REPS = int(os.environ.get("REPS", "10"))
for tile, (name, M, N, K, akc, bkc) in [(t, sh) for sh in SHAPES for t in TILES]:
    _lib.load().pk_gemm_bf16_set_tile(tile)
    A = torch.randn((M, up(K, 64)) if akc else (K, up(M, 64)), device="cuda").to(torch.bfloat16)
    B = torch.randn((N, up(K, 64)) if bkc else (K, up(N, 64)), device="cuda").to(torch.bfloat16)
    ldc = up(N, int(os.environ.get("LDC_ALIGN", "1")))  # row pitch of the fp32 output (floats)
    C = torch.empty(M, ldc, device="cuda")
    sk = F_._splitk_bf(F_._tiles_bf(M, N), K) if not akc else 1

    def run():
        F_.gemm_bf16(M, N, K, A, A.shape[1], akc, B, B.shape[1], bkc, C, ldc, splitk=sk)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = REPS
    for _ in range(n):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print("%s  tile %3d  ldc %5d  M=%6d N=%5d K=%6d splitk=%2d  %.3f ms  %.0f TFLOP/s" % (name, tile, ldc, M, N, K, sk, ms, 2.0 * M * N * K / ms / 1e9))
