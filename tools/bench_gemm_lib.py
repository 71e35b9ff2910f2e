#!/usr/bin/env python
"""Reference point for DESIGN.md 5.2 / 11.8 / 12: what the vendor library (torch.mm on bf16 -> hipBLASLt / rocBLAS) reaches on
the headline's GEMM shapes, next to pk_gemm_bf16 CALLED THE WAY THE PRODUCT CALLS IT (round-4 review: the first version
called pk_gemm_bf16 without the product's split-K and pitches and so compared unlike with unlike):
  * operand layouts of the training step (tools/bench_gemm.py's table): projections / heads k-contiguous x k-contiguous,
    dX k-contiguous x k-major, the weight gradients k-major x k-major with the library's own split-K;
  * operand pitches rounded up to 64 elements, the 1938-column output at a 128-byte row pitch.
NOT on the product path (the product never calls a library GEMM for these); the library writes a bf16 result (half the
output traffic of pk_gemm_bf16's fp32 result), so it is an upper reference.
    python tools/bench_gemm_lib.py [out.json]"""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F_ = importlib.import_module("pytorch-kaldi_amd.functional")
TB = 64000
# name, M, N, K, a_kc, b_kc (1 = k-contiguous operand rows, 0 = k-major: the reduction runs over the tensor's rows)
SHAPES = [("projection", TB, 1100, 1104, 1, 1), ("head_fwd", TB, 1938, 1100, 1, 1), ("dX", TB, 1100, 1100, 1, 0),
          ("dW", 1100, 1104, TB, 0, 0), ("dU_one_direction", 550, 550, TB - 128, 0, 0), ("head_dW", 1938, 1100, TB, 0, 0)]


def up(n, m):
    return (n + m - 1) // m * m


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


res = {}
for name, M, N, K, akc, bkc in SHAPES:
    # product-side operands (pitched), library-side operands (dense) over the same values
    A = torch.randn((M, up(K, 64)) if akc else (K, up(M, 64)), device="cuda").to(torch.bfloat16)
    B = torch.randn((N, up(K, 64)) if bkc else (K, up(N, 64)), device="cuda").to(torch.bfloat16)
    (A[:, K:] if akc else A[:, M:]).zero_()  # the product's operands are zero-padded up to their pitch
    (B[:, K:] if bkc else B[:, N:]).zero_()
    a_lib = (A[:, :K] if akc else A[:, :M].t()).contiguous()          # [M, K]
    b_lib = (B[:, :K].t() if bkc else B[:, :N]).contiguous()          # [K, N]
    ldc = up(N, 32)  # 128-byte row pitch of the fp32 output
    C = torch.empty(M, ldc, device="cuda")
    sk = F_._splitk_bf(F_._tiles_bf(M, N), K) if not akc else 1
    out = {"splitk": sk}
    for label, fn in (("torch_mm_bf16_out", lambda: torch.mm(a_lib, b_lib)),
                      ("pk_gemm_bf16_fp32_out", lambda: F_.gemm_bf16(M, N, K, A, A.shape[1], akc, B, B.shape[1], bkc, C, ldc, splitk=sk))):
        try:
            ms = timed(fn)
            out[label] = {"ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}
        except Exception as e:  # noqa: BLE001
            out[label] = {"error": "%s: %s" % (type(e).__name__, e)}
    try:  # same values: the two sides must agree
        ref = torch.mm(a_lib[:256].float() if akc else a_lib[:256].float(), b_lib.float())
        got = F_.gemm_bf16(M, N, K, A, A.shape[1], akc, B, B.shape[1], bkc, C, ldc, splitk=sk)[:256, :N]
        out["max_rel_diff_first_rows"] = float((got - ref).abs().max() / ref.abs().max())
    except Exception as e:  # noqa: BLE001
        out["check_error"] = "%s: %s" % (type(e).__name__, e)
    res[name] = {"M": M, "N": N, "K": K, "a_kc": akc, "b_kc": bkc, **out}
    print(name, json.dumps(out), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
