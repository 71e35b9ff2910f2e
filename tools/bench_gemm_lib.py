#!/usr/bin/env python
"""Reference point for DESIGN.md 5.2 / 11.8: what the vendor library (torch.mm on bf16 -> hipBLASLt / rocBLAS) reaches on the
headline's GEMM shapes, next to pk_gemm_bf16.  NOT on the product path (the product never calls a library GEMM for these);
the library writes a bf16 result (half the output traffic of pk_gemm_bf16's fp32 result), so it is an upper reference.
    python tools/bench_gemm_lib.py [out.json]"""
import importlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F_ = importlib.import_module("pytorch-kaldi_amd.functional")
SHAPES = [("projection", 64000, 1100, 1104), ("head_fwd", 64000, 1938, 1104), ("dX", 64000, 1100, 1100), ("dW", 1100, 1104, 64000),
          ("dU_one_direction", 1100, 550, 64000)]
res = {}
for name, M, N, K in SHAPES:
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    b = torch.randn(K, N, device="cuda").to(torch.bfloat16)
    bt = b.t().contiguous()  # [N, K]: k-contiguous, the layout pk_gemm_bf16 takes for weights
    c = torch.empty(M, N, device="cuda")
    out = {}
    for label, fn in (("torch_mm_bf16_out", lambda: torch.mm(a, b)),
                      ("torch_mm_nt_bf16_out", lambda: torch.mm(a, bt.t())),
                      ("pk_gemm_bf16_fp32_out", lambda: F_.gemm_bf16(M, N, K, a, K, 1, bt, K, 1, c, N))):
        try:
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 20
            out[label] = {"ms": round(ms, 4), "tflops": round(2.0 * M * N * K / ms / 1e9, 1)}
        except Exception as e:  # noqa: BLE001
            out[label] = {"error": "%s: %s" % (type(e).__name__, e)}
    res[name] = {"M": M, "N": N, "K": K, **out}
    print(name, json.dumps(out), flush=True)
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
