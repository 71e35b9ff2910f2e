#!/usr/bin/env python
"""Micro-benchmark of the conv1d+max_pool entry points on the SincNet / CNN layer shapes (batch 128):
forward, and backward (filter gradient + data gradient), reported against the packed-fp32 VALU peak.
`python tools/bench_conv.py [B]`; extra shapes via CONV_SHAPES="Cin,L,Cout,K,pool;..."."""
import ctypes
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
_lib = importlib.import_module("pytorch-kaldi_amd._lib")

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
SHAPES = [(1, 3200, 128, 129, 3), (128, 1024, 60, 5, 3), (60, 340, 60, 5, 3), (60, 112, 60, 3, 2),
          (1, 3200, 128, 65, 3), (1, 3200, 128, 33, 3)]
if os.environ.get("CONV_SHAPES"):
    SHAPES = [tuple(int(v) for v in s.split(",")) for s in os.environ["CONV_SHAPES"].split(";")]
PEAK = 256 * 4 * 32 * 2.4e9  # lane-FMA/s: 256 CUs x 4 SIMDs x 32 (v_pk_fma_f32) x 2.4 GHz

lib = _lib.load()
P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for Cin, L, Cout, K, pool in SHAPES:
    Lc = L - K + 1
    Lp = Lc // pool
    x = torch.randn(B, Cin, L, device="cuda")
    w = torch.randn(Cout, Cin, K, device="cuda") / (Cin * K) ** 0.5
    b = torch.randn(Cout, device="cuda")
    y = torch.empty(B, Cout, Lp, device="cuda")
    arg = torch.empty(B, Cout, Lp, device="cuda", dtype=torch.int32)
    work = torch.empty(int(lib.pk_conv_fwd_work_floats(Cin, Cout, K)), device="cuda")
    dy = torch.randn(B, Cout, Lp, device="cuda")
    dw, db, dx = torch.empty_like(w), torch.empty_like(b), torch.empty_like(x)
    part = torch.empty(int(lib.pk_conv_partial_floats(B, Cin, L, Cout, K, pool)), device="cuda")
    fwd = lambda: _lib.check(lib.pk_conv1d_pool_fwd(st, P(x), P(w), P(b), B, Cin, L, Cout, K, pool, P(y), P(arg), P(work)), "f")
    bwf = lambda: _lib.check(lib.pk_conv1d_pool_bwd(st, P(x), P(w), P(dy), P(arg), B, Cin, L, Cout, K, pool, P(dw), P(db), None,
                                                    P(part)), "b")
    bwa = lambda: _lib.check(lib.pk_conv1d_pool_bwd(st, P(x), P(w), P(dy), P(arg), B, Cin, L, Cout, K, pool, P(dw), P(db), P(dx),
                                                    P(part)), "b")
    tf, tw, ta = timed(fwd), timed(bwf), timed(bwa)
    macs = B * Lp * pool * Cout * Cin * K
    print(f"Cin={Cin:3d} L={L:4d} Cout={Cout:3d} K={K:3d} pool={pool}  fwd {tf:7.3f} ms ({macs / tf / 1e9 / (PEAK / 1e12) * 100:5.1f}% of "
          f"VALU peak)  filter-grad {tw:7.3f} ms ({macs / tw / 1e9 / (PEAK / 1e12) * 100:5.1f}%)  "
          f"data-grad {ta - tw:7.3f} ms ({B * L * Cout * Cin * K / (ta - tw) / 1e9 / (PEAK / 1e12) * 100:5.1f}%)", flush=True)
