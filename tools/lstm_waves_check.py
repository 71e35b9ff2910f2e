"""Eight-wave LSTM kernels (pk_rec_persist2_lstm.hip) against the four-wave ones on the same inputs, then timing.

    python tools/lstm_waves_check.py [--time]

Forward: per gate the MFMA accumulation order is the same in both kernels; they differ where an fp32 expression
contracts differently and a bf16 rounding of h_t flips (measured 1e-7 .. 2e-4 norm-relative, tolerance 1e-3).
Backward: the K split changes the order of the fp32 sum over the gates (measured <= 2.7e-3 on bf16-rounded operands,
tolerance 1e-2).  One JSON line per case; tests/test_gpu_lstm_waves.py holds the same cases.
"""
import argparse
import importlib
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
F_amd = importlib.import_module("pytorch-kaldi_amd.functional")
lib = importlib.import_module("pytorch-kaldi_amd._lib").load()


def rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def opts(lay, act, bidir):
    n = len(lay)
    j = lambda v: ",".join([str(v)] * n)  # noqa: E731
    return {"lstm_lay": ",".join(map(str, lay)), "lstm_drop": j(0.2), "lstm_use_laynorm_inp": "False",
            "lstm_use_batchnorm_inp": "False", "lstm_use_laynorm": j(False), "lstm_use_batchnorm": j(True),
            "lstm_bidir": str(bidir), "lstm_act": j(act), "lstm_orthinit": "True", "use_cuda": "True", "to_do": "train"}


def run(net, x, cot, masks):
    net.zero_grad()
    xe = x.clone().requires_grad_(True)
    y = net(xe, drop_masks=masks)
    (y * cot).sum().backward()
    torch.cuda.synchronize()
    return y.detach().clone(), xe.grad.clone(), {k: p.grad.detach().clone() for k, p in net.named_parameters() if p.grad is not None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    args = ap.parse_args()
    F_amd.set_precision("bf16")
    F_amd.set_rec_algo("persistent")
    worst, bad = 0.0, 0
    cases = [(550, 12, 5, True, "tanh", 0), (40, 9, 3, True, "tanh", 0), (20, 7, 4, False, "tanh", 1), (14, 5, 33, True, "relu", 0),
             (576, 3, 2, True, "tanh", 0), (24, 4, 300, True, "elu", 0), (550, 40, 128, True, "tanh", 0), (8, 1, 2, False, "tanh", 1)]
    for H, T, B, bidir, act, safe in cases:
        torch.manual_seed(21)
        net = nn_amd.LSTM(opts([H, H], act, bidir), 23).cuda().train()
        g = torch.Generator().manual_seed(13)
        x = torch.randn(T, B, 23, generator=g).cuda()
        cot = torch.randn(T, B, net.out_dim, generator=g).cuda()
        masks = [torch.bernoulli(torch.full((B * (2 if bidir else 1), H), 0.8), generator=g) for _ in range(2)]
        lib.pk_persist2_set_mode(safe)
        res = {}
        for w in (4, 8):
            lib.pk_persist2_set_lstm_waves(w)
            lib.pk_persist2_error_reset()
            res[w] = run(net, x, cot, masks)
            res[w] += (int(lib.pk_persist2_error_count()),)
        e_y, e_dx = rel(res[8][0], res[4][0]), rel(res[8][1], res[4][1])
        e_g = max(rel(res[8][2][k], v) for k, v in res[4][2].items())
        ok = e_y < 1e-3 and e_dx < 1e-2 and e_g < 1e-2 and res[8][3] == 0 and res[4][3] == 0
        worst = max(worst, e_y, e_dx, e_g)
        print(json.dumps({"H": H, "T": T, "B": B, "bidir": bidir, "act": act, "safe": safe, "y": e_y, "dx": e_dx, "grads": e_g,
                          "errors": [res[4][3], res[8][3]], "ok": ok}), flush=True)
        bad += 0 if ok else 1
    lib.pk_persist2_set_mode(0)
    if args.time:
        T, B, H = 500, 128, 550
        torch.manual_seed(3)
        net = nn_amd.LSTM(opts([H], "tanh", True), 40).cuda().train()
        x = torch.randn(T, B, 40, device="cuda")
        cot = torch.randn(T, B, net.out_dim, device="cuda")
        for w in (4, 8, 4, 8):
            lib.pk_persist2_set_lstm_waves(w)
            for _ in range(2):
                run(net, x, cot, None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4):
                net.zero_grad()
                y = net(x)
                (y * cot).sum().backward()
            e1.record()
            torch.cuda.synchronize()
            print(json.dumps({"waves": w, "layer_fwd_bwd_ms": e0.elapsed_time(e1) / 4, "T": T, "B": B, "H": H}), flush=True)
    print("OK worst" if bad == 0 else "MISMATCH in %d cases, worst" % bad, worst)
    sys.exit(0 if bad == 0 else 1)


if __name__ == "__main__":
    main()
