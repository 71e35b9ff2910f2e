#!/usr/bin/env python
"""Ties bench.py's `cpu_baseline` (the oracle, a torch-CPU PORT of the reference path) to the speed of the REFERENCE
itself.  Build container only (/root/reference does not travel to the GPU box).

Times one training step - forward_model, zero_grad, loss_final.backward(), optimizer.step(), i.e. what run_nn's
elapsed_time_chunk covers (core.py:567-701) - of the unscaled Li-GRU recipe (5 x 550 bidirectional + 1938 / 48 heads)
  (a) with the reference's own classes through its own utils.model_init / optimizer_init / forward_model, and
  (b) with oracle/pk_oracle.py on the same parameters, batch and drop masks,
at the bench sample (T=50, B=8) and at the metric's sequence length (T=500, B=8), and writes the ratio to
profiles/r02_cpu_port_vs_reference.json.  The port unbinds the projections once instead of indexing them inside the time
loop, so its autograd does not zero-fill a (T, 2B, H) tensor per step per gate: the port is FASTER than
the reference, increasingly so with T; with index_like_reference=True (what bench.py's cpu_baseline uses) the port
runs the reference's own indexing pattern and the two agree in speed.
"""
import configparser
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PK_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import torch  # noqa: E402

import pk_oracle as O  # noqa: E402
import utils as ref_utils  # noqa: E402  (the reference's)


def build():
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REF, "cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg"))
    cfg["exp"]["to_do"], cfg["exp"]["use_cuda"] = "train", "False"
    cfg["architecture2"]["dnn_lay"], cfg["architecture3"]["dnn_lay"] = "1938", "48"
    nfea = 40
    fea_dict = {"fmllr": ["fmllr", "lst", "opts", "0", "0", 0, nfea, nfea]}
    lab_dict = {"lab_cd": ["lab_cd", "f", "o", nfea], "lab_mono": ["lab_mono", "f", "o", nfea + 1]}
    arch_dict = {"liGRU_layers": ["architecture1", "liGRU_layers", True], "MLP_layers": ["architecture2", "MLP_layers", False],
                 "MLP_layers2": ["architecture3", "MLP_layers2", False]}
    model = cfg["model"]["model"].split("\n")
    iod = {"fmllr": fea_dict["fmllr"][5:]}
    torch.manual_seed(2234)
    nns, costs = ref_utils.model_init(iod, model, cfg, arch_dict, False, False, "train")
    opts = ref_utils.optimizer_init(nns, cfg, arch_dict)
    return cfg, fea_dict, lab_dict, arch_dict, model, iod, nns, costs, opts


def batch(T, B, seed, nfea=40):
    g = torch.Generator().manual_seed(seed)
    inp = torch.randn(T, B, nfea + 2, generator=g)
    inp[:, :, nfea] = torch.randint(0, 1938, (T, B), generator=g).float()
    inp[:, :, nfea + 1] = torch.randint(0, 48, (T, B), generator=g).float()
    return inp


def time_reference(T, B, steps):
    cfg, fea_dict, lab_dict, arch_dict, model, iod, nns, costs, opts = build()

    def step(i):
        outs = ref_utils.forward_model(fea_dict, lab_dict, arch_dict, model, nns, costs, batch(T, B, i), iod, T, B, "train", [])
        for o in opts.values():
            o.zero_grad()
        outs["loss_final"].backward()
        for o in opts.values():
            o.step()

    step(0)
    t0 = time.time()
    for i in range(steps):
        step(1 + i)
    return (time.time() - t0) / steps


def time_port(T, B, steps, like_ref=False):
    cfg, fea_dict, lab_dict, arch_dict, model, iod, nns, costs, _ = build()
    sds, opts = {}, []
    for n, net in nns.items():
        sd = {k: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in k)
              for k, v in net.state_dict().items()}
        sds[n] = sd
        opts.append(torch.optim.RMSprop([v for v in sd.values() if v.requires_grad], lr=4e-4, alpha=0.95, eps=1e-8))
    o1, o2, o3 = (dict(cfg["architecture%d" % i]) for i in (1, 2, 3))

    def step(i):
        inp = batch(T, B, i)
        out1 = O.recurrent_forward("liGRU", o1, sds["liGRU_layers"], inp[:, :, :40], index_like_reference=like_ref)
        loss, _, _, _ = O.two_head_loss(out1, sds["MLP_layers"], o2, sds["MLP_layers2"], o3,
                                        inp[:, :, 40].reshape(-1).long(), inp[:, :, 41].reshape(-1).long())
        for o in opts:
            o.zero_grad()
        loss.backward()
        for o in opts:
            o.step()

    step(0)
    t0 = time.time()
    for i in range(steps):
        step(1 + i)
    return (time.time() - t0) / steps


def main():
    cores = len(os.sched_getaffinity(0))
    torch.set_num_threads(cores)
    model = next((ln.split(":", 1)[1].strip() for ln in open("/proc/cpuinfo") if ln.startswith("model name")), "")
    out = {"host": model, "cores": cores, "torch": torch.__version__, "network": "liGRU 5x550 bidirectional + 1938/48 heads "
           "(cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg), fwd + bwd + RMSprop step, fp32", "points": []}
    for T, B, n_ref, n_port in ((50, 8, 5, 10), (500, 8, 1, 3)):
        tr_ = time_reference(T, B, n_ref)
        tp_ = time_port(T, B, n_port)
        tq_ = time_port(T, B, n_ref, like_ref=True)  # projections indexed inside the loop, as the reference does
        out["points"].append({"T": T, "B": B, "reference_s_per_step": round(tr_, 3), "port_s_per_step": round(tp_, 3),
                              "port_indexed_like_reference_s_per_step": round(tq_, 3),
                              "reference_frames_per_s": round(T * B / tr_, 1), "port_frames_per_s": round(T * B / tp_, 1),
                              "port_indexed_like_reference_frames_per_s": round(T * B / tq_, 1),
                              "port_over_reference": round(tr_ / tp_, 2),
                              "port_indexed_like_reference_over_reference": round(tr_ / tq_, 2)})
        print(out["points"][-1], flush=True)
    path = os.path.join(ROOT, "profiles", "r02_cpu_port_vs_reference.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
