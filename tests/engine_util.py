"""Helpers for the GPU parity tests: build an engine module from a golden fixture."""
import importlib

import torch

from golden_util import Golden  # noqa: F401

nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
F_amd = importlib.import_module("pytorch-kaldi_amd.functional")
REC = ("liGRU", "LSTM", "GRU", "minimalGRU", "RNN")


def build_engine(meta, sd, device="cuda"):
    opts = dict(meta["options"])
    opts["use_cuda"] = "True"
    net = getattr(nn_amd, meta["arch_class"])(opts, meta["inp_dim"])
    missing = net.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    net.to(device)
    net.train() if meta["training"] else net.eval()
    return net


def run_engine(net, meta, x_cpu, masks, cot_cpu=None):
    x = x_cpu.detach().clone().cuda().requires_grad_(True)
    if meta["arch_class"] in REC:
        y = net(x, drop_masks=masks) if masks else net(x)
    else:
        y = net(x)
    grads = None
    if cot_cpu is not None:
        (y * cot_cpu.cuda()).sum().backward()
        grads = {k: (p.grad.detach().cpu() if p.grad is not None else None) for k, p in net.named_parameters()}
    torch.cuda.synchronize()
    return y.detach().cpu(), (x.grad.detach().cpu() if x.grad is not None else None), grads
