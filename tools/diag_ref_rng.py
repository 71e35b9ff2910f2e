#!/usr/bin/env python
"""Diagnosis of functional._RefRng (the device mirror of torch's CPU mt19937): where a state / mask mismatch sits."""
import importlib, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F_ = importlib.import_module("pytorch-kaldi_amd.functional")
dev = torch.device("cuda")
for pre in (0, 37):
    for shapes in ([(1, 100, 0.2)], [(1, 700, 0.2)], [(256, 550, 0.2)], [(256, 550, 0.2), (10, 7, 0.5), (1, 1, 0.1)]):
        torch.manual_seed(5)
        if pre:
            torch.rand(pre)
        ref = [torch.bernoulli(torch.Tensor(r, h).fill_(1 - p)) for r, h, p in shapes]
        after = torch.get_rng_state().clone()
        torch.manual_seed(5)
        if pre:
            torch.rand(pre)
        F_._RefRng.dev = None
        got = [F_.ref_rng_mask(r, h, p, dev) for r, h, p in shapes]
        torch.cuda.synchronize()
        mirror = F_._RefRng.dev.cpu().numpy().view(np.uint32)
        want = F_._RefRng._parse(after)
        F_._RefRng.sync_back()
        now = torch.get_rng_state()
        diff = (now != after).nonzero().flatten().tolist()
        print("pre", pre, shapes, "masks", [bool(torch.equal(g.cpu(), r)) for g, r in zip(got, ref)], "state words differ", int((mirror[:624] != want[:624]).sum()),
              "left/next", mirror[624:].tolist(), want[624:].tolist(), "bytes differing", diff[:6], len(diff))
