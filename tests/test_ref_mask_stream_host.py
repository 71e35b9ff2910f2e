"""The reference's drop-mask stream at the level of torch's CPU generator (no GPU): the oracle's numpy restatement of
at::mt19937 + bernoulli_(Tensor p) is pinned against torch itself, and the product's state parser / writer
(functional._RefRng._parse / _build, what PK_MASK_RNG=reference uploads to and reads back from the device) against
torch.get_rng_state / set_rng_state."""
import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pk_oracle as O  # noqa: E402

F_ = importlib.import_module("pytorch-kaldi_amd.functional")


def _ref_masks(shapes):
    return [torch.bernoulli(torch.Tensor(r, h).fill_(1 - p)) for r, h, p in shapes]


@pytest.mark.parametrize("seed,pre", [(1234, 0), (2234, 37), (7, 623), (7, 624), (7, 625), (99, 5000)])
def test_numpy_engine_reproduces_torch_bernoulli_and_its_state(seed, pre):
    shapes = [(256, 550, 0.2), (10, 7, 0.5), (1, 1, 0.1), (3, 624, 0.25), (2, 1247, 0.9)]
    torch.manual_seed(seed)
    if pre:
        torch.rand(pre)  # (foreign draws: the stream is picked up wherever the generator stands)
    base = torch.get_rng_state()
    st = F_._RefRng._parse(base)
    assert st.dtype == np.uint32 and st.shape == (626,)
    want = _ref_masks(shapes)
    after = torch.get_rng_state()
    for (r, h, p), w in zip(shapes, want):
        keep = float(torch.tensor(1 - p, dtype=torch.float32))
        m, st = O.mt19937_bernoulli_np(st, r * h, keep)
        assert np.array_equal(m.reshape(r, h), w.numpy()), (r, h, p)
    rebuilt = F_._RefRng._build(base, st)
    assert torch.equal(rebuilt, after), "generator state after the draws differs from torch's own"
    # and the state written back continues the reference's stream on the host
    torch.set_rng_state(rebuilt)
    a = torch.bernoulli(torch.Tensor(5, 9).fill_(0.5))
    torch.set_rng_state(after)
    assert torch.equal(a, torch.bernoulli(torch.Tensor(5, 9).fill_(0.5)))


def test_state_layout_round_trip():
    torch.manual_seed(5)
    torch.rand(1000)
    s = torch.get_rng_state()
    assert s.numel() == 5056, "torch's CPU generator state is not the mt19937 layout the device mirror parses"
    assert torch.equal(F_._RefRng._build(s, F_._RefRng._parse(s)), s)
    # a freshly seeded engine: word 0 is the seed, left = 1, next = 0 (at::mt19937::init_with_uint32)
    torch.manual_seed(4357)
    w = F_._RefRng._parse(torch.get_rng_state())
    assert int(w[0]) == 4357 and int(w[624]) == 1 and int(w[625]) == 0
