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


def _advance_mirror(n, keep):
    """What pk_mt19937_bernoulli does to the mirror, by the oracle's numpy engine (the mirror lives on the CPU here)."""
    R = F_._RefRng
    m, st = O.mt19937_bernoulli_np(R.dev.numpy().view(np.uint32), n, keep)
    R.dev.copy_(torch.from_numpy(st.view(np.int32).copy()))
    R.ahead = True
    return m


def _reset_mirror():
    R = F_._RefRng
    R.dev, R.base, R.ahead = None, None, False


def test_mirror_bookkeeping_adopt_advance_write_back():
    """functional._RefRng without a GPU (the mirror as a CPU tensor, advanced by the oracle's engine instead of the HIP
    kernel): adopt() uploads once and then leaves an untouched generator alone, sync_back() leaves torch's generator where
    the reference's own draws would, a foreign draw on the CPU generator makes the next adopt() start from the CPU state."""
    R = F_._RefRng
    _reset_mirror()
    try:
        torch.manual_seed(77)
        torch.rand(5)
        start = torch.get_rng_state().clone()
        R.adopt("cpu")
        first = R.dev
        got = [_advance_mirror(300, np.float32(0.8)), _advance_mirror(1000, np.float32(0.5))]
        R.adopt("cpu")  # the CPU generator has not moved: the mirror (which is ahead) stays
        assert R.dev is first and R.ahead
        R.sync_back()
        assert not R.ahead
        mine = torch.get_rng_state().clone()
        torch.set_rng_state(start)
        want = [torch.bernoulli(torch.Tensor(300).fill_(0.8)), torch.bernoulli(torch.Tensor(1000).fill_(0.5))]
        assert torch.equal(torch.get_rng_state(), mine)
        for g_, w in zip(got, want):
            assert np.array_equal(g_, w.numpy())
        # a foreign draw while the mirror is ahead: the CPU state wins, the mirror starts again from it
        R.adopt("cpu")
        _advance_mirror(10, np.float32(0.5))
        torch.rand(3)
        now = torch.get_rng_state().clone()
        R.adopt("cpu")
        assert not R.ahead and np.array_equal(R.dev.numpy().view(np.uint32), R._parse(now))
        R.sync_back()  # nothing to write back
        assert torch.equal(torch.get_rng_state(), now)
    finally:
        _reset_mirror()
