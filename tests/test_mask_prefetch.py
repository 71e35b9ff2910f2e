"""PK_MASK_RNG=reference with the draws off the critical path (pytorch-kaldi_amd/nn.py::_MaskPrefetcher): the helper thread
draws the NEXT forward call's recurrent drop masks with the reference's own call (neural_networks.py:1102-1107:
torch.bernoulli(torch.Tensor(rows, H).fill_(1 - p)) on the global CPU generator) - the masks handed out, and the
generator state left behind, must be exactly what drawing them on the spot gives, whatever the shapes do."""
import importlib

import torch

nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")


def _direct(calls):
    return [[torch.bernoulli(torch.Tensor(r, h).fill_(1 - p)) for (r, h, p) in sig] for sig in calls]


def _through(pf, calls):
    return [[pf.get(i, len(sig), *s_) for i, s_ in enumerate(sig)] for sig in calls]


FULL = [(8, 5, 0.2), (8, 7, 0.2), (8, 7, 0.1)]
CALLS = [FULL] * 3 + [[(4, 5, 0.2), (4, 7, 0.2), (4, 7, 0.1)]] + [FULL] * 2 + [[(8, 5, 0.2), (6, 7, 0.2), (8, 7, 0.1)], FULL]


def test_prefetched_masks_are_the_reference_stream():
    torch.manual_seed(5)
    ref = _direct(CALLS)
    tail_ref = torch.rand(3)
    torch.manual_seed(5)
    pf = nn_amd._MaskPrefetcher()
    got = _through(pf, CALLS)
    nn_amd.drain_mask_prefetch()  # the masks drawn ahead for a call that never comes go back into the generator
    tail = torch.rand(3)
    for ca, cb in zip(ref, got):
        for a, b in zip(ca, cb):
            assert torch.equal(a, b)
    assert torch.equal(tail, tail_ref)


def test_reseeding_between_calls_wins_over_the_draws_made_ahead():
    torch.manual_seed(7)
    ref = _direct(CALLS[:2])
    torch.manual_seed(9)
    pf = nn_amd._MaskPrefetcher()
    _through(pf, CALLS[:1])        # leaves a helper thread drawing the next call's masks from seed 9's stream
    nn_amd.drain_mask_prefetch()   # what core.run_nn_dp does in front of torch.manual_seed
    torch.manual_seed(7)
    pf = nn_amd._MaskPrefetcher()
    got = _through(pf, CALLS[:2])
    nn_amd.drain_mask_prefetch()
    for ca, cb in zip(ref, got):
        for a, b in zip(ca, cb):
            assert torch.equal(a, b)
