"""PK_MASK_RNG=reference_host (the host-side form of the reference stream; PK_MASK_RNG=reference draws it on the device since round 5: tests/test_gpu_kernels.py::test_reference_mask_stream_on_the_device) with the draws off the critical path (pytorch-kaldi_amd/nn.py::_MaskPrefetcher): the helper thread
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


def test_two_modules_sharing_the_generator_follow_the_reference_stream():
    """Two recurrent modules in one model (cfg/TIMIT_baselines/TIMIT_rev/TIMIT_joint_training_liGRU_fbank.cfg): their
    forward calls alternate on the ONE CPU generator.  Whatever one of them drew ahead while it was alone is given back
    the moment the other asks for a mask; from then on both draw on the spot, in call order."""
    sig_a = [(8, 5, 0.2), (8, 7, 0.2)]
    sig_b = [(8, 9, 0.1), (8, 4, 0.1), (8, 4, 0.3)]
    order = [sig_a, sig_a, sig_b, sig_a, sig_b, sig_a, sig_b, sig_b, sig_a]   # A runs alone first (it draws ahead), then B joins
    for trial in range(3):
        torch.manual_seed(11 + trial)
        ref = _direct(order)
        tail_ref = torch.rand(3)
        torch.manual_seed(11 + trial)
        pa, pb = nn_amd._MaskPrefetcher(), None
        got = []
        for sig in order:
            if sig is sig_b and pb is None:
                pb = nn_amd._MaskPrefetcher()
            pf = pa if sig is sig_a else pb
            got.append([pf.get(i, len(sig), *s_) for i, s_ in enumerate(sig)])
        nn_amd.drain_mask_prefetch()
        tail = torch.rand(3)
        for ca, cb in zip(ref, got):
            for a, b in zip(ca, cb):
                assert torch.equal(a, b)
        assert torch.equal(tail, tail_ref)


def test_a_prefetcher_dies_with_its_module_and_its_survivor_draws_ahead_again():
    import gc

    torch.manual_seed(3)
    ref = _direct([FULL] * 4)
    torch.manual_seed(3)
    pa, pb = nn_amd._MaskPrefetcher(), nn_amd._MaskPrefetcher()
    got = [_through(pa, [FULL])[0]]
    del pb
    gc.collect()
    got += _through(pa, [FULL] * 3)
    assert pa._done is not None or pa._ahead is not None   # alone again: a set is being drawn ahead
    nn_amd.drain_mask_prefetch()
    for ca, cb in zip(ref, got):
        for a, b in zip(ca, cb):
            assert torch.equal(a, b)


def test_an_exception_in_the_helper_reaches_the_caller_and_the_helper_survives(monkeypatch):
    torch.manual_seed(2)
    pf = nn_amd._MaskPrefetcher()
    _through(pf, [FULL])
    pf._join()                       # the set drawn ahead is complete; now make the NEXT ahead-of-time draw fail
    calls = {"n": 0}
    real = nn_amd._MaskPrefetcher._draw

    def flaky(rows, H, p):
        calls["n"] += 1
        if calls["n"] == 2:
            raise RuntimeError("boom")
        return real(rows, H, p)

    monkeypatch.setattr(nn_amd._MaskPrefetcher, "_draw", staticmethod(flaky))
    _through(pf, [FULL])             # served from the good set; starts the failing one
    try:
        _through(pf, [FULL])
        raised = False
    except RuntimeError:
        raised = True
    assert raised
    monkeypatch.setattr(nn_amd._MaskPrefetcher, "_draw", staticmethod(real))
    _through(pf, [FULL] * 2)         # the helper thread is still there: no hang
    nn_amd.drain_mask_prefetch()
