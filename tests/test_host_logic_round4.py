"""Host-side logic added in round 4 (no GPU): the training-step context and what it unlocks, the private alias the fused
heads share, deferred side-stream launches' bookkeeping."""
import importlib

import torch

F_ = importlib.import_module("pytorch-kaldi_amd.functional")


def _flat_param(shape):
    q = torch.nn.Parameter(torch.zeros(*shape))
    q.grad = torch.zeros(*shape)
    q._pk_flat = True
    return q


def test_affine_views_only_inside_the_training_step_context(monkeypatch):
    """functional.direct_affine_ok (decided in FORWARD: a recurrent layer takes its BatchNorm scales / shifts as detached
    views of the flat buffer): only inside accumulating_backward, with autograd recording, flat-bucket parameters with a
    pre-allocated gradient, perf mode, no data-parallel listener, PK_EXPERIMENT direct_grads not 0."""
    monkeypatch.delenv("PK_EXPERIMENT", raising=False)
    old = F_.settings.precision
    q, plain = _flat_param((6,)), torch.nn.Parameter(torch.zeros(6))
    plain.grad = torch.zeros(6)
    try:
        F_.set_precision("bf16")
        monkeypatch.setattr(F_._Side, "listener", None)
        assert not F_.direct_affine_ok([q])                 # nobody declared a training step
        with F_.accumulating_backward():
            assert F_.direct_affine_ok([q])
            assert not F_.direct_affine_ok([q, plain])
            with torch.no_grad():
                assert not F_.direct_affine_ok([q])         # evaluation: nothing to route
            monkeypatch.setattr(F_._Side, "listener", lambda params: None)
            assert not F_.direct_affine_ok([q])
            monkeypatch.setattr(F_._Side, "listener", None)
            monkeypatch.setenv("PK_EXPERIMENT", "direct_grads=0")
            assert not F_.direct_affine_ok([q])
            monkeypatch.delenv("PK_EXPERIMENT")
            F_.set_precision("fp32")
            assert not F_.direct_affine_ok([q])             # parity mode: node by node
        assert F_.accumulating_backward.depth == 0
        try:
            with F_.accumulating_backward():
                raise RuntimeError("x")
        except RuntimeError:
            pass
        assert F_.accumulating_backward.depth == 0          # the context unwinds on exceptions
    finally:
        F_.set_precision(old)


def test_heads_share_one_private_alias_of_their_input():
    """functional._head_input: every fused head built on the same tensor (same object, same version) gets the SAME alias;
    another tensor or an in-place change gets a new one; the alias's backward hands the heads' summed gradient to the
    input - next to whatever else consumes it - and lets go of the alias and of the shared-gradient table."""
    old = F_._DxShare.on
    F_._DxShare.on = True
    try:
        x = torch.randn(5, 3, requires_grad=True)
        h = x * 1.0
        a1, a2 = F_._head_input(h), F_._head_input(h)
        assert a1 is a2 and a1 is not h and a1.data_ptr() == h.data_ptr()
        other = x * 2.0
        b1 = F_._head_input(other)
        assert b1 is not a1 and F_._head_input(other) is b1
        with torch.no_grad():
            assert F_._head_input(h) is h                   # nothing to share without a backward pass
        assert F_._head_input(torch.randn(5, 3)) is not None
        # gradients: two "heads" on the alias, a third consumer on the tensor itself, in between in creation order
        h2 = x * 1.0
        s = F_._head_input(h2)
        c1 = (s * 2.0).sum()
        extra = (h2 * h2).sum()
        c2 = (F_._head_input(h2) * 3.0).sum()
        F_._DxShare.table["probe"] = (F_._DxShare.epoch, torch.zeros(1), set())
        (c1 + extra + c2).backward()
        assert torch.allclose(x.grad, 5.0 + 2.0 * x.detach())
        assert F_._DxShare.alias is None and not F_._DxShare.table
        F_._DxShare.on = False
        assert F_._head_input(h) is h
    finally:
        F_._DxShare.on = old
        F_._DxShare.alias = None
        F_._DxShare.table.clear()


def test_deferred_side_launches_are_kept_in_order_and_flushed_by_join(monkeypatch):
    """functional.side_launch(defer=True) only queues (PK_EXPERIMENT side_late / side_defer_heads on); flush_deferred_side hands
    the queue to side_launch in order; join_side flushes first.  (The streams themselves need a GPU: the launch is
    replaced by a recorder here.)"""
    calls = []
    monkeypatch.delenv("PK_EXPERIMENT", raising=False)
    monkeypatch.setattr(F_.settings, "side_late", True)
    monkeypatch.setattr(F_._Side, "deferred", [])
    monkeypatch.setattr(F_._Side, "pending", False)
    monkeypatch.setattr(F_._Side, "stream", None)
    real = F_.side_launch

    def recorder(fn, keep, params=None, defer=False):
        if defer:
            return real(fn, keep, params, defer=True)
        calls.append(fn())

    monkeypatch.setattr(F_, "side_launch", recorder)
    F_.side_launch(lambda: "a", (), None, defer=True)
    F_.side_launch(lambda: "b", (), None, defer=True)
    assert calls == [] and len(F_._Side.deferred) == 2 and F_._Side.pending
    F_.flush_deferred_side()
    assert calls == ["a", "b"] and F_._Side.deferred == []
    F_.side_launch(lambda: "c", (), None, defer=True)
    F_.join_side()                                          # (no stream was ever made: nothing to wait for)
    assert calls == ["a", "b", "c"] and not F_._Side.pending
    monkeypatch.setenv("PK_EXPERIMENT", "side_defer_heads=0")
    monkeypatch.setattr(F_, "side_launch", real)
    ran = []
    monkeypatch.setattr(F_.torch.cuda, "current_stream", lambda: (_ for _ in ()).throw(AssertionError("launched at once")))
    try:
        F_.side_launch(lambda: ran.append(1), (), None, defer=True)
    except AssertionError as e:
        assert "launched at once" in str(e)                 # the switch turns the queue off: the call goes straight through
    assert F_._Side.deferred == []
