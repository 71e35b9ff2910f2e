"""HIP-graph capture of a whole training step for the launch-bound recipes.

The non-sequential recipes (MLP on 128 frames, SincNet on 128 chunks) spend 30-60 us of ideal GPU work in ~70
kernel launches: the step is bound by launch latency, not by any kernel.  Every call the engine
makes is an asynchronous launch on torch's current stream (C-ABI kernels, hipMemsetAsync, torch element-wise ops,
the device RNG for drop masks), so the step - forward_model, backward, fused optimizer - is captured once into a HIP
graph (torch.cuda.CUDAGraph) and replayed per batch from a static input buffer: one launch per step.

Measured (MI355X, tools/graph_probe.py): timit_mlp 2.47 -> 0.91 ms/step.  The sequence recipes gain nothing (their
kernels run for milliseconds) and the Li-GRU step is slower under capture (27.3 vs 26.3 ms: the side-stream
weight-gradient overlap is flattened), so graphs are used for fixed-shape, non-sequence batches only.
"""
import torch

from . import _lib


class GraphedStep:
    """step_fn(inp) -> dict/tuple/tensor of device tensors.  ``capture(inp)`` records one call (nothing runs),
    ``__call__(inp)`` copies the batch into the static buffer, replays and returns the static outputs.

    The caller runs a few eager steps first (lazy one-time initialisation - kernel attributes, allocator pools -
    must not happen inside the capture).  Fused optimizers whose step count is a kernel ARGUMENT (Adam's bias
    correction) cannot be replayed: pass them in ``optimizers`` and capture refuses them; for the others the python
    side step counters are advanced per replay so that checkpoints keep counting."""

    def __init__(self, step_fn, optimizers=()):
        self.step_fn = step_fn
        self.optimizers = list(optimizers)
        for o in self.optimizers:
            if getattr(o, "kind", None) == "adam":
                raise _lib.PkError("hipGraph replay bakes kernel arguments: Adam's step count cannot be captured")
        self.graph = None
        self.static_inp = None
        self.out = None

    def capture(self, inp):
        self.static_inp = inp.clone()
        from . import functional as F_
        F_.label_check_counter(inp.device)  # the bad-label counter must exist before the capture that adds into it
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        steps0 = [getattr(o, "steps", 0) for o in self.optimizers]
        with torch.cuda.graph(self.graph):
            self.out = self.step_fn(self.static_inp)
        for o, s0 in zip(self.optimizers, steps0):  # the recorded call did not run
            if hasattr(o, "steps"):
                o.steps = s0
        return self

    def __call__(self, inp):
        if self.graph is None:
            raise _lib.PkError("GraphedStep: capture() first")
        self.static_inp.copy_(inp, non_blocking=True)
        self.graph.replay()
        for o in self.optimizers:
            if hasattr(o, "steps"):
                o.steps += 1
        return self.out
