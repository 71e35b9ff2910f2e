"""Data parallelism for the chunk loop: one process per GPU, gradients all-reduced
over RCCL/xGMI.

The reference's only multi-GPU mechanism is ``torch.nn.DataParallel``
(core.py:103-104, 537-538), which scatters dim 0 - the TIME axis of (T, B, F)
sequence batches.  Here the batch (utterance) axis is sharded instead: rank r
takes columns [r*B/N, (r+1)*B/N) of every (T, B, F) batch (rows for 2-D batches),
BatchNorm statistics stay per replica (what DataParallel does too), and the only
exchange is one gradient all-reduce per step, issued bucket by bucket while
backward is still running so that it overlaps with the remaining BPTT kernels.

``GradReducer`` works on plain modules (per-parameter grads are packed into a
bucket) or on ``optim.FlatParams`` (buckets are slices of the flat gradient
buffer: zero copies).  Backend: "nccl" (= RCCL on ROCm) on GPUs, "gloo" on CPU
for the tests.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Rendezvous from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {}
        if backend == "nccl":
            kw["device_id"] = torch.device("cuda", torch.cuda.current_device())
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world, local


def shard_batch(inp, rank, world):
    """Rank's share of one batch: columns of (T, B, F), rows of (N, F).  B must divide evenly
    (the chunk planner pads the last batch like the reference drops it, core.py:552-556)."""
    if world == 1:
        return inp
    dim = 1 if inp.dim() == 3 else 0
    n = inp.shape[dim]
    if n % world != 0:
        raise ValueError("batch of %d does not shard over %d ranks" % (n, world))
    per = n // world
    return inp.narrow(dim, rank * per, per)


class GradReducer:
    """Average gradients across ranks, overlapping the all-reduce with backward.

    Buckets are filled in REVERSE parameter order (the order backward produces
    gradients).  A bucket is launched from the post-accumulate-grad hook of its
    last parameter; ``finish()`` waits for all of them (and launches whatever did
    not fire, e.g. parameters that received no gradient this step).
    """

    def __init__(self, modules, bucket_bytes=32 << 20, flats=None, group=None, overlap=True, force=False, wire=None):
        """overlap=True (default): a bucket's all-reduce is launched as soon as every gradient in it has been produced,
        while BPTT of the lower layers still runs.  Gradients arrive two ways: through autograd (BatchNorm / bias
        gradients: post-accumulate-grad hooks) and - in perf mode - from the weight-gradient GEMMs that
        functional.side_launch runs on the side stream and that accumulate straight into the flat .grad buffer
        (functional.set_side_listener reports them).  The all-reduce is enqueued from the SIDE stream, behind those
        GEMMs, so the main stream (the backward dependency chain) never waits for a weight gradient.
        overlap=False: every bucket is reduced in finish(), after backward.  force: build buckets on one rank too
        (tests: the whole path runs over a one-rank RCCL communicator).
        wire: "fp32" (default) or "bf16" (PK_DP_WIRE): the format a bucket travels in.  bf16 halves the payload of the
        ring (26.8-62.5 MB per step and rank at the BASELINE configurations; xGMI is per-link bound): every
        rank's share is rounded once (2^-9 relative) and the ring adds in bf16; the fp32 flat gradient is overwritten with
        the widened sum.  Graded against the fp32 wire in tests/test_dp_gloo.py."""
        self.group = group
        self.overlap = overlap
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._by_param = {}
        self.flats = flats  # dict name -> FlatParams or None
        self.buckets = []   # each: dict(params=[...], flat=tensor or None, pending=int)
        self.handles = []
        self._hooks = []
        self.active = False
        # PK_DP_TRACE=1: HIP events at every bucket's hand-over to the all-reduce and around finish(): timeline() then
        # says how far ahead of the end of backward each bucket left and how much of the exchange finish() still waited for
        self.trace = os.environ.get("PK_DP_TRACE", "0") == "1"
        self.wire = wire or os.environ.get("PK_DP_WIRE", "fp32")
        if self.wire not in ("fp32", "bf16"):
            raise ValueError("GradReducer: wire format %r (fp32 or bf16)" % (self.wire,))
        self._ev = []
        self._timelines = []
        if self.world == 1 and not force:
            return
        self.active = True
        if overlap:
            from . import functional as _F
            _F.set_side_listener(self._side_done)
        for name, mod in modules.items():
            flat = flats[name] if flats else None
            params = [p for p in mod.parameters() if p.requires_grad]
            if flat is not None:
                # (the never-used parameters sit behind n_active with a zero gradient: nothing to exchange)
                order = sorted(((o, p) for o, p, u in zip(flat.offsets, flat.params, flat.unused) if not u),
                               key=lambda t: -t[0])
                cur, cur_hi = [], None
                for off, p in order:
                    if cur_hi is None:
                        cur_hi = off + p.numel()
                    cur.append(p)
                    if (cur_hi - off) * 4 >= bucket_bytes:
                        self._add_bucket(cur, flat.grad[off:cur_hi])
                        cur, cur_hi = [], None
                if cur:
                    self._add_bucket(cur, flat.grad[0:cur_hi])
            else:
                cur, size = [], 0
                for p in reversed(params):
                    cur.append(p)
                    size += p.numel() * 4
                    if size >= bucket_bytes:
                        self._add_bucket(cur, None)
                        cur, size = [], 0
                if cur:
                    self._add_bucket(cur, None)

    def _add_bucket(self, params, flat_slice):
        # "expect" = how many ready-signals each parameter gives per step.  The reference always creates ln / bn
        # sub-modules it never calls (18 of 62 tensors in the Li-GRU recipe): their hooks never fire.
        # A module applied twice in one [model] (shared architecture) has its weight-gradient GEMM side-launched twice.
        # Step 1 therefore only COUNTS the signals per parameter (its buckets are reduced in finish()); from step 2 on
        # a bucket is launched when every parameter that signalled in step 1 has signalled as often again.
        b = {"params": list(params), "flat": flat_slice, "pending": len(params), "fired": False, "seen": {},
             "expect": None, "got": {}, "idx": len(self.buckets)}
        self.buckets.append(b)
        for p in params:
            self._by_param[id(p)] = b
        if self.overlap:
            for p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(lambda _p, b=b: self._ready(b, _p)))

    def _side_done(self, params):
        """functional.side_launch: the GEMMs that accumulate into these parameters' .grad are enqueued on the side
        stream (a bucket launched now, from that stream, runs behind them)."""
        for p in params:
            b = self._by_param.get(id(p))
            if b is not None:
                self._ready(b, p)

    def _ready(self, b, p):
        """One gradient contribution of p is enqueued (autograd: once per backward; side stream: once per
        side_launch).  Which path a Linear takes (functional: M >= 4096 rows -> side stream) depends on the shard
        shape only, and shards are equal on every rank (shard_batch), so all ranks signal - and launch their
        collectives - in the same order."""
        k = id(p)
        if b["expect"] is None:  # step 1: learn the pattern, reduce in finish()
            b["seen"][k] = b["seen"].get(k, 0) + 1
            return
        if b["fired"]:
            raise RuntimeError("data parallel: a gradient of a %s parameter arrived after its bucket had been handed to "
                               "the all-reduce (the step produced more gradient contributions than the first step of "
                               "this run did); build the GradReducer with overlap=False for graphs that change from "
                               "step to step" % (tuple(p.shape),))
        want = b["expect"].get(k)
        if want is None:
            return  # a parameter that was silent in step 1 fired now: it is still covered by finish()
        got = b["got"].get(k, 0) + 1
        b["got"][k] = got
        if got == want:
            b["pending"] -= 1
            if b["pending"] == 0:
                self._launch(b)

    def _mark(self, tag, stream=None):
        if self.trace and torch.cuda.is_available():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(stream if stream is not None else torch.cuda.current_stream())
            self._ev.append((tag, ev))

    def timeline(self):
        """PK_DP_TRACE=1: per finished step a dict {"buckets": [(index, bytes, ms of the hand-over relative to the start
        of finish(): negative = while backward was still running)], "exposed_ms": what finish() waited for}."""
        return list(self._timelines)

    def _launch(self, b):
        b["fired"] = True
        if b["flat"] is not None:
            buf = b["flat"]
        else:
            grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in b["params"]]
            buf = torch.cat([g.reshape(-1) for g in grads])
            b["packed"] = (buf, grads)
        side = None
        if buf.is_cuda:
            from . import functional as _F
            _F.flush_deferred_side()  # (postponed weight-gradient launches of this bucket's parameters: enqueue them first)
            side = _F._Side.stream if _F._Side.pending else None
        if side is not None:
            # behind the weight-gradient GEMMs of this bucket (side stream) AND behind the gradients autograd has
            # accumulated so far (main stream); the main stream itself does not wait
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._mark(("bucket", b["idx"], buf.numel() * buf.element_size()), side)
                h = self._reduce(b, buf)
        else:
            self._mark(("bucket", b["idx"], buf.numel() * buf.element_size()))
            h = self._reduce(b, buf)
        self.handles.append((h, b))

    def _reduce(self, b, buf):
        buf.div_(self.world)
        if self.wire == "bf16":
            w = buf.to(torch.bfloat16)
            b["wire"] = (w, buf)
            return dist.all_reduce(w, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reset(self):
        """Forget a step that did not complete (a failed HIP-graph capture, an exception inside backward): drop the
        handles and wire copies of buckets that had been handed over and re-arm every bucket, keeping what step 1 taught
        (``expect``).  Gradients are NOT touched: the caller runs the step again from zero_grad."""
        self.handles = []
        self._ev = []
        for b in self.buckets:
            b.pop("wire", None)
            b.pop("packed", None)
            if b["expect"] is None:
                b["seen"] = {}
            self._rearm(b)

    @staticmethod
    def _rearm(b):
        # (a bucket none of whose parameters signalled in step 1 never reaches 0: finish() reduces it)
        b["pending"] = len(b["expect"]) if b["expect"] else len(b["params"]) + 1
        b["got"] = {}
        b["fired"] = False

    def finish(self):
        """Call after backward(): completes every bucket and re-arms for the next step."""
        if not self.active:
            return
        self._mark(("finish",))
        for b in self.buckets:
            if not b["fired"]:
                self._launch(b)
        from .functional import join_side
        join_side()  # the side stream's weight gradients (and the reductions enqueued behind them) before anything reads .grad
        for h, b in self.handles:
            h.wait()
            if "wire" in b:  # the widened sum back into the fp32 bucket
                w, buf = b.pop("wire")
                # w was allocated on the stream the bucket was launched from (the side stream under overlap); its real
                # cross-stream consumer is this copy on the main stream: tell the allocator before w is dropped
                if w.is_cuda and not torch.cuda.is_current_stream_capturing():
                    w.record_stream(torch.cuda.current_stream())
                buf.copy_(w)
            if b["flat"] is None:
                buf, grads = b.pop("packed")
                o = 0
                for p, g in zip(b["params"], grads):
                    n = g.numel()
                    if p.grad is None:
                        p.grad = buf[o:o + n].view_as(p).clone()
                    else:
                        p.grad.copy_(buf[o:o + n].view_as(p))
                    o += n
        self.handles = []
        if self.trace and self._ev:
            self._mark(("done",))
            torch.cuda.current_stream().synchronize()  # (tracing only: the events are read back per step)
            t0 = next(ev for tag, ev in self._ev if tag[0] == "finish")
            rec = {"buckets": [(tag[1], tag[2], round(t0.elapsed_time(ev), 3)) for tag, ev in self._ev if tag[0] == "bucket"],
                   "exposed_ms": round(t0.elapsed_time(self._ev[-1][1]), 3)}
            self._timelines.append(rec)
            self._ev = []
        for b in self.buckets:
            if b["expect"] is None:
                b["expect"] = dict(b["seen"])
            self._rearm(b)
