"""Host-side autograd wrappers around the C-ABI kernels of libpk_amd.so.

Everything here is plumbing: torch owns the memory, autograd owns the graph,
and each ``torch.autograd.Function`` hands raw device pointers plus the current
HIP stream to ``include/pk_amd.h`` entry points.  There is deliberately no
torch-op fallback: a CPU tensor or a missing library raises.
"""
import ctypes
import os
import weakref

import torch

from . import _lib

PREC = {"fp32": 0, "bf16": 1}
ACT = {"linear": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "leaky_relu": 4, "elu": 5}
CELL = {"liGRU": 0, "RNN": 1, "LSTM": 2, "GRU": 3, "minimalGRU": 4}
REC_STEPWISE, REC_PERSISTENT = 0, 1


class _Settings:
    """Process-wide numerics / algorithm switches.

    precision : "fp32" (exact fp32 MFMA - parity mode, default) or "bf16"
                (bf16 MFMA operands, fp32 accumulate / state / master weights).
    rec_algo  : "auto" | "stepwise" | "persistent".
    """

    def __init__(self):
        self.precision = os.environ.get("PK_PRECISION", "fp32")
        self.rec_algo = os.environ.get("PK_REC_ALGO", "auto")
        # recurrent drop masks: "device" (default) = Bernoulli(1-p) drawn on the GPU RNG (the reference's distribution, not
        # its stream), no host work in the step; "reference" = the reference's STREAM - the masks
        # torch.bernoulli(torch.Tensor(rows,H).fill_(1-p)) gives on the CPU generator, same seed -> bit-identical masks -
        # drawn ON THE DEVICE from a mirror of that generator (round 5: _RefRng, csrc/pk_rng.hip; ~0.2 ms per 256 x 550
        # mask, in line on the caller's stream); "reference_host" = the same
        # stream by making the reference's own call on the host (a forward call ahead on a helper thread: ~40 ms per step
        # at BASELINE config 2, host-bound)
        self.mask_rng = os.environ.get("PK_MASK_RNG", "device")
        assert self.mask_rng in ("reference", "reference_host", "device"), self.mask_rng
        # perf mode: weight-gradient GEMMs (dW, dU) of a recurrent layer / a Linear run on a second HIP stream, next to
        # the following layer's recurrent backward, and accumulate straight into the parameters' flat .grad buffer
        # (only for parameters owned by optim.FlatParams; see side_launch / join_side)
        self.wgrad_side = _lib.experiment("wgrad_side", "1") != "0"
        # the liGRU / RNN persistent kernels keep their exchange buffers filled themselves (0 = whole-buffer fill on a third stream)
        self.self_fill = _lib.experiment("rec_self_fill", "1") != "0"
        # weight-gradient GEMMs of a recurrent layer go to the side stream behind the layer's LAST main-stream kernel
        # (its dX GEMM), so that they run next to the layer below's latency-bound recurrence and not next to this
        # layer's bandwidth-bound BatchNorm backward / dX GEMM (0 = as soon as their operands exist)
        self.side_late = _lib.experiment("side_late", "1") != "0"
        # ... and size their split-K grids for the CUs that recurrence leaves free (0 = for the whole device)
        self.side_cus = _lib.experiment("side_cus", "0") != "0"
        # perf mode: a cost_nll line directly behind a fused output layer is computed by head_nll (one pass, no dense
        # one-hot gradient); 0 = through the caller's nn.NLLLoss on the log-posteriors, which is what the reference's
        # own forward_model does with this package's classes (utils.py:2361)
        self.fused_cost = _lib.experiment("fused_cost", "1") != "0"
        assert self.precision in PREC, self.precision
        assert self.rec_algo in ("auto", "stepwise", "persistent"), self.rec_algo


settings = _Settings()


def set_precision(p):
    assert p in PREC, p
    settings.precision = p


def set_mask_rng(m):
    assert m in ("reference", "reference_host", "device"), m
    settings.mask_rng = m


def set_rec_algo(a):
    assert a in ("auto", "stepwise", "persistent"), a
    settings.rec_algo = a


# ----------------------------------------------------------------------------
# Kink-forced test mode (protocol 3b).  A ReLU recurrence is piecewise linear: a 1e-7
# forward difference that moves a pre-activation a_t across 0 changes WHICH linear piece is differentiated, and two
# correct fp32 runs of the reference then differ by 4e-4 (T=100) .. 3e-3 (T=500) in gradient norm.  To grade
# gradients at 1e-4 at any T the backward pass can be told the reference's own pattern (a_t > 0): the saved
# pre-activation of every (step, row, unit) keeps its magnitude and takes the reference's sign before BPTT runs, so
# both sides differentiate the same piece.  Test infrastructure: off unless set_forced_kinks() is called.
# ----------------------------------------------------------------------------
class _Kinks:
    queue = None   # list of bool tensors (T, R, H) in the reference's row order (rows >= B = time-reversed copies),
    report = None  # one per recurrent layer call, consumed in call order; report: (flipped, total, max |a| flipped)


def set_forced_kinks(patterns):
    """patterns: list of (T, R, H) bool tensors, one per recurrent layer in forward order, or None to switch off.
    Returns the report list that fills up as layers consume their pattern."""
    _Kinks.queue = None if patterns is None else list(patterns)
    _Kinks.report = []
    return _Kinks.report


_KINK_SLOT = {"liGRU": 1, "RNN": 0, "GRU": 2, "minimalGRU": 1}


def _force_kinks(S, cell, T, B, ndir, H):
    if _Kinks.queue is None:
        return
    if cell not in _KINK_SLOT:
        raise _lib.PkError("kink-forced mode covers the cells whose candidate goes through act(a): %s" % cell)
    k = _Kinks.queue.pop(0).to(S.device)
    NS = S.shape[2] // H
    flipped, worst = 0, 0.0
    for d in range(ndir):  # S[d] is indexed by ORIGINAL time; the reference's reversed rows run backwards in time
        a = S[d].view(T, B, NS, H)[:, :, _KINK_SLOT[cell], :]
        kd = k[:, d * B:(d + 1) * B]
        if d == 1:
            kd = torch.flip(kd, [0])
        diff = (a > 0) != kd
        n = int(diff.sum())
        if n:
            flipped += n
            worst = max(worst, float(a[diff].abs().max()))
        mag = a.abs().clamp_min(1e-30)
        a.copy_(torch.where(kd, mag, -mag))
    _Kinks.report.append((flipped, ndir * T * B * H, worst))


# ----------------------------------------------------------------------------
# Injected nn.Dropout masks (test infrastructure, like the kink patterns above): the reference draws the masks of its
# nn.Dropout modules (MLP :139-148, CNN / SincNet :1546-1552, :1655-1661) from torch's dropout RNG; a config-scale
# comparison with a reference run needs the very masks that run drew (tests/golden/scale_*: recovered by forward hooks,
# oracle/make_golden.py::_DropoutTap).  Off unless set_forced_dropout() is called.
# ----------------------------------------------------------------------------
class _Drops:
    queue = None


def set_forced_dropout(masks):
    """masks: list of 0/1 tensors, one per dropout call with p > 0 in forward order, or None to switch off."""
    _Drops.queue = None if masks is None else list(masks)


class _Decisions:
    relu = None    # queue of bool patterns (input > 0) for feed-forward ReLU calls (NormActDropFn), in call order
    pool = None    # queue of (pool, window-offset uint8 tensors) for conv1d_pool calls, in call order
    report = None  # [(kind, differing entries, entries)] as the calls consume their patterns


def set_forced_decisions(relu=None, pool=None):
    """Test mode for feed-forward stacks (the counterpart of set_forced_kinks): `relu` = list of bool tensors, the ReLU
    pattern (pre-activation > 0) of another run for every ReLU call in forward order; `pool` = list of (pool length,
    uint8 offsets of the arg-max inside its window) for every conv1d_pool call.  Backward then routes gradients through
    THOSE decisions (forward values are untouched).  Returns the report list; None / None switches it off."""
    _Decisions.relu = None if relu is None else list(relu)
    _Decisions.pool = None if pool is None else list(pool)
    _Decisions.report = []
    return _Decisions.report


class _RefRng:
    """PK_MASK_RNG=reference: the reference's drop-mask stream, drawn on the device (csrc/pk_rng.hip).  A mirror of torch's
    default CPU generator (mt19937: 624 words, left, next) lives on the device and is advanced there, one launch per
    mask; torch's own generator is brought up to date by sync_back() - run_nn_dp's end of chunk and
    nn.drain_mask_prefetch() call it, and so does the next adopt() - so that whatever draws from it afterwards continues
    the reference's stream.  Between two sync points the CPU generator lags behind: nothing in the engine's step draws
    from it (batch padding uses python's random, nn.Dropout masks the device generator).  A re-seed or a foreign draw is
    noticed at the next mask (the CPU state no longer equals the one the mirror descends from) and the mirror is rebuilt
    from the CPU state."""
    dev = None        # int32 [626] on the device: state words, left, next
    base = None       # the CPU state (ByteTensor, 5056 bytes) the mirror descends from
    ahead = False     # the mirror has drawn since `base`

    WORDS, OFF_LEFT, OFF_NEXT, OFF_STATE = 624, 8, 16, 24

    @classmethod
    def _parse(cls, st):
        import numpy as np
        b = st.numpy().tobytes()
        words = np.frombuffer(b, dtype="<u8", count=cls.WORDS, offset=cls.OFF_STATE).astype(np.uint32)
        left = int(np.frombuffer(b, dtype="<i4", count=1, offset=cls.OFF_LEFT)[0])
        nxt = int(np.frombuffer(b, dtype="<u8", count=1, offset=cls.OFF_NEXT)[0])
        return np.concatenate([words, np.array([left, nxt], dtype=np.uint32)])

    @classmethod
    def _build(cls, base, mirror):
        """base state bytes with the engine fields replaced by the mirror's [626] words"""
        import numpy as np
        b = bytearray(base.numpy().tobytes())
        b[cls.OFF_STATE:cls.OFF_STATE + 8 * cls.WORDS] = mirror[:cls.WORDS].astype("<u8").tobytes()
        b[cls.OFF_LEFT:cls.OFF_LEFT + 4] = np.array([int(mirror[cls.WORDS])], dtype="<i4").tobytes()
        b[cls.OFF_NEXT:cls.OFF_NEXT + 8] = np.array([int(mirror[cls.WORDS + 1])], dtype="<u8").tobytes()
        return torch.frombuffer(b, dtype=torch.uint8).clone()

    @classmethod
    def adopt(cls, device):
        """Make the device mirror current: nothing to do while the CPU generator still holds the state it descends from."""
        device = torch.empty(0, device=device).device  # ("cuda" -> "cuda:0": the mirror is compared by its tensor's device)
        st = torch.get_rng_state()
        if st.numel() != 5056:
            raise _lib.PkError("PK_MASK_RNG=reference: torch's CPU generator state has %d bytes, not the 5056 of the "
                               "mt19937 layout this build mirrors" % st.numel())
        if cls.dev is not None and cls.dev.device == device and cls.base is not None and torch.equal(st, cls.base):
            return
        if cls.ahead and cls.dev is not None and cls.base is not None and torch.equal(st, cls.base):
            # (another device asks while the mirror is ahead of an untouched CPU generator: its draws are part of the
            # stream - write them back first, then the new device's mirror descends from that state)
            cls.sync_back()
            st = torch.get_rng_state()
        if cls.ahead:  # (somebody re-seeded or drew on the CPU while the mirror was ahead: the CPU state wins)
            cls.ahead = False
        cls.base = st.clone()
        import numpy as np
        cls.dev = torch.from_numpy(cls._parse(st).view(np.int32).copy()).to(device)
        cls.ahead = False

    @classmethod
    def masks(cls, shapes, device):
        """[(rows, H, p)] -> [mask]: the masks the reference's torch.bernoulli(torch.Tensor(rows, H).fill_(1 - p)) calls would
        give, in that order (its masks are unscaled).  Drawn in line on the caller's stream (~0.2 ms each at 256 x 550, one
        workgroup; DESIGN.md 12.8 has the history of a form that drew them on a stream of their own)."""
        cls.adopt(device)
        lib = _lib.load()
        out = []
        for rows, H, p in shapes:
            m = torch.empty(rows, H, device=cls.dev.device, dtype=torch.float32)
            keep = float(torch.tensor(1 - p, dtype=torch.float32))  # fill_(1 - p): the float32 the reference compares with
            _lib.check(lib.pk_mt19937_bernoulli(_stream(), ctypes.c_void_p(cls.dev.data_ptr()), rows * H, keep, _p(m)),
                       "pk_mt19937_bernoulli")
            out.append(m)
        cls.ahead = True
        return out

    @classmethod
    def mask(cls, rows, H, p, device):
        return cls.masks([(rows, H, p)], device)[0]

    @classmethod
    def sync_back(cls):
        """torch's CPU generator <- the mirror (one 2.5 KB read-back; waits for the draws enqueued so far)."""
        if cls.dev is None or not cls.ahead:
            return
        import numpy as np
        mirror = cls.dev.to("cpu").numpy().view(np.uint32)
        if not torch.equal(torch.get_rng_state(), cls.base):
            cls.ahead = False  # the CPU generator moved on by itself meanwhile: its state wins (see adopt)
            return
        new = cls._build(cls.base, mirror)
        torch.set_rng_state(new)
        cls.base = torch.get_rng_state().clone()
        cls.ahead = False


class StepFence:
    """A bound on how far the host may run AHEAD of the GPU in an eager training loop: call it once per step; it records
    an event and waits for the one from `depth` steps ago (PK_STEPS_IN_FLIGHT, default 4; 0 = no bound).

    Why: the engine hands big tensors to another stream (the operands of the weight-gradient GEMMs: side_launch) and marks
    them with record_stream - the caching allocator may reuse such a block only once the GPU has passed that use.  A host that enqueues a 25-35 ms step in 4 ms gets dozens of steps ahead, every one of them holding
    its own 7-9 GB of activations: the pool runs into the 288 GB, the allocator starts freeing and re-allocating device
    memory (synchronous calls, seconds each) and single steps take 2-6 s (measured: tools/diag_slow_steps.py,
    profiles/r05_host_lead.json).  The reference never gets there - it reads the loss back after every batch
    (core.py:689).  Four steps in flight keep the GPU's queue full; the wait is on the host only."""

    def __init__(self, depth=None):
        import collections
        self.depth = int(os.environ.get("PK_STEPS_IN_FLIGHT", "4")) if depth is None else int(depth)
        self.q = collections.deque()

    def __call__(self):
        if self.depth <= 0 or not torch.cuda.is_available():
            return
        ev = torch.cuda.Event()
        ev.record()
        self.q.append(ev)
        if len(self.q) > self.depth:
            self.q.popleft().synchronize()


def ref_rng_mask(rows, H, p, device):
    return _RefRng.mask(rows, H, p, device)


class _Ahead:
    """Masks of the nn.Dropout calls a forward is about to make, drawn together (masks_ahead)."""
    queue = []
    ones = {}  # (device, n) -> tensor of ones the fused dropout kernel is applied to


def masks_ahead(specs, device):
    """specs: [(rows, cols, p)] in call order.  A feed-forward stack at batch 128 is launch-bound: instead of two
    launches per nn.Dropout call (Bernoulli draw, scale) the masks of all calls with the same p are drawn as ONE flat
    tensor - one launch per distinct p; dropout_mask() then hands out its slices in order.  (Same distribution,
    another use of the device generator's stream than call-by-call draws.)"""
    _Ahead.queue = []
    if _Drops.queue is not None or len(specs) < 2:
        return
    out = [None] * len(specs)
    for p in sorted(set(q for _, _, q in specs)):
        idx = [i for i, sp in enumerate(specs) if sp[2] == p]
        sizes = [_up(specs[i][0] * specs[i][1], 4) for i in idx]  # 16-byte aligned slices
        # (one launch: the fused dropout kernel on a cached tensor of ones gives mask / (1 - p) directly - a Bernoulli draw
        # followed by a scale launch were two of the ~12 stock launches of a 0.3 ms MLP step)
        n = sum(sizes)
        ones = _Ahead.ones.get((str(device), n))
        if ones is None:
            ones = _Ahead.ones[(str(device), n)] = torch.ones(n, device=device, dtype=torch.float32)
        flat = torch.nn.functional.dropout(ones, p, True)
        o = 0
        for i, n in zip(idx, sizes):
            out[i] = flat[o:o + specs[i][0] * specs[i][1]].view(specs[i][0], specs[i][1])
            o += n
    _Ahead.queue = [(tuple(sp[:2]), sp[2], m) for sp, m in zip(specs, out)]


def dropout_mask(like, p):
    """The Bernoulli(1-p) / (1-p) mask of one nn.Dropout call on a tensor shaped `like` (device RNG)."""
    if _Drops.queue is not None:
        m = _Drops.queue.pop(0)
        if tuple(m.shape) != tuple(like.shape):
            raise _lib.PkError("forced dropout mask %s for a %s tensor" % (tuple(m.shape), tuple(like.shape)))
        return m.to(device=like.device, dtype=torch.float32) / (1.0 - p)
    if _Ahead.queue:
        shape, q, m = _Ahead.queue.pop(0)
        if shape == tuple(like.shape) and q == p and m.device == like.device:
            return m
        _Ahead.queue = []  # the forward took another route than announced: back to call-by-call draws
    return torch.empty_like(like).bernoulli_(1.0 - p).div_(1.0 - p)


# ----------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------
def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.PkError("pytorch-kaldi_amd runs on the GPU only (got a %s tensor); there is no CPU fallback. "
                               "Set use_cuda=True in the cfg." % t.device)
        if t is not None and t.dtype != torch.float32:
            raise _lib.PkError("pytorch-kaldi_amd kernels take fp32 tensors (got %s)" % t.dtype)


def _rows2d(x):
    """View x as [rows, cols] with unit column stride (copy only if it must)."""
    if x.dim() != 2:
        x = x.reshape(-1, x.shape[-1])
    if x.stride(1) != 1 or (x.shape[0] > 1 and x.stride(0) < x.shape[1]):
        x = x.contiguous()
    return x


def _new(*shape, like):
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


def _splitk(out_tiles, K):
    """Split the reduction when a GEMM has few output tiles and a long K (dW, dU)."""
    if K < 4096:
        return 1
    # four 128 x 128 workgroups fit a CU (34 KB of LDS each) and one wave per SIMD hides nothing: fill 1024 slots, not 256
    # (the 1100 x 1104 x 64 000 weight gradient ran as 243 workgroups - 3 splits - at 66 TFLOP/s)
    slots = int(_lib.experiment("f32_splitk_slots", "1024"))
    s = max(1, slots // max(1, out_tiles))
    return int(min(16, s, max(1, K // 1024)))


def _small_m_splitk(M, N, K):
    """Exact-fp32 products of a small batch (an MLP layer at 128 frames is 8 output tiles on 256 CUs, each a chain of 64
    k-tiles: 70 us): the reduction is split over the chip (round 6; the perf mode's twin is pk_gemm_bf16_small_splitk).
    Reductions shorter than 512 keep one chain - the module fixtures and the 30-step trajectory fixture, whose bits are
    pinned, are all below it."""
    if M > 256 or K < 512 or _lib.experiment("f32_small_splitk", "1") == "0":
        return 1
    return int(max(1, min(16, 512 // max(1, _tiles(M, N)), K // 64)))


def gemm(M, N, K, A, a_rs, a_cs, B, b_rs, b_cs, C, ldc, alpha=1.0, beta=0.0, bias=None, splitk=1, prec=None):
    lib = _lib.load()
    ws = None
    if splitk == 1 and a_cs == 1 and (prec or settings.precision) == "fp32":
        splitk = _small_m_splitk(M, N, K)
    if splitk > 1:
        ws = torch.empty(splitk * M * N, device=C.device, dtype=torch.float32)
    rc = lib.pk_gemm(_stream(), PREC[prec or settings.precision], M, N, K, alpha, _p(A), a_rs, a_cs, _p(B), b_rs, b_cs,
                     beta, _p(C), ldc, _p(bias), splitk, _p(ws))
    _lib.check(rc, "pk_gemm")
    return C


def _tiles(M, N):
    return ((M + 127) // 128) * ((N + 127) // 128)


def _tiles_bf(M, N):
    """The output shape of a bf16-operand GEMM, as _splitk_bf takes it."""
    return (M, N)


def _up(n, m):
    return (n + m - 1) // m * m


def weight_bf16(w):
    """The bf16 copy of a 2-D weight at the plain pitch (what cvt_bf16(w) gives).  A weight that lives in an
    optim.FlatParams buffer gets a PERSISTENT copy which the fused optimizer step refreshes on its way out
    (pk_fused_step): the per-layer conversion launch of every step is gone.  The copy is trusted only while the weight's
    version counter stands where it stood when the copy was made (the fused step writes through raw pointers and does not
    move it; load_state_dict / any torch in-place op does, and the copy is then redone)."""
    sh = getattr(w, "_pk_shadow", None)
    if sh is not None:
        view, version, alive, flat_version = sh
        if alive() is not None:
            owner = getattr(w, "_pk_owner", None)
            owner = owner() if owner is not None else None
            fv = owner.flat._version if owner is not None else flat_version
            # (the Parameter's own counter AND the flat buffer's: an in-place op or a collective on FlatParams.flat
            # rewrites the weight without touching the Parameter's counter; raw-pointer writers other than the fused
            # step must call FlatParams.invalidate_shadows())
            if w._version == version and fv == flat_version:
                return view
            cvt_bf16(w.detach(), out=view)  # somebody rewrote the weight through torch: resynchronise
            w._pk_shadow = (view, w._version, alive, fv)
            return view
    owner = getattr(w, "_pk_owner", None)
    owner = owner() if owner is not None else None
    if owner is not None and w.is_contiguous():
        owner.want_shadow(w)  # from the next optimizer step on
    return cvt_bf16(w.contiguous())


def cvt_bf16(x2, nseg=1, seglen=None, segpad=None, out=None):
    """fp32 [rows, cols] (unit column stride) -> bf16 [rows, pitch]; the columns are `nseg` segments of
    `seglen`, each placed at a pitch of `segpad` (multiple of 8 so that every segment starts 16-byte
    aligned); padding is zero.  pitch = nseg*segpad rounded up to 64.  out: write into this [rows, pitch] bf16 tensor."""
    lib = _lib.load()
    rows, cols = x2.shape
    if seglen is None:
        nseg, seglen = 1, cols
    if segpad is None:
        segpad = _up(seglen, 8)
    assert nseg * seglen == cols and x2.stride(1) == 1
    pitch = _up(nseg * segpad, 64)
    if out is None:
        out = torch.empty(rows, pitch, device=x2.device, dtype=torch.bfloat16)
    assert tuple(out.shape) == (rows, pitch) and out.dtype == torch.bfloat16 and out.is_contiguous()
    _lib.check(lib.pk_cvt_bf16(_stream(), _p(x2), x2.stride(0), rows, nseg, seglen, segpad, _p(out), pitch),
               "pk_cvt_bf16")
    return out


_SMALL_SPLITK = _lib.experiment("small_splitk", "1") != "0"


def gemm_bf16(M, N, K, A, lda, a_kc, B, ldb, b_kc, C, ldc, alpha=1.0, beta=0.0, bias=None, splitk=1):
    """C[M,N] fp32 = alpha * A.B + beta*C + bias on bf16 operands (see include/pk_amd.h: pk_gemm_bf16).
    A / B may be tensors or (tensor, element_offset) pairs."""
    lib = _lib.load()
    ws = None
    if splitk == 1 and a_kc and M <= 128 and _SMALL_SPLITK:
        # a small-batch product is as slow as its largest per-workgroup footprint: the reduction is split over the grid
        # (pk_gemm_bf16_small_splitk: 1 = not worth a second launch)
        splitk = int(lib.pk_gemm_bf16_small_splitk(M, N, K))
    if splitk > 1:
        ws = torch.empty(splitk * M * N, device=C.device, dtype=torch.float32)

    def ptr(t):
        if isinstance(t, tuple):
            return ctypes.c_void_p(t[0].data_ptr() + 2 * t[1])
        return ctypes.c_void_p(t.data_ptr())

    rc = lib.pk_gemm_bf16(_stream(), M, N, K, alpha, ptr(A), lda, int(a_kc), ptr(B), ldb, int(b_kc), beta, _p(C), ldc,
                          _p(bias), splitk, _p(ws))
    _lib.check(rc, "pk_gemm_bf16")
    return C


def gemm_bf16_bn_stats(M, N, K, A, lda, a_kc, B, ldb, b_kc, C, ldc, bias=None, gates=None):
    """C = A.B (+ bias) on bf16 operands together with the BatchNorm statistics of C's columns (biased variance):
    taken in the GEMM epilogue where the shape runs on the 256-tile (pk_gemm_bf16_stats), by pk_bn_stats otherwise.
    -> (mean, var), or (mean, var, scale, shift) with gates = the arguments of bn_finalize_gates behind (mean, var):
    (gamma, beta, eps, H, running_means, running_vars, batches, momentum, count) - the merge of the statistics partials
    and the finalize step are then one launch."""
    lib = _lib.load()

    def done(mean, var):
        return (mean, var) if gates is None else (mean, var) + tuple(bn_finalize_gates(mean, var, *gates))

    if _lib.experiment("gemm_stats", "1") == "0":  # A/B switch: statistics by a second pass over C
        gemm_bf16(M, N, K, A, lda, a_kc, B, ldb, b_kc, C, ldc, bias=bias)
        return done(*bn_stats(C if ldc == N else C.as_strided((M, N), (ldc, 1))))
    stats = torch.empty(int(lib.pk_gemm_bf16_stats_floats(M, N)), device=C.device, dtype=torch.float32)
    rb = ctypes.c_int(0)
    rc = lib.pk_gemm_bf16_stats(_stream(), M, N, K, 1.0, _p(A), lda, int(a_kc), _p(B), ldb, int(b_kc), _p(C), ldc, _p(bias),
                                _p(stats), ctypes.byref(rb))
    _lib.check(rc, "pk_gemm_bf16_stats")
    if rb.value == 0:
        return done(*bn_stats(C if ldc == N else C.as_strided((M, N), (ldc, 1))))
    mean, var = torch.empty(N, device=C.device), torch.empty(N, device=C.device)
    if gates is not None:
        gamma, beta, eps, H, rms, rvs, nbs, momentum, count = gates
        G = len(rms)
        scale, shift = torch.empty_like(mean), torch.empty_like(mean)
        arr = lambda ts: (ctypes.c_void_p * G)(*[t.data_ptr() for t in ts])
        _lib.check(lib.pk_bn_stats_merge_finalize_gates(_stream(), _p(stats), rb.value, G, H, _p(mean), _p(var), _p(gamma),
                                                        _p(beta), eps, _p(scale), _p(shift), arr(rms), arr(rvs), arr(nbs),
                                                        momentum, float(count)), "pk_bn_stats_merge_finalize_gates")
        return mean, var, scale, shift
    _lib.check(lib.pk_bn_stats_merge(_stream(), _p(stats), rb.value, N, _p(mean), _p(var)), "pk_bn_stats_merge")
    return mean, var


def _splitk_bf(out_tiles, K):
    """Split the reduction of the dW / dU shapes (few output tiles, K = T*B rows) over the chip.  `out_tiles` is an
    (M, N) pair: the library knows which block tile the shape takes."""
    M, N = out_tiles
    return int(_lib.load().pk_gemm_bf16_auto_splitk_cus(int(M), int(N), int(K), int(_SideCus.cus)))


class _SideCus:
    """CUs a split-K GEMM should size its grid for (0 = the whole device).  The weight-gradient GEMMs of a recurrent
    layer run on the side stream next to the recurrence of the layer below, which holds C x Pn CUs for its whole
    launch (pk_rec_plan_cus): sized for the whole chip, their 250 (dW) / 500 (dU) work items took three / two rounds
    on the CUs that are left."""
    cus = 0


def _sized_for(cus, fn):
    def run():
        old, _SideCus.cus = _SideCus.cus, cus
        try:
            fn()
        finally:
            _SideCus.cus = old
    return run


def bf16_mode():
    return settings.precision == "bf16"



# ----------------------------------------------------------------------------
# Second stream for work that is off the backward dependency chain (weight gradients)
# ----------------------------------------------------------------------------
class _Side:
    stream = None
    pending = False
    listener = None  # dp.GradReducer (overlap mode): told which parameters' gradients were just enqueued on the side stream
    deferred = []    # side_launch(defer=True) calls waiting for the next recurrence launch (flush_deferred_side)


def set_side_listener(fn):
    _Side.listener = fn


def side_targets_ok(params):
    """Weight gradients may bypass autograd only when every target parameter has a pre-allocated .grad that is a
    view of an optim.FlatParams buffer (its consumers - fused optimizer step, gradient all-reduce, zero_grad -
    call join_side() first)."""
    return bf16_mode() and side_targets_any_ok(params)


def side_targets_any_ok(params):
    """side_targets_ok without the precision condition.  Round 6: the exact-fp32 mode takes the side stream for its
    weight-gradient GEMMs too (RecLayerFn, LinearFn: 33 ms of a 149 ms fp32 LSTM step ran serially next to recurrences
    that leave 112 CUs idle); PK_EXPERIMENT f32_wgrad_side=0 keeps them on the main stream."""
    if not bf16_mode() and _lib.experiment("f32_wgrad_side", "1") == "0":
        return False
    return (settings.wgrad_side and params is not None and len(params) > 0
            and all(getattr(q, "_pk_flat", False) and q.grad is not None and q.requires_grad for q in params))


class accumulating_backward:
    """`with accumulating_backward(): out = net(x); ...; loss.backward()` - the caller states that this is an ordinary
    training step whose backward pass ACCUMULATES into the parameters' .grad, which is what lets kernels add their
    gradients to the flat .grad themselves (direct_grads_ok, decided in backward: small-batch weight / BatchNorm / bias
    gradients; direct_affine_ok, decided in forward: the BatchNorm affine of recurrent layers).  Without it every such
    gradient is returned to autograd, so torch.autograd.grad() and other gradient-only callers get what they ask for.
    core.run_nn's step and bench.py wrap forward and backward in it."""
    depth = 0

    def __enter__(self):
        accumulating_backward.depth += 1
        return self

    def __exit__(self, *exc):
        accumulating_backward.depth -= 1
        return False


def direct_grads_ok(params):
    """Small-batch layers (an MLP step of 128 frames is launch-bound): a gradient may be ACCUMULATED into the parameter's
    pre-allocated flat .grad by the kernel that produces it, on the current stream, instead of being returned to autograd
    (whose AccumulateGrad node is one more add launch per parameter).  Only inside `with accumulating_backward():` (the
    engine's own step: a backward pass that is known to accumulate), under the conditions of the side-stream weight
    gradients, and only while no data-parallel reducer listens for gradient hooks.  PK_EXPERIMENT direct_grads=0 turns it off."""
    return (accumulating_backward.depth > 0 and _Side.listener is None and _lib.experiment("direct_grads", "1") != "0"
            and torch.is_grad_enabled() is False and side_targets_ok(params))


def direct_affine_ok(params):
    """direct_grads_ok for a decision that has to be taken in FORWARD (a recurrent layer's BatchNorm scales / shifts are
    handed to its node as detached views of the flat buffer - no torch.cat per layer and step, no AccumulateGrad adds
    behind the node - only when the whole step runs inside `with accumulating_backward():`)."""
    return (accumulating_backward.depth > 0 and _Side.listener is None and _lib.experiment("direct_grads", "1") != "0"
            and torch.is_grad_enabled() and side_targets_ok(params))


_DEBUG_SKIP_SIDE = _lib.experiment("debug_skip_side", "0") == "1"  # timing experiments only: drops the weight-gradient work


def _make_side_stream():
    # (hardware queue priorities were tried both ways - side stream lowest, main stream highest - to let a recurrence
    # launched into a chip full of weight-gradient workgroups take the CUs first: no effect on the step time)
    return torch.cuda.Stream()


def side_launch(fn, keep, params=None, defer=False):
    """Run fn() on the side stream after everything enqueued so far on the current stream; `keep` are the tensors
    fn reads or writes that autograd may free before the side stream is done; `params`: the parameters whose .grad fn
    accumulates into (the data-parallel reducer launches a bucket's all-reduce behind them, on this stream).

    defer = True (PK_EXPERIMENT side_late and PK_EXPERIMENT side_defer_heads, default on): the whole call - the stream dependency included - is postponed until the
    next recurrent layer's backward is about to launch its recurrence (flush_deferred_side), or until join_side().  For
    the output layers on top of a recurrent stack: their weight-gradient GEMMs fill the whole chip for 0.3 ms, and the
    short kernels of the NEXT head's backward on the main stream queued behind their workgroups (0.29 ms per step in the
    round-4 timeline); next to the recurrence, which leaves 112 CUs idle, the same work is free."""
    if defer and settings.side_late and _lib.experiment("side_defer_heads", "1") != "0":
        _Side.deferred.append((fn, keep, params))
        _Side.pending = True
        return
    flush_deferred_side()
    main = torch.cuda.current_stream()
    if _Side.stream is None:
        _Side.stream = _make_side_stream()
    side = _Side.stream
    side.wait_stream(main)
    with torch.cuda.stream(side):
        if not _DEBUG_SKIP_SIDE:
            fn()
    for t in keep:
        if t is not None:
            t.record_stream(side)
    _Side.pending = True
    if _Side.listener is not None and params:
        _Side.listener(params)


def flush_deferred_side():
    """Launch what side_launch(defer=True) postponed, in order, behind everything enqueued so far on the current stream."""
    if _Side.deferred:
        items, _Side.deferred = _Side.deferred, []
        for fn, keep, params in items:
            side_launch(fn, keep, params)


def join_side():
    """Make the current stream wait for the side stream (before anything reads or rewrites .grad)."""
    flush_deferred_side()
    if _Side.pending:
        if _Side.stream is not None:
            torch.cuda.current_stream().wait_stream(_Side.stream)
        _Side.pending = False


def adjacent_view(ts):
    """The row-wise concatenation of `ts` as a VIEW, when the tensors sit back to back in one storage (the gates of a
    recurrent layer after optim.FlatParams packed them: pk_flat_groups), else None.  No autograd history."""
    if not ts:
        return None
    base = ts[0]
    if len(ts) == 1:
        return base.detach() if base.is_contiguous() else None
    ptr, rows = base.data_ptr(), 0
    for t in ts:
        if (not t.is_contiguous() or t.dtype != base.dtype or t.shape[1:] != base.shape[1:] or t.data_ptr() != ptr
                or t.untyped_storage().data_ptr() != base.untyped_storage().data_ptr()):
            return None
        ptr += t.numel() * t.element_size()
        rows += t.shape[0]
    if base.data_ptr() % 16 != 0:
        return None
    return base.detach().as_strided((rows,) + tuple(base.shape[1:]), base.stride())


def _accumulate_rows(params, rows):
    """params[i].grad += rows[i] (row blocks of a concatenated weight gradient); one launch when the gradients sit back
    to back in the flat buffer and the row blocks are slices of one matrix."""
    with torch.no_grad():
        gview = adjacent_view([q.grad for q in params])
        whole = adjacent_view(list(rows)) if gview is not None else None
        if whole is not None and whole.shape == gview.shape:
            gview.add_(whole)
            return
        for q, r in zip(params, rows):
            q.grad.add_(r)

# ----------------------------------------------------------------------------
# Linear:  y = x W^T + b        (nn.Linear; neural_networks.py:111, 139-148)
# ----------------------------------------------------------------------------
_XB_LAST = None  # (source tensor [kept alive: its memory cannot be reused], data_ptr, shape, strides, version, bf16 copy)


def _cvt_bf16_shared(x2):
    """bf16 copy of a Linear input, shared by consecutive layers that read the SAME tensor (the senone and the
    monophone head both take the last recurrent layer's output: one 64000 x 1100 conversion instead of two)."""
    global _XB_LAST
    key = (x2.data_ptr(), tuple(x2.shape), tuple(x2.stride()), x2._version)
    if _XB_LAST is not None and _XB_LAST[1] == key and x2.numel() >= (1 << 20):
        return _XB_LAST[2]
    xb = cvt_bf16(x2)
    _XB_LAST = (x2, key, xb) if x2.numel() >= (1 << 20) else None
    return xb


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _need_gpu(x, weight, bias)
        x2 = _rows2d(x)
        weight = weight.contiguous()
        M, K = x2.shape
        N = weight.shape[0]
        y = _new(M, N, like=x2)
        ctx.bf = bf16_mode()
        if ctx.bf:  # perf mode: operands rounded to bf16 once, saved in bf16 for backward
            xb, wb = _cvt_bf16_shared(x2), cvt_bf16(weight)
            gemm_bf16(M, N, K, xb, xb.shape[1], 1, wb, wb.shape[1], 1, y, N, bias=bias)
            ctx.save_for_backward(xb, wb)
        else:
            gemm(M, N, K, x2, x2.stride(0), 1, weight, 1, K, y, N, bias=bias)
            ctx.save_for_backward(x2, weight)
        ctx.dims = (M, N, K)
        ctx.has_bias = bias is not None
        ctx.in_shape = x.shape
        ctx.wparam = weight if isinstance(weight, torch.nn.Parameter) else None
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        M, N, K = ctx.dims
        if (not ctx.bf and dy.dim() == 2 and dy.stride(1) == 1 and dy.stride(0) % 4 == 0 and dy.stride(0) >= N
                and dy.data_ptr() % 16 == 0):
            dy2 = dy  # (a re-pitched gradient - LogSoftmaxFn.backward - is read at its own pitch)
        else:
            dy2 = _rows2d(dy.contiguous())
        ldy = dy2.stride(0)
        dx = dw = db = None
        if ctx.bf:
            dx, dw = _linear_bwd_bf16(ctx, cvt_bf16(dy2), x2, weight, dy2)
            if ctx.has_bias and ctx.needs_input_grad[2]:
                db = colsum(dy2)
            return dx, dw, db
        if ctx.needs_input_grad[0]:
            dx = _new(M, K, like=dy2)
            gemm(M, K, N, dy2, ldy, 1, weight, K, 1, dx, K)
            dx = dx.view(ctx.in_shape)
        if ctx.needs_input_grad[1]:
            wp = ctx.wparam
            sk = _splitk(_tiles(N, K), M)
            if M >= 4096 and wp is not None and wp.is_contiguous() and side_targets_any_ok([wp]):
                # (round 6) off the dependency chain in the exact-fp32 mode too: on the side stream, straight into the flat .grad
                # (postponed like the perf mode's heads: the product enters the side stream when the recurrence below launches
                # and leaves 112 CUs idle, instead of sharing the chip with this layer's dX GEMM on the dependency chain)
                side_launch(lambda: gemm(N, K, M, dy2, 1, ldy, x2, x2.stride(0), 1, wp.grad, K, beta=1.0, splitk=sk), (dy2, x2), [wp],
                            defer=True)
            else:
                dw = _new(N, K, like=dy2)
                gemm(N, K, M, dy2, 1, ldy, x2, x2.stride(0), 1, dw, K, splitk=sk)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dy2)
        return dx, dw, db


class _DxShare:
    """Several perf-mode output layers on ONE input (the senone and the monophone head both read the last recurrent
    layer's output): each head's backward produces a full dX (64 000 x 1100 fp32 = 282 MB at the benchmarked shape) and
    autograd adds them with an element-wise kernel (0.2 ms, 846 MB of traffic).  Instead the first head to run backward
    allocates dX and every later head's dX GEMM accumulates into it (beta = 1) and returns no gradient of its own.
    Keyed by the input's identity at forward time; the table is cleared whenever a new forward pass builds heads, i.e.
    after the backward pass that used it.  PK_EXPERIMENT head_dx_share=0 switches it off.

    Accumulating in place into a tensor autograd already holds is only sound while nobody else adds to that buffer, so
    the heads do not consume x itself: they consume ONE private alias of it (_ShareIn, made by the first head built on x
    and handed to every later one).  The alias has no other consumers, its gradient buffer therefore only ever sees
    the heads' contributions - the first as the shared tensor, the rest as None (with the concatenated operand below:
    all of them None, and _ShareIn.backward produces dX itself) - and whatever else reads x adds to x's own buffer,
    behind _ShareIn.backward, which also lets go of the shared dX."""
    on = _lib.experiment("head_dx_share", "1") != "0"
    epoch = 0
    epoch_fwd = 0
    table = {}  # key -> (epoch, dx tensor, contributors)
    alias = None  # (weakref to x, its private alias) of the head input seen last
    # Round 5: ONE input-gradient GEMM for all heads on one input.  The heads' dz (bf16) are column slices of one
    # buffer [M, sum of the heads' padded widths], the heads' weights rows of one [same sum, pitch] bf16 copy, and
    # dX = dz_cat . W_cat is launched once, by _ShareIn.backward - which autograd runs behind every head that takes part
    # in this backward pass (a head that does not: its slice is zeroed there).  At the BASELINE shape the monophone
    # head's own dX GEMM cost 0.21 ms for 7 GFLOP: it read and rewrote the 282 MB the senone head had just written.
    cat = {}       # key -> dict(epoch, heads=[dict(id, N, off, wb)], width, dz, done, dims, in_shape)
    wcat = {}      # (ids of the weight copies) -> persistent zero-padded bf16 buffer [width, pitch]


def _cat_register(key, ctx_id, N, wb, M):
    """Forward of a fused-cost head on the shared input `key`: reserve its columns of the concatenated operand."""
    if not (_DxShare.on and _lib.experiment("head_dx_cat", "1") != "0" and M >= 4096):
        return
    c = _DxShare.cat.get(key)
    if c is None or c["epoch"] != _DxShare.epoch_fwd:
        c = _DxShare.cat[key] = {"epoch": _DxShare.epoch_fwd, "heads": [], "width": 0, "dz": None, "done": set()}
        for k in [k for k, v in _DxShare.cat.items() if v["epoch"] != _DxShare.epoch_fwd]:
            del _DxShare.cat[k]
    c["heads"].append({"id": ctx_id, "N": N, "off": c["width"], "wb": wb})
    c["width"] += _up(N, 64)


def _cat_slot(ctx, M):
    """Backward of a head: (cat, head entry) when its dz belongs into a shared operand of at least two heads."""
    key = getattr(ctx, "dx_share", None)
    c = _DxShare.cat.get(key) if key is not None else None
    if c is None or len(c["heads"]) < 2:
        return None
    pitches = {h["wb"].shape[1] for h in c["heads"]}
    mine = [h for h in c["heads"] if h["id"] == id(ctx)]
    if len(pitches) != 1 or len(mine) != 1 or id(ctx) in c["done"]:
        return None
    if c["dz"] is None:
        c["dz"] = torch.empty(M, c["width"], device=c["heads"][0]["wb"].device, dtype=torch.bfloat16)
    return c, mine[0]


def _cat_finish(key, g):
    """_ShareIn.backward: every head that takes part has written its slice of dz_cat.  The heads returned NO gradient of
    their own (nothing uninitialised ever enters autograd's accumulation); this node produces dX = dz_cat . W_cat with
    ONE GEMM - plus g, whatever else back-propagated into the alias (a head's log-posteriors used by a second cost)."""
    c = _DxShare.cat.get(key)
    if c is None or not c["done"] or c["dz"] is None:
        return g
    heads, dz = c["heads"], c["dz"]
    for h in heads:  # a head whose cost did not reach the loss contributes nothing
        if h["id"] not in c["done"]:
            dz[:, h["off"]:h["off"] + _up(h["N"], 64)].zero_()
    wkey = tuple((h["wb"].data_ptr(), h["N"], h["off"]) for h in heads)
    wc = _DxShare.wcat.get(wkey)
    pitch = heads[0]["wb"].shape[1]
    if wc is None:
        _DxShare.wcat.clear()  # (one set of heads at a time: the copies are 4-9 MB)
        wc = _DxShare.wcat[wkey] = torch.zeros(c["width"], pitch, device=dz.device, dtype=torch.bfloat16)
    for h in heads:  # the weights moved this step: refresh their rows (pad rows stay zero)
        wc[h["off"]:h["off"] + h["N"]].copy_(h["wb"])
    M, K = c["dims"]
    dx2 = _new(M, K, like=dz)
    gemm_bf16(M, K, c["width"], dz, dz.shape[1], 1, wc, pitch, 0, dx2, K)
    del _DxShare.cat[key]
    dx = dx2.view(c["in_shape"])
    return dx if g is None else dx.add_(g)


class _ShareIn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.set_materialize_grads(False)  # (heads of a shared operand return None: no 282 MB zero tensor in their place)
        ctx.key = _dx_share_key(x)
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        if _DxShare.cat:
            g = _cat_finish(ctx.key, g)
        _DxShare.table.clear()  # every head in this backward pass has contributed: nothing keeps dX alive from here
        _DxShare.alias = None
        return g


def _head_input(x):
    """The tensor the fused heads on x consume (see _DxShare)."""
    if not (_DxShare.on and x.requires_grad and torch.is_grad_enabled() and x.is_contiguous()):
        return x
    a = _DxShare.alias
    if a is not None and a[0]() is x and a[1]._version == x._version:
        return a[1]
    xs = _ShareIn.apply(x)
    _DxShare.alias = (weakref.ref(x), xs)
    return xs


def _dx_share_key(x):
    return (x.data_ptr(), x._version, tuple(x.shape), tuple(x.stride()))


def _linear_bwd_bf16(ctx, dyb, xb, wb, like, cat=None):
    """dX and dW of a perf-mode Linear from the bf16 copy of the output gradient (ctx: dims, in_shape, wparam, and
    xseg = (nseg, seglen, segpad) when xb is a re-pitched twin of the input: wb is then the plain-pitch weight copy).
    dyb may be a column slice of a wider buffer (its row pitch is its stride).  cat: the shared operand of several heads
    on one input (_cat_slot) - dX is then ONE GEMM over all of them, launched by _ShareIn.backward."""
    M, N, K = ctx.dims
    xseg = getattr(ctx, "xseg", None)
    Kx = K if xseg is None else xseg[0] * xseg[2]  # columns of xb that carry data (pad columns are zero)
    dx = dw = None
    ldy = dyb.stride(0)
    if ctx.needs_input_grad[0] and cat is not None:
        c = cat[0]  # (dx stays None: _ShareIn.backward, which autograd runs behind every head, produces it)
        c["done"].add(id(ctx))
        c["dims"], c["in_shape"] = (M, K), ctx.in_shape
    elif ctx.needs_input_grad[0]:  # dx[m,k] = sum_n dy[m,n] W[n,k]: A k-contiguous, B = W is k-major
        key = getattr(ctx, "dx_share", None)
        shared = _DxShare.table.get(key) if key is not None else None
        if shared is not None and shared[0] == _DxShare.epoch and id(ctx) not in shared[2]:
            # another head on the same input already produced dX in this backward pass: accumulate into it
            # (a head that finds ITSELF among the contributors is in a second backward pass over a retained graph)
            gemm_bf16(M, K, N, dyb, ldy, 1, wb, wb.shape[1], 0, shared[1], K, beta=1.0)
            shared[2].add(id(ctx))
            dx = None
        else:
            dx = _new(M, K, like=like)
            gemm_bf16(M, K, N, dyb, ldy, 1, wb, wb.shape[1], 0, dx, K)
            if key is not None:
                _DxShare.table[key] = (_DxShare.epoch, dx, {id(ctx)})
            dx = dx.view(ctx.in_shape)
    if ctx.needs_input_grad[1]:  # dw[n,k] = sum_m dy[m,n] x[m,k]: both operands k-major
        wp = ctx.wparam
        side = M >= 4096 and wp is not None and wp.is_contiguous() and side_targets_ok([wp])
        sk = _splitk_bf(_tiles_bf(N, Kx), M)
        if xseg is None:
            if side:  # off the dependency chain: accumulate into the flat .grad on the side stream (beta = 1)
                side_launch(lambda: gemm_bf16(N, K, M, dyb, ldy, 0, xb, xb.shape[1], 0, wp.grad, K, beta=1.0,
                                              splitk=sk), (dyb, xb), [wp])
            elif M <= 512 and wp is not None and wp.is_contiguous() and direct_grads_ok([wp]):
                # small batch: straight into the flat .grad on this stream (no AccumulateGrad add behind it)
                gemm_bf16(N, K, M, dyb, ldy, 0, xb, xb.shape[1], 0, wp.grad, K, beta=1.0, splitk=sk)
            else:
                dw = _new(N, K, like=like)
                gemm_bf16(N, K, M, dyb, ldy, 0, xb, xb.shape[1], 0, dw, K, splitk=sk)
        else:  # the product comes out with the input's segment pitch: its segments are copied / added back
            nseg, seglen, segpad = xseg
            dwp = _new(N, Kx, like=like)

            def run():
                gemm_bf16(N, Kx, M, dyb, ldy, 0, xb, xb.shape[1], 0, dwp, Kx, splitk=sk)
                if side:
                    with torch.no_grad():
                        for s_ in range(nseg):
                            wp.grad[:, s_ * seglen:(s_ + 1) * seglen].add_(dwp[:, s_ * segpad:s_ * segpad + seglen])

            if side:  # (the input is a recurrent layer's output: next to that layer's backward recurrence)
                side_launch(run, (dyb, xb, dwp), [wp], defer=True)
            else:
                run()
                dw = torch.cat([dwp[:, s_ * segpad:s_ * segpad + seglen] for s_ in range(nseg)], 1)
    return dx, dw


def input_twin(x2):
    """(xb, xseg) when the 2-D Linear input x2 still carries the bf16 copy its producer published (a recurrent layer's
    exchange buffer: direction halves at a pitch of Hp), else None."""
    tw = getattr(x2, "_pk_twin", None)
    if tw is None:
        return None
    xb, xseg, version = tw
    if x2._version != version or x2.dim() != 2 or xb.shape[0] != x2.shape[0] or xseg[0] * xseg[1] != x2.shape[1]:
        return None
    return xb, xseg


class LinearLogSoftmaxFn(torch.autograd.Function):
    """log_softmax(x W^T + b) of a perf-mode output layer (neural_networks.py:139-148 with dnn_act = softmax and
    neither normalisation nor dropout on that layer: every shipped recipe's heads) as ONE autograd node, so that what
    travels between its two halves never takes the fp32 round trip through HBM:
      forward   the GEMM writes z at a 128-byte-aligned row pitch (1938 columns = 7752-byte rows would otherwise take
                the scalar-store epilogue: 0.50 -> 0.35 ms) and the LogSoftmax reads it there;
      backward  dz = dy - exp(y) rowsum(dy) is produced directly as the bf16 operand of the dX / dW GEMMs, with its
                column sums (the bias gradient) taken on the way (pk_logsoftmax_bwd_bf16).
    Same arithmetic as LinearFn + LogSoftmaxFn in perf mode (operands rounded to bf16 once, fp32 accumulation, bias
    gradient from the unrounded dz).  xb / wb: the bf16 copies of x / weight (made by the caller, shared with
    HeadNllFn)."""

    @staticmethod
    def forward(ctx, x, weight, bias, xb, wb, wb_plain, xseg):
        _need_gpu(x, weight, bias)
        lib = _lib.load()
        x2 = _rows2d(x)
        M, K = x2.shape
        N = weight.shape[0]
        ldz = _up(N, 32)
        z = torch.empty(M, ldz, device=x2.device, dtype=torch.float32)
        Kx = K if xseg is None else xseg[0] * xseg[2]
        gemm_bf16(M, N, Kx, xb, xb.shape[1], 1, wb, wb.shape[1], 1, z, ldz, bias=bias)
        y = _new(M, N, like=x2)
        # (with the arg-max position of every row: the cost behind this output - head_nll - then needs one gathered load
        # and one compare per row instead of a second pass over y)
        amax = torch.empty(M, device=x2.device, dtype=torch.int32)
        _lib.check(lib.pk_logsoftmax_fwd_ld_argmax(_stream(), _p(z), ldz, M, N, _p(y), _p(amax)), "pk_logsoftmax_fwd_ld_argmax")
        ctx.save_for_backward(xb, wb_plain, y)
        ctx.xseg = xseg
        ctx.dims = (M, N, K)
        ctx.has_bias = bias is not None
        ctx.in_shape = x.shape
        ctx.wparam = weight if isinstance(weight, torch.nn.Parameter) else None
        ctx.mark_non_differentiable(amax)
        return y, amax

    @staticmethod
    def backward(ctx, dy, _damax=None):
        lib = _lib.load()
        xb, wb, y = ctx.saved_tensors
        M, N, K = ctx.dims
        dy2 = dy.contiguous()
        ldb = _up(N, 64)
        dzb = torch.empty(M, ldb, device=y.device, dtype=torch.bfloat16)
        part = _new(int(lib.pk_logsoftmax_bwd_bf16_partial_floats(M, N)), like=y)
        db = _new(N, like=y)
        _lib.check(lib.pk_logsoftmax_bwd_bf16(_stream(), _p(dy2), _p(y), M, N, _p(dzb), ldb, _p(part), _p(db)),
                   "pk_logsoftmax_bwd_bf16")
        dx, dw = _linear_bwd_bf16(ctx, dzb, xb, wb, y)
        return dx, dw, (db if ctx.has_bias and ctx.needs_input_grad[2] else None), None, None, None, None


def linear_log_softmax_ok(x, weight):
    """The fused head covers perf mode, 2-D inputs and rows of up to 2048 classes."""
    return bf16_mode() and x.dim() == 2 and weight.shape[0] <= 2048


def linear_log_softmax(x, weight, bias=None):
    """-> log-posteriors y.  y carries what head_nll needs to put the mean-NLL cost of the recipe directly behind
    (x, weight, bias): `y._pk_head`.  When x is the output of a perf-mode recurrent layer, the bf16 copy that layer
    published is the GEMM operand (the weight copy is re-pitched to match): the 282 MB activation is not converted
    again."""
    with torch.no_grad():
        w = weight.contiguous()
        x2 = _rows2d(x)
        tw = input_twin(x2)
        wb_plain = weight_bf16(weight)
        if tw is None:
            xb, xseg, wb = _cvt_bf16_shared(x2), None, wb_plain
        elif tw[1][0] == 1 and tw[0].shape[1] == wb_plain.shape[1]:
            # one segment at the weight copy's own pitch (the bf16 copy an MLP layer published): nothing is re-pitched -
            # the plain path (one weight copy; the weight gradient goes straight into .grad, no segment copies)
            xb, xseg, wb = tw[0], None, wb_plain
        else:
            xb, xseg = tw
            wb = cvt_bf16(w, *xseg)
            assert wb.shape[1] == xb.shape[1]
    _DxShare.epoch += 1   # (a new forward pass: whatever the previous backward pass shared is history)
    _DxShare.table.clear()
    _DxShare.epoch_fwd = _DxShare.epoch  # heads registered behind this point belong to this forward pass (_cat_register)
    xs = _head_input(x)
    y, amax = LinearLogSoftmaxFn.apply(xs, weight if weight.is_contiguous() else w, bias, xb, wb, wb_plain, xseg)
    y._pk_head = (xs, weight, bias, xb, wb_plain, xseg, y._version, amax)
    return y


class HeadNllFn(torch.autograd.Function):
    """loss = NLLLoss()(y, lab) for y = linear_log_softmax(x, weight, bias) (utils.py:2361; core.py:631-642), as a node
    whose inputs are (x, weight, bias): the log-posteriors are the ones the head already computed (nothing is
    recomputed), and backward goes from the scalar straight to dz = (dloss / count) (exp(y) - onehot(lab)) in bf16 -
    the dense one-hot gradient of the cost (a 496 MB zero fill, a scatter and a second read at the BASELINE shape) is
    never formed.  Autograd runs y's own node only if something else back-propagates through y; the two paths then add
    up in (x, weight, bias) as they should.  The same forward pass counts the frame errors of the cost_err line."""

    @staticmethod
    def forward(ctx, x, weight, bias, y, lab, xb, wb, xseg, ignore_index, amax=None):
        lib = _lib.load()
        ctx.set_materialize_grads(False)  # (no zero tensor for the statistics output in backward)
        ctx.xseg = xseg
        M, N = y.shape
        lab = lab.contiguous()
        out4 = _new(4, like=y)
        loss = torch.empty((), device=y.device, dtype=torch.float32)  # the differentiable output: written by the same launch
        part = _new(int(lib.pk_nll_err_partial_floats(M)), like=y)
        # bad labels are counted IN PLACE into the device's persistent counter by the same launch (HIP-graph safe; the loss
        # of such a batch is NaN; the count is reported at the next point where the host waits for the GPU anyway)
        if amax is not None and _lib.experiment("head_argmax", "1") != "0":
            # the rows' arg-max positions came with y (pk_logsoftmax_fwd_ld_argmax): y is not read a second time
            _lib.check(lib.pk_nll_err_fwd_argmax(_stream(), _p(y), _p(lab), _p(amax), int(ignore_index), M, N, _p(part), _p(out4),
                                                 _p(loss), _p(label_check_counter(y.device))), "pk_nll_err_fwd_argmax")
        else:
            _lib.check(lib.pk_nll_err_fwd(_stream(), _p(y), _p(lab), int(ignore_index), M, N, _p(part), _p(out4), _p(loss),
                                          _p(label_check_counter(y.device))), "pk_nll_err_fwd")
        ctx.save_for_backward(xb, wb, y, lab, out4)
        ctx.dx_share = _dx_share_key(x) if (_DxShare.on and x.is_contiguous()) else None
        if ctx.dx_share is not None and x.requires_grad:
            _cat_register(ctx.dx_share, id(ctx), N, wb, M)  # (wb: the plain-pitch copy of the weight)
        ctx.dims = (M, N, x.shape[-1])
        ctx.has_bias = bias is not None
        ctx.in_shape = x.shape
        ctx.wparam = weight if isinstance(weight, torch.nn.Parameter) else None
        ctx.ignore_index = int(ignore_index)
        ctx.mark_non_differentiable(out4)
        return loss, out4

    @staticmethod
    def backward(ctx, dloss, _d4):
        lib = _lib.load()
        if dloss is None:
            return (None,) * 10
        xb, wb, y, lab, out4 = ctx.saved_tensors
        M, N, K = ctx.dims
        dl = dloss.contiguous().float()
        ldb = _up(N, 64)
        part = _new(int(lib.pk_logsoftmax_bwd_bf16_partial_floats(M, N)), like=y)
        db = _new(N, like=y)
        cnt = ctypes.c_void_p(out4.data_ptr() + 8)
        cat = _cat_slot(ctx, M) if ctx.needs_input_grad[0] else None
        if cat is not None:  # my columns of the operand all heads on this input share
            dz, off = cat[0]["dz"], cat[1]["off"]
            dzb = dz[:, off:off + ldb]
            _lib.check(lib.pk_nll_logsoftmax_bwd_bf16_p(_stream(), _p(y), _p(lab), _p(dl), cnt, ctx.ignore_index, M, N,
                                                        ctypes.c_void_p(dzb.data_ptr()), ldb, dz.shape[1], _p(part), _p(db)),
                       "pk_nll_logsoftmax_bwd_bf16_p")
        else:
            dzb = torch.empty(M, ldb, device=y.device, dtype=torch.bfloat16)
            _lib.check(lib.pk_nll_logsoftmax_bwd_bf16(_stream(), _p(y), _p(lab), _p(dl), cnt, ctx.ignore_index, M, N, _p(dzb),
                                                      ldb, _p(part), _p(db)), "pk_nll_logsoftmax_bwd_bf16")
        dx, dw = _linear_bwd_bf16(ctx, dzb, xb, wb, y, cat)
        return dx, dw, (db if ctx.has_bias and ctx.needs_input_grad[2] else None), None, None, None, None, None, None, None


def linear(x, weight, bias=None):
    return LinearFn.apply(x, weight, bias)


def colsum(g, g2=None):
    lib = _lib.load()
    M, N = g.shape
    out = _new(N, like=g)
    part = _new(int(lib.pk_bn_partial_floats(M, N)), like=g)
    _lib.check(lib.pk_colsum(_stream(), _p(g), _p(g2), g.stride(0), M, N, _p(part), _p(out)), "pk_colsum")
    return out


class _LabelCheck:
    bad = {}  # device -> persistent scalar: labels outside [0, classes) seen by head_nll since the last raise_if_bad_labels()


def label_check_counter(device):
    """The persistent per-device counter (allocate it BEFORE a HIP-graph capture: the captured in-place add then
    accumulates on every replay)."""
    dev = torch.empty(0, device=device).device
    if dev not in _LabelCheck.bad:
        _LabelCheck.bad[dev] = torch.zeros((), device=dev, dtype=torch.float32)
    assert _LabelCheck.bad[dev].dtype == torch.float32  # pk_nll_err_fwd* add into it in place through a `float*`
    return _LabelCheck.bad[dev]


def note_label_check(stats):
    """Accumulate the bad-label count of one head_nll call IN PLACE into the persistent device scalar (no host sync
    here; the loss of such a batch is NaN - pk_nll_err_fwd - and the count is reported at the next point where the host
    waits for the GPU anyway).  In place, so that a HIP-graph replay of the step keeps counting."""
    del stats  # (round 4: pk_nll_err_fwd accumulates the count itself - HeadNllFn passes the counter)


def raise_if_bad_labels():
    """Call after a host sync (core.run_nn_dp: once per chunk)."""
    n = 0
    for c in _LabelCheck.bad.values():
        n += int(float(c))
        c.zero_()
    if n > 0:
        raise _lib.PkError("cost_nll: %d label(s) outside [0, classes) in this chunk" % n)


def head_nll(y, lab, ignore_index=-100):
    """(loss, stats) of the mean-NLL cost on the output y of linear_log_softmax, or None when y is not one (any more).
    stats (device, 4 floats): loss, frame error rate, counted rows, labels outside [0, classes)."""
    head = getattr(y, "_pk_head", None)
    if head is None or y.dim() != 2 or lab.dim() != 1 or lab.shape[0] != y.shape[0] or lab.dtype != torch.int64:
        return None
    x, weight, bias, xb, wb, xseg, version, amax = head
    if y._version != version or not lab.is_cuda:
        return None
    loss, out4 = HeadNllFn.apply(x, weight, bias, y.detach(), lab, xb, wb, xseg, ignore_index, amax)
    return loss, out4


# ----------------------------------------------------------------------------
# BatchNorm1d(momentum=0.05) + activation + dropout mask, fused element-wise
# (neural_networks.py:139-148 MLP; :1546-1552 CNN after pooling)
# ----------------------------------------------------------------------------
def bn_stats(x2):
    lib = _lib.load()
    M, N = x2.shape
    mean, var = _new(N, like=x2), _new(N, like=x2)
    part = _new(int(lib.pk_bn_partial_floats(M, N)), like=x2)
    _lib.check(lib.pk_bn_stats(_stream(), _p(x2), x2.stride(0), M, N, _p(part), _p(mean), _p(var)), "pk_bn_stats")
    return mean, var


def bn_finalize(mean, var, gamma, beta, eps, running_mean=None, running_var=None, momentum=0.05, count=1.0):
    lib = _lib.load()
    N = mean.numel()
    scale, shift = torch.empty_like(mean), torch.empty_like(mean)
    _lib.check(lib.pk_bn_finalize(_stream(), N, _p(mean), _p(var), _p(gamma), _p(beta), eps, _p(scale), _p(shift),
                                  _p(running_mean), _p(running_var), momentum, float(count)), "pk_bn_finalize")
    return scale, shift


def bn_finalize_gates(mean, var, gamma, beta, eps, H, running_means, running_vars, batches, momentum, count):
    """bn_finalize for the concatenated gates of a recurrent layer in training mode: scale / shift, and the
    running statistics + num_batches_tracked of every gate's own BatchNorm1d updated by the same launch."""
    lib = _lib.load()
    G = len(running_means)
    scale, shift = torch.empty_like(mean), torch.empty_like(mean)
    arr = lambda ts: (ctypes.c_void_p * G)(*[t.data_ptr() for t in ts])
    rm, rv, nb = arr(running_means), arr(running_vars), arr(batches)
    _lib.check(lib.pk_bn_finalize_gates(_stream(), G, H, _p(mean), _p(var), _p(gamma), _p(beta), eps, _p(scale), _p(shift),
                                        rm, rv, nb, momentum, float(count)), "pk_bn_finalize_gates")
    return scale, shift


class NormActDropFn(torch.autograd.Function):
    """y = mask * act(BN(x))   (BN optional).  x: [M, N]."""

    @staticmethod
    def forward(ctx, x, gamma, beta, running_mean, running_var, use_bn, training, eps, momentum, act, mask):
        _need_gpu(x, gamma, beta, mask)
        lib = _lib.load()
        x2 = _rows2d(x)
        M, N = x2.shape
        scale = shift = mean = var = None
        if use_bn:
            if training:
                mean, var = bn_stats(x2)
                scale, shift = bn_finalize(mean, var, gamma, beta, eps, running_mean, running_var, momentum, M)
            else:
                mean, var = running_mean, running_var
                scale, shift = bn_finalize(mean, var, gamma, beta, eps)
        a = _new(M, N, like=x2)
        _lib.check(lib.pk_affine_act_fwd(_stream(), _p(x2), x2.stride(0), M, N, _p(scale), _p(shift), ACT[act], None,
                                         _p(a), N), "pk_affine_act_fwd")
        y = a
        if mask is not None:
            y = _new(M, N, like=x2)
            _lib.check(lib.pk_affine_act_fwd(_stream(), _p(a), N, M, N, None, None, 0, _p(mask), _p(y), N),
                       "pk_affine_act_fwd")
        if _Decisions.relu is not None and act == "relu":
            # test mode: backward takes the ReLU derivative from the saved OUTPUT (a > 0); give it another run's pattern
            pat = _Decisions.relu.pop(0).to(a.device).reshape(M, N)
            _Decisions.report.append(("relu", int(((a > 0) != pat).sum()), M * N))
            a = torch.where(pat, a.clamp_min(1e-30), torch.zeros_like(a))
        ctx.save_for_backward(x2, gamma, mean, var, a, mask, scale)
        ctx.cfg = (use_bn, training, eps, act)
        ctx.in_shape = x.shape
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, gamma, mean, var, a, mask, scale = ctx.saved_tensors
        use_bn, training, eps, act = ctx.cfg
        M, N = x2.shape
        dy2 = _rows2d(dy.contiguous())
        g = _new(M, N, like=x2)
        _lib.check(lib.pk_act_bwd(_stream(), _p(dy2), _p(a), _p(mask), ACT[act], M * N, _p(g)), "pk_act_bwd")
        dgamma = dbeta = None
        if not use_bn:
            return g.view(ctx.in_shape), None, None, None, None, None, None, None, None, None, None
        part = _new(int(lib.pk_bn_partial_floats(M, N)), like=x2)
        sum_g, sum_gx = _new(N, like=x2), _new(N, like=x2)
        _lib.check(lib.pk_bn_bwd_reduce(_stream(), _p(g), None, N, _p(x2), x2.stride(0), M, N, _p(mean), _p(var), eps,
                                        _p(part), _p(sum_g), _p(sum_gx)), "pk_bn_bwd_reduce")
        dgamma, dbeta = sum_gx, sum_g
        dx = _new(M, N, like=x2)
        if training:
            _lib.check(lib.pk_bn_bwd_apply(_stream(), _p(g), None, N, _p(x2), x2.stride(0), M, N, _p(mean), _p(var), eps,
                                           _p(gamma), _p(sum_g), _p(sum_gx), float(M), _p(dx), N), "pk_bn_bwd_apply")
        else:  # running statistics are constants: dx = g * gamma * invstd
            zero = torch.zeros_like(scale)
            _lib.check(lib.pk_affine_act_fwd(_stream(), _p(g), N, M, N, _p(scale), _p(zero), 0, None, _p(dx), N),
                       "pk_affine_act_fwd")
        return dx.view(ctx.in_shape), dgamma, dbeta, None, None, None, None, None, None, None, None


class LinearBnActFn(torch.autograd.Function):
    """drop(act(bn(x W^T + b))) of ONE perf-mode MLP layer on a batch of up to 128 rows as one launch
    (pk_linear_bn_act_bf16: the whole batch sits in one row tile of the GEMM, so the BatchNorm statistics are a reduction
    inside the workgroup that owns the columns).  Forward replaces LinearFn + NormActDropFn + the next layer's bf16
    conversion (seven launches); backward runs the same kernels those two nodes run (activation / BatchNorm backward on
    the saved z, a, statistics; dX / dW / db from the bf16 copies).  y carries its bf16 twin for the next layer."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, running_mean, running_var, eps, momentum, act, mask, xb):
        _need_gpu(x, weight, bias, gamma, beta, mask)
        ctx.set_materialize_grads(False)  # no zero tensor for the (non-differentiable) bf16 twin in backward: one fill launch less
        lib = _lib.load()
        x2 = _rows2d(x)
        M, K = x2.shape
        N = weight.shape[0]
        wb = weight_bf16(weight)
        if xb is None:
            xb = cvt_bf16(x2)
        z, a = _new(M, N, like=x2), _new(M, N, like=x2)
        y = _new(M, N, like=x2) if mask is not None else a
        yb = torch.empty(M, N, device=x2.device, dtype=torch.bfloat16)
        mean, var = _new(N, like=x2), _new(N, like=x2)
        sk = int(lib.pk_gemm_bf16_small_splitk(M, N, K)) if _SMALL_SPLITK else 1
        if sk > 1:  # the product split along K over the whole chip, then the layer epilogue from the slabs
            ws = torch.empty(sk * M * N, device=x2.device, dtype=torch.float32)
            _lib.check(lib.pk_linear_bn_act_bf16_sk(_stream(), M, N, K, _p(xb), xb.shape[1], _p(wb), wb.shape[1], _p(bias),
                                                    _p(gamma), _p(beta), float(eps), float(momentum), _p(running_mean),
                                                    _p(running_var), ACT[act], _p(mask), _p(z), _p(a),
                                                    _p(y) if mask is not None else None, _p(yb), N, _p(mean), _p(var), sk, _p(ws)),
                       "pk_linear_bn_act_bf16_sk")
        else:
            _lib.check(lib.pk_linear_bn_act_bf16(_stream(), M, N, K, _p(xb), xb.shape[1], _p(wb), wb.shape[1], _p(bias), _p(gamma),
                                                 _p(beta), float(eps), float(momentum), _p(running_mean), _p(running_var),
                                                 ACT[act], _p(mask), _p(z), _p(a), _p(y) if mask is not None else None, _p(yb), N,
                                                 _p(mean), _p(var)), "pk_linear_bn_act_bf16")
        if _Decisions.relu is not None and act == "relu":  # test mode (see NormActDropFn)
            pat = _Decisions.relu.pop(0).to(a.device).reshape(M, N)
            _Decisions.report.append(("relu", int(((a > 0) != pat).sum()), M * N))
            a = torch.where(pat, a.clamp_min(1e-30), torch.zeros_like(a))
        ctx.save_for_backward(xb, wb, z, gamma, mean, var, a, mask)
        ctx.cfg = (float(eps), act)
        ctx.dims = (M, N, K)
        ctx.has_bias = bias is not None
        ctx.in_shape = x.shape
        ctx.wparam = weight if isinstance(weight, torch.nn.Parameter) else None
        ctx.bnparams = [gamma, beta] if (isinstance(gamma, torch.nn.Parameter) and isinstance(beta, torch.nn.Parameter)) else None
        ctx.bparam = bias if isinstance(bias, torch.nn.Parameter) else None
        ctx.mark_non_differentiable(yb)
        return y, yb

    @staticmethod
    def backward(ctx, dy, _dyb):
        lib = _lib.load()
        if dy is None:
            return (None,) * 12
        xb, wb, z, gamma, mean, var, a, mask = ctx.saved_tensors
        eps, act = ctx.cfg
        M, N, K = ctx.dims
        dy2 = _rows2d(dy.contiguous())
        sum_g, sum_gx = _new(N, like=z), _new(N, like=z)
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        if lib.pk_bn_act_bwd_small_covers(M, N) == 1 and _lib.experiment("mlp_fused_bwd", "1") != "0":
            # one launch: activation / mask backward, both BatchNorm reductions, BatchNorm backward, the bf16 operand
            direct = ctx.bnparams is not None and direct_grads_ok(ctx.bnparams)
            dzb = torch.empty(M, _up(N, 64), device=z.device, dtype=torch.bfloat16)
            # the bias gradient (column sums of dz) comes out of the same launch: into the flat .grad when that is allowed,
            # else as a tensor for autograd
            direct_b = need_db and ctx.bparam is not None and direct_grads_ok([ctx.bparam])
            db = _new(N, like=z) if (need_db and not direct_b) else None
            _lib.check(lib.pk_bn_act_bwd_small(_stream(), _p(dy2), _p(a), _p(mask), ACT[act], _p(z), _p(mean), _p(var), eps,
                                               _p(gamma), M, N, _p(dzb), dzb.shape[1], None, _p(sum_g), _p(sum_gx),
                                               _p(ctx.bnparams[1].grad) if direct else None,
                                               _p(ctx.bnparams[0].grad) if direct else None, _p(db),
                                               _p(ctx.bparam.grad) if direct_b else None), "pk_bn_act_bwd_small")
            dx, dw = _linear_bwd_bf16(ctx, dzb, xb, wb, z)
            if direct:
                return dx, dw, db, None, None, None, None, None, None, None, None, None
            return dx, dw, db, sum_gx, sum_g, None, None, None, None, None, None, None
        g = _new(M, N, like=z)
        _lib.check(lib.pk_act_bwd(_stream(), _p(dy2), _p(a), _p(mask), ACT[act], M * N, _p(g)), "pk_act_bwd")
        part = _new(int(lib.pk_bn_partial_floats(M, N)), like=z)
        _lib.check(lib.pk_bn_bwd_reduce(_stream(), _p(g), None, N, _p(z), N, M, N, _p(mean), _p(var), eps, _p(part), _p(sum_g),
                                        _p(sum_gx)), "pk_bn_bwd_reduce")
        dz = _new(M, N, like=z)
        _lib.check(lib.pk_bn_bwd_apply(_stream(), _p(g), None, N, _p(z), N, M, N, _p(mean), _p(var), eps, _p(gamma), _p(sum_g),
                                       _p(sum_gx), float(M), _p(dz), N), "pk_bn_bwd_apply")
        dx, dw = _linear_bwd_bf16(ctx, cvt_bf16(dz), xb, wb, dz)
        db = colsum(dz) if need_db else None
        return dx, dw, db, sum_gx, sum_g, None, None, None, None, None, None, None


def linear_bn_act_ok(x, weight, training, use_bn, act):
    """The one-launch MLP layer covers perf mode, training-mode BatchNorm, 2-D batches of up to 128 rows."""
    return (bf16_mode() and training and use_bn and x.dim() == 2 and act in ACT and x.is_cuda
            and _lib.experiment("mlp_fused", "1") != "0"
            and _lib.load().pk_linear_bn_act_bf16_covers(x.shape[0], weight.shape[0], x.shape[1]) == 1)


def linear_bn_act(x, weight, bias, bn, act, mask, count=True):
    """-> y = mask * act(bn(x W^T + b)); y._pk_yb = its bf16 copy (what the next layer's GEMM reads)."""
    twin = getattr(x, "_pk_yb", None)
    xb = twin[0] if (twin is not None and twin[1] == x._version and twin[0].shape[0] == x.shape[0]) else None
    if count:
        with torch.no_grad():
            bn.num_batches_tracked += 1
    y, yb = LinearBnActFn.apply(x, weight, bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.eps, bn.momentum,
                                act, mask, xb)
    y._pk_yb = (yb, y._version)
    if yb.shape[1] % 64 == 0:  # also in the form the output layers look for (input_twin): one segment, no re-pitching
        y._pk_twin = (yb, (1, yb.shape[1], yb.shape[1]), y._version)
    return y


def norm_act_drop(x, bn, use_bn, training, act, mask=None, eps=None):
    """bn: an nn.BatchNorm1d used as a parameter container (or None)."""
    if use_bn:
        if training:
            with torch.no_grad():
                bn.num_batches_tracked += 1
        return NormActDropFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, True, training,
                                   bn.eps if eps is None else eps, bn.momentum, act, mask)
    return NormActDropFn.apply(x, None, None, None, None, False, training, 0.0, 0.0, act, mask)


# ----------------------------------------------------------------------------
# LayerNorm of the reference (neural_networks.py:23-33)
# ----------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _need_gpu(x, gamma, beta)
        lib = _lib.load()
        F_ = gamma.numel()
        x2 = x.contiguous().view(-1, F_)
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean, rinv = _new(rows, like=x2), _new(rows, like=x2)
        g = gamma.contiguous().view(-1)
        b = beta.contiguous().view(-1)
        _lib.check(lib.pk_layernorm_fwd(_stream(), _p(x2), rows, F_, _p(g), _p(b), eps, _p(y), _p(mean), _p(rinv)),
                   "pk_layernorm_fwd")
        ctx.save_for_backward(x2, g, mean, rinv)
        ctx.eps = eps
        ctx.shapes = (x.shape, gamma.shape)
        return y.view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, g, mean, rinv = ctx.saved_tensors
        rows, F_ = x2.shape
        dy2 = dy.contiguous().view(rows, F_)
        dx, dgx = torch.empty_like(x2), torch.empty_like(x2)
        _lib.check(lib.pk_layernorm_bwd(_stream(), _p(dy2), _p(x2), rows, F_, _p(g), _p(mean), _p(rinv), ctx.eps, _p(dx),
                                        _p(dgx)), "pk_layernorm_bwd")
        xs, gs = ctx.shapes
        return dx.view(xs), colsum(dgx).view(gs), colsum(dy2).view(gs), None


def layer_norm(x, gamma, beta, eps=1e-6):
    return LayerNormFn.apply(x, gamma, beta, eps)


class SincBankFn(torch.autograd.Function):
    """The band-pass bank of SincConv (neural_networks.py:1789-1800) as one launch each way (pk_sinc.hip): filters
    [N, 1, K] from low_hz_ / band_hz_ [N, 1].  Replaces ~15 forward and ~25 backward element-wise torch launches of a
    3 ms step; backward is analytic (d lowpass(c)[k] / dc = 2 cos(2 pi c j_k))."""

    @staticmethod
    def forward(ctx, low_hz, band_hz, n_, window, sample_rate, min_low, min_band):
        _need_gpu(low_hz, band_hz, n_, window)
        lib = _lib.load()
        N, K = low_hz.shape[0], window.numel()
        lo, ba = low_hz.contiguous().view(-1), band_hz.contiguous().view(-1)
        n1, w1 = n_.contiguous().view(-1).float(), window.contiguous().view(-1).float()
        filt = _new(N, 1, K, like=lo)
        mx = _new(N, like=lo)
        ks = torch.empty(N, device=lo.device, dtype=torch.int32)
        _lib.check(lib.pk_sinc_bank_fwd(_stream(), _p(lo), _p(ba), _p(n1), _p(w1), N, K, float(sample_rate), float(min_low),
                                        float(min_band), _p(filt), _p(mx), ctypes.c_void_p(ks.data_ptr())), "pk_sinc_bank_fwd")
        ctx.save_for_backward(lo, ba, n1, w1, mx, ks)
        ctx.cfg = (N, K, float(sample_rate), float(min_low), float(min_band), low_hz.shape, band_hz.shape)
        return filt

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        lo, ba, n1, w1, mx, ks = ctx.saved_tensors
        N, K, sr, min_low, min_band, s_lo, s_ba = ctx.cfg
        g = g.contiguous()
        dlow, dband = _new(N, like=lo), _new(N, like=lo)
        _lib.check(lib.pk_sinc_bank_bwd(_stream(), _p(g), _p(lo), _p(ba), _p(n1), _p(w1), _p(mx), ctypes.c_void_p(ks.data_ptr()), N,
                                        K, sr, min_low, min_band, _p(dlow), _p(dband)), "pk_sinc_bank_bwd")
        return dlow.view(s_lo), dband.view(s_ba), None, None, None, None, None


def sinc_bank(low_hz, band_hz, n_, window, sample_rate, min_low, min_band):
    return SincBankFn.apply(low_hz, band_hz, n_, window, sample_rate, min_low, min_band)


class LnLastActDropFn(torch.autograd.Function):
    """drop(act(LayerNorm(z))) behind a conv layer's max-pool as ONE launch each way (pk_ln_last_act_drop_*), the CNN /
    SincNet flavour of the reference's LayerNorm (features [C, L], statistics over the last dim:
    neural_networks.py:1510-1512 + :1546-1552, :1639-1641 + :1655-1661).  Replaces layer_norm_last (two fills, the
    normalisation, two broadcast launches) + NormActDropFn (activation, mask) - six launches forward, about ten backward -
    with the same arithmetic in the same order."""

    @staticmethod
    def forward(ctx, z, gamma, beta, eps, act, mask):
        _need_gpu(z, gamma, beta, mask)
        lib = _lib.load()
        z = z.contiguous()
        B, C, L = z.shape
        g, b = gamma.contiguous(), beta.contiguous()
        assert tuple(g.shape) == (C, L) and tuple(b.shape) == (C, L), (tuple(g.shape), (C, L))
        a = torch.empty_like(z)
        y = torch.empty_like(z) if mask is not None else None
        if mask is not None:
            mask = mask.contiguous()
        mean, rinv = _new(B * C, like=z), _new(B * C, like=z)
        _lib.check(lib.pk_ln_last_act_drop_fwd(_stream(), _p(z), B, C, L, _p(g), _p(b), float(eps), ACT[act], _p(mask), _p(a),
                                               _p(y), _p(mean), _p(rinv)), "pk_ln_last_act_drop_fwd")
        out = y if mask is not None else a
        if _Decisions.relu is not None and act == "relu":  # test mode (see NormActDropFn)
            pat = _Decisions.relu.pop(0).to(a.device).reshape(B, C, L)
            _Decisions.report.append(("relu", int(((a > 0) != pat).sum()), a.numel()))
            a = torch.where(pat, a.clamp_min(1e-30), torch.zeros_like(a))
        ctx.save_for_backward(z, g, mean, rinv, a, mask)
        ctx.cfg = (float(eps), act)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        z, g, mean, rinv, a, mask = ctx.saved_tensors
        eps, act = ctx.cfg
        B, C, L = z.shape
        dy = dy.contiguous()
        dz = torch.empty_like(z)
        pg = _new(B, 2 * C * L, like=z)
        _lib.check(lib.pk_ln_last_act_drop_bwd(_stream(), _p(dy), _p(z), _p(a), _p(mask), B, C, L, _p(g), _p(mean), _p(rinv), eps,
                                               ACT[act], _p(dz), _p(pg)), "pk_ln_last_act_drop_bwd")
        both = colsum(pg)
        return dz, both[:C * L].view(C, L), both[C * L:].view(C, L), None, None, None


def ln_last_act_drop(z, gamma, beta, eps, act, mask=None):
    return LnLastActDropFn.apply(z, gamma, beta, eps, act, mask)


def layer_norm_last(x, gamma, beta, eps=1e-6):
    """LayerNorm whose affine parameters are [C, L] but whose statistics run over
    the last dim only (the CNN/SincNet flavour, neural_networks.py:1510-1512):
    normalise rows of length L without affine, then apply gamma/beta with torch
    broadcasting (tiny element-wise plumbing)."""
    L = x.shape[-1]
    ones = torch.ones(L, device=x.device)
    zeros = torch.zeros(L, device=x.device)
    xn = LayerNormFn.apply(x, ones, zeros, eps)
    return gamma * xn + beta


# ----------------------------------------------------------------------------
# LogSoftmax(dim=1)  ('softmax' activation, neural_networks.py:53-54)
# ----------------------------------------------------------------------------
class LogSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _need_gpu(x)
        lib = _lib.load()
        x2 = x.contiguous()
        rows, N = x2.shape
        y = torch.empty_like(x2)
        _lib.check(lib.pk_logsoftmax_fwd(_stream(), _p(x2), rows, N, _p(y)), "pk_logsoftmax_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (y,) = ctx.saved_tensors
        rows, N = y.shape
        dy2 = dy.contiguous()
        if N % 4 != 0 and N <= 2048 and rows >= 4096:
            # a row-streaming head whose class count is not a multiple of 4 (1938 senones): the gradient is written at a pitch
            # of N rounded up to 4, so that the dX / dW GEMMs of the Linear in front read 16-byte aligned rows (the LDS-DMA
            # form of pk_gemm: 74 -> ~105 TFLOP/s on those two products); LinearFn.backward takes the view as it is
            ld = _up(N, 4)
            buf = torch.empty(rows, ld, device=y.device, dtype=torch.float32)
            _lib.check(lib.pk_logsoftmax_bwd_ld(_stream(), _p(dy2), _p(y), rows, N, _p(buf), ld), "pk_logsoftmax_bwd_ld")
            return buf[:, :N]
        dx = torch.empty_like(y)
        _lib.check(lib.pk_logsoftmax_bwd(_stream(), _p(dy2), _p(y), rows, N, _p(dx)), "pk_logsoftmax_bwd")
        return dx


def log_softmax(x):
    if x.dim() != 2:
        raise _lib.PkError("log_softmax expects a 2-D (rows, classes) input as the reference's LogSoftmax(dim=1)")
    return LogSoftmaxFn.apply(x)


# ----------------------------------------------------------------------------
# One recurrent layer: projections + (BatchNorm) + time loop, both directions.
# neural_networks.py:412-481 (LSTM), 589-653 (GRU), 1092-1153 (liGRU), ...
# ----------------------------------------------------------------------------
def ln_persistent_ok(cell, H):
    """Per-step LayerNorm of h_t inside the persistent time loop: every cell in perf mode (second-generation kernels); in
    fp32 liGRU / RNN on the second-generation kernels and LSTM / GRU / minimalGRU on the fourth (LSTM's first-generation fp32
    kernels do not have it).  PK_EXPERIMENT rec_ln_persist=0 sends such layers back to the step-wise algorithm."""
    if _lib.experiment("rec_ln_persist", "1") == "0" or H < 2:
        return False
    if bf16_mode():
        return cell in ("liGRU", "RNN", "LSTM", "GRU", "minimalGRU")
    if cell in ("LSTM", "GRU", "minimalGRU"):
        return _lib.experiment("rec_f32_gen4", "1")[:1] != "0"
    return cell in ("liGRU", "RNN") and _lib.experiment("rec_f32_gen", "")[:1] != "1"


def choose_rec_algo(cell, H, use_ln):
    want = settings.rec_algo
    # exact fp32: LSTM / GRU / minimalGRU run on the fourth-generation persistent kernels, with or without per-step LayerNorm
    # (pk_rec_persist4_f32.hip; PK_EXPERIMENT rec_f32_gen4=0 keeps the first-generation LSTM kernels / the step-wise GRU -
    # the library reads the same switch, pk_rec4f_covers)
    gen4 = (not bf16_mode() and cell in ("LSTM", "GRU", "minimalGRU") and _lib.experiment("rec_f32_gen4", "1")[:1] != "0"
            and (not use_ln or ln_persistent_ok(cell, H)))
    ok = cell in ("liGRU", "RNN", "LSTM") and H <= 576 and (not use_ln or ln_persistent_ok(cell, H))
    if cell in ("GRU", "minimalGRU"):  # bf16 two-phase kernels on this (general) path: the layers that normalise h_t
        ok = H <= 576 and ((use_ln and ln_persistent_ok(cell, H)) or gen4)
    # the first-generation exact-fp32 kernels exchange pairs of fp32 values: LSTM without generation 4, liGRU / RNN when
    # PK_EXPERIMENT rec_f32_gen=1 keeps them (the library reads the same switch, pk_rec_persist.hip::use_gen2_f32)
    if not bf16_mode() and not gen4 and (cell == "LSTM" or _lib.experiment("rec_f32_gen", "")[:1] == "1"):
        ok = ok and H % 2 == 0
    if want == "persistent":
        if not ok:
            raise _lib.PkError("persistent recurrence does not cover cell=%s H=%d laynorm=%s" % (cell, H, use_ln))
        return REC_PERSISTENT
    if want == "stepwise":
        return REC_STEPWISE
    return REC_PERSISTENT if ok else REC_STEPWISE


_DU_SPLITK = 16  # pk_rec.hip DU_SPLITK: the same split of the reduction (pk_gemm clamps it to 32-deep slices), so the same sums


def _gemm_km_f32(M, N, K, A, a_ld, bflat, b_off, b_ld, oflat, overwrite, realign=True):
    """oflat[M, N] (+)= A^T . Bm in exact fp32, both operands k-major (the dU products): A [K][a_ld], Bm = the N columns of
    bflat's [K][b_ld] rows that start at element b_off.  The direction halves of a layer output / the slots of the saved
    state start at multiples of H = 550 floats: 8-byte aligned.  The LDS-DMA form of the kernel wants 16-byte aligned
    pieces (pk_gemm.hip: 41 -> ~90 TFLOP/s on these shapes), so such a product is taken over the columns from the
    aligned boundary below b_off on (`shift` extra columns: 0.4 % more work) into a temporary whose first `shift` columns
    are dropped - the same sums in the same order, added to oflat by the same fp32 add the kernel's beta = 1 does.
    realign = False keeps the product on the register-staged kernel: next to a fourth-generation recurrence (LSTM / GRU /
    minimalGRU, whose backward pass is bound by its polls through the XCD's L2) the faster product costs the recurrence
    more than it returns - timit_lstm 120.8 ms per step without, 123.6 with (rec4_bwd 5.68 -> 6.24 ms per launch,
    profiles/r06_fp32_gemm_dma.json) - while the Li-GRU's second-generation kernels gain (74.4 -> 71.6)."""
    Bm = bflat[b_off:]
    shift = (Bm.data_ptr() % 16) // 4
    if not realign or shift == 0 or b_ld % 4 != 0 or a_ld % 4 != 0 or b_off < shift or _lib.experiment("f32_du_shift", "1") == "0":
        gemm(M, N, K, A, 1, a_ld, Bm, b_ld, 1, oflat, N, beta=0.0 if overwrite else 1.0, splitk=_DU_SPLITK, prec="fp32")
        return
    Nw = N + shift
    tmp = torch.empty(M, Nw, device=oflat.device, dtype=torch.float32)
    gemm(M, Nw, K, A, 1, a_ld, bflat[b_off - shift:], b_ld, 1, tmp, Nw, beta=0.0, splitk=_DU_SPLITK, prec="fp32")
    o2 = oflat[:M * N].view(M, N)
    with torch.no_grad():
        if overwrite:
            o2.copy_(tmp[:, shift:])
        else:
            o2.add_(tmp[:, shift:])


def _deferred_dU_f32(cell, T, B, ndir, H, G, NS, Y, S, dP2, out, accumulate):
    """dU[G*H, H] (+)= sum over directions and steps of dgate_t^T . (vector that fed U_g at step t) in exact fp32: what
    pk_rec_bwd does behind its recurrence when it is handed a dU (pk_rec.hip::deferred_dU), as separate launches so that
    they can run on the side stream.  out: [G*H, H] fp32 - a temporary (accumulate = False) or the gates' view of the flat
    gradient (accumulate = True: every product adds)."""
    TB, GH, YH = T * B, G * H, ndir * H
    two_phase = cell in ("GRU", "minimalGRU")
    Gh = G - 1 if two_phase else G
    Kh = (T - 1) * B
    dflat, yflat, sflat, oflat = dP2.reshape(-1), Y.reshape(-1), S.reshape(-1), out.reshape(-1)
    first = not accumulate
    realign = cell in ("liGRU", "RNN") or _lib.experiment("f32_du_shift", "") == "all"  # (see _gemm_km_f32)
    if Kh == 0 and not accumulate:
        out[:Gh * H].zero_()
    for d in range(ndir if Kh > 0 else 0):
        # rows whose previous state exists: dir 0 -> ts >= 1 (h at ts-1); dir 1 -> ts <= T-2 (h at ts+1)
        A = dflat[(d * TB + (0 if d else B)) * GH:]
        _gemm_km_f32(Gh * H, H, Kh, A, GH, yflat, (B if d else 0) * YH + d * H, YH, oflat, first, realign)
        first = False
    if two_phase:  # candidate gate: dU_h = sum dA^T . (r*h) or (z*h), saved in S, same row
        slot = 3 if cell == "GRU" else 2
        for d in range(ndir):
            A = dflat[d * TB * GH + Gh * H:]
            _gemm_km_f32(H, H, TB, A, GH, sflat, d * TB * NS * H + slot * H, NS * H, oflat[Gh * H * H:], not (accumulate or d),
                         realign)


def _deferred_dU_bf16(lib, cell, T, B, ndir, H, G, Y, S, dP2, dU, Yb=None, dGb=None, Xb=None):
    """dU[G*H, H] = sum over directions and steps of dgate_t^T . (vector that fed U_g at step t), as
    k-major x k-major bf16 GEMMs over the (T-1)*B rows (Appendix C of SURVEY.md; pk_rec.hip deferred_dU
    is the fp32 twin).  Y's two direction halves are re-pitched to a multiple of 8 so that the reversed
    half starts 16-byte aligned."""
    TB, GH = T * B, G * H
    two_phase = cell in ("GRU", "minimalGRU")
    Gh = G - 1 if two_phase else G
    Kh = (T - 1) * B
    Hp = _up(H, 8)
    if Yb is None:  # (the persistent bf16 kernels hand both buffers over in exactly this layout)
        Yb = cvt_bf16(Y.view(TB, ndir * H), ndir, H, Hp)
    Yp = Yb.shape[1]
    if dGb is None:
        dGb = cvt_bf16(dP2.view(ndir * TB, GH), G, H, Hp)  # gate g of a row starts at column g*Hp (16-byte aligned)
    Gp = dGb.shape[1]
    # one GEMM per direction over all gates fed by h_{t-1}: with the gates re-pitched to Hp the output has
    # Hp - H dead rows per gate (zero gate gradients), dropped when the rows are copied into dU
    Mp = Gh * H if Hp == H else Gh * Hp
    out = dU[:Gh * H] if Hp == H else _new(Mp, H, like=dU)
    if Kh == 0:
        dU[:Gh * H].zero_()
    else:
        for d in range(ndir):
            # rows whose previous state exists: dir 0 -> ts >= 1 (h at ts-1); dir 1 -> ts <= T-2 (h at ts+1)
            a_off = (d * TB + (0 if d else B)) * Gp
            b_off = (B if d else 0) * Yp + d * Hp
            gemm_bf16(Mp, H, Kh, (dGb, a_off), Gp, 0, (Yb, b_off), Yp, 0, out, H, beta=0.0 if d == 0 else 1.0,
                      splitk=_splitk_bf(_tiles_bf(Mp, H), Kh))
        if Hp != H:
            for g in range(Gh):
                dU[g * H:(g + 1) * H].copy_(out[g * Hp:g * Hp + H])
    if two_phase:  # candidate gate: dU_h = sum dA^T . (r*h) or (z*h), saved in S
        slot = 3 if cell == "GRU" else 2
        for d in range(ndir):
            if Xb is not None:  # the persistent kernels publish r*h / z*h as bf16, direction d at column d*Hp
                gh, gh_ld = (Xb, d * Hp), Xb.shape[1]
            else:
                gh = cvt_bf16(S[d][:, slot * H:(slot + 1) * H])
                gh_ld = gh.shape[1]
            gemm_bf16(H, H, TB, (dGb, d * TB * Gp + Gh * Hp), Gp, 0, gh, gh_ld, 0, dU[Gh * H:], H,
                      beta=0.0 if d == 0 else 1.0, splitk=_splitk_bf(_tiles_bf(H, H), TB))


LN_EPS = 1e-6  # the reference's LayerNorm(features, eps=1e-6), neural_networks.py:23-27


class RecLayerFn(torch.autograd.Function):
    """y, bn_mean, bn_var = f(x, Wcat, bcat, Ucat, gamma, beta, mask)

    x [T,B,D]; Wcat [G*H, D]; bcat [G*H] or None; Ucat [G*H, H]; gamma/beta
    [G*H] or None (BatchNorm on the projections); mask [R,H] or None.
    """

    @staticmethod
    def forward(ctx, x, Wcat, bcat, Ucat, gamma, beta, running_mean, running_var, mask, ln_gamma, ln_beta, cfg):
        _need_gpu(x, Wcat, bcat, Ucat, gamma, beta, mask, ln_gamma, ln_beta)
        lib = _lib.load()
        cell, act, H, bidir, use_bn, training, eps, momentum, mask_scalar = cfg[:9]
        # (round 6) the gates' parameters and whether their gradients bypass autograd: exact-fp32 weight-gradient GEMMs on
        # the side stream, accumulating into the flat .grad (decided by nn._Recurrent.forward, which then hands Wcat / Ucat
        # over detached)
        ctx.wparams, ctx.uparams = (cfg[9], cfg[10]) if len(cfg) > 10 else (None, None)
        ctx.side_w, ctx.side_u = (bool(cfg[11]), bool(cfg[12])) if len(cfg) > 12 else (False, False)
        T, B, D = x.shape
        x2 = _rows2d(x)
        G = lib.pk_rec_num_gates(CELL[cell])
        NS = lib.pk_rec_num_saved(CELL[cell])
        ndir = 2 if bidir else 1
        TB, GH = T * B, G * H
        Wcat = Wcat.contiguous()
        Ucat = Ucat.contiguous()
        # K1: input projections for all steps at once, on the NON-duplicated batch (the reversed
        # half of the reference's cat([x, flip(x)]) is the same rows read backwards in time)
        P = _new(TB, GH, like=x2)
        bf = bf16_mode()
        xb = Wb = None
        if bf:  # perf mode: bf16 operands (rounded once, kept for backward), fp32 accumulate / P
            xb, Wb = cvt_bf16(x2), cvt_bf16(Wcat)
            gemm_bf16(TB, GH, D, xb, xb.shape[1], 1, Wb, Wb.shape[1], 1, P, GH)
        else:
            gemm(TB, GH, D, x2, x2.stride(0), 1, Wcat, 1, D, P, GH)
        mean = var = None
        if use_bn:
            if training:
                mean, var = bn_stats(P)
                # duplicating every row leaves mean / biased var unchanged; only the unbiased
                # running_var factor sees the reference's row count ndir*T*B
                pscale, pshift = bn_finalize(mean, var, gamma, beta, eps, running_mean, running_var, momentum, ndir * TB)
            else:
                mean, var = running_mean, running_var
                pscale, pshift = bn_finalize(mean, var, gamma, beta, eps)
        else:
            pscale = torch.ones(GH, device=x.device)
            pshift = bcat.contiguous() if bcat is not None else torch.zeros(GH, device=x.device)
        Y = _new(T, B, ndir * H, like=x2)
        S = _new(ndir, TB, NS * H, like=x2)
        use_ln = ln_gamma is not None
        algo = choose_rec_algo(cell, H, use_ln)
        prec = PREC[settings.precision]
        if algo == REC_PERSISTENT:
            _lib.raise_if_persist_failed()
        LNS = None
        n_work = int(lib.pk_rec_work_floats(CELL[cell], T, B, int(bidir), H))
        n_lnwork = 0
        if use_ln:  # per-step LayerNorm of h_t: statistics and pre-LN h of every (step, row), saved for backward
            ln_gamma, ln_beta = ln_gamma.contiguous(), ln_beta.contiguous()
            LNS = _new(int(lib.pk_rec_ln_saved_floats(T, B, int(bidir), H)), like=x2)
            if algo == REC_PERSISTENT:  # + the row-statistics exchange of the persistent kernels
                n_lnwork = int(lib.pk_rec_ln_work_floats(T, B, int(bidir), H))
        work = _new(n_work + (0 if bf else n_lnwork), like=x2)
        Yb = None
        ctx.Xb = None
        if bf and algo == REC_PERSISTENT:
            # perf mode: second-generation persistent kernel; its bf16 exchange buffer Yb is also the
            # k-major operand of the dU GEMM in backward (nothing is converted afterwards)
            Hp = _up(H, 8)
            Yb = torch.empty(TB, _up(ndir * Hp, 64), device=x.device, dtype=torch.bfloat16)
            two_phase = cell in ("GRU", "minimalGRU")
            if use_ln and two_phase:
                lnwork = _new(n_lnwork, like=x2)
                ctx.Xb = torch.empty_like(Yb)  # second mailbox of the two-phase cells: r*h (GRU) / z*h (minimalGRU)
                rc = lib.pk_rec2p_fwd_bf16_ln(_stream(), CELL[cell], ACT[act], T, B, int(bidir), H, _p(P), _p(pscale),
                                              _p(pshift), _p(Ucat), _p(mask), float(mask_scalar), _p(ln_gamma), _p(ln_beta),
                                              LN_EPS, _p(Y), _p(S), _p(LNS), _p(Yb), _p(ctx.Xb), Yb.shape[1],
                                              2 if settings.self_fill else 0, _p(lnwork))
            elif use_ln:
                lnwork = _new(n_lnwork, like=x2)
                rc = lib.pk_rec_fwd_bf16_ln(_stream(), CELL[cell], ACT[act], T, B, int(bidir), H, _p(P), _p(pscale),
                                            _p(pshift), _p(Ucat), _p(mask), float(mask_scalar), _p(ln_gamma), _p(ln_beta),
                                            LN_EPS, _p(Y), _p(S), _p(LNS), _p(Yb), Yb.shape[1],
                                            2 if settings.self_fill else 0, _p(lnwork))
            else:
                rc = lib.pk_rec_fwd_bf16(_stream(), CELL[cell], ACT[act], T, B, int(bidir), H, _p(P), _p(pscale),
                                         _p(pshift), _p(Ucat), _p(mask), float(mask_scalar), _p(Y), _p(S), _p(Yb),
                                         Yb.shape[1], 2 if settings.self_fill else 0)
            _lib.check(rc, "pk_rec_fwd_bf16")
        else:
            rc = lib.pk_rec_fwd(_stream(), algo, prec, CELL[cell], ACT[act], T, B, int(bidir), H, _p(P), _p(pscale),
                                _p(pshift), _p(Ucat), _p(mask), float(mask_scalar), _p(ln_gamma), _p(ln_beta), _p(Y),
                                _p(S), _p(LNS), _p(work))
            _lib.check(rc, "pk_rec_fwd")
        _force_kinks(S, cell, T, B, ndir, H)
        ctx.bf = bf
        ctx.Yb = Yb
        if bf:
            ctx.save_for_backward(xb, Wb, Ucat, P, mean, var, gamma, mask, Y, S, pscale, ln_gamma, LNS)
        else:
            ctx.save_for_backward(x2, Wcat, Ucat, P, mean, var, gamma, mask, Y, S, pscale, ln_gamma, LNS)
        ctx.cfg = cfg[:9]
        ctx.algo, ctx.prec = algo, prec
        ctx.in_shape = x.shape
        ctx.has_bias = bcat is not None
        if use_bn and training:
            ctx.mark_non_differentiable(mean, var)
            return Y, mean, var
        return Y, None, None

    @staticmethod
    def backward(ctx, dY, _dm, _dv):
        lib = _lib.load()
        x2, Wcat, Ucat, P, mean, var, gamma, mask, Y, S, pscale, ln_gamma, LNS = ctx.saved_tensors
        cell, act, H, bidir, use_bn, training, eps, momentum, mask_scalar = ctx.cfg
        T, B, D = ctx.in_shape
        G = lib.pk_rec_num_gates(CELL[cell])
        ndir = 2 if bidir else 1
        TB, GH = T * B, G * H
        dY = dY.contiguous()
        dP2 = _new(ndir, TB, GH, like=dY)
        dU = _new(GH, H, like=dY)
        bf = ctx.bf
        n_lnwork = 0
        if ln_gamma is not None and ctx.algo == REC_PERSISTENT:
            n_lnwork = int(lib.pk_rec_ln_work_floats(T, B, int(bidir), H))
        work = _new(int(lib.pk_rec_work_floats(CELL[cell], T, B, int(bidir), H)) + (0 if ctx.Yb is not None else n_lnwork),
                    like=dY)
        dlg = dlb = None
        if ln_gamma is not None:
            dlg, dlb = _new(H, like=dY), _new(H, like=dY)
        dGb = None
        if ctx.Yb is not None:
            Hp = _up(H, 8)
            dGb = torch.empty(ndir * TB, _up(G * Hp, 64), device=dY.device, dtype=torch.bfloat16)
            if ctx.Xb is not None:  # two-phase cells (with per-step LayerNorm): the gate gradients come back as bf16 only
                lnwork = _new(n_lnwork, like=dY)
                rc = lib.pk_rec2p_bwd_bf16_ln(_stream(), CELL[cell], ACT[act], T, B, int(bidir), H, _p(Ucat), _p(mask),
                                              float(mask_scalar), _p(ln_gamma), LN_EPS, _p(Y), _p(S), _p(LNS), _p(dY),
                                              _p(dGb), dGb.shape[1], 2 if settings.self_fill else 0, _p(lnwork),
                                              _p(dlg), _p(dlb))
                flat = dP2.view(ndir * TB, GH)
                for g_ in range(G):
                    flat[:, g_ * H:(g_ + 1) * H].copy_(dGb[:, g_ * Hp:g_ * Hp + H])
            elif ln_gamma is not None:
                lnwork = _new(n_lnwork, like=dY)
                rc = lib.pk_rec_bwd_bf16_ln(_stream(), CELL[cell], ACT[act], T, B, int(bidir), H, _p(Ucat), _p(mask),
                                            float(mask_scalar), _p(ln_gamma), LN_EPS, _p(Y), _p(S), _p(LNS), _p(dY),
                                            _p(dP2), _p(dGb), dGb.shape[1], 2 if settings.self_fill else 0, _p(lnwork),
                                            _p(dlg), _p(dlb))
            else:
                rc = lib.pk_rec_bwd_bf16(_stream(), CELL[cell], ACT[act], T, B, int(bidir), H, _p(Ucat), _p(mask),
                                         float(mask_scalar), _p(Y), _p(S), _p(dY), _p(dP2), _p(dGb), dGb.shape[1],
                                         2 if settings.self_fill else 0)
            _lib.check(rc, "pk_rec_bwd_bf16")
        else:
            side_u = (not bf) and ctx.side_u and side_targets_any_ok(ctx.uparams)
            flush_deferred_side()  # (the output layers' postponed weight gradients: next to this recurrence)
            rc = lib.pk_rec_bwd(_stream(), ctx.algo, ctx.prec, CELL[cell], ACT[act], T, B, int(bidir), H, _p(Ucat),
                                _p(mask), float(mask_scalar), _p(ln_gamma), _p(Y), _p(S), _p(LNS), _p(dY), _p(dP2),
                                None if (bf or ctx.side_u) else _p(dU), _p(dlg), _p(dlb), _p(work))
            _lib.check(rc, "pk_rec_bwd")
            if ctx.side_u and not side_u:  # (decided in forward, no longer possible: the gradient goes back through autograd... which has no edge)
                raise _lib.PkError("recurrent layer: the flat gradient buffer the weights' gradients were to be added to is gone "
                                   "between forward and backward (optim.FlatParams.zero_grad() re-aliases it)")
        NS_ = lib.pk_rec_num_saved(CELL[cell])

        def do_dU_f32():
            gview = adjacent_view([q.grad for q in ctx.uparams])
            if gview is not None:
                _deferred_dU_f32(cell, T, B, ndir, H, G, NS_, Y, S, dP2, gview, True)
            else:
                _deferred_dU_f32(cell, T, B, ndir, H, G, NS_, Y, S, dP2, dU, False)
                _accumulate_rows(ctx.uparams, [dU[g * H:(g + 1) * H] for g in range(G)])

        if bf:
            _deferred_dU_bf16(lib, cell, T, B, ndir, H, G, Y, S, dP2, dU, ctx.Yb, dGb, ctx.Xb)
            ctx.Yb = ctx.Xb = None
        g1 = dP2[0]
        g2 = dP2[1] if bidir else None
        dgamma = dbeta = dbias = None
        # the projection gradient is the operand of the dX / dW GEMMs: with BatchNorm (whose backward writes it at any pitch)
        # a row of 3 x 550 = 1650 floats - the GRU - is laid out at 1652, so that rows start 16-byte aligned and the products
        # take the LDS-DMA form of pk_gemm (libri_gru fp32 143.5 -> 137.1 ms, profiles/r06_fp32_gemm_dma.json)
        ldp = GH
        if use_bn and not bf and GH % 4 != 0:
            ldp = _up(GH, 4)
        dPraw = _new(TB, ldp, like=dY)
        if use_bn:
            part = _new(int(lib.pk_bn_partial_floats(TB, GH)), like=dY)
            sum_g, sum_gx = _new(GH, like=dY), _new(GH, like=dY)
            _lib.check(lib.pk_bn_bwd_reduce(_stream(), _p(g1), _p(g2), GH, _p(P), GH, TB, GH, _p(mean), _p(var), eps,
                                            _p(part), _p(sum_g), _p(sum_gx)), "pk_bn_bwd_reduce")
            dgamma, dbeta = sum_gx, sum_g
            if training:
                _lib.check(lib.pk_bn_bwd_apply(_stream(), _p(g1), _p(g2), GH, _p(P), GH, TB, GH, _p(mean), _p(var), eps,
                                               _p(gamma), _p(sum_g), _p(sum_gx), float(TB), _p(dPraw), ldp),
                           "pk_bn_bwd_apply")
            else:
                gs = g1
                if g2 is not None:
                    gs = _new(TB, GH, like=dY)
                    _lib.check(lib.pk_add(_stream(), _p(g1), _p(g2), TB * GH, _p(gs)), "pk_add")
                zero = torch.zeros_like(pscale)
                _lib.check(lib.pk_affine_act_fwd(_stream(), _p(gs), GH, TB, GH, _p(pscale), _p(zero), 0, None,
                                                 _p(dPraw), ldp), "pk_affine_act_fwd")
        else:
            if g2 is not None:
                _lib.check(lib.pk_add(_stream(), _p(g1), _p(g2), TB * GH, _p(dPraw)), "pk_add")
            else:
                dPraw = g1
            if ctx.has_bias:
                dbias = colsum(dPraw)
        dx = dW = None
        dW = _new(GH, D, like=dY)
        if bf:
            xb, Wb = x2, Wcat
            dPb = cvt_bf16(dPraw)
            if ctx.needs_input_grad[0]:
                dx = _new(TB, D, like=dY)
                gemm_bf16(TB, D, GH, dPb, dPb.shape[1], 1, Wb, Wb.shape[1], 0, dx, D)
                dx = dx.view(T, B, D)
            gemm_bf16(GH, D, TB, dPb, dPb.shape[1], 0, xb, xb.shape[1], 0, dW, D, splitk=_splitk_bf(_tiles_bf(GH, D), TB))
            return dx, dW, dbias, dU, dgamma, dbeta, None, None, None, dlg, dlb, None
        if ctx.needs_input_grad[0]:
            dx = _new(TB, D, like=dY)
            gemm(TB, D, GH, dPraw, ldp, 1, Wcat, D, 1, dx, D)
            dx = dx.view(T, B, D)
        # weight gradients are off the dependency chain (next on it: the recurrence of the layer below, which leaves 112 CUs
        # idle): with flat-bucket parameters they run on the side stream, behind the dX GEMM, and add straight into .grad
        side_w = ctx.side_w and side_targets_any_ok(ctx.wparams)
        if ctx.side_w and not side_w:
            raise _lib.PkError("recurrent layer: the flat gradient buffer of the input weights is gone between forward and backward")
        if ctx.side_u:
            side_launch(do_dU_f32, (Y, S, dP2, dU), ctx.uparams)

        def do_dW_f32():
            gview = adjacent_view([q.grad for q in ctx.wparams])
            sk = _splitk(_tiles(GH, D), TB)
            if gview is not None:
                gemm(GH, D, TB, dPraw, 1, ldp, x2, x2.stride(0), 1, gview, D, beta=1.0, splitk=sk)
            else:
                gemm(GH, D, TB, dPraw, 1, ldp, x2, x2.stride(0), 1, dW, D, splitk=sk)
                _accumulate_rows(ctx.wparams, [dW[g * H:(g + 1) * H] for g in range(G)])

        if side_w:
            side_launch(do_dW_f32, (dPraw, x2, dW), ctx.wparams)
        else:
            gemm(GH, D, TB, dPraw, 1, ldp, x2, x2.stride(0), 1, dW, D, splitk=_splitk(_tiles(GH, D), TB))
        return (dx, None if side_w else dW, dbias, None if ctx.side_u else dU, dgamma, dbeta, None, None, None, dlg, dlb, None)


def perf_path_ok(cell, H, use_ln, use_bn, training):
    """The bf16 pipeline below covers liGRU / RNN / LSTM layers without per-step LayerNorm; BatchNorm
    backward through frozen statistics (eval-mode module with autograd on) stays on the general path."""
    if not bf16_mode() or settings.rec_algo == "stepwise":
        return False
    if cell not in ("liGRU", "RNN", "LSTM", "GRU", "minimalGRU") or H > 576 or use_ln:
        return False
    return training or not use_bn or not torch.is_grad_enabled()


class _Prefill:
    """0xFF fill of the persistent kernels' exchange buffers on a third stream (next to the projection GEMM)."""
    stream = None

    def __init__(self):
        self.done = 0

    def start(self, *bufs):
        if not settings.wgrad_side:  # PK_EXPERIMENT wgrad_side=0 keeps the whole step on one stream
            return
        main = torch.cuda.current_stream()
        if _Prefill.stream is None:
            _Prefill.stream = torch.cuda.Stream()
        st = _Prefill.stream
        st.wait_stream(main)  # the fresh blocks may still be in use by work enqueued earlier on the main stream
        with torch.cuda.stream(st):
            for b in bufs:
                if b is not None:
                    b.view(torch.int16).fill_(-1)
        for b in bufs:
            if b is not None:
                b.record_stream(st)
        self.done = 1

    def wait(self):
        if self.done:
            torch.cuda.current_stream().wait_stream(_Prefill.stream)


class RecLayerPerfFn(torch.autograd.Function):
    """Perf-mode (bf16 MFMA operands) recurrent layer: same math as RecLayerFn, but every GEMM operand
    lives in HBM as bf16 and nothing is converted twice:

      xb   bf16 layer input - either converted here (first layer) or the previous layer's Yb, the bf16
           copy of its output that the persistent kernel publishes anyway (direction halves at a pitch
           of Hp = H rounded up to 8; the weight copy Wb is re-pitched the same way);
      Yb / dGb  the kernels' exchange buffers, reused as the k-major operands of the dU GEMMs;
      dPb  BatchNorm backward (pk_bn_bwd_bf16) reads dGb and writes the projection gradient as bf16:
           the fp32 gate-gradient slabs and dP never exist.

    y, bn_mean, bn_var, Yb = f(x, xb_in, Wcat, bcat, Ucat, gamma, beta, running_mean, running_var, mask)
    """

    @staticmethod
    def forward(ctx, x, xb_in, Wcat, bcat, Ucat, gamma, beta, running_mean, running_var, mask, cfg, edge=None):
        ctx.set_materialize_grads(False)  # no 147 MB zero tensor for the (non-differentiable) bf16 twin in backward
        _need_gpu(x, Wcat, bcat, Ucat, gamma, beta, mask)
        lib = _lib.load()
        cell, act, H, bidir, use_bn, training, eps, momentum, mask_scalar, xseg = cfg[:10]
        ctx.wparams, ctx.uparams = (cfg[10], cfg[11]) if len(cfg) > 10 else (None, None)
        ctx.side_w, ctx.side_u = (bool(cfg[12]), bool(cfg[13])) if len(cfg) > 13 else (False, False)
        # (gamma parameters, beta parameters) when the BatchNorm affine was handed over DETACHED (views of the flat
        # buffer): backward adds d gamma / d beta to their flat .grad inside pk_bn_bwd_bf16 (`edge` keeps the node alive)
        ctx.affine = cfg[15] if len(cfg) > 15 else None
        T, B, D = x.shape
        G = lib.pk_rec_num_gates(CELL[cell])
        NS = lib.pk_rec_num_saved(CELL[cell])
        ndir = 2 if bidir else 1
        TB, GH = T * B, G * H
        Wcat = Wcat.contiguous()
        Ucat = Ucat.contiguous()
        if xb_in is None:
            xseg = None
            xb, Wb, K = cvt_bf16(_rows2d(x)), cvt_bf16(Wcat), D
        else:
            nseg, seglen, segpad = xseg
            assert nseg * seglen == D and xb_in.shape[0] == TB
            xb, Wb, K = xb_in, cvt_bf16(Wcat, nseg, seglen, segpad), nseg * segpad
            assert Wb.shape[1] == xb.shape[1]
        # the kernels' bf16 exchange buffers must hold the 0xFF "not written yet" pattern: they are filled on a third
        # stream next to the projection GEMM / BatchNorm statistics instead of in front of the recurrence (the backward
        # one, dGb, too: 295 MB that would otherwise be filled on the critical path of backward)
        Hp = _up(H, 8)
        Yb = torch.empty(TB, _up(ndir * Hp, 64), device=x.device, dtype=torch.bfloat16)
        two_phase = cell in ("GRU", "minimalGRU")
        # (the liGRU / RNN kernels write the pattern themselves, a few steps ahead of their own publishes: prefilled = 2)
        self_fill = settings.self_fill and lib.pk_rec_self_fill(CELL[cell]) == 1
        dGb = None
        if any(ctx.needs_input_grad) and not self_fill:  # a backward pass will follow: its exchange buffer is filled now
            dGb = torch.empty(ndir * TB, _up(G * Hp, 64), device=x.device, dtype=torch.bfloat16)
        Xb = torch.empty_like(Yb) if two_phase else None  # two exchanges per step: h and r*h (z*h)
        fill = _Prefill()
        if not self_fill:
            fill.start(Yb, Xb, dGb)
        P = _new(TB, GH, like=Wcat)
        mean = var = None
        bn_bufs = cfg[14] if len(cfg) > 14 else None  # per-gate (running_mean, running_var, num_batches_tracked) lists
        pscale = pshift = None
        if use_bn and training and bn_bufs is not None:
            # the statistics come out of the projection GEMM's epilogue; their merge, scale / shift and the running
            # statistics of every gate's module are one launch behind it
            mean, var, pscale, pshift = gemm_bf16_bn_stats(TB, GH, K, xb, xb.shape[1], 1, Wb, Wb.shape[1], 1, P, GH, gates=(
                gamma, beta, eps, H, bn_bufs[0], bn_bufs[1], bn_bufs[2], momentum, ndir * TB))
        elif use_bn and training:
            mean, var = gemm_bf16_bn_stats(TB, GH, K, xb, xb.shape[1], 1, Wb, Wb.shape[1], 1, P, GH)
        else:
            gemm_bf16(TB, GH, K, xb, xb.shape[1], 1, Wb, Wb.shape[1], 1, P, GH)
        if use_bn:
            if pscale is not None:
                pass
            elif training:
                pscale, pshift = bn_finalize(mean, var, gamma, beta, eps, running_mean, running_var, momentum, ndir * TB)
            else:
                mean, var = running_mean, running_var
                pscale, pshift = bn_finalize(mean, var, gamma, beta, eps)
        else:
            pscale = torch.ones(GH, device=x.device)
            pshift = bcat.contiguous() if bcat is not None else torch.zeros(GH, device=x.device)
        # A validation / forward chunk (torch.no_grad, core.py:644-671: nothing here needs a gradient) saves nothing for a
        # backward pass: no S (563 MB per Li-GRU layer at the BASELINE shape), and an INNER layer of a stack does not write
        # its fp32 output either - the next layer reads the bf16 copy Yb (cfg[16], set by nn._Recurrent.forward; the
        # tensor returned in its place carries the shape only).
        infer = not any(ctx.needs_input_grad) and _Kinks.queue is None and _lib.experiment("fwd_nosave", "1") != "0"
        keep_y = bool(cfg[16]) if len(cfg) > 16 else True
        Y = _new(T, B, ndir * H, like=Wcat)
        S = None if infer else _new(ndir, TB, NS * H, like=Wcat)
        Yarg = Y if (keep_y or not infer) else None
        _lib.raise_if_persist_failed()
        fill.wait()
        if two_phase:
            rc = lib.pk_rec2p_fwd_bf16(_stream(), CELL[cell], ACT[act], T, B, int(bidir), H, _p(P), _p(pscale),
                                       _p(pshift), _p(Ucat), _p(mask), float(mask_scalar), _p(Yarg), _p(S), _p(Yb), _p(Xb),
                                       Yb.shape[1], 2 if self_fill else fill.done)
            _lib.check(rc, "pk_rec2p_fwd_bf16")
        else:
            rc = lib.pk_rec_fwd_bf16(_stream(), CELL[cell], ACT[act], T, B, int(bidir), H, _p(P), _p(pscale), _p(pshift),
                                     _p(Ucat), _p(mask), float(mask_scalar), _p(Yarg), _p(S), _p(Yb), Yb.shape[1],
                                     2 if self_fill else fill.done)
            _lib.check(rc, "pk_rec_fwd_bf16")
        if S is not None:
            _force_kinks(S, cell, T, B, ndir, H)
        ctx.self_fill = self_fill
        ctx.dGb = dGb if fill.done else None
        ctx.Xb = Xb
        ctx.save_for_backward(xb, Wb, Wcat, Ucat, P, mean, var, gamma, mask, Y, S, Yb)
        ctx.cfg = cfg[:9] + (xseg,)
        ctx.in_shape = x.shape
        ctx.has_bias = bcat is not None
        if use_bn and training:
            ctx.mark_non_differentiable(mean, var, Yb)
            return Y, mean, var, Yb
        ctx.mark_non_differentiable(Yb)
        return Y, None, None, Yb

    @staticmethod
    def backward(ctx, dY, _dm, _dv, _dyb):
        if dY is None:  # the layer output did not reach the loss
            return (None,) * 12
        lib = _lib.load()
        xb, Wb, Wcat, Ucat, P, mean, var, gamma, mask, Y, S, Yb = ctx.saved_tensors
        cell, act, H, bidir, use_bn, training, eps, momentum, mask_scalar, xseg = ctx.cfg
        if use_bn and not training:
            raise _lib.PkError("perf-mode recurrent layer: backward through frozen BatchNorm statistics is not covered")
        T, B, D = ctx.in_shape
        G = lib.pk_rec_num_gates(CELL[cell])
        ndir = 2 if bidir else 1
        TB, GH = T * B, G * H
        Hp = _up(H, 8)
        dY = dY.contiguous()
        dGb, prefilled = ctx.dGb, 1  # filled with the "not written" pattern during forward (third stream)
        ctx.dGb = None
        if dGb is None:
            dGb = torch.empty(ndir * TB, _up(G * Hp, 64), device=dY.device, dtype=torch.bfloat16)
            prefilled = 2 if ctx.self_fill else 0
        Gp = dGb.shape[1]
        flush_deferred_side()  # the output layers' weight gradients: ready when this recurrence is, next to it on the idle CUs
        if ctx.Xb is not None:
            rc = lib.pk_rec2p_bwd_bf16(_stream(), CELL[cell], ACT[act], T, B, int(bidir), H, _p(Ucat), _p(mask),
                                       float(mask_scalar), _p(Y), _p(S), _p(dY), _p(dGb), Gp, prefilled)
            _lib.check(rc, "pk_rec2p_bwd_bf16")
        else:
            rc = lib.pk_rec_bwd_bf16(_stream(), CELL[cell], ACT[act], T, B, int(bidir), H, _p(Ucat), _p(mask),
                                     float(mask_scalar), _p(Y), _p(S), _p(dY), None, _p(dGb), Gp, prefilled)
            _lib.check(rc, "pk_rec_bwd_bf16")
        # weight gradients are off the dependency chain (the next thing on it is the layer below's recurrent
        # backward): with flat-bucket parameters they run on the side stream and accumulate straight into .grad
        side_u, side_w = ctx.side_u, ctx.side_w  # decided in forward: those weights were handed over detached
        Xb = ctx.Xb
        ctx.Xb = None
        dU = _new(GH, H, like=dY)

        def do_dU():
            _deferred_dU_bf16(lib, cell, T, B, ndir, H, G, Y, S, None, dU, Yb, dGb, Xb)
            if side_u:
                _accumulate_rows(ctx.uparams, [dU[g * H:(g + 1) * H] for g in range(G)])

        # (the bottom layer has no dX GEMM and nothing below it to hide behind: its dU starts at once, next to its own
        # BatchNorm backward, and only the short dW is left for the tail of the step)
        late = settings.side_late and bool(ctx.needs_input_grad[0])
        if side_u and not late:
            side_launch(do_dU, (Y, S, Yb, dGb, Xb, dU), ctx.uparams)
        elif not side_u:
            do_dU()
        # BatchNorm backward (or plain sum of the two directions) straight from dGb -> bf16 projection gradient
        dPb = torch.empty(TB, _up(GH, 64), device=dY.device, dtype=torch.bfloat16)
        part = _new(int(lib.pk_bn_partial_floats(TB, GH)), like=dY)
        sum_g = _new(GH, like=dY)
        sum_gx = _new(GH, like=dY) if use_bn else None
        g1 = ctypes.c_void_p(dGb.data_ptr() + 2 * TB * Gp) if bidir else None
        acc_g = acc_b = None
        if use_bn and ctx.affine is not None:  # decided in forward: d gamma / d beta go straight into the flat .grad
            acc_g = adjacent_view([q.grad for q in ctx.affine[0]])
            acc_b = adjacent_view([q.grad for q in ctx.affine[1]])
            if acc_g is None or acc_b is None:
                raise _lib.PkError("perf-mode recurrent layer: the BatchNorm gradients left the flat buffer between forward "
                                   "and backward (optim.FlatParams.zero_grad() re-aliases them)")
        rc = lib.pk_bn_bwd_bf16(_stream(), _p(dGb), g1, Gp, G, H, _p(P), GH, TB, _p(mean) if use_bn else None,
                                _p(var) if use_bn else None, eps, _p(gamma) if use_bn else None, float(TB), _p(part),
                                _p(sum_g), _p(sum_gx), _p(dPb), dPb.shape[1], _p(acc_b), _p(acc_g))
        _lib.check(rc, "pk_bn_bwd_bf16")
        dgamma = dbeta = dbias = None
        if use_bn and acc_g is None:
            dgamma, dbeta = sum_gx, sum_g
        elif not use_bn and ctx.has_bias:
            dbias = sum_g
        # dW[n,d] = sum_m dP[m,n] x[m,d]: both operands k-major; with a re-pitched input the columns come out re-pitched
        Kx = D if xseg is None else xseg[0] * xseg[2]
        dWp = _new(GH, Kx, like=dY)
        dW = None

        def do_dW():
            nonlocal dW
            gview = adjacent_view([q.grad for q in ctx.wparams]) if side_w else None
            sk = _splitk_bf(_tiles_bf(GH, Kx), TB)
            if gview is not None and xseg is None:  # straight into the flat gradient of the layer's gates (beta = 1)
                gemm_bf16(GH, Kx, TB, dPb, dPb.shape[1], 0, xb, xb.shape[1], 0, gview, Kx, beta=1.0, splitk=sk)
                return
            gemm_bf16(GH, Kx, TB, dPb, dPb.shape[1], 0, xb, xb.shape[1], 0, dWp, Kx, splitk=sk)
            if gview is not None:  # re-pitched input: one add per direction segment, all gates at once
                nseg, seglen, segpad = xseg
                with torch.no_grad():
                    for s_ in range(nseg):
                        gview[:, s_ * seglen:(s_ + 1) * seglen].add_(dWp[:, s_ * segpad:s_ * segpad + seglen])
                return
            if xseg is None:
                dW = dWp
            else:
                nseg, seglen, segpad = xseg
                dW = torch.cat([dWp[:, s_ * segpad:s_ * segpad + seglen] for s_ in range(nseg)], 1)
            if side_w:
                _accumulate_rows(ctx.wparams, [dW[g * H:(g + 1) * H] for g in range(G)])

        if side_w and not late:
            side_launch(do_dW, (dPb, xb, dWp), ctx.wparams)
        dx = None
        if ctx.needs_input_grad[0]:  # dx[m,d] = sum_n dP[m,n] W[n,d]: A k-contiguous, B = W (plain pitch) k-major
            Wb2 = Wb if xseg is None else cvt_bf16(Wcat)
            dx = _new(TB, D, like=dY)
            gemm_bf16(TB, D, GH, dPb, dPb.shape[1], 1, Wb2, Wb2.shape[1], 0, dx, D)
            dx = dx.view(T, B, D)
        if late:  # behind the dX GEMM: next on the main stream is the recurrence of the layer below
            free = 0
            if settings.side_cus:
                held = int(lib.pk_rec_plan_cus(ndir * B, H))
                free = max(int(lib.pk_num_cu()) - held, 0) if held > 0 else 0
            if side_u:
                side_launch(_sized_for(free, do_dU), (Y, S, Yb, dGb, Xb, dU), ctx.uparams)
            if side_w:
                side_launch(_sized_for(free, do_dW), (dPb, xb, dWp), ctx.wparams)
        if not side_w:
            do_dW()
        return dx, None, (None if side_w else dW), dbias, (None if side_u else dU), dgamma, dbeta, None, None, None, None, None


# ----------------------------------------------------------------------------
# conv1d + max_pool1d (neural_networks.py:1546-1552, 1655-1661, 1805-1813)
# ----------------------------------------------------------------------------
# Round 5: "1" - layers with at least 8 input channels take bf16 MFMA operands in perf mode (graded on whole tensors against
# the bf16-operand model, tests/test_gpu_reference_pins.py; the sinc layer - one input channel, 129-tap filters on the raw
# waveform - stays on the exact-fp32 kernels: its gradients are not good enough in bf16).
CONV_BF16_DEFAULT = "1"


def conv_bf16_mode():
    """PK_CONV_BF16: "0" exact-fp32 convolutions, "1" bf16 MFMA operands for layers with >= 8 input channels, "2" for every
    covered layer (perf mode only).  One place for the default: the tests' bf16-operand model follows it."""
    return os.environ.get("PK_CONV_BF16", CONV_BF16_DEFAULT)


class ConvPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, pool):
        _need_gpu(x, w, bias)
        lib = _lib.load()
        x = x.contiguous()
        w = w.contiguous()
        B, Cin, L = x.shape
        Cout, _, K = w.shape
        Lp = (L - K + 1) // pool
        y = _new(B, Cout, Lp, like=x)
        arg = torch.empty(B, Cout, Lp, device=x.device, dtype=torch.int32)
        # perf mode, opt-in: the implicit-GEMM kernels on the matrix pipe (bf16 operands, pk_conv_bf16.hip).  PK_CONV_BF16=1:
        # layers with at least 8 input channels, =2: every covered layer; default 0: the exact-fp32 kernels.  Measured on
        # timit_sincnet with =2: 3.36 vs 3.91 ms per step - but with the raw waveform and the 129-tap sinc filters rounded
        # to bf16 the recipe-scale fixture's worst parameter gradient sits 30 % from the reference's (bf16-operand model
        # vs reference), and the network's own noise floor exceeds the fixed grading limits: not a
        # default until the first layer's treatment is settled
        mode = conv_bf16_mode()
        ctx.conv_bf = (bf16_mode() and mode in ("1", "2") and (Cin >= 8 or mode == "2")
                       and lib.pk_conv_bf16_covers(Cin, Cout, K, pool) == 1)
        if ctx.conv_bf:
            work = _new(int(lib.pk_conv_bf16_work_floats(B, Cin, L, Cout, K, pool, 0)), like=x)
            _lib.check(lib.pk_conv1d_pool_fwd_bf16(_stream(), _p(x), _p(w), _p(bias), B, Cin, L, Cout, K, pool, _p(y),
                                                   ctypes.c_void_p(arg.data_ptr()), _p(work)), "pk_conv1d_pool_fwd_bf16")
        else:
            work = _new(int(lib.pk_conv_fwd_work_floats(Cin, Cout, K)), like=x)
            _lib.check(lib.pk_conv1d_pool_fwd(_stream(), _p(x), _p(w), _p(bias), B, Cin, L, Cout, K, pool, _p(y),
                                              ctypes.c_void_p(arg.data_ptr()), _p(work)), "pk_conv1d_pool_fwd")
        if _Decisions.pool is not None:
            # test mode: backward routes dy to another run's arg-max positions (absolute index = window start + offset)
            fpool, off = _Decisions.pool.pop(0)
            assert fpool == pool and tuple(off.shape) == (B, Cout, Lp), (fpool, pool, tuple(off.shape), (B, Cout, Lp))
            forced = (torch.arange(Lp, device=x.device, dtype=torch.int32) * pool)[None, None, :] + off.to(x.device).to(torch.int32)
            _Decisions.report.append(("pool", int((forced != arg).sum()), arg.numel()))
            arg = forced.contiguous()
        ctx.save_for_backward(x, w, arg)
        ctx.pool = pool
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w, arg = ctx.saved_tensors
        B, Cin, L = x.shape
        Cout, _, K = w.shape
        pool = ctx.pool
        dy = dy.contiguous()
        dw = torch.empty_like(w)
        db = _new(Cout, like=x) if ctx.has_bias else None
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        if ctx.conv_bf:
            work = _new(int(lib.pk_conv_bf16_work_floats(B, Cin, L, Cout, K, pool, 1)), like=x)
            dx_mfma = dx
            if dx is not None and Cin < 8:
                # a layer with a handful of input channels (SincNet's first: one): its data gradient is a matrix-vector
                # product per position - the exact-fp32 kernel, not an MFMA tile that is 15/16 padding
                wk = _new(int(lib.pk_conv_fwd_work_floats(Cin, Cout, K)), like=x)
                _lib.check(lib.pk_conv1d_pool_dgrad(_stream(), _p(w), _p(dy), ctypes.c_void_p(arg.data_ptr()), B, Cin, L, Cout,
                                                    K, pool, _p(dx), _p(wk)), "pk_conv1d_pool_dgrad")
                dx_mfma = None
            _lib.check(lib.pk_conv1d_pool_bwd_bf16(_stream(), _p(x), _p(w), _p(dy), ctypes.c_void_p(arg.data_ptr()), B, Cin,
                                                   L, Cout, K, pool, _p(dw), _p(dx_mfma), _p(work)), "pk_conv1d_pool_bwd_bf16")
            if db is not None:
                db = dy.sum(dim=(0, 2))
            return dx, dw, db, None
        part = _new(int(lib.pk_conv_partial_floats(B, Cin, L, Cout, K, pool)), like=x)
        _lib.check(lib.pk_conv1d_pool_bwd(_stream(), _p(x), _p(w), _p(dy), ctypes.c_void_p(arg.data_ptr()), B, Cin, L,
                                          Cout, K, pool, _p(dw), _p(db), _p(dx), _p(part)), "pk_conv1d_pool_bwd")
        return dx, dw, db, None


def conv1d_pool(x, w, bias, pool):
    return ConvPoolFn.apply(x, w, bias, pool)
