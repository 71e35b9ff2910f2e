"""Kernel-level numerics of libpk_amd.so on the GPU, each against a plain fp64/fp32
torch-CPU evaluation of the same op (a torch reference is appropriate here: these
are floating-point kernels; the model-level parity lives in test_gpu_parity.py)."""
import ctypes
import importlib

import pytest
import torch
import torch.nn.functional as TF

from golden_util import rel_err

pytestmark = pytest.mark.gpu
F_ = importlib.import_module("pytorch-kaldi_amd.functional")
_lib = importlib.import_module("pytorch-kaldi_amd._lib")


def test_mfma_fragment_layouts():
    lib = _lib.load()
    bad = ctypes.c_int(-1)
    rc = lib.pk_selftest_mfma(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(bad))
    _lib.check(rc, "pk_selftest_mfma")
    assert bad.value == 0


def test_permlane16_swap_lane_mapping():
    """The lane mapping of v_permlane16_swap_b32 the third-generation recurrences build their publish chunks with."""
    lib = _lib.load()
    bad = ctypes.c_int(-1)
    _lib.check(lib.pk_selftest_permlane(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(bad)),
               "pk_selftest_permlane")
    assert bad.value == 0


def test_dpp_row_sum_reaches_every_lane():
    """The 16-lane all-reduce the per-step LayerNorm of the persistent recurrences sums its row statistics with."""
    lib = _lib.load()
    bad = ctypes.c_int(-1)
    _lib.check(lib.pk_selftest_dpp_row_sum(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.byref(bad)),
               "pk_selftest_dpp_row_sum")
    assert bad.value == 0


def test_grouped_dropout_masks():
    """masks_ahead draws the masks of a stack's nn.Dropout calls together (two launches per distinct p); dropout_mask then
    hands them out in call order and falls back to a fresh draw when the forward takes another route."""
    dev = torch.device("cuda")
    specs = [(128, 1024, 0.15), (128, 1024, 0.15), (128, 512, 0.3), (128, 1024, 0.15)]
    torch.manual_seed(5)
    F_.masks_ahead(specs, dev)
    got = [F_.dropout_mask(torch.empty(r, c, device=dev), p) for r, c, p in specs]
    assert F_._Ahead.queue == []
    for (r, c, p), m in zip(specs, got):
        assert tuple(m.shape) == (r, c) and m.data_ptr() % 16 == 0
        vals = torch.unique(m).cpu()
        assert all(abs(float(v)) < 1e-12 or abs(float(v) - 1.0 / (1.0 - p)) < 1e-6 for v in vals)
        assert abs(float(m.mean()) - 1.0) < 0.02
    assert not torch.equal(got[0], got[1])
    F_.masks_ahead(specs, dev)
    m = F_.dropout_mask(torch.empty(7, 9, device=dev), 0.5)  # not what was announced: a fresh draw, the queue is dropped
    assert tuple(m.shape) == (7, 9) and F_._Ahead.queue == []


def test_single_hip_runtime():
    """libpk_amd.so must bind to the HIP runtime torch already mapped (one runtime per
    process): exactly one libamdhip64 in /proc/self/maps."""
    _lib.load()
    torch.zeros(1).cuda()
    libs = set()
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            libs.add(line.split()[-1])
    assert len(libs) == 1, libs


GEMM_SHAPES = [
    # M, N, K, a_kc, b_kc, splitk
    (1, 1, 1, True, True, 1),
    (37, 53, 29, True, True, 1),
    (128, 128, 64, True, False, 1),
    (300, 130, 550, False, True, 1),
    (257, 1100, 40, True, True, 1),
    (130, 70, 5000, False, False, 8),
    (1100, 40, 6000, False, False, 16),
    (512, 1938, 1100, True, True, 1),
    # the LDS-DMA form of the exact-fp32 kernel (aligned operands; K % 4 == 0 / M, N % 4 == 0): all four layouts, ragged
    # row / column tiles, a ragged last k-tile (1100 = 68 x 16 + 12), split-K
    (640, 260, 1100, True, False, 1),
    (260, 1100, 2000, False, False, 3),
    (200, 132, 76, False, True, 1),
    (1024, 1100, 1104, True, True, 1),
    (132, 128, 12, True, True, 1),
    (2200, 1100, 4096, False, False, 4),
    (1100, 552, 3000, False, False, 3),
]


@pytest.mark.parametrize("M,N,K,a_kc,b_kc,splitk", GEMM_SHAPES)
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_gemm(M, N, K, a_kc, b_kc, splitk, prec):
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    bias = torch.randn(N, generator=g)
    Ad = (A if a_kc else A.t().contiguous()).cuda()   # storage [M,K] or [K,M]
    Bd = (B.t().contiguous() if b_kc else B).cuda()   # storage [N,K] or [K,N]
    a_rs, a_cs = (K, 1) if a_kc else (1, M)
    b_rs, b_cs = (1, K) if b_kc else (N, 1)
    C = C0.clone().cuda()
    F_.gemm(M, N, K, Ad, a_rs, a_cs, Bd, b_rs, b_cs, C, N, alpha=0.5, beta=2.0, bias=bias.cuda(), splitk=splitk, prec=prec)
    torch.cuda.synchronize()
    ref = 0.5 * (A.double() @ B.double()) + 2.0 * C0.double() + bias.double()
    tol = 2e-6 if prec == "fp32" else 1e-2
    assert rel_err(C, ref) < tol


@pytest.mark.parametrize("M,N,K,a_kc,b_kc,splitk", [c for c in GEMM_SHAPES if c[0] * c[1] > 10000])
def test_gemm_f32_forms_bit_identical(M, N, K, a_kc, b_kc, splitk):
    """The LDS-DMA form of the exact-fp32 kernel against the register-staged form: every output element is one fmaf chain
    over ascending k in both (same MFMA, same k pairs in the same order, same split boundaries) - equal bit for bit."""
    import importlib

    lib = importlib.import_module("pytorch-kaldi_amd._lib").load()
    g = torch.Generator().manual_seed(M + 5 * N + 11 * K)
    A = torch.randn((M, K) if a_kc else (K, M), generator=g).cuda()
    B = torch.randn((N, K) if b_kc else (K, N), generator=g).cuda()
    a_rs, a_cs = (K, 1) if a_kc else (1, M)
    b_rs, b_cs = (1, K) if b_kc else (N, 1)
    outs = []
    for form in (1, 0):
        lib.pk_gemm_f32_set_form(form)
        try:
            C = torch.zeros(M, N).cuda()
            F_.gemm(M, N, K, A, a_rs, a_cs, B, b_rs, b_cs, C, N, splitk=splitk, prec="fp32")
            torch.cuda.synchronize()
            outs.append(C.cpu())
        finally:
            lib.pk_gemm_f32_set_form(0)
    assert torch.equal(outs[0], outs[1])


def test_gemm_f32_8_byte_aligned_rows():
    """A k-major operand whose rows are 8-byte aligned (1100 of the 1650 columns of the GRU's gate gradients): the LDS-DMA
    form fetches 8-byte aligned 16-byte pieces.  Against fp64 and bit for bit against the first form."""
    import importlib

    lib = importlib.import_module("pytorch-kaldi_amd._lib").load()
    g = torch.Generator().manual_seed(11)
    K, M, N = 3001, 1100, 552
    A = torch.randn(K, 1650, generator=g).cuda()
    B = torch.randn(K, 1104, generator=g).cuda()
    outs = []
    for form in (1, 0):
        lib.pk_gemm_f32_set_form(form)
        try:
            C = torch.zeros(M, N).cuda()
            F_.gemm(M, N, K, A, 1, 1650, B, 1104, 1, C, N, splitk=3, prec="fp32")
            torch.cuda.synchronize()
            outs.append(C.cpu())
        finally:
            lib.pk_gemm_f32_set_form(0)
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[1], A[:, :M].double().t().cpu() @ B[:, :N].double().cpu()) < 2e-6


@pytest.mark.parametrize("b_off", [0, 550, 3 * 1100 + 550, 1650])
@pytest.mark.parametrize("overwrite", [True, False])
def test_gemm_km_f32_offsets(b_off, overwrite):
    """The dU products of the exact-fp32 mode (functional._gemm_km_f32): k-major x k-major over the rows of wider
    matrices, the second operand starting at a multiple of H = 550 floats (8-byte aligned: taken from the 16-byte
    boundary below, the extra columns dropped; 550 columns = a ragged last 16-byte piece).  Against fp64, and bit for bit
    against the register-staged form of the kernel on the unshifted operand."""
    import importlib

    lib = importlib.import_module("pytorch-kaldi_amd._lib").load()
    g = torch.Generator().manual_seed(b_off + 3)
    K, M, N, a_ld, b_ld = 3000, 1100, 550, 1100, 2200
    A = torch.randn(K + 8, a_ld, generator=g).cuda()
    Bw = torch.randn(K + 8, b_ld, generator=g).cuda()
    C0 = torch.randn(M, N, generator=g).cuda()
    aflat, bflat = A.reshape(-1)[2 * a_ld:], Bw.reshape(-1)
    Bsub = bflat[b_off:b_off + K * b_ld].view(K, b_ld)[:, :N] if b_off + K * b_ld <= bflat.numel() else None
    outs = []
    for form in (1, 0):
        lib.pk_gemm_f32_set_form(form)
        try:
            C = C0.clone()
            F_._gemm_km_f32(M, N, K, aflat, a_ld, bflat, b_off, b_ld, C.view(-1), overwrite)
            torch.cuda.synchronize()
            outs.append(C.cpu())
        finally:
            lib.pk_gemm_f32_set_form(0)
    assert torch.equal(outs[0], outs[1])
    ref = A[2:2 + K, :M].double().t() @ Bsub.double()
    if not overwrite:
        ref = ref + C0.double()
    assert rel_err(outs[1], ref) < 2e-6


@pytest.mark.parametrize("M,N,two", [(64000, 1100, True), (1000, 2200, True), (333, 132, False), (80, 48, True),
                                     (4000, 1650, True), (77, 30, False)])
def test_bn_bwd_f32_forms_bit_identical(M, N, two):
    """The 16-byte / 8-byte forms of the exact-fp32 BatchNorm-backward passes (pk_bn_bwd_reduce / pk_bn_bwd_apply on aligned
    operands; 1650 and 30 columns take the 8-byte form) against the scalar forms (the same call with the gate gradient one float off alignment): the same row lanes,
    rows and expressions per column - sums and outputs equal bit for bit; and against fp64."""
    import ctypes
    import importlib

    _lib = importlib.import_module("pytorch-kaldi_amd._lib")
    lib = _lib.load()
    gen = torch.Generator().manual_seed(M + N)
    g = torch.randn(M, N, generator=gen).cuda()
    g2 = torch.randn(M, N, generator=gen).cuda() if two else None
    x = (torch.randn(M, N, generator=gen) * 2 + 0.3).cuda()
    gamma = (torch.rand(N, generator=gen) + 0.5).cuda()
    mean, var = x.mean(0), x.var(0, unbiased=False)
    p = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(gt):
        part = torch.empty(int(lib.pk_bn_partial_floats(M, N)), device="cuda")
        sg, sx, dx = torch.empty(N, device="cuda"), torch.empty(N, device="cuda"), torch.empty(M, N, device="cuda")
        _lib.check(lib.pk_bn_bwd_reduce(st, p(gt), p(g2), N, p(x), N, M, N, p(mean), p(var), 1e-5, p(part), p(sg), p(sx)), "reduce")
        _lib.check(lib.pk_bn_bwd_apply(st, p(gt), p(g2), N, p(x), N, M, N, p(mean), p(var), 1e-5, p(gamma), p(sg), p(sx),
                                       float(M), p(dx), N), "apply")
        torch.cuda.synchronize()
        return sg.cpu(), sx.cpu(), dx.cpu()

    odd = torch.empty(M * N + 1, device="cuda")[1:].view(M, N)  # 4-byte aligned only: the scalar forms
    odd.copy_(g)
    a, b = run(g), run(odd)
    for name, u, v in zip(("sum_g", "sum_gx", "dx"), a, b):
        assert torch.equal(u, v), (name, float((u - v).abs().max()))
    gd = g.double() + (g2.double() if two else 0)
    inv = 1.0 / torch.sqrt(var.double() + 1e-5)
    xh = (x.double() - mean.double()) * inv
    sg_ref, sx_ref = gd.sum(0), (gd * xh).sum(0)
    dx_ref = gamma.double() * inv * (gd - sg_ref / M - xh * sx_ref / M)
    assert rel_err(a[0], sg_ref) < 1e-5 and rel_err(a[1], sx_ref) < 1e-5 and rel_err(a[2], dx_ref) < 1e-5


@pytest.mark.parametrize("b_kc", [False, True])
@pytest.mark.parametrize("M,N,K", [(640, 1100, 1938), (130, 48, 18), (4096, 260, 50)])
def test_gemm_f32_ragged_k_pitched(M, N, K, b_kc):
    """k-contiguous operands whose K is not a multiple of 4 at a 16-byte aligned pitch (the 1938-senone head's gradient at
    a pitch of 1940): the LDS-DMA form fetches the 16-byte piece that straddles the end of the reduction and zeroes its
    tail in LDS - the pad columns hold NaN here.  Against fp64 and bit for bit against the first form."""
    import importlib

    lib = importlib.import_module("pytorch-kaldi_amd._lib").load()
    g = torch.Generator().manual_seed(M + N + K)
    Kp = (K + 3) // 4 * 4
    A = torch.full((M, Kp), float("nan"))
    A[:, :K] = torch.randn(M, K, generator=g)
    if b_kc:
        B = torch.full((N, Kp), float("nan"))
        B[:, :K] = torch.randn(N, K, generator=g)
        b_rs, b_cs, Bref = 1, Kp, B[:, :K].t()
    else:
        B = torch.randn(K, N, generator=g)
        b_rs, b_cs, Bref = N, 1, B
    Ad, Bd = A.cuda(), B.cuda()
    outs = []
    for form in (1, 0):
        lib.pk_gemm_f32_set_form(form)
        try:
            C = torch.zeros(M, N).cuda()
            F_.gemm(M, N, K, Ad, Kp, 1, Bd, b_rs, b_cs, C, N, prec="fp32")
            torch.cuda.synchronize()
            outs.append(C.cpu())
        finally:
            lib.pk_gemm_f32_set_form(0)
    assert torch.equal(outs[0], outs[1])
    assert rel_err(outs[1], A[:, :K].double() @ Bref.double()) < 2e-6


def test_logsoftmax_bwd_pitched_head():
    """LogSoftmaxFn.backward of a row-streaming head with 1938 classes hands its gradient over at a pitch of 1940 (pad
    zeroed); LinearFn.backward reads it there: x / weight / bias gradients against fp64 torch."""
    g = torch.Generator().manual_seed(5)
    M, K, N = 4096, 64, 1938
    x, w, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) * 0.1, torch.randn(N, generator=g)
    lab = torch.randint(0, N, (M,), generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    TF.nll_loss(TF.log_softmax(TF.linear(xr, wr, br), 1), lab).backward()
    xe, we, be = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
    F_.set_precision("fp32")
    TF.nll_loss(F_.log_softmax(F_.linear(xe, we, be)), lab.cuda()).backward()
    for a, r in ((xe, xr), (we, wr), (be, br)):
        assert rel_err(a.grad, r.grad) < 5e-6


def test_gemm_strided_rows_and_unaligned():
    """forward_model hands column slices (row stride = feat + labels, utils.py:2321)."""
    g = torch.Generator().manual_seed(1)
    wide = torch.randn(100, 43, generator=g).cuda()
    A = wide[:, 1:41]  # unaligned base, ld 43
    W = torch.randn(64, 40, generator=g).cuda()
    C = torch.empty(100, 64).cuda()
    F_.gemm(100, 64, 40, A, 43, 1, W, 1, 40, C, 64, prec="fp32")
    assert rel_err(C, A.cpu().double() @ W.cpu().double().t()) < 2e-6


def test_linear_autograd():
    g = torch.Generator().manual_seed(2)
    x = torch.randn(50, 33, generator=g)
    w = torch.randn(21, 33, generator=g)
    b = torch.randn(21, generator=g)
    cot = torch.randn(50, 21, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    (TF.linear(xr, wr, br) * cot.double()).sum().backward()
    xe, we, be = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
    (F_.linear(xe, we, be) * cot.cuda()).sum().backward()
    for a, r in ((xe, xr), (we, wr), (be, br)):
        assert rel_err(a.grad, r.grad) < 2e-6


@pytest.mark.parametrize("M,N", [(7, 5), (1000, 70), (128000 // 8, 1100)])
def test_batchnorm_act_drop(M, N):
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, N, generator=g) * 2 + 0.5
    bn = torch.nn.BatchNorm1d(N, momentum=0.05)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(N, generator=g) + 0.5)
        bn.bias.copy_(torch.randn(N, generator=g))
    mask = torch.bernoulli(torch.full((M, N), 0.85), generator=g) / 0.85
    cot = torch.randn(M, N, generator=g)
    xr = x.clone().requires_grad_(True)
    yr = torch.relu(bn(xr)) * mask
    (yr * cot).sum().backward()
    bn2 = torch.nn.BatchNorm1d(N, momentum=0.05)
    bn2.load_state_dict({k: v for k, v in bn.state_dict().items()})
    with torch.no_grad():
        bn2.running_mean.zero_(); bn2.running_var.fill_(1.0); bn2.num_batches_tracked.zero_()
    bn2.cuda()
    xe = x.clone().cuda().requires_grad_(True)
    ye = F_.norm_act_drop(xe, bn2, True, True, "relu", mask.cuda())
    (ye * cot.cuda()).sum().backward()
    assert rel_err(ye, yr) < 1e-5
    assert rel_err(xe.grad, xr.grad) < 1e-4
    assert rel_err(bn2.weight.grad, bn.weight.grad) < 1e-4
    assert rel_err(bn2.bias.grad, bn.bias.grad) < 1e-4
    assert rel_err(bn2.running_mean, bn.running_mean) < 1e-5
    assert rel_err(bn2.running_var, bn.running_var) < 1e-5


@pytest.mark.parametrize("rows,F", [(3, 7), (64, 550), (16, 3200)])
def test_layernorm(rows, F):
    import pk_oracle as O

    g = torch.Generator().manual_seed(rows + F)
    x = torch.randn(rows, F, generator=g)
    gamma = torch.rand(F, generator=g) + 0.5
    beta = torch.randn(F, generator=g)
    cot = torch.randn(rows, F, generator=g)
    xr, gr, br = (t.double().requires_grad_(True) for t in (x, gamma, beta))
    (O.layer_norm(xr, gr, br) * cot.double()).sum().backward()
    xe, ge, be = (t.clone().cuda().requires_grad_(True) for t in (x, gamma, beta))
    y = F_.layer_norm(xe, ge, be, 1e-6)
    (y * cot.cuda()).sum().backward()
    assert rel_err(y, O.layer_norm(x.double(), gamma.double(), beta.double())) < 1e-5
    for a, r in ((xe, xr), (ge, gr), (be, br)):
        assert rel_err(a.grad, r.grad) < 1e-5


@pytest.mark.parametrize("seed", [1, 4234])
def test_reference_mask_stream_on_the_device(seed):
    """PK_MASK_RNG=reference (functional._RefRng, pk_mt19937_bernoulli): the masks the reference draws with
    torch.bernoulli(torch.Tensor(rows, H).fill_(1 - p)) on the global CPU generator (neural_networks.py:1102-1107), layer
    after layer, call after call - bit-identical from the device mirror of that generator, and the CPU generator
    bit-identical afterwards (so that whatever draws next continues the reference's stream).  Also: a re-seed between
    two masks is noticed; a generator that has drawn before (left != 1) is mirrored mid-block."""
    dev = torch.device("cuda")
    shapes = [(256, 550, 0.2), (256, 550, 0.2), (10, 7, 0.5), (3, 1100, 0.25), (1, 1, 0.1), (256, 550, 0.2)]
    torch.manual_seed(seed)
    torch.rand(37)  # (the generator is mid-block when the first mask is asked for)
    ref = [torch.bernoulli(torch.Tensor(r, h).fill_(1 - p)) for r, h, p in shapes]
    after_ref = torch.get_rng_state().clone()
    nxt_ref = torch.rand(5)
    torch.manual_seed(seed)
    torch.rand(37)
    F_._RefRng.dev = None
    got = [F_.ref_rng_mask(r, h, p, dev) for r, h, p in shapes]
    torch.cuda.synchronize()
    F_._RefRng.sync_back()
    assert torch.equal(torch.get_rng_state(), after_ref)
    assert torch.equal(torch.rand(5), nxt_ref)
    for m, r in zip(got, ref):
        assert torch.equal(m.cpu(), r)
    # a re-seed between two masks: the mirror follows the CPU generator
    F_.ref_rng_mask(4, 4, 0.5, dev)
    torch.manual_seed(seed + 1)
    r2 = torch.bernoulli(torch.Tensor(20, 30).fill_(0.7))
    torch.manual_seed(seed + 1)
    assert torch.equal(F_.ref_rng_mask(20, 30, 0.3, dev).cpu(), r2)
    F_._RefRng.sync_back()


@pytest.mark.parametrize("N,K", [(128, 129), (80, 251), (7, 3), (16, 33)])
def test_sinc_bank_in_one_launch(N, K):
    """pk_sinc_bank_fwd / _bwd (SincConv.forward up to self.filters, neural_networks.py:1789-1800): the module's bank
    against the oracle's synthesis - filters against its fp32 evaluation (what the reference computes) and against fp64,
    the ANALYTIC parameter gradients against autograd through the fp64 evaluation."""
    import pk_oracle as O

    nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
    torch.manual_seed(N + K)
    conv = nn_amd.SincConv(1, N, K, sample_rate=16000, min_low_hz=50, min_band_hz=50)
    with torch.no_grad():  # move the mel initialisation around a little, both signs (abs() in the synthesis)
        conv.low_hz_.mul_(1.0 + 0.1 * torch.randn(N, 1)).mul_(torch.where(torch.rand(N, 1) > 0.8, -1.0, 1.0))
        conv.band_hz_.mul_(1.0 + 0.1 * torch.randn(N, 1)).mul_(torch.where(torch.rand(N, 1) > 0.8, -1.0, 1.0))
    lo, ba = conv.low_hz_.detach().clone(), conv.band_hz_.detach().clone()
    cot = torch.randn(N, 1, conv.kernel_size)
    ref32 = O.sinc_filters(lo, ba, K, 16000, 50, 50)
    lo64, ba64 = lo.double().requires_grad_(True), ba.double().requires_grad_(True)
    ref64 = O.sinc_filters(lo64, ba64, K, 16000, 50, 50)
    (ref64 * cot.double()).sum().backward()
    conv = conv.cuda()
    f = conv.filters()
    (f * cot.cuda()).sum().backward()
    assert tuple(f.shape) == tuple(ref32.shape)
    assert rel_err(f, ref32) < 2e-6 and rel_err(f, ref64) < 2e-5   # (fp32 sin / division of arguments up to ~200 rad)
    assert rel_err(conv.low_hz_.grad, lo64.grad) < 1e-4, rel_err(conv.low_hz_.grad, lo64.grad)
    assert rel_err(conv.band_hz_.grad, ba64.grad) < 1e-4, rel_err(conv.band_hz_.grad, ba64.grad)


@pytest.mark.parametrize("B,C,L,act,drop", [(3, 5, 36, "relu", True), (4, 60, 340, "relu", True), (2, 128, 1024, "leaky_relu", False),
                                           (5, 7, 112, "tanh", True), (2, 3, 1027, "relu", False), (3, 2, 3, "linear", True)])
def test_conv_layer_tail_in_one_launch(B, C, L, act, drop):
    """pk_ln_last_act_drop_fwd / _bwd: drop(act(LayerNorm(z))) of a conv layer (features [C, L], statistics over the last
    dim: neural_networks.py:1510-1512, 1546-1552) against an fp64 evaluation of the oracle's LayerNorm + activation + mask,
    and against the launches it replaces (layer_norm_last + norm_act_drop)."""
    import pk_oracle as O

    g = torch.Generator().manual_seed(B * 1000 + C * 10 + L)
    z = torch.randn(B, C, L, generator=g) * 2 + 0.3
    gamma = torch.rand(C, L, generator=g) + 0.5
    beta = torch.randn(C, L, generator=g) * 0.3
    cot = torch.randn(B, C, L, generator=g)
    mask = ((torch.rand(B, C, L, generator=g) > 0.2).float() / 0.8) if drop else None
    zr, gr, br = (t.double().requires_grad_(True) for t in (z, gamma, beta))
    yr = O.activation(act, O.layer_norm(zr, gr, br))
    if mask is not None:
        yr = yr * mask.double()
    (yr * cot.double()).sum().backward()
    ze, ge, be = (t.clone().cuda().requires_grad_(True) for t in (z, gamma, beta))
    y = F_.ln_last_act_drop(ze, ge, be, 1e-6, act, None if mask is None else mask.cuda())
    (y * cot.cuda()).sum().backward()
    assert rel_err(y, yr) < 1e-5
    for a, r in ((ze, zr), (ge, gr), (be, br)):
        assert rel_err(a.grad, r.grad) < 2e-5, (rel_err(a.grad, r.grad), tuple(r.shape))
    with torch.no_grad():  # the launches it replaces: the same values
        m2 = None if mask is None else mask.cuda().reshape(B * C, L)
        old = F_.norm_act_drop(F_.layer_norm_last(z.cuda(), gamma.cuda(), beta.cuda(), 1e-6).reshape(B * C, L), None, False, True,
                               act, m2).view(B, C, L)
    assert rel_err(y.detach(), old) < 2e-6  # (another summation order in the row statistics, one fused multiply-add)


@pytest.mark.parametrize("rows,N", [(1, 1), (5, 48), (7, 64), (9, 200), (33, 1000), (300, 1938), (3, 2048), (6, 3400)])
def test_logsoftmax(rows, N):
    g = torch.Generator().manual_seed(rows + N)
    x = torch.randn(rows, N, generator=g) * 4
    cot = torch.randn(rows, N, generator=g)
    xr = x.double().requires_grad_(True)
    (TF.log_softmax(xr, 1) * cot.double()).sum().backward()
    xe = x.clone().cuda().requires_grad_(True)
    y = F_.log_softmax(xe)
    (y * cot.cuda()).sum().backward()
    assert rel_err(y, TF.log_softmax(x.double(), 1)) < 1e-6
    assert rel_err(xe.grad, xr.grad) < 1e-5


CONV_SHAPES = [
    # B, Cin, L, Cout, K, pool
    (2, 1, 200, 8, 33, 3),
    (3, 8, 56, 6, 5, 2),
    (2, 6, 26, 5, 3, 2),
    (4, 1, 3200, 128, 129, 3),
    (4, 60, 340, 60, 5, 3),
    (2, 3, 50, 17, 7, 1),
    (3, 40, 100, 80, 10, 3),     # cnn_len_filt = 10,3,3 / cnn_max_pool_len = 3,2,1 (TIMIT_CNN cfg)
    (3, 80, 90, 60, 3, 2),
    (3, 60, 43, 60, 3, 1),
    (130, 20, 700, 33, 5, 3),    # more (batch, tile) pairs than reduction slices; ragged channel tiles
    (2, 5, 2000, 20, 251, 3),    # long filters with several input channels
    (2, 2, 64, 3, 4, 5),         # pool > 3: generic forward kernel
]


@pytest.mark.parametrize("B,Cin,L,Cout,K,pool", CONV_SHAPES)
def test_conv_pool(B, Cin, L, Cout, K, pool):
    g = torch.Generator().manual_seed(B + Cin + L)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    Lp = (L - K + 1) // pool
    cot = torch.randn(B, Cout, Lp, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = TF.max_pool1d(TF.conv1d(xr, wr, br), pool)
    (yr * cot.double()).sum().backward()
    xe, we, be = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
    ye = F_.conv1d_pool(xe, we, be, pool)
    (ye * cot.cuda()).sum().backward()
    assert rel_err(ye, yr) < 1e-5
    assert rel_err(xe.grad, xr.grad) < 1e-5
    assert rel_err(we.grad, wr.grad) < 1e-5
    assert rel_err(be.grad, br.grad) < 1e-5


def _rb64(t):
    return t.to(torch.bfloat16).to(t.dtype)


class _Bf16ConvRef(torch.autograd.Function):
    """fp64 evaluation of the bf16-operand convolution: x, w and the output gradient rounded to bf16."""

    @staticmethod
    def forward(ctx, x, w):
        xb, wb = _rb64(x), _rb64(w)
        ctx.save_for_backward(xb, wb)
        return TF.conv1d(xb, wb)

    @staticmethod
    def backward(ctx, g):
        xb, wb = ctx.saved_tensors
        gb = _rb64(g)
        return torch.nn.grad.conv1d_input(xb.shape, wb, gb), torch.nn.grad.conv1d_weight(xb, wb.shape, gb)


@pytest.mark.parametrize("B,Cin,L,Cout,K,pool", CONV_SHAPES + [(128, 1, 3200, 128, 129, 3), (16, 128, 1024, 60, 5, 3),
                                                               (5, 60, 112, 60, 3, 2)])
def test_conv_pool_bf16(B, Cin, L, Cout, K, pool, monkeypatch):
    """The perf-mode convolutions (pk_conv_bf16.hip: implicit GEMM on the bf16 matrix pipe, eight
    shifted copies of the staged window) against an fp64 evaluation of the same bf16-operand algorithm: outputs, arg-max
    routing, dx, dw, db."""
    monkeypatch.setenv("PK_CONV_BF16", "2")  # every covered layer, also those with fewer than 8 input channels
    lib = _lib.load()
    if lib.pk_conv_bf16_covers(Cin, Cout, K, pool) != 1:
        pytest.skip("layer not covered by the bf16 kernels (pool width / channel count): fp32 kernels")
    g = torch.Generator().manual_seed(B + Cin + L)
    x = torch.randn(B, Cin, L, generator=g)
    w = torch.randn(Cout, Cin, K, generator=g) / (Cin * K) ** 0.5
    b = torch.randn(Cout, generator=g)
    Lp = (L - K + 1) // pool
    cot = torch.randn(B, Cout, Lp, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    yr = TF.max_pool1d(_Bf16ConvRef.apply(xr, wr) + br.view(1, -1, 1), pool)
    (yr * cot.double()).sum().backward()
    F_.set_precision("bf16")
    try:
        xe, we, be = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
        ye = F_.conv1d_pool(xe, we, be, pool)
        (ye * cot.cuda()).sum().backward()
        torch.cuda.synchronize()
    finally:
        F_.set_precision("fp32")
    assert rel_err(ye, yr) < 2e-5
    # (an arg-max that flips between the fp32 and the fp64 accumulation re-routes one element of dz: 1 / sqrt(N) of a norm)
    # (layers with fewer than 8 input channels take their data gradient from the exact-fp32 kernel: it differs from this
    # bf16-rounded reference by the rounding itself, measured 2.4e-3)
    assert rel_err(xe.grad, xr.grad) < (2e-3 if Cin >= 8 else 1e-2)
    assert rel_err(we.grad, wr.grad) < 2e-3
    assert rel_err(be.grad, br.grad) < 2e-5


def test_optimizers_match_torch():
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(4)
    p0 = torch.randn(10007, generator=g)
    grads = [torch.randn(10007, generator=g) for _ in range(3)]
    # RMSprop as utils.optimizer_init builds it (lr 4e-4, alpha .95, eps 1e-8)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.RMSprop([pr], lr=4e-4, alpha=0.95, eps=1e-8)
    pe, sq = p0.clone().cuda(), torch.zeros(10007).cuda()
    for gr in grads:
        pr.grad = gr.clone()
        opt.step()
        ge = gr.cuda()
        _lib.check(lib.pk_rmsprop_step(st, pe.data_ptr(), ge.data_ptr(), sq.data_ptr(), 10007, 4e-4, 0.95, 1e-8, 0.0), "rms")
    assert rel_err(pe, pr) < 1e-6
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([pr], lr=0.08, momentum=0.9, weight_decay=1e-4)
    pe, buf = p0.clone().cuda(), torch.zeros(10007).cuda()
    for i, gr in enumerate(grads):
        pr.grad = gr.clone()
        opt.step()
        ge = gr.cuda()
        _lib.check(lib.pk_sgd_step(st, pe.data_ptr(), ge.data_ptr(), buf.data_ptr(), 10007, 0.08, 0.9, 1e-4, int(i == 0)), "sgd")
    assert rel_err(pe, pr) < 1e-6
    # Adam as utils.py:2130-2146 builds it, with and without amsgrad
    for ams in (False, True):
        pr = p0.clone().requires_grad_(True)
        opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, amsgrad=ams)
        pe = p0.clone().cuda()
        m, v, vm = (torch.zeros(10007).cuda() for _ in range(3))
        for i, gr in enumerate(grads):
            pr.grad = gr.clone()
            opt.step()
            ge = gr.cuda()
            _lib.check(lib.pk_adam_step(st, pe.data_ptr(), ge.data_ptr(), m.data_ptr(), v.data_ptr(),
                                        vm.data_ptr() if ams else None, 10007, 1e-3, 0.9, 0.999, 1e-8, 1e-4, i + 1), "adam")
        assert rel_err(pe, pr) < 1e-6


@pytest.mark.parametrize("kind,kw", [("rmsprop", {"weight_decay": 0.05}), ("sgd", {"momentum": 0.9, "weight_decay": 0.05}),
                                     ("adam", {"weight_decay": 0.05})])
def test_fused_optimizer_skips_never_used_parameters_like_torch(kind, kw):
    """The reference's classes carry ln / bn sub-modules forward never calls; torch leaves their .grad None and its
    optimizers skip them - with weight decay / momentum / Adam a ZERO gradient would still move them.  The fused
    optimizers keep them behind the active range: untouched, and without a state entry in state_dict()."""
    optim_ = importlib.import_module("pytorch-kaldi_amd.optim")
    nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
    opts = {"ligru_lay": "24,16", "ligru_drop": "0.0,0.0", "ligru_use_laynorm_inp": "False", "ligru_use_batchnorm_inp": "False",
            "ligru_use_laynorm": "False,True", "ligru_use_batchnorm": "True,False", "ligru_bidir": "True",
            "ligru_act": "relu,relu", "ligru_orthinit": "True", "use_cuda": "True", "to_do": "train"}
    torch.manual_seed(3)
    net_e = nn_amd.liGRU(dict(opts), 9).cuda().train()
    net_t = nn_amd.liGRU(dict(opts), 9).cuda().train()
    net_t.load_state_dict(net_e.state_dict())
    x = torch.randn(7, 3, 9, generator=torch.Generator().manual_seed(1)).cuda()
    mk = {"rmsprop": torch.optim.RMSprop, "sgd": torch.optim.SGD, "adam": torch.optim.Adam}[kind]
    lr = 0.01
    opt_t = mk(net_t.parameters(), lr=lr, **kw)
    opt_e = optim_.FusedOptimizer(optim_.FlatParams(net_e), kind, lr, **kw)
    unused = {n for n, p in net_e.named_parameters() if any(p is q for q in net_e.pk_unused_parameters())}
    assert unused == {"ln.0.gamma", "ln.0.beta", "bn_wh.1.weight", "bn_wh.1.bias", "bn_wz.1.weight", "bn_wz.1.bias"}
    before = {n: p.detach().clone() for n, p in net_e.named_parameters()}
    for _ in range(3):
        for net, opt in ((net_t, opt_t), (net_e, opt_e)):
            opt.zero_grad()
            net(x).square().mean().backward()
            opt.step()
    torch.cuda.synchronize()
    for (n, pe), (_, pt) in zip(net_e.named_parameters(), net_t.named_parameters()):
        if n in unused:
            assert pt.grad is None and torch.equal(pe, before[n]) and torch.equal(pt, before[n]), n
        else:  # (Adam's m / (sqrt(v) + eps) amplifies last-bit differences of near-zero gradient elements)
            assert rel_err(pe, pt) < (1e-4 if kind == "adam" else 1e-5), n
    assert sorted(opt_e.state_dict()["state"]) == sorted(opt_t.state_dict()["state"])


@pytest.mark.parametrize("kind", ["rmsprop", "sgd", "adam"])
def test_fused_optimizer_continues_a_torch_checkpoint(kind):
    """optimizer_par written by torch.optim (the reference, core.py:708-722) -> fused optimizer -> same next steps,
    and the fused state_dict loads back into torch.optim."""
    optim_ = importlib.import_module("pytorch-kaldi_amd.optim")
    mk = {"rmsprop": lambda ps: torch.optim.RMSprop(ps, lr=4e-4, alpha=0.95, eps=1e-8),
          "sgd": lambda ps: torch.optim.SGD(ps, lr=0.08, momentum=0.9),
          "adam": lambda ps: torch.optim.Adam(ps, lr=1e-3)}[kind]
    torch.manual_seed(3)
    ref = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5)).cuda()
    eng = torch.nn.Sequential(torch.nn.Linear(33, 17), torch.nn.Linear(17, 5)).cuda()
    eng.load_state_dict(ref.state_dict())
    x = torch.randn(64, 33).cuda()
    ropt = mk(list(ref.parameters()))

    def rstep(net, opt):
        opt.zero_grad()
        net(x).pow(2).mean().backward()
        opt.step()

    for _ in range(2):
        rstep(ref, ropt)
    eng.load_state_dict(ref.state_dict())
    flat = optim_.FlatParams(eng)
    kw = {"rmsprop": dict(alpha=0.95, eps=1e-8), "sgd": dict(momentum=0.9), "adam": dict()}[kind]
    fopt = optim_.FusedOptimizer(flat, kind, {"rmsprop": 4e-4, "sgd": 0.08, "adam": 1e-3}[kind], **kw)
    fopt.load_state_dict(ropt.state_dict())
    for _ in range(2):
        rstep(ref, ropt)
        rstep(eng, fopt)
    for a, b in zip(eng.parameters(), ref.parameters()):
        assert rel_err(a, b) < 1e-5
    # and back: torch continues from the fused optimizer's state
    back = mk(list(ref.parameters()))
    back.load_state_dict(fopt.state_dict())
    rstep(ref, back)
    rstep(eng, fopt)
    for a, b in zip(eng.parameters(), ref.parameters()):
        assert rel_err(a, b) < 1e-5


# ---- perf-mode GEMM on bf16 operands (LDS-DMA staging, transpose reads for k-major operands) --------
def _bf_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


def test_cvt_bf16_segments():
    g = torch.Generator().manual_seed(8)
    x = torch.randn(37, 2 * 550, generator=g).cuda()
    out = F_.cvt_bf16(x, 2, 550, 552)
    assert out.shape == (37, 1152) and out.dtype == torch.bfloat16
    ref = torch.zeros(37, 1152, dtype=torch.bfloat16)
    ref[:, :550] = x[:, :550].cpu().to(torch.bfloat16)
    ref[:, 552:1102] = x[:, 550:].cpu().to(torch.bfloat16)
    assert torch.equal(out.cpu().view(torch.int16), ref.view(torch.int16))
    wide = torch.randn(9, 45, generator=g).cuda()
    out = F_.cvt_bf16(wide[:, 1:41])  # strided, unaligned source
    assert torch.equal(out.cpu()[:, :40].view(torch.int16), wide[:, 1:41].cpu().to(torch.bfloat16).view(torch.int16))
    assert float(out[:, 40:].float().abs().max()) == 0.0


BF_GEMM_SHAPES = [
    # M, N, K, a_kc, b_kc, splitk
    (16, 16, 64, 1, 1, 1),
    (128, 128, 64, 1, 1, 1),
    (128, 128, 64, 0, 0, 1),
    (128, 128, 64, 1, 0, 1),
    (128, 128, 64, 0, 1, 1),
    (37, 53, 40, 1, 1, 1),
    (300, 1100, 1100, 1, 1, 1),
    (2050, 130, 552, 1, 0, 1),
    (1100, 40, 3000, 0, 0, 8),
    (550, 550, 2111, 0, 0, 5),
    (1100, 1100, 6400, 0, 0, 16),
    (1, 1, 8, 1, 1, 1),
    (5, 1938, 1100, 1, 1, 1),
    # the 256 x 256 eight-phase kernel (M, N >= 384): every operand layout, a ragged last k-tile (1100 = 17 x 64 + 12),
    # one k-tile only, a single short k-tile, out-of-range quadrants (N = 1100: the last tile column has 76 columns)
    (512, 512, 192, 1, 1, 1),
    (512, 512, 192, 0, 0, 1),
    (512, 512, 192, 1, 0, 1),
    (512, 512, 192, 0, 1, 1),
    (700, 1100, 1100, 1, 1, 1),
    (390, 1938, 1100, 1, 0, 1),
    (384, 384, 64, 1, 1, 1),
    (384, 400, 8, 0, 0, 1),
    (640, 900, 136, 1, 0, 1),
    (1100, 1104, 6400, 0, 0, 4),
    (1938, 1100, 5000, 0, 0, 3),
    # small-batch products: the four-stage ring of the 128-tile (<= 64 tiles, K >= 256, k-contiguous A)
    (128, 1024, 1024, 1, 1, 1),
    (128, 1024, 440, 1, 1, 1),
    (128, 1024, 1000, 1, 0, 1),
    (100, 1944, 264, 1, 1, 1),
    (77, 136, 3000, 1, 0, 1),
    (1024, 1024, 128, 0, 0, 1),
    # round 4: the reduction of a small-batch product split over the grid (pk_gemm_bf16_small_splitk: output layers with 48
    # / 1938 columns, their input gradient over K = 1938, a two-row batch), weight gradients over a short reduction on
    # 64 x 64 tiles (ragged rows / columns / reduction)
    (128, 48, 1024, 1, 1, 1),
    (128, 1024, 1938, 1, 0, 1),
    (2, 64, 512, 1, 1, 1),
    (128, 3300, 1024, 1, 0, 1),
    (1938, 1024, 128, 0, 0, 1),
    (100, 72, 37, 0, 0, 1),
    (1024, 3300, 128, 0, 0, 1),
]


@pytest.mark.parametrize("tile", [0, 128, 256])
@pytest.mark.parametrize("M,N,K,a_kc,b_kc,splitk", BF_GEMM_SHAPES)
def test_gemm_bf16_operands(M, N, K, a_kc, b_kc, splitk, tile):
    """tile: 0 = the library's own choice, 128 / 256 = that block tile forced for every shape."""
    import importlib

    lib = importlib.import_module("pytorch-kaldi_amd._lib").load()
    lib.pk_gemm_bf16_set_tile(tile)
    try:
        _gemm_bf16_case(M, N, K, a_kc, b_kc, splitk)
    finally:
        lib.pk_gemm_bf16_set_tile(0)


def _gemm_bf16_case(M, N, K, a_kc, b_kc, splitk):
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K + a_kc + 2 * b_kc)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(K, N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    bias = torch.randn(N, generator=g)
    Ab = F_.cvt_bf16((A if a_kc else A.t().contiguous()).cuda())   # [M][Kp] or [K][Mp]
    Bb = F_.cvt_bf16((B.t().contiguous() if b_kc else B).cuda())   # [N][Kp] or [K][Np]
    C = C0.clone().cuda()
    F_.gemm_bf16(M, N, K, Ab, Ab.shape[1], a_kc, Bb, Bb.shape[1], b_kc, C, N, alpha=0.5, beta=2.0, bias=bias.cuda(),
                 splitk=splitk)
    torch.cuda.synchronize()
    ref = 0.5 * (_bf_round(A) @ _bf_round(B)) + 2.0 * C0.double() + bias.double()
    assert rel_err(C, ref) < 2e-5  # operands are exactly the bf16-rounded values; only fp32 accumulation differs


def test_gemm_bf16_offset_operands():
    """The dU shape: sub-matrices addressed by element offsets into larger bf16 buffers."""
    g = torch.Generator().manual_seed(77)
    T, Bn, H, GH = 6, 8, 24, 48
    dG = torch.randn(T * Bn, GH, generator=g)
    Y = torch.randn(T * Bn, 2 * H, generator=g)
    dGb, Yb = F_.cvt_bf16(dG.cuda()), F_.cvt_bf16(Y.cuda(), 2, H, H)
    K = (T - 1) * Bn
    C = torch.empty(GH, H).cuda()
    F_.gemm_bf16(GH, H, K, (dGb, Bn * dGb.shape[1]), dGb.shape[1], 0, (Yb, H), Yb.shape[1], 0, C, H)
    ref = _bf_round(dG[Bn:]).t() @ _bf_round(Y[:K, H:])
    assert rel_err(C, ref) < 2e-5


def test_linear_autograd_bf16_mode():
    g = torch.Generator().manual_seed(21)
    x = torch.randn(700, 330, generator=g)
    w = torch.randn(210, 330, generator=g) / 18
    b = torch.randn(210, generator=g)
    cot = torch.randn(700, 210, generator=g)
    xr, wr, br = (t.double().requires_grad_(True) for t in (x, w, b))
    (TF.linear(xr, wr, br) * cot.double()).sum().backward()
    F_.set_precision("bf16")
    try:
        xe, we, be = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
        y = F_.linear(xe, we, be)
        (y * cot.cuda()).sum().backward()
    finally:
        F_.set_precision("fp32")
    assert rel_err(y, TF.linear(x.double(), w.double(), b.double())) < 1e-2
    for a, r in ((xe, xr), (we, wr), (be, br)):
        assert rel_err(a.grad, r.grad) < 1e-2


@pytest.mark.parametrize("rows,K,N", [(5, 24, 1), (70, 40, 48), (300, 330, 1938), (64 * 40 + 3, 64, 2048), (9, 16, 200)])
def test_fused_output_layer_matches_the_two_nodes(rows, K, N):
    """LinearLogSoftmaxFn (perf mode: padded GEMM output, dz written once as bf16, bias gradient on the way) against
    LinearFn + LogSoftmaxFn in the same mode: same GEMM kernels on the same rounded operands -> outputs, dX and dW
    bit for bit, the bias gradient to summation order; and both against fp64 torch on bf16-rounded operands."""
    g = torch.Generator().manual_seed(rows + N)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) / (K ** 0.5)
    b = torch.randn(N, generator=g)
    cot = torch.randn(rows, N, generator=g)
    F_.set_precision("bf16")
    try:
        outs = []
        for fused in (True, False):
            xe, we, be = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
            y = F_.linear_log_softmax(xe, we, be) if fused else F_.log_softmax(F_.linear(xe, we, be))
            (y * cot.cuda()).sum().backward()
            torch.cuda.synchronize()
            outs.append((y.detach(), xe.grad, we.grad, be.grad))
    finally:
        F_.set_precision("fp32")
    (y1, dx1, dw1, db1), (y2, dx2, dw2, db2) = outs
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2) and torch.equal(dw1, dw2)
    assert rel_err(db1, db2) < 1e-5
    rb = lambda t: t.to(torch.bfloat16).double()
    xr, wr = rb(x).requires_grad_(True), rb(w).requires_grad_(True)
    br = b.double().requires_grad_(True)
    yr = TF.log_softmax(TF.linear(xr, wr, br), 1)
    (yr * cot.double()).sum().backward()
    assert rel_err(y1, yr) < 1e-5
    assert rel_err(db1, br.grad) < 1e-5
    assert rel_err(dx1, xr.grad) < 1e-2 and rel_err(dw1, wr.grad) < 1e-2  # dz rounded to bf16 for the two GEMMs


@pytest.mark.parametrize("rows,K,N,ignored", [(6, 24, 1, 0), (70, 40, 48, 3), (300, 330, 1938, 0), (64 * 40 + 3, 64, 2048, 40),
                                               (9, 16, 200, 9)])
def test_head_nll_matches_nll_loss_behind_the_head(rows, K, N, ignored):
    """functional.head_nll (cost straight behind the head's inputs, one-hot gradient never formed, frame errors counted
    in the same pass) against torch's nll_loss / argmax on the same head output: loss to summation order, error rate
    exactly, dX / dW bit for bit (same dz bits into the same GEMMs), bias gradient to summation order."""
    g = torch.Generator().manual_seed(rows * 3 + N)
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(N, K, generator=g) / (K ** 0.5)
    b = torch.randn(N, generator=g)
    lab = torch.randint(0, N, (rows,), generator=g)
    lab[:ignored] = -100
    F_.set_precision("bf16")
    try:
        res = []
        for fused in (True, False):
            xe, we, be = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
            y = F_.linear_log_softmax(xe, we, be)
            if fused:
                loss, stats = F_.head_nll(y, lab.cuda())
                err = stats[1]
                assert float(stats[2]) == rows - ignored and float(stats[3]) == 0
            else:
                loss = TF.nll_loss(y, lab.cuda())
                err = (y.argmax(1) != lab.cuda()).float().mean()
            (loss * 1.7).backward()
            torch.cuda.synchronize()
            res.append((loss.detach(), err, xe.grad, we.grad, be.grad))
    finally:
        F_.set_precision("fp32")
    (l1, e1, dx1, dw1, db1), (l2, e2, dx2, dw2, db2) = res
    if ignored == rows:
        assert torch.isnan(l1) and torch.isnan(l2)
        assert float(dx1.abs().max()) == 0.0  # torch: 0 * nan-free zeros; the engine: scale 0
        return
    assert abs(float(l1) - float(l2)) <= 2e-6 * abs(float(l2))
    assert float(e1) == pytest.approx(float(e2), abs=1e-7)
    assert torch.equal(dx1, dx2) and torch.equal(dw1, dw2)
    assert rel_err(db1, db2) < 1e-5


@pytest.mark.parametrize("rows,K,N", [(300, 32, 48), (1000, 64, 1938), (257, 16, 200), (64, 40, 1024), (5000, 24, 64)])
def test_cost_from_the_arg_max_positions_equals_the_second_pass(rows, K, N, monkeypatch):
    """Round 4: the fused output layer writes every row's arg-max position next to the log-posteriors and the cost takes
    one gathered load + one compare per row (pk_logsoftmax_fwd_ld_argmax / pk_nll_err_fwd_argmax) instead of reading the
    rows again (pk_nll_err_fwd).  Same four numbers: error rate, counted rows and bad labels EXACTLY - on logits built to
    tie constantly (small integers: the first index must win), with NaN rows (column 0 by convention), ignored and
    out-of-range labels - the loss to summation order; and the positions themselves against torch on tie-free rows."""
    g = torch.Generator().manual_seed(rows + N)
    x = torch.randint(-1, 2, (rows, K), generator=g).float()
    w = torch.randint(-1, 2, (N, K), generator=g).float() * 0.5
    b = torch.randint(-2, 3, (N,), generator=g).float() * 0.25
    x[3] = float("nan")
    x[rows - 2] = float("nan")
    lab = torch.randint(0, N, (rows,), generator=g)
    lab[3] = lab[rows - 2] = -100          # (their log-posteriors are NaN: keep the loss finite)
    lab[7:17] = -100
    F_.set_precision("bf16")
    try:
        out = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("PK_EXPERIMENT", "head_argmax=" + mode)
            y = F_.linear_log_softmax(x.cuda(), w.cuda(), b.cuda())
            assert y._pk_head[-1] is not None and y._pk_head[-1].dtype == torch.int32
            loss, stats = F_.head_nll(y, lab.cuda())
            bad = lab.clone()
            bad[20] = N + 3
            _, stats_bad = F_.head_nll(y, bad.cuda())
            torch.cuda.synchronize()
            out[mode] = (float(loss), stats.clone(), stats_bad.clone(), y.detach().clone(), y._pk_head[-1].clone())
        F_.label_check_counter("cuda").zero_()
    finally:
        F_.set_precision("fp32")
    (l1, s1, sb1, y1, a1), (l0, s0, sb0, y0, _) = out["1"], out["0"]
    assert torch.equal(y1, y0, ) or torch.equal(torch.nan_to_num(y1), torch.nan_to_num(y0))
    assert torch.equal(s1[1:], s0[1:]) and torch.equal(sb1[1:], sb0[1:]) and float(sb1[3]) == 1.0
    assert abs(l1 - l0) <= 2e-6 * abs(l0) and abs(float(s1[0]) - float(s0[0])) <= 2e-6 * abs(l0)
    # the positions: first index of the row maximum; NaN rows -> 0
    yc = y1.cpu()
    first = torch.full((rows,), -1, dtype=torch.int64)
    for r in range(rows):
        row = yc[r]
        first[r] = 0 if bool(torch.isnan(row).any()) else int((row == row.max()).nonzero()[0])
    assert torch.equal(a1.cpu().long(), first)


def test_head_nll_declines_what_it_does_not_cover():
    F_.set_precision("bf16")
    try:
        x = torch.randn(8, 16, device="cuda", requires_grad=True)
        w = torch.randn(5, 16, device="cuda", requires_grad=True)
        y = F_.linear_log_softmax(x, w, None)
        lab = torch.randint(0, 5, (8,), device="cuda")
        assert F_.head_nll(F_.log_softmax(F_.linear(x, w, None)), lab) is None     # not a fused head
        assert F_.head_nll(y, lab.int()) is None                                    # labels must be int64
        assert F_.head_nll(y, lab[:4]) is None
        bad = lab.clone()
        bad[2] = 7
        loss, stats = F_.head_nll(y, bad)
        assert bool(torch.isnan(loss))   # a label outside [0, classes) poisons the loss (torch: device-side assert) ...
        loss, stats = F_.head_nll(y, bad)  # ... and is counted in place by the same launch (a HIP-graph replay keeps counting)
        F_.note_label_check(stats)         # (round 4: a no-op, kept for callers of the round-3 protocol)
        assert float(F_.label_check_counter("cuda")) == 2.0 and float(stats[3]) == 1.0
        with pytest.raises(_lib.PkError):
            F_.raise_if_bad_labels()
        assert float(F_.label_check_counter("cuda")) == 0.0
        F_.raise_if_bad_labels()
        y2 = F_.linear_log_softmax(x, w, None)
        with torch.no_grad():
            y2.add_(1.0)
        assert F_.head_nll(y2, lab) is None                                         # modified in place since
    finally:
        F_.set_precision("fp32")


@pytest.mark.parametrize("M,N,K,bias", [(16384 + 100, 1100, 200, False), (16384, 1024, 72, True), (300, 1100, 64, False)])
def test_projection_gemm_with_batchnorm_statistics(M, N, K, bias):
    """pk_gemm_bf16_stats: the BatchNorm statistics of the projection taken in the GEMM epilogue (256-tile shapes; the
    third shape takes the fallback: plain GEMM + pk_bn_stats) against torch on the very matrix the GEMM wrote."""
    g = torch.Generator().manual_seed(M + N + K)
    x = (torch.randn(M, K, generator=g) + 0.3).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    b = (torch.randn(N, generator=g) * 3).cuda() if bias else None
    xb, wb = F_.cvt_bf16(x), F_.cvt_bf16(w)
    C = torch.empty(M, N, device="cuda")
    mean, var = F_.gemm_bf16_bn_stats(M, N, K, xb, xb.shape[1], 1, wb, wb.shape[1], 1, C, N, bias=b)
    torch.cuda.synchronize()
    ref = x.to(torch.bfloat16).double() @ w.to(torch.bfloat16).double().t()
    if bias:
        ref = ref + b.double()
    assert rel_err(C, ref) < 1e-5
    Cd = C.double()
    assert rel_err(mean, Cd.mean(0)) < 1e-5
    assert rel_err(var, Cd.var(0, unbiased=False)) < 1e-5


def test_output_layer_on_the_published_bf16_twin():
    """A perf-mode output layer behind a recurrent layer reads the bf16 copy that layer published (direction halves at
    a pitch of Hp, weight copy re-pitched to match) instead of converting the fp32 activation again: same operands, the
    reduction merely walks the zero pad columns too -> results equal to summation order; gradients likewise."""
    g = torch.Generator().manual_seed(77)
    rows, H, N = 4200, 70, 130  # two direction halves of 70 units at a pitch of 72
    x = torch.randn(rows, 2 * H, generator=g)
    w = torch.randn(N, 2 * H, generator=g) / 12
    b = torch.randn(N, generator=g)
    lab = torch.randint(0, N, (rows,), generator=g).cuda()
    F_.set_precision("bf16")
    try:
        res = []
        for twin in (True, False):
            xe, we, be = (t.clone().cuda().requires_grad_(True) for t in (x, w, b))
            if twin:
                xb = F_.cvt_bf16(xe.detach(), 2, H, 72)
                xe._pk_twin = (xb, (2, H, 72), xe._version)
            y = F_.linear_log_softmax(xe, we, be)
            loss, _ = F_.head_nll(y, lab)
            loss.backward()
            torch.cuda.synchronize()
            res.append((y.detach(), xe.grad, we.grad, be.grad))
        assert F_.input_twin(torch.zeros(3, 4, device="cuda")) is None
    finally:
        F_.set_precision("fp32")
    for a, r in zip(*res):
        assert rel_err(a, r) < 1e-5


# --------------------------------------------------------------------------------
# the GEMM shapes the timed step actually launches (BASELINE configs[1]: T*B = 64 000 rows, H = 550, 1938 / 48 classes)
# --------------------------------------------------------------------------------
BASELINE_GEMMS = [
    # name, M, N, K, a_kc, b_kc, split-K (0 = the library's own choice for the shape)
    ("projection", 64000, 1100, 1104, 1, 1, 1),
    ("projection_layer0", 64000, 1100, 40, 1, 1, 1),
    ("dX", 64000, 1100, 1100, 1, 0, 1),
    ("head_1938", 64000, 1938, 1104, 1, 1, 1),
    ("head_48", 64000, 48, 1104, 1, 1, 1),
    ("dW", 1100, 1104, 64000, 0, 0, 0),
    ("dW_layer0", 1100, 40, 64000, 0, 0, 0),
    ("dU", 1104, 550, 63872, 0, 0, 0),
    ("head_dW_1938", 1938, 1100, 64000, 0, 0, 0),
    ("head_dW_48", 48, 1100, 64000, 0, 0, 0),
]


@pytest.mark.parametrize("name,M,N,K,a_kc,b_kc,splitk", BASELINE_GEMMS)
def test_gemm_bf16_baseline_shapes(name, M, N, K, a_kc, b_kc, splitk):
    """pk_gemm_bf16 on the shapes of the benchmarked step, with the tile and split-K the library picks for them (the
    256-tile for the row-streaming and weight-gradient shapes, split-K 4-28 over the 64 000-row reductions).  The fp64
    reference is evaluated on a sample of output rows (every 61st) plus the column sums of the whole output - a
    checksum that sees every row."""
    g = torch.Generator().manual_seed(len(name) * 1000 + M % 977 + N)
    A = torch.randn(M, K, generator=g) / 8
    B = torch.randn(K, N, generator=g) / 8
    Ab = F_.cvt_bf16((A if a_kc else A.t().contiguous()).cuda())
    Bb = F_.cvt_bf16((B.t().contiguous() if b_kc else B).cuda())
    C = torch.full((M, N), float("nan"), device="cuda")
    sk = splitk or F_._splitk_bf(F_._tiles_bf(M, N), K)
    F_.gemm_bf16(M, N, K, Ab, Ab.shape[1], a_kc, Bb, Bb.shape[1], b_kc, C, N, splitk=sk)
    torch.cuda.synchronize()
    Ar, Br = _bf_round(A), _bf_round(B)
    idx = torch.arange(0, M, 61)
    ref_rows = Ar[idx] @ Br
    assert rel_err(C[idx.cuda()], ref_rows) < 2e-5, (name, sk)
    ref_colsum = Ar.sum(0, keepdim=True) @ Br
    got = C.double().sum(0, keepdim=True)
    scale = float((Ar.abs().sum(0, keepdim=True) @ Br.abs()).norm())  # the size of what cancels in a column sum
    assert float((got.cpu() - ref_colsum).norm()) < 2e-6 * scale, (name, sk)
    assert bool(torch.isfinite(C).all())


@pytest.mark.parametrize("K", [1104, 40])
def test_projection_statistics_at_250_row_blocks(K):
    """pk_gemm_bf16_stats at the benchmarked projection shape: 64 000 rows = 250 row blocks of the 256-tile, 1100
    columns, merged by pk_bn_stats_merge - mean / biased variance of every column against fp64 on the matrix the GEMM
    wrote, and that matrix against fp64 on sampled rows."""
    M, N = 64000, 1100
    g = torch.Generator().manual_seed(K)
    x = torch.randn(M, K, generator=g) + 0.25
    w = torch.randn(N, K, generator=g) / K ** 0.5
    xb, wb = F_.cvt_bf16(x.cuda()), F_.cvt_bf16(w.cuda())
    C = torch.empty(M, N, device="cuda")
    mean, var = F_.gemm_bf16_bn_stats(M, N, K, xb, xb.shape[1], 1, wb, wb.shape[1], 1, C, N)
    torch.cuda.synchronize()
    idx = torch.arange(0, M, 61)
    assert rel_err(C[idx.cuda()], _bf_round(x[idx]) @ _bf_round(w).t()) < 2e-5
    Cd = C.double()
    assert rel_err(mean, Cd.mean(0)) < 1e-5
    assert rel_err(var, Cd.var(0, unbiased=False)) < 1e-5


def test_output_layers_on_one_input_share_their_input_gradient():
    """Two perf-mode heads behind one activation (senone + monophone): the second head's dX GEMM accumulates into the
    first one's buffer (functional._DxShare) instead of autograd adding two 282 MB tensors.  Same gradients as the
    unshared path, also over a second backward pass through a retained graph."""
    g = torch.Generator().manual_seed(5)
    rows, K = 4200, 144
    x0 = torch.randn(rows, K, generator=g)
    w1, w2 = torch.randn(37, K, generator=g) / 12, torch.randn(11, K, generator=g) / 12
    l1, l2 = torch.randint(0, 37, (rows,), generator=g).cuda(), torch.randint(0, 11, (rows,), generator=g).cuda()
    F_.set_precision("bf16")
    res = {}
    try:
        for share in (True, False):
            F_._DxShare.on = share
            x = x0.clone().cuda().requires_grad_(True)
            h = x * 1.0  # a non-leaf input, like a recurrent layer's output
            a, b = w1.clone().cuda().requires_grad_(True), w2.clone().cuda().requires_grad_(True)
            y1, y2 = F_.linear_log_softmax(h, a, None), F_.linear_log_softmax(h, b, None)
            loss = F_.head_nll(y1, l1)[0] + 0.5 * F_.head_nll(y2, l2)[0]
            loss.backward(retain_graph=True)
            g1 = x.grad.clone()
            loss.backward()  # second pass over the retained graph: gradients double
            torch.cuda.synchronize()
            res[share] = (g1, x.grad.clone(), a.grad.clone(), b.grad.clone())
    finally:
        F_._DxShare.on = True
        F_.set_precision("fp32")
    for got, ref in zip(res[True], res[False]):
        assert rel_err(got, ref) < 1e-6
    assert rel_err(res[True][1], 2 * res[True][0]) < 1e-6


@pytest.mark.gpu
def test_shared_input_gradient_with_another_consumer_in_between():
    """The shared dX must survive a third consumer of the same activation whose gradient reaches autograd's buffer
    BETWEEN the two heads' backward nodes (node order = reverse creation order: head 2's cost, the extra term, head 1's
    cost): the heads consume a private alias of the input (functional._ShareIn), so the in-place accumulation never
    races autograd's own out-of-place add.  Also: a head left out of the loss simply does not contribute."""
    g = torch.Generator().manual_seed(6)
    rows, K = 1300, 96
    x0 = torch.randn(rows, K, generator=g)
    w1, w2 = torch.randn(21, K, generator=g) / 10, torch.randn(9, K, generator=g) / 10
    l1, l2 = torch.randint(0, 21, (rows,), generator=g).cuda(), torch.randint(0, 9, (rows,), generator=g).cuda()
    F_.set_precision("bf16")
    res = {}
    try:
        for share in (True, False):
            F_._DxShare.on = share
            for both in (True, False):
                x = x0.clone().cuda().requires_grad_(True)
                h = x * 1.0
                a, b = w1.clone().cuda().requires_grad_(True), w2.clone().cuda().requires_grad_(True)
                y1, y2 = F_.linear_log_softmax(h, a, None), F_.linear_log_softmax(h, b, None)
                c1 = F_.head_nll(y1, l1)[0]
                extra = (h * h).mean()
                c2 = F_.head_nll(y2, l2)[0]
                loss = c1 + 3.0 * extra + (0.5 * c2 if both else 0.0)
                loss.backward()
                torch.cuda.synchronize()
                res[share, both] = (x.grad.clone(), a.grad.clone())
                assert (b.grad is not None) == both
                assert F_._DxShare.alias is None and not F_._DxShare.table  # (the shared dX is let go once x has its gradient)
    finally:
        F_._DxShare.on = True
        F_.set_precision("fp32")
    for both in (True, False):
        for got, ref in zip(res[True, both], res[False, both]):
            assert rel_err(got, ref) < 1e-6, both


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,K,act,masked", [(128, 1024, 1024, "relu", True), (128, 1024, 440, "relu", True), (37, 200, 520, "tanh", False),
                                             (2, 8, 128, "relu", False), (128, 1024, 3300, "relu", True)])
def test_small_batch_layer_two_launch_form_matches_the_one_launch_form(M, N, K, act, masked):
    """pk_linear_bn_act_bf16_sk (product split along K over the whole chip + layer epilogue from the slabs) against
    pk_linear_bn_act_bf16 (one launch): same outputs up to the fp32 summation order of the split reduction."""
    import importlib

    _lib = importlib.import_module("pytorch-kaldi_amd._lib")
    lib = _lib.load()
    sk = int(lib.pk_gemm_bf16_small_splitk(M, N, K))
    assert sk >= 2
    g = torch.Generator().manual_seed(M + N + K)
    x = torch.randn(M, K, generator=g).cuda()
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda()
    bias, gamma, beta = torch.randn(N, generator=g).cuda(), (1 + 0.1 * torch.randn(N, generator=g)).cuda(), torch.randn(N, generator=g).cuda()
    mask = ((torch.rand(M, N, generator=g) > 0.15).float() / 0.85).cuda() if masked else None
    xb, wb = F_.cvt_bf16(x), F_.cvt_bf16(w)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    outs = []
    for form in (0, 1):
        z, a, y = (torch.empty(M, N).cuda() for _ in range(3))
        yb = torch.empty(M, N, dtype=torch.bfloat16).cuda()
        mean, var = torch.empty(N).cuda(), torch.empty(N).cuda()
        rm, rv = torch.zeros(N).cuda(), torch.ones(N).cuda()
        args = [st, M, N, K, p(xb), xb.shape[1], p(wb), wb.shape[1], p(bias), p(gamma), p(beta), 1e-5, 0.05, p(rm), p(rv), F_.ACT[act],
                p(mask), p(z), p(a), p(y) if masked else None, p(yb), N, p(mean), p(var)]
        if form == 0:
            _lib.check(lib.pk_linear_bn_act_bf16(*args), "one launch")
        else:
            ws = torch.empty(sk * M * N).cuda()
            _lib.check(lib.pk_linear_bn_act_bf16_sk(*args, sk, p(ws)), "two launches")
        torch.cuda.synchronize()
        outs.append((z, a, y if masked else a, yb.float(), mean, var, rm, rv))
    for i, (u, v) in enumerate(zip(*outs)):
        assert rel_err(v, u.double()) < (8e-3 if i == 3 else 2e-5), (i, rel_err(v, u.double()))
    # and against the arithmetic itself (fp64, bf16-rounded operands)
    zr = _bf_round(x.cpu()) @ _bf_round(w.cpu()).t() + bias.cpu().double()
    mu, vr = zr.mean(0), zr.var(0, unbiased=False)
    ar = (zr - mu) / (vr + 1e-5).sqrt() * gamma.cpu().double() + beta.cpu().double()
    ar = ar.clamp_min(0) if act == "relu" else ar.tanh()
    assert rel_err(outs[1][0], zr) < 2e-5 and rel_err(outs[1][1], ar) < 1e-4 and rel_err(outs[1][4], mu) < 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("M,N,act,masked", [(128, 1024, "relu", True), (37, 200, "tanh", False), (2, 8, "relu", False), (100, 50, "relu", True),
                                           (128, 3300, "relu", True)])
def test_bn_act_bwd_small_against_the_arithmetic(M, N, act, masked):
    """pk_bn_act_bwd_small (activation / mask backward, both BatchNorm reductions, BatchNorm backward, bf16 operand, bias
    gradient) against an fp64 evaluation of neural_networks.py:139-148 backwards; N = 50 takes the element-by-element path."""
    import importlib

    _lib = importlib.import_module("pytorch-kaldi_amd._lib")
    lib = _lib.load()
    g = torch.Generator().manual_seed(3 * M + N)
    z = torch.randn(M, N, generator=g)
    gamma = 1 + 0.1 * torch.randn(N, generator=g)
    mean, var = z.double().mean(0), z.double().var(0, unbiased=False)
    xh = (z.double() - mean) / (var + 1e-5).sqrt()
    pre = xh * gamma.double() + 0.3
    a = pre.clamp_min(0) if act == "relu" else pre.tanh()
    mask = ((torch.rand(M, N, generator=g) > 0.15).float() / 0.85) if masked else None
    dy = torch.randn(M, N, generator=g)
    gg = dy.double() * (mask.double() if masked else 1.0) * ((a > 0).double() if act == "relu" else 1 - a * a)
    sg, sgx = gg.sum(0), (gg * xh).sum(0)
    dz = gamma.double() / (var + 1e-5).sqrt() * (gg - sg / M - xh * sgx / M)
    ldb = (N + 63) // 64 * 64
    dev = lambda t: None if t is None else t.float().cuda()
    zc, ac, mc, dyc, gc, mnc, vrc = dev(z), dev(a), dev(mask), dev(dy), dev(gamma), dev(mean), dev(var)
    dzb = torch.full((M, ldb), 7.0, dtype=torch.bfloat16).cuda()
    dzf, s_g, s_gx, db = torch.empty(M, N).cuda(), torch.empty(N).cuda(), torch.empty(N).cuda(), torch.empty(N).cuda()
    accb, accg, accbias = torch.ones(N).cuda(), torch.ones(N).cuda(), torch.ones(N).cuda()
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.pk_bn_act_bwd_small(st, p(dyc), p(ac), p(mc), F_.ACT[act], p(zc), p(mnc), p(vrc), 1e-5, p(gc), M, N, p(dzb), ldb,
                                       p(dzf), p(s_g), p(s_gx), p(accb), p(accg), p(db), p(accbias)), "pk_bn_act_bwd_small")
    torch.cuda.synchronize()
    tol = 2e-5 if M > 2 else 1e-4  # (two rows: xhat = +-1 up to eps, the backward cancels to rounding level)
    assert rel_err(dzf, dz) < tol and rel_err(dzb[:, :N].float(), dz) < 6e-3
    assert float(dzb[:, N:].float().abs().max()) == 0.0 if ldb > N else True
    assert rel_err(s_g, sg) < 2e-5 and rel_err(s_gx, sgx) < 2e-5
    assert rel_err(accb - 1, sg) < 2e-5 and rel_err(accg - 1, sgx) < 2e-5
    assert float((db - dzf.sum(0)).abs().max()) < 1e-4 * float(dzf.abs().max()) * M and float((accbias - 1 - db).abs().max()) < 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("T,B,G,H,bidir,use_bn", [(7, 6, 2, 550, True, True), (5, 4, 2, 24, False, True), (4, 8, 4, 36, True, True),
                                                 (6, 4, 1, 50, True, False)])
def test_bn_bwd_bf16_sums_into_the_flat_gradient_and_zeroes_the_pad_itself(T, B, G, H, bidir, use_bn):
    """pk_bn_bwd_bf16 (BatchNorm backward of a recurrent layer's projections from the bf16 gate gradients; round 4: the sums
    are also ADDED to acc_beta / acc_gamma - the flat .grad of the gates' BatchNorm shifts / scales - and the pad columns of
    the bf16 output are zeroed by the launch that finishes the sums) against an fp64 evaluation on the same bf16 inputs.
    The output buffer starts out as NaN patterns: whatever the launch leaves beyond G*H columns must be zero."""
    lib = _lib.load()
    TB, GH = T * B, G * H
    Hp = (H + 7) // 8 * 8
    Gp = (G * Hp + 63) // 64 * 64
    ndir = 2 if bidir else 1
    g = torch.Generator().manual_seed(T * 100 + H)
    dG = torch.randn(ndir * TB, Gp, generator=g).to(torch.bfloat16)
    P = torch.randn(TB, GH, generator=g) * 1.5 + 0.2
    gamma = 1 + 0.1 * torch.randn(GH, generator=g)
    mean, var = P.double().mean(0), P.double().var(0, unbiased=False)
    # plain-layout fp64 view of the gate gradients (gate g at column g*Hp; both directions add up)
    gsum = torch.zeros(TB, GH, dtype=torch.float64)
    for d in range(ndir):
        for k in range(G):
            gsum[:, k * H:(k + 1) * H] += dG[d * TB:(d + 1) * TB, k * Hp:k * Hp + H].double()
    if use_bn:
        xh = (P.double() - mean) / (var + 1e-5).sqrt()
        sg, sgx = gsum.sum(0), (gsum * xh).sum(0)
        dP = gamma.double() / (var + 1e-5).sqrt() * (gsum - sg / TB - xh * sgx / TB)
    else:
        sg, sgx, dP = gsum.sum(0), None, gsum
    opitch = (GH + 63) // 64 * 64
    dGc, Pc = dG.cuda(), P.cuda()
    out = torch.full((TB, opitch), float("nan"), dtype=torch.bfloat16).cuda()
    part = torch.empty(int(lib.pk_bn_partial_floats(TB, GH))).cuda()
    s_g, s_gx = torch.empty(GH).cuda(), torch.empty(GH).cuda()
    accb, accg = torch.full((GH,), 2.0).cuda(), torch.full((GH,), -1.0).cuda()
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g1 = ctypes.c_void_p(dGc.data_ptr() + 2 * TB * Gp) if bidir else None
    dev = lambda t: t.float().cuda()
    mc, vc, gc = dev(mean), dev(var), dev(gamma)
    _lib.check(lib.pk_bn_bwd_bf16(st, p(dGc), g1, Gp, G, H, p(Pc), GH, TB, p(mc) if use_bn else None, p(vc) if use_bn else None,
                                  1e-5, p(gc) if use_bn else None, float(TB), p(part), p(s_g), p(s_gx) if use_bn else None, p(out),
                                  opitch, p(accb), p(accg) if use_bn else None), "pk_bn_bwd_bf16")
    torch.cuda.synchronize()
    assert rel_err(out[:, :GH].float(), dP) < 8e-3
    if opitch > GH:
        assert float(out[:, GH:].float().abs().max()) == 0.0  # (NaN would fail the comparison too)
    assert rel_err(s_g, sg) < 2e-5 and rel_err(accb - 2.0, sg) < 2e-5
    if use_bn:
        assert rel_err(s_gx, sgx) < 2e-5 and rel_err(accg + 1.0, sgx) < 2e-5
    else:
        assert float((accg + 1.0).abs().max()) == 0.0  # untouched


@pytest.mark.gpu
@pytest.mark.parametrize("M,G,H", [(64000, 2, 550), (1500, 3, 40), (512, 4, 36)])
def test_statistics_merge_and_finalize_in_one_launch(M, G, H):
    """pk_bn_stats_merge_finalize_gates against pk_bn_stats_merge followed by pk_bn_finalize_gates on the partials of one
    pk_gemm_bf16_stats call: same mean / var / scale / shift and the same running statistics and batch counters, bit for bit
    (the merge order is the same; only the launch boundary is gone)."""
    lib = _lib.load()
    N, K = G * H, 96
    g = torch.Generator().manual_seed(M + H)
    A = torch.randn(M, K, generator=g).to(torch.bfloat16).cuda()
    Bm = (torch.randn(N, K, generator=g) / 8).to(torch.bfloat16).cuda()
    C = torch.empty(M, N).cuda()
    stats = torch.empty(int(lib.pk_gemm_bf16_stats_floats(M, N))).cuda()
    rb = ctypes.c_int(0)
    p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.pk_gemm_bf16_stats(st, M, N, K, 1.0, p(A), K, 1, p(Bm), K, 1, p(C), N, None, p(stats), ctypes.byref(rb)), "gemm")
    if rb.value == 0:
        pytest.skip("this shape does not take the statistics epilogue")
    gamma, beta = (1 + 0.1 * torch.randn(N, generator=g)).cuda(), (0.1 * torch.randn(N, generator=g)).cuda()
    res = []
    for merged in (False, True):
        mean, var, scale, shift = (torch.empty(N).cuda() for _ in range(4))
        rms, rvs = [torch.full((H,), 0.5).cuda() for _ in range(G)], [torch.full((H,), 2.0).cuda() for _ in range(G)]
        nbs = [torch.tensor(3, dtype=torch.int64).cuda() for _ in range(G)]
        arr = lambda ts: (ctypes.c_void_p * G)(*[t.data_ptr() for t in ts])
        if merged:
            _lib.check(lib.pk_bn_stats_merge_finalize_gates(st, p(stats), rb.value, G, H, p(mean), p(var), p(gamma), p(beta), 1e-5,
                                                            p(scale), p(shift), arr(rms), arr(rvs), arr(nbs), 0.05, float(2 * M)), "merged")
        else:
            _lib.check(lib.pk_bn_stats_merge(st, p(stats), rb.value, N, p(mean), p(var)), "merge")
            _lib.check(lib.pk_bn_finalize_gates(st, G, H, p(mean), p(var), p(gamma), p(beta), 1e-5, p(scale), p(shift), arr(rms),
                                                arr(rvs), arr(nbs), 0.05, float(2 * M)), "finalize")
        torch.cuda.synchronize()
        res.append([mean, var, scale, shift] + rms + rvs + nbs)
    for u, v in zip(*res):
        assert torch.equal(u, v)
    assert rel_err(res[1][0], C.double().mean(0)) < 1e-5 and int(res[1][-1]) == 4


@pytest.mark.gpu
def test_fused_step_matches_torch_zeroes_the_gradient_and_refreshes_the_bf16_copies():
    """pk_fused_step (round 4) against torch.optim for the three optimizers, with its two side jobs: the gradient is zero
    afterwards, and the bf16 copies of the 2-D weights named in the segment table equal bf16(new weights) at their pitch
    (pad columns stay zero, everything outside the named weights stays untouched)."""
    lib = _lib.load()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(14)
    n = 64 * 300
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) for _ in range(3)]
    # two "weights" inside the bucket: [40 x 100] at element 128 (pitch 128) and [7 x 64] at element 6400 (pitch 64)
    segs = torch.tensor([[128, 40, 100, 128, 0], [6400, 7, 64, 64, 40 * 128]], dtype=torch.int64).cuda()
    for kind, mk in ((0, lambda q: torch.optim.RMSprop([q], lr=4e-4, alpha=0.95, eps=1e-8, weight_decay=1e-4)),
                     (1, lambda q: torch.optim.SGD([q], lr=0.08, momentum=0.9, weight_decay=1e-4)),
                     (2, lambda q: torch.optim.Adam([q], lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-4, amsgrad=True))):
        pr = p0.clone().requires_grad_(True)
        opt = mk(pr)
        pe = p0.clone().cuda()
        s0, s1, s2 = (torch.zeros(n).cuda() for _ in range(3))
        shadow = torch.zeros(40 * 128 + 7 * 64 + 64, dtype=torch.bfloat16).cuda()
        shadow[-64:] = 3.0
        h = [(0.95, 1e-8, 0.0), (0.9, 0.0, 0.0), (0.9, 0.999, 1e-8)][kind]
        for i, gr in enumerate(grads):
            pr.grad = gr.clone()
            opt.step()
            ge = gr.cuda()
            _lib.check(lib.pk_fused_step(st, kind, pe.data_ptr(), ge.data_ptr(), s0.data_ptr(), s1.data_ptr(), s2.data_ptr(), n,
                                         [4e-4, 0.08, 1e-3][kind], h[0], h[1], h[2], 1e-4, i + 1, int(i != 1), segs.data_ptr(), 2,
                                         shadow.data_ptr()), "fused step")
            torch.cuda.synchronize()
            assert float(ge.abs().max()) == (0.0 if i != 1 else float(gr.abs().max()))
        assert rel_err(pe, pr) < 1e-6
        w1 = pe[128:128 + 4000].view(40, 100)
        c1 = shadow[:40 * 128].view(40, 128)
        assert torch.equal(c1[:, :100], w1.to(torch.bfloat16)) and float(c1[:, 100:].float().abs().max()) == 0.0
        w2 = pe[6400:6400 + 448].view(7, 64)
        assert torch.equal(shadow[40 * 128:40 * 128 + 448].view(7, 64), w2.to(torch.bfloat16))
        assert float((shadow[-64:].float() - 3.0).abs().max()) == 0.0


@pytest.mark.gpu
def test_weight_copies_kept_by_the_optimizer_follow_every_way_a_weight_changes():
    """functional.weight_bf16: after the first optimizer step a flat 2-D weight has a persistent bf16 copy; it equals
    bf16(weight) after fused steps (raw-pointer writes), after a torch in-place change (version counter) and after
    load_state_dict."""
    optim_ = importlib.import_module("pytorch-kaldi_amd.optim")
    lin = torch.nn.Linear(100, 40).cuda()
    flat = optim_.FlatParams(lin)
    opt = optim_.FusedOptimizer(flat, "sgd", 0.1)
    opt.zero_in_step = True
    w = lin.weight
    first = F_.weight_bf16(w)          # no copy yet: a fresh conversion, and the request is noted
    assert getattr(w, "_pk_shadow", None) is None
    for step in range(3):
        opt.zero_grad()
        w.grad.add_(torch.randn_like(w))
        opt.step()
        torch.cuda.synchronize()
        assert float(flat.grad.abs().max()) == 0.0
        view = F_.weight_bf16(w)
        assert view.data_ptr() == w._pk_shadow[0].data_ptr() and view.shape == (40, 128)
        assert torch.equal(view[:, :100], w.detach().to(torch.bfloat16)) and float(view[:, 100:].float().abs().max()) == 0.0
    assert not torch.equal(first[:, :100], view[:, :100])
    with torch.no_grad():
        w.mul_(0.5)
    assert torch.equal(F_.weight_bf16(w)[:, :100], w.detach().to(torch.bfloat16))
    sd = {k: v.clone() + 1.0 for k, v in lin.state_dict().items()}
    lin.load_state_dict(sd)
    assert torch.equal(F_.weight_bf16(w)[:, :100], w.detach().to(torch.bfloat16))
    opt.zero_grad()                       # (the gradient was left clean by the last step: no fill, still zero)
    assert float(flat.grad.abs().max()) == 0.0
