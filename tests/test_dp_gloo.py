"""Data-parallel gradient exchange (pytorch-kaldi_amd/dp.py) on CPU: world_size 2, gloo.
N-rank result must equal the per-shard gradients averaged (SURVEY.md 8e parity definition)."""
import importlib
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _model(seed):
    torch.manual_seed(seed)
    return torch.nn.Sequential(torch.nn.Linear(6, 9), torch.nn.Tanh(), torch.nn.Linear(9, 5), torch.nn.Tanh(),
                               torch.nn.Linear(5, 3))


def _batch():
    g = torch.Generator().manual_seed(3)
    return torch.randn(7, 8, 6, generator=g), torch.randn(7, 8, 3, generator=g)


def _grouped(net):
    """A layout statement like the recurrent modules make (pk_flat_groups): layer by layer from the input up, a layer's
    weight and bias back to back - the reducer's buckets then complete in backward order."""
    net.pk_flat_groups = lambda: [[net[0].weight, net[0].bias], [net[2].weight, net[2].bias], [net[4].weight, net[4].bias]]
    return net


def _worker(rank, world, port, use_flat, bucket_bytes, out, overlap=True, wire="fp32"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    DP = importlib.import_module("pytorch-kaldi_amd.dp")
    OPT = importlib.import_module("pytorch-kaldi_amd.optim")
    r, w, _ = DP.init_from_env("gloo")
    assert (r, w) == (rank, world)
    net = _model(0)
    # an unused parameter, like the reference's never-called ln/bn sub-modules (grad stays None / zero)
    net.unused = torch.nn.Parameter(torch.ones(4))
    if use_flat == "grouped":
        _grouped(net)
    flats = {"net": OPT.FlatParams(net)} if use_flat else None
    if use_flat == "grouped":  # weight and bias of a layer are neighbours in the flat buffer, layers in order
        f = flats["net"]
        off = {id(p): o for p, o in zip(f.params, f.offsets)}
        for i in (0, 2, 4):
            assert off[id(net[i].bias)] == off[id(net[i].weight)] + net[i].weight.numel()
        assert off[id(net[0].weight)] < off[id(net[2].weight)] < off[id(net[4].weight)] < off[id(net.unused)] < f.n_active
    red = DP.GradReducer({"net": net}, bucket_bytes=bucket_bytes, flats=flats, overlap=overlap, wire=wire)
    x, y = _batch()
    for step in range(2):  # two steps: buckets must re-arm
        if flats:
            flats["net"].zero_grad()
        else:
            net.zero_grad()
        xs, ys = DP.shard_batch(x, rank, world), DP.shard_batch(y, rank, world)
        ((net(xs) - ys) ** 2).mean().backward()
        red.finish()
    grads = {k: (p.grad.clone() if p.grad is not None else None) for k, p in net.named_parameters()}
    if rank == 0:
        torch.save(grads, out)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("use_flat,bucket_bytes,overlap", [(False, 1 << 20, True), (False, 64, True), (True, 1 << 20, True),
                                                           (True, 128, True), (True, 128, False),
                                                           ("grouped", 128, True), ("grouped", 1 << 20, False)])
def test_two_rank_allreduce_equals_shard_average(tmp_path, use_flat, bucket_bytes, overlap):
    world = 2
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(world, _free_port(), use_flat, bucket_bytes, out, overlap), nprocs=world, join=True)
    got = torch.load(out)
    x, y = _batch()
    ref = None
    for r in range(world):
        net = _model(0)
        xs, ys = x[:, r * 4:(r + 1) * 4], y[:, r * 4:(r + 1) * 4]
        ((net(xs) - ys) ** 2).mean().backward()
        g = {k: p.grad for k, p in net.named_parameters()}
        ref = g if ref is None else {k: ref[k] + g[k] for k in g}
    for k, v in ref.items():
        assert torch.allclose(got[k], v / world, rtol=1e-6, atol=1e-7), k
    assert got["unused"] is None or float(got["unused"].abs().max()) == 0.0


@pytest.mark.parametrize("use_flat,overlap", [(True, True), (False, False), ("grouped", True)])
def test_bf16_wire_is_the_fp32_exchange_up_to_bf16_rounding(tmp_path, use_flat, overlap):
    """GradReducer(wire="bf16") (PK_DP_WIRE): the buckets travel as bf16 - every rank's share rounded once, the sum formed
    in bf16 - and come back widened into the fp32 gradients.  Against the fp32 wire on the same two shards: within
    2^-7 of each tensor's largest entry (the shares and their sum are rounded to 2^-9 relative each; a share can be larger than the
    sum), and not bit-identical (the switch does something)."""
    world = 2
    outs = {}
    for wire in ("fp32", "bf16"):
        out = str(tmp_path / (wire + ".pt"))
        mp.spawn(_worker, args=(world, _free_port(), use_flat, 128, out, overlap, wire), nprocs=world, join=True)
        outs[wire] = torch.load(out)
    differs = False
    for k, v in outs["fp32"].items():
        if v is None or float(v.abs().max()) == 0.0:
            continue
        d = float((outs["bf16"][k] - v).abs().max())
        assert d <= 2.0 ** -7 * float(v.abs().max()), (k, d)
        differs = differs or d > 0.0
    assert differs


def test_shard_batch_shapes():
    DP = importlib.import_module("pytorch-kaldi_amd.dp")
    x = torch.arange(2 * 8 * 3).reshape(2, 8, 3)
    assert torch.equal(DP.shard_batch(x, 1, 4), x[:, 2:4])
    y = torch.arange(8 * 3).reshape(8, 3)
    assert torch.equal(DP.shard_batch(y, 3, 4), y[6:8])
    with pytest.raises(ValueError):
        DP.shard_batch(x, 0, 3)


def _core_worker(rank, world, port, out):
    """The data-parallel plumbing of core.run_nn_dp on CPU modules: column assignment, reducer over the fused
    optimizers' flat buckets, loss / error averaging."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    DP = importlib.import_module("pytorch-kaldi_amd.dp")
    OPT = importlib.import_module("pytorch-kaldi_amd.optim")
    core = importlib.import_module("pytorch-kaldi_amd.core")
    DP.init_from_env("gloo")
    nns = {"trunk": _model(0), "head": torch.nn.Linear(3, 2)}
    torch.manual_seed(1)
    nns["head"].reset_parameters()
    optimizers = {k: OPT.FusedOptimizer(OPT.FlatParams(m), "sgd", 0.1) for k, m in nns.items()}
    reducer = core.make_reducer(nns, optimizers, world)
    x, y = _batch()
    local, cols = core.rank_columns(x.shape[1], rank, world)
    assert local == 4 and list(cols) == list(range(rank * 4, rank * 4 + 4))
    for o in optimizers.values():
        o.zero_grad()
    loss = ((nns["head"](nns["trunk"](x[:, cols])) - y[:, cols, :2]) ** 2).mean()
    loss.backward()
    reducer.finish()
    l, e = core.mean_over_ranks(loss.detach(), torch.tensor(float(rank)), world)
    if rank == 0:
        torch.save({"grads": {k: o.flat.grad.clone() for k, o in optimizers.items()}, "loss": l, "err": e}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_run_nn_dp_plumbing_two_ranks(tmp_path):
    out = str(tmp_path / "core.pt")
    mp.spawn(_core_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    OPT = importlib.import_module("pytorch-kaldi_amd.optim")
    nns = {"trunk": _model(0), "head": torch.nn.Linear(3, 2)}
    torch.manual_seed(1)
    nns["head"].reset_parameters()
    flats = {k: OPT.FlatParams(m) for k, m in nns.items()}
    x, y = _batch()
    ref = {k: torch.zeros_like(f.grad) for k, f in flats.items()}
    losses = []
    for r in range(2):
        for f in flats.values():
            f.zero_grad()
        cols = list(range(r * 4, r * 4 + 4))
        loss = ((nns["head"](nns["trunk"](x[:, cols])) - y[:, cols, :2]) ** 2).mean()
        loss.backward()
        losses.append(float(loss))
        for k, f in flats.items():
            ref[k] += f.grad / 2
    for k in ref:
        assert torch.allclose(got["grads"][k], ref[k], atol=1e-6), k
    assert abs(float(got["loss"]) - sum(losses) / 2) < 1e-6 and abs(float(got["err"]) - 0.5) < 1e-6


def test_bucket_waits_for_every_contribution_of_a_parameter_used_twice():
    """A module applied twice in one [model] (shared architecture) has its weight-gradient GEMM side-launched twice
    per step: the bucket must not leave for the all-reduce after the first one (dp.GradReducer counts, in step 1, how
    many signals each parameter gives, and only reduces in finish() during that step)."""
    DP = importlib.import_module("pytorch-kaldi_amd.dp")
    net = torch.nn.Linear(3, 2)
    red = DP.GradReducer({"net": net}, bucket_bytes=1 << 20, overlap=False, force=True)
    assert len(red.buckets) == 1
    b = red.buckets[0]
    launched = []

    def fake_launch(bk):
        bk["fired"] = True
        launched.append(dict(bk["got"]))

    red._launch = fake_launch
    w, bias = net.weight, net.bias
    # step 1: the weight signals twice, the bias once - nothing leaves before finish()
    for p in (w, bias, w):
        red._ready(b, p)
    assert launched == []
    red.finish()
    assert len(launched) == 1 and b["expect"] == {id(w): 2, id(bias): 1}
    # step 2: after (weight, bias) the bucket is NOT complete; the second weight signal completes it
    red._ready(b, w)
    red._ready(b, bias)
    assert len(launched) == 1
    red._ready(b, w)
    assert len(launched) == 2
    # a contribution that arrives after the bucket left is an error, not a silent race
    with pytest.raises(RuntimeError):
        red._ready(b, w)
    red.handles = []
    red.finish()
    # step 3: fewer signals than learnt -> the bucket is reduced by finish()
    red._ready(b, w)
    assert len(launched) == 2
    red.finish()
    assert len(launched) == 3
