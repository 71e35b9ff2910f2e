"""Worker of tests/test_gpu_dp_two_ranks.py: one rank of a 2-process data-parallel run on ONE GPU
(python -m torch.distributed.run --nproc-per-node 2 ... both ranks use cuda:0; backend from PK_DP_BACKEND).

Also usable stand-alone as the single-process reference: --reference runs every shard in turn on the same parameters
and averages the gradients (the N-GPU parity definition of SURVEY.md 8e: per-replica BatchNorm statistics, gradients
averaged over shards).
"""
import argparse
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def build(prec):
    F_ = importlib.import_module("pytorch-kaldi_amd.functional")
    nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
    OPT = importlib.import_module("pytorch-kaldi_amd.optim")
    F_.set_precision(prec)
    if os.environ.get("PK_PERSIST2_SAFE"):  # diagnostics: the placement-independent exchange for every cluster
        importlib.import_module("pytorch-kaldi_amd._lib").load().pk_persist2_set_mode(1)
    rec = {"ligru_lay": "72,72", "ligru_drop": "0.0,0.0", "ligru_use_laynorm_inp": "False", "ligru_use_batchnorm_inp": "False",
           "ligru_use_laynorm": "False,False", "ligru_use_batchnorm": "True,True", "ligru_bidir": "True",
           "ligru_act": "relu,relu", "ligru_orthinit": "True", "use_cuda": "True", "to_do": "train"}
    head = {"dnn_lay": "37", "dnn_drop": "0.0", "dnn_use_laynorm_inp": "False", "dnn_use_batchnorm_inp": "False",
            "dnn_use_batchnorm": "False", "dnn_use_laynorm": "False", "dnn_act": "softmax", "use_cuda": "True",
            "to_do": "train"}
    torch.manual_seed(5)
    nns = {"rec": nn_amd.liGRU(rec, 24).cuda().train(), "head": nn_amd.MLP(head, 144).cuda().train()}
    opts = {k: OPT.FusedOptimizer(OPT.FlatParams(m), "rmsprop", 2e-3, alpha=0.95, eps=1e-8) for k, m in nns.items()}
    return F_, nns, opts


def batch(step):
    g = torch.Generator().manual_seed(100 + step)
    x = torch.randn(300, 16, 24, generator=g)  # 4800 rows per 16 sequences: the Linear weight gradients take the side stream too
    lab = torch.randint(0, 37, (300, 16), generator=g)
    return x.cuda(), lab.cuda()


def loss_of(nns, x, lab):
    T, B, _ = x.shape
    return torch.nn.functional.nll_loss(nns["head"](nns["rec"](x).reshape(T * B, -1)), lab.reshape(-1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--prec", default="fp32")
    ap.add_argument("--overlap", type=int, default=0)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--reference", action="store_true")
    ap.add_argument("--solo", default="", help="one rank: '' | 'plain' (no reducer) | 'nccl' (one-rank RCCL communicator, forced buckets)")
    a = ap.parse_args()
    DP = importlib.import_module("pytorch-kaldi_amd.dp")
    _lib = importlib.import_module("pytorch-kaldi_amd._lib")
    world = 2
    if a.solo:
        # the whole data-parallel path - buckets over the flat gradient buffer, hooks + side-stream notifications, async
        # all_reduce on RCCL's stream, finish() - on ONE rank, where the result must equal the plain loop bit for bit
        if a.solo == "nccl":
            os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29577")
            torch.cuda.set_device(0)
            torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        F_, nns, opts = build(a.prec)
        red = None
        if a.solo == "nccl":
            red = DP.GradReducer(nns, flats={k: o.flat for k, o in opts.items()}, bucket_bytes=64 << 10,
                                 overlap=bool(a.overlap), force=True)
            assert red.active and len(red.buckets) > 1
        for step in range(a.steps):
            x, lab = batch(step)
            for o in opts.values():
                o.zero_grad()
            loss_of(nns, x, lab).backward()
            if red is not None:
                red.finish()
            for o in opts.values():
                o.step()
        torch.cuda.synchronize()
        _lib.raise_if_persist_failed()
        torch.save({"params": {k: o.flat.flat.cpu() for k, o in opts.items()}}, a.out)
        if a.solo == "nccl":
            torch.distributed.destroy_process_group()
        return
    if a.reference:
        F_, nns, opts = build(a.prec)
        g0 = None
        raw = {}
        for step in range(a.steps):
            x, lab = batch(step)
            acc = {k: torch.zeros_like(o.flat.grad) for k, o in opts.items()}
            sd = {k: {n: b.clone() for n, b in m.named_buffers()} for k, m in nns.items()}
            for r in range(world):
                for k, m in nns.items():  # every replica starts the step from the same running statistics
                    for n, b in m.named_buffers():
                        b.copy_(sd[k][n])
                for o in opts.values():
                    o.zero_grad()
                loss_of(nns, DP.shard_batch(x, r, world), DP.shard_batch(lab, r, world)).backward()
                F_.join_side()
                for k, o in opts.items():
                    acc[k] += o.flat.grad / world
                if step == 0:
                    raw[r] = {k: o.flat.grad.cpu().clone() for k, o in opts.items()}
                if r == 0:  # DataParallel keeps replica 0's running statistics
                    keep = {k: {n: b.clone() for n, b in m.named_buffers()} for k, m in nns.items()}
            for k, m in nns.items():
                for n, b in m.named_buffers():
                    b.copy_(keep[k][n])
            for k, o in opts.items():
                o.flat.grad.copy_(acc[k])
                o.step()
            if g0 is None:
                g0 = {k: v.cpu() for k, v in acc.items()}
        torch.cuda.synchronize()
        _lib.raise_if_persist_failed()
        torch.save({"params": {k: o.flat.flat.cpu() for k, o in opts.items()}, "grad0": g0, "raw": raw}, a.out)
        return
    os.environ["LOCAL_RANK"] = "0"  # both ranks on the one GPU of the box
    rank, w, _ = DP.init_from_env(os.environ.get("PK_DP_BACKEND", "nccl"))
    assert w == world
    F_, nns, opts = build(a.prec)
    red = DP.GradReducer(nns, flats={k: o.flat for k, o in opts.items()}, bucket_bytes=64 << 10, overlap=bool(a.overlap))
    g0 = None
    for step in range(a.steps):
        x, lab = batch(step)
        for o in opts.values():
            o.zero_grad()
        loss_of(nns, DP.shard_batch(x, rank, world), DP.shard_batch(lab, rank, world)).backward()
        if step == 0:
            F_.join_side()
            torch.save({k: o.flat.grad.cpu().clone() for k, o in opts.items()}, a.out + ".raw%d" % rank)
        red.finish()
        if g0 is None:
            g0 = {k: o.flat.grad.cpu() for k, o in opts.items()}
        for o in opts.values():
            o.step()
    torch.cuda.synchronize()
    _lib.raise_if_persist_failed()
    if rank == 0:
        torch.save({"params": {k: o.flat.flat.cpu() for k, o in opts.items()}, "grad0": g0}, a.out)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
