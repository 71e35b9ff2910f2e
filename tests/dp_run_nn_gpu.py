"""Worker of tests/test_gpu_dp_run_nn.py: core.run_nn_dp itself on N ranks of ONE GPU (gloo over device tensors - RCCL
refuses several ranks per device), strong-scaling shapes: a global batch of 128 sequences, every rank keeps 128 / N
columns (N = 8: 16 sequences = 32 rows = two-cluster persistent launches).

--reference: one process that replays the same chunk the N-GPU parity definition's way: every batch, every
shard in turn from the same parameters and running statistics, gradients averaged, rank 0's running statistics kept,
one fused optimizer step.  Both write {arch: state_dict} of the trained networks.

TEST INFRASTRUCTURE (no oracle needed: the engine is compared with itself, shard by shard)."""
import argparse
import configparser
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

T, B, NB, H, LAY = 40, 128, 1, 550, 2


def recipe():
    R = importlib.import_module("pytorch-kaldi_amd.recipes")
    rcp = R.recipe("timit_ligru", n_lay=LAY, H=H)
    a1 = rcp["cfg"]["architecture1"]
    a1["ligru_drop"] = ",".join(["0.0"] * LAY)  # (drop masks come from each process's own device generator)
    # ONE optimizer step of plain SGD: the update is lr x the averaged gradient, element by element - the parameters after
    # the chunk ARE the gradient exchange, linearly.  (RMSprop's first step is +-lr / sqrt(1 - alpha) whatever |g| is, and a
    # second step in bf16 amplifies the summation-order noise of the first through T steps of ReLU recurrence: three
    # RMSprop steps on 8 ranks differ from the replay in a quarter of the elements - measured - without any rank being wrong.)
    for sec in ("architecture1", "architecture2", "architecture3"):
        rcp["cfg"][sec].update({"arch_opt": "sgd", "arch_lr": "0.05", "opt_momentum": "0.0", "opt_weight_decay": "0.0",
                                "opt_dampening": "0.0", "opt_nesterov": "False"})
    return rcp


def chunk(rcp):
    g = torch.Generator().manual_seed(31)
    n_snt = NB * B
    data = torch.randn(n_snt * T, rcp["nfea"] + 2, generator=g)
    data[:, rcp["nfea"]] = torch.randint(0, rcp["n_cd"], (n_snt * T,), generator=g).float()
    data[:, rcp["nfea"] + 1] = torch.randint(0, rcp["n_mono"], (n_snt * T,), generator=g).float()
    end = np.arange(1, n_snt + 1, dtype=np.int64) * T
    return ["utt%05d" % i for i in range(n_snt)], data, end


def write_cfg(rcp, tmp):
    cfg = configparser.ConfigParser()
    for sec in rcp["cfg"].sections():
        cfg[sec] = dict(rcp["cfg"][sec])
    cfg["exp"].update({"seed": "1234", "out_folder": tmp, "save_gpumem": "False", "production": "False",
                       "out_info": os.path.join(tmp, "chunk.info")})
    cfg["model"] = {"model": "\n".join(rcp["model"])}
    cfg["forward"] = {"forward_out": "out_dnn2", "normalize_posteriors": "False", "normalize_with_counts_from": "none",
                      "require_decoding": "False"}
    cfg["batches"] = {"batch_size_train": str(B), "batch_size_valid": str(B)}
    path = os.path.join(tmp, "chunk.cfg")
    with open(path, "w") as f:
        cfg.write(f)
    return path, cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--tmp", required=True)
    ap.add_argument("--prec", default="bf16")
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--reference", action="store_true")
    a = ap.parse_args()
    F_ = importlib.import_module("pytorch-kaldi_amd.functional")
    core = importlib.import_module("pytorch-kaldi_amd.core")
    _lib = importlib.import_module("pytorch-kaldi_amd._lib")
    F_.set_precision(a.prec)
    rcp = recipe()
    names, data, end = chunk(rcp)
    fea_dict = {k: list(v) for k, v in rcp["fea_dict"].items()}
    if not a.reference:
        os.environ["LOCAL_RANK"] = "0"      # every rank on the one GPU of the box
        os.environ["PK_DP_BACKEND"] = "gloo"
        DP = importlib.import_module("pytorch-kaldi_amd.dp")
        rank, world, _ = DP.init_from_env("gloo")
        assert world == a.world
        tmp = os.path.join(a.tmp, "dp")
        if rank == 0:
            os.makedirs(tmp, exist_ok=True)
        torch.distributed.barrier()
        path, _ = write_cfg(rcp, tmp) if rank == 0 else (os.path.join(tmp, "chunk.cfg"), None)
        torch.distributed.barrier()
        core.run_nn_dp(names, data.cuda(), end, fea_dict, rcp["lab_dict"], rcp["arch_dict"], path, False, path,
                       reader=lambda *args: None)
        if rank == 0:
            sds = {}
            for arch, (sec, _, _) in rcp["arch_dict"].items():
                ck = torch.load(os.path.join(tmp, "chunk_%s.pkl" % sec), weights_only=False)
                sds[arch] = {k: v.cpu() for k, v in ck["model_par"].items()}
            info = configparser.ConfigParser()
            info.read(os.path.join(tmp, "chunk.info"))
            torch.save({"sd": sds, "loss": float(info["results"]["loss"])}, a.out)
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
        return
    # ---- the shard-by-shard replay in one process
    U = importlib.import_module("pytorch-kaldi_amd.utils")
    OPT = importlib.import_module("pytorch-kaldi_amd.optim")
    tmp = os.path.join(a.tmp, "ref")
    os.makedirs(tmp, exist_ok=True)
    _, cfg = write_cfg(rcp, tmp)
    torch.manual_seed(1234)
    iod = fea_dict
    model = cfg["model"]["model"].split("\n")
    nns, costs = U.model_init(iod, model, cfg, rcp["arch_dict"], True, False, "train")
    opts = OPT.fused_optimizer_init(nns, cfg, rcp["arch_dict"])
    init = {k: {n: v.detach().cpu().clone() for n, v in m.state_dict().items()} for k, m in nns.items()}
    dev = data.cuda()  # (zero_in_step stays off here: several backward passes per optimizer step)
    world, local = a.world, B // a.world
    loss_sum = 0.0
    for i in range(NB):
        full = dev[i * B * T:(i + 1) * B * T].view(B, T, -1).transpose(0, 1).contiguous()  # (T, B, .): equal-length sentences, no padding
        acc = {k: torch.zeros_like(o.flat.grad) for k, o in opts.items()}
        sd0 = {k: {n: b.clone() for n, b in m.named_buffers()} for k, m in nns.items()}
        keep = None
        for r in range(world):
            for k, m in nns.items():
                for n, b in m.named_buffers():
                    b.copy_(sd0[k][n])
            inp = full[:, r * local:(r + 1) * local].contiguous()
            with F_.accumulating_backward():
                outs = U.forward_model(fea_dict, rcp["lab_dict"], rcp["arch_dict"], model, nns, costs, inp, iod, T, local, "train", [])
                for o in opts.values():
                    o.zero_grad()
                outs["loss_final"].backward()
            F_.join_side()
            for k, o in opts.items():
                acc[k] += o.flat.grad / world
            loss_sum += float(outs["loss_final"]) / world
            if r == 0:
                keep = {k: {n: b.clone() for n, b in m.named_buffers()} for k, m in nns.items()}
        for k, m in nns.items():
            for n, b in m.named_buffers():
                b.copy_(keep[k][n])
        for k, o in opts.items():
            o.flat.grad.copy_(acc[k])
            o.step()
    torch.cuda.synchronize()
    _lib.raise_if_persist_failed()
    torch.save({"sd": {k: {n: v.detach().cpu() for n, v in m.state_dict().items()} for k, m in nns.items()}, "loss": loss_sum / NB, "init": init}, a.out)


if __name__ == "__main__":
    main()
