"""The data-parallel chunk function (pytorch-kaldi_amd/core.py::run_nn_dp) against the reference's own chunk loop.

tests/golden/chunk_ligru_run_nn.npz was produced by oracle/make_golden.py::chunk_case running the reference's
core.run_nn (core.py:439-753) on an in-memory synthetic chunk: train from scratch -> ck0, continue from ck0 -> ck1 +
loss/err, validate with ck1, forward with ck1 (Kaldi ark with log-prior normalisation).  The CPU tests check the
host pieces (batch assembly, ark writer, torch-format optimizer state); the GPU test replays chunk 2, the
validation and the forward pass through the engine from the reference-written ck0 / ck1 checkpoints.
"""
import configparser
import importlib
import io
import os
import random
import struct

import numpy as np
import pytest
import torch

from golden_util import Golden, rel_err

core = importlib.import_module("pytorch-kaldi_amd.core")
optim_ = importlib.import_module("pytorch-kaldi_amd.optim")

CASE = "chunk_ligru_run_nn"


def reference_batch(data_set, data_end_index, snt_index, batch_size):
    """Checker: the padding loop exactly as core.py:581-598 walks it (one random.randint per sentence)."""
    end = np.asarray(data_end_index)
    lens = np.diff(end, prepend=0)
    max_len = int(max(lens[snt_index:snt_index + batch_size]))
    inp = torch.zeros(max_len, batch_size, data_set.shape[1])
    beg = 0 if snt_index == 0 else int(end[snt_index - 1])
    for k in range(batch_size):
        n = int(end[snt_index]) - beg
        left = random.randint(0, max_len - n)
        inp[left:left + n, k, :] = data_set[beg:beg + n, :]
        beg = int(end[snt_index])
        snt_index += 1
    return max_len, inp


def test_batch_assembler_matches_reference_padding_loop():
    g = Golden(CASE)
    data, end = g.t("data_set"), g.arrays["data_end_index"]
    asm = core.BatchAssembler(data, end, torch.device("cpu"))
    for cols in (None, [2, 3], [0]):
        random.seed(77)
        want = [reference_batch(data, end, s, 4) for s in (0, 4)]
        state_after = random.getstate()
        random.seed(77)
        for (max_len, inp), s in zip(want, (0, 4)):
            got_len, got = asm.batch(s, 4, cols)
            assert got_len == max_len
            ref = inp if cols is None else inp[:, cols, :]
            assert torch.equal(got, ref)
        assert random.getstate() == state_after  # same RNG consumption whatever columns a rank keeps


@pytest.mark.gpu
def test_batch_assembler_device_path_with_the_host_running_ahead():
    """The device path uploads (left, lens, beg) through a ring of pinned slots guarded by events and builds the row
    map on the GPU.  200 batches are enqueued behind a long-running kernel without any host sync - the host is far
    ahead of the DMA engine, the situation in which an unguarded staging buffer hands a batch the padding of a later
    one - and every batch must equal the reference's padding loop bit for bit."""
    rng = np.random.RandomState(3)
    lens = rng.randint(5, 60, size=64)
    end = np.cumsum(lens)
    data = torch.randn(int(end[-1]), 7, generator=torch.Generator().manual_seed(1))
    dev = torch.device("cuda")
    asm = core.BatchAssembler(data.cuda(), end, dev)
    starts = [int(s) for s in rng.randint(0, 64 - 8, size=200)]
    random.seed(5)
    want = [reference_batch(data, end, s, 8) for s in starts]
    random.seed(5)
    busy = torch.randn(4096, 4096, device=dev)
    for _ in range(20):  # keep the GPU queue occupied so that the copies below stay pending while the host loops
        busy = busy @ busy
        busy = busy / busy.norm()
    got = [asm.batch(s, 8) for s in starts]
    torch.cuda.synchronize()
    for (wl, w), (gl, g_) in zip(want, got):
        assert wl == gl and torch.equal(g_.cpu(), w)


def test_sentence_lengths_and_counts(tmp_path):
    assert list(core.sentence_lengths([3, 7, 12])) == [3, 4, 5]
    p = tmp_path / "counts"
    p.write_text("[ 3 1 4 1 5 ]\n")
    c = core.load_counts(str(p))
    assert c.dtype == np.float32 and list(c) == [3, 1, 4, 1, 5]


def parse_ark(buf):
    """(key, matrix) records of a binary Kaldi float-matrix ark."""
    out, i = [], 0
    while i < len(buf):
        j = buf.index(b" ", i)
        key = buf[i:j].decode("latin1")
        assert buf[j + 1:j + 3] == b"\0B" and buf[j + 3:j + 6] == b"FM "
        assert buf[j + 6] == 4 and buf[j + 11] == 4
        rows, cols = struct.unpack("<I", buf[j + 7:j + 11])[0], struct.unpack("<I", buf[j + 12:j + 16])[0]
        n = rows * cols * 4
        out.append((key, np.frombuffer(buf[j + 16:j + 16 + n], dtype=np.float32).reshape(rows, cols)))
        i = j + 16 + n
    return out


def test_write_mat_reproduces_reference_ark_bytes():
    g = Golden(CASE)
    ark = bytes(g.arrays["ark"])
    recs = parse_ark(ark)
    assert [k for k, _ in recs] == g.meta["data_name"]
    f = io.BytesIO()
    for key, m in recs:
        core.write_mat(f, m, key)
    assert f.getvalue() == ark
    with pytest.raises(TypeError):
        core.write_mat(io.BytesIO(), np.zeros((2, 2), dtype=np.int32), "x")


def checkpoint_from_fixture(g, ck, arch):
    model_par = {k: v for k, v in g.group("%s/%s/model_par/" % (ck, arch)).items()}
    state = {}
    for k, v in g.group("%s/%s/opt/" % (ck, arch)).items():
        idx, name = k.split("/")
        state.setdefault(int(idx), {})[name] = v
    return {"model_par": model_par, "optimizer_par": {"state": state, "param_groups": g.meta["param_groups"][ck + "/" + arch]}}


def test_fused_optimizer_speaks_torch_state_dicts():
    """optimizer_par written by the reference (torch.optim.RMSprop) -> FusedOptimizer -> the same dict again."""
    g = Golden(CASE)
    ck = checkpoint_from_fixture(g, "ck0", "architecture2")
    n_params = len(ck["optimizer_par"]["param_groups"][0]["params"])
    params = [torch.nn.Parameter(torch.zeros_like(ck["optimizer_par"]["state"][i]["square_avg"])) if i in
              ck["optimizer_par"]["state"] else torch.nn.Parameter(torch.zeros(1)) for i in range(n_params)]
    mod = torch.nn.Module()
    mod.ps = torch.nn.ParameterList(params)
    opt = optim_.FusedOptimizer(optim_.FlatParams(mod), "rmsprop", 1.0, alpha=0.5, eps=1.0)
    opt.load_state_dict(ck["optimizer_par"])
    grp = ck["optimizer_par"]["param_groups"][0]
    assert opt.param_groups[0]["lr"] == grp["lr"] and opt.alpha == grp["alpha"] and opt.eps == grp["eps"]
    sd = opt.state_dict()
    assert sd["param_groups"][0]["params"] == grp["params"]
    for i, ent in ck["optimizer_par"]["state"].items():
        assert torch.equal(sd["state"][i]["square_avg"], ent["square_avg"])
        assert float(sd["state"][i]["step"]) == float(ent["step"])
    # and torch accepts what the fused optimizer emits
    twin = torch.optim.RMSprop(params, lr=1.0)
    twin.load_state_dict(sd)
    assert twin.param_groups[0]["alpha"] == grp["alpha"]


def engine_cfg(g, tag, out):
    return g.meta["cfgs"][tag].replace("{OUT}", str(out)).replace("arch_library = neural_networks",
                                                                   "arch_library = pytorch-kaldi_amd.nn") \
        .replace("use_cuda = False", "use_cuda = True")


@pytest.mark.gpu
def test_run_nn_dp_replays_the_reference_chunk_loop(tmp_path):
    g = Golden(CASE)
    meta = g.meta
    data, end = g.arrays["data_set"], g.arrays["data_end_index"]
    for ck in ("ck0", "ck1"):
        for i in (1, 2, 3):
            torch.save(checkpoint_from_fixture(g, ck, "architecture%d" % i), tmp_path / ("%s_architecture%d.pkl" % (ck, i)))
    with open(tmp_path / "counts", "w") as f:
        f.write("[ " + " ".join(str(int(c)) for c in g.arrays["counts"]) + " ]\n")

    def reader(cfg_file, is_production, shared_list, output_folder):
        shared_list.extend([meta["data_name"], end, {k: list(v) for k, v in meta["fea_dict"].items()},
                            meta["lab_dict"], meta["arch_dict"], data])

    def run(tag, first):
        path = tmp_path / (tag + ".cfg")
        path.write_text(engine_cfg(g, tag, tmp_path))
        args = ([None] * 6) if first else [meta["data_name"], torch.from_numpy(data).cuda(), end,
                                          {k: list(v) for k, v in meta["fea_dict"].items()}, meta["lab_dict"],
                                          meta["arch_dict"]]
        nxt = core.run_nn_dp(*args, str(path), first, str(path), reader=reader)
        assert nxt[0] == meta["data_name"] and nxt[1].is_cuda and nxt[1].shape == data.shape
        info = configparser.ConfigParser()
        info.read(tmp_path / (tag + ".info"))
        return info["results"]

    ref1 = {i: checkpoint_from_fixture(g, "ck1", "architecture%d" % i) for i in (1, 2, 3)}
    ref0 = {i: checkpoint_from_fixture(g, "ck0", "architecture%d" % i) for i in (1, 2, 3)}
    # keep the reference's ck1 aside: the engine overwrites ck1_*.pkl
    for i in (1, 2, 3):
        os.rename(tmp_path / ("ck1_architecture%d.pkl" % i), tmp_path / ("refck1_architecture%d.pkl" % i))
    res = run("ck1", first=True)  # chunk 2 of training, continued from the reference's ck0
    assert abs(float(res["loss"]) - meta["info"]["ck1"]["loss"]) < 1e-4 * abs(meta["info"]["ck1"]["loss"])
    assert abs(float(res["err"]) - meta["info"]["ck1"]["err"]) < 1e-6
    for i in (1, 2, 3):
        got = torch.load(tmp_path / ("ck1_architecture%d.pkl" % i), weights_only=False)
        for k, v in ref1[i]["model_par"].items():
            if not v.is_floating_point():
                assert int(got["model_par"][k]) == int(v), k
                continue
            assert rel_err(got["model_par"][k], v) < 1e-5, k
            upd_ref = v.double() - ref0[i]["model_par"][k].double()
            if float(upd_ref.norm()) > 0:
                upd = got["model_par"][k].cpu().double() - ref0[i]["model_par"][k].double()
                assert rel_err(upd, upd_ref) < 5e-3, (k, rel_err(upd, upd_ref))
        for idx, ent in ref1[i]["optimizer_par"]["state"].items():
            assert rel_err(got["optimizer_par"]["state"][idx]["square_avg"], ent["square_avg"]) < 1e-3, (i, idx)
            assert float(got["optimizer_par"]["state"][idx]["step"]) == float(ent["step"])
    # validation and forward from the reference's own ck1
    for i in (1, 2, 3):
        os.replace(tmp_path / ("refck1_architecture%d.pkl" % i), tmp_path / ("ck1_architecture%d.pkl" % i))
    res = run("valid", first=False)
    assert abs(float(res["loss"]) - meta["info"]["valid"]["loss"]) < 1e-4 * abs(meta["info"]["valid"]["loss"])
    assert abs(float(res["err"]) - meta["info"]["valid"]["err"]) < 1e-6
    run_res = run("forward", first=False)
    assert "loss" not in run_res
    got = parse_ark(open(tmp_path / "forward_out_dnn2_to_decode.ark", "rb").read())
    want = parse_ark(bytes(g.arrays["ark"]))
    assert [k for k, _ in got] == [k for k, _ in want]
    for (_, a), (_, b) in zip(got, want):
        assert a.shape == b.shape
        assert rel_err(torch.from_numpy(a.copy()), torch.from_numpy(b.copy())) < 1e-4


@pytest.mark.gpu
def test_run_nn_dp_hip_graph_replay_equals_eager(tmp_path, monkeypatch):
    """Non-sequence recipe (an MLP trunk in place of the Li-GRU): run_nn_dp replays the training step as a HIP graph after
    three eager batches; the chunk must end with the same parameters, optimizer state and loss as the eager loop."""
    g = Golden(CASE)
    meta = g.meta
    cp = configparser.ConfigParser()
    cp.read_string(engine_cfg(g, "ck0", tmp_path))
    a1 = cp["architecture1"]
    for k in [k for k in a1 if k.startswith("ligru_")]:
        del a1[k]
    a1["arch_class"], a1["arch_seq_model"] = "MLP", "False"
    a1.update({"dnn_lay": "32,32", "dnn_drop": "0.0,0.0", "dnn_use_laynorm_inp": "False", "dnn_use_batchnorm_inp": "False",
               "dnn_use_batchnorm": "True,True", "dnn_use_laynorm": "False,False", "dnn_act": "relu,relu"})
    cp["batches"]["batch_size_train"] = "8"
    arch_dict = {k: list(v) for k, v in meta["arch_dict"].items()}
    arch_dict["liGRU_layers"][2] = False
    data, end = g.arrays["data_set"], g.arrays["data_end_index"]  # 80+ frames: 10 batches of 8

    def reader(cfg_file, is_production, shared_list, output_folder):
        shared_list.extend([meta["data_name"], end, {k: list(v) for k, v in meta["fea_dict"].items()}, meta["lab_dict"],
                            arch_dict, data])

    results = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("PK_HIPGRAPH", mode)
        out = tmp_path / ("g" + mode)
        out.mkdir()
        cp["exp"]["out_folder"] = str(out)
        cp["exp"]["out_info"] = str(out / "ck.info")
        path = out / "ck.cfg"
        with open(path, "w") as f:
            cp.write(f)
        core.run_nn_dp(*([None] * 6), str(path), True, str(path), reader=reader)
        info = configparser.ConfigParser()
        info.read(out / "ck.info")
        cks = [torch.load(out / ("ck_architecture%d.pkl" % i), weights_only=False) for i in (1, 2, 3)]
        results[mode] = (float(info["results"]["loss"]), float(info["results"]["err"]), cks)
    assert data.shape[0] // 8 >= 6
    assert results["0"][0] == results["1"][0] and results["0"][1] == results["1"][1]
    for ce, cg in zip(results["0"][2], results["1"][2]):
        for k, v in ce["model_par"].items():
            assert torch.equal(v.cpu(), cg["model_par"][k].cpu()), k
        for idx, ent in ce["optimizer_par"]["state"].items():
            assert torch.equal(ent["square_avg"].cpu(), cg["optimizer_par"]["state"][idx]["square_avg"].cpu())
            assert float(ent["step"]) == float(cg["optimizer_par"]["state"][idx]["step"])
