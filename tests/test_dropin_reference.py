"""The drop-in, driven by the REFERENCE's own caller (build container only: /root/reference does not travel).

A PyTorch-Kaldi user switches engines by editing one cfg line per architecture (arch_library).  Everything that then
touches the classes is the reference's code, not this package's mirror of it: utils.model_init instantiates
``arch_class(options, inp_dim)`` through importlib (utils.py:2047-2057), utils.optimizer_init builds torch optimizers
over ``.parameters()`` (utils.py:2106-2164), core.run_nn loads the previous chunk's ``.pkl`` into both
(core.py:523-535).  These tests run exactly those functions, imported from /root/reference unmodified, on the shipped
cfg files with ``arch_library = pytorch-kaldi_amd.nn``.
"""
import configparser
import importlib
import os
import sys

import pytest
import torch

from golden_util import Golden

REF = os.environ.get("PK_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(REF, "utils.py")),
                                reason="needs the reference checkout (build container only)")


@pytest.fixture(scope="module")
def ref_utils():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    saved = sys.modules.pop("utils", None), sys.modules.pop("data_io", None)
    try:
        yield importlib.import_module("utils")
    finally:
        sys.path.remove(REF)
        for name, mod in zip(("utils", "data_io"), saved):
            sys.modules.pop(name, None)
            if mod is not None:
                sys.modules[name] = mod


def _cfg(path, library, n_out):
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REF, path))
    cfg["exp"]["to_do"] = "train"
    cfg["exp"]["use_cuda"] = "False"  # construction only: the engine's forward needs the GPU
    for sec in cfg.sections():
        if sec.startswith("architecture"):
            cfg[sec]["arch_library"] = library
            for k, v in cfg[sec].items():
                if v in n_out:
                    cfg[sec][k] = str(n_out[v])
    return cfg


RECIPES = [
    ("cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg", 40),
    ("cfg/TIMIT_baselines/TIMIT_LSTM_fmllr.cfg", 40),
    ("cfg/TIMIT_baselines/TIMIT_MLP_fmllr.cfg", 440),
    ("cfg/TIMIT_baselines/TIMIT_SincNet_raw.cfg", 3200),
    ("cfg/Librispeech_baselines/libri_GRU_fmllr.cfg", 40),
]


def _build(ref_utils, path, library, feat_dim):
    n_out = {"N_out_lab_cd": 1938, "N_out_lab_mono": 48}
    cfg = _cfg(path, library, n_out)
    model = cfg["model"]["model"].split("\n")
    arch_dict = {}
    for sec in cfg.sections():
        if sec.startswith("architecture"):
            arch_dict[cfg[sec]["arch_name"]] = [sec, cfg[sec]["arch_name"], bool(ref_utils.strtobool(cfg[sec]["arch_seq_model"]))]
    fea = [ln for ln in model if "compute" in ln][0].split(",")[1].strip(" )")
    inp_out_dict = {fea: [0, feat_dim, feat_dim]}
    nns, costs = ref_utils.model_init(inp_out_dict, model, cfg, arch_dict, False, False, "train")
    opts = ref_utils.optimizer_init(nns, cfg, arch_dict)
    return cfg, nns, costs, opts, inp_out_dict


@pytest.mark.parametrize("path,feat_dim", RECIPES)
def test_reference_model_init_builds_the_engine_classes(ref_utils, path, feat_dim):
    """utils.model_init + optimizer_init of the reference on the SHIPPED cfg with arch_library switched: same
    architectures, out_dims, parameter names / shapes / order, optimizer types and hyper-parameters as with the
    reference's own neural_networks - and, with the same seed, the same initial weights."""
    nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
    torch.manual_seed(1234)
    _, nns_e, costs_e, opts_e, iod_e = _build(ref_utils, path, "pytorch-kaldi_amd.nn", feat_dim)
    torch.manual_seed(1234)
    _, nns_r, costs_r, opts_r, iod_r = _build(ref_utils, path, "neural_networks", feat_dim)
    assert list(nns_e) == list(nns_r) and list(costs_e) == list(costs_r)
    assert iod_e == iod_r  # every out_dim the interpreter derived
    for name in nns_r:
        e, r = nns_e[name], nns_r[name]
        assert type(e).__module__ == nn_amd.__name__ and type(e).__name__ == type(r).__name__
        assert e.out_dim == r.out_dim and e.training == r.training
        pe, pr = list(e.named_parameters()), list(r.named_parameters())
        assert [k for k, _ in pe] == [k for k, _ in pr]  # the order indexes torch optimizer state
        sde, sdr = e.state_dict(), r.state_dict()
        assert list(sde) == list(sdr)
        for k in sdr:
            assert sde[k].shape == sdr[k].shape and sde[k].dtype == sdr[k].dtype, (name, k)
            assert torch.equal(sde[k], sdr[k]), (name, k)  # same seed -> same initialisation
        assert type(opts_e[name]) is type(opts_r[name])
        ge, gr = opts_e[name].state_dict()["param_groups"], opts_r[name].state_dict()["param_groups"]
        assert ge == gr
        # a reference-written checkpoint of this architecture loads into the engine's module and its optimizer
        ck = {"model_par": r.state_dict(), "optimizer_par": opts_r[name].state_dict()}
        e.load_state_dict(ck["model_par"])
        opts_e[name].load_state_dict(ck["optimizer_par"])


def test_reference_written_checkpoint_loads_through_the_reference_caller(ref_utils):
    """core.py:523-535 with a REAL reference checkpoint: tests/golden/chunk_ligru_run_nn.npz holds the .pkl contents
    the reference's run_nn wrote after its first training chunk (model_par + RMSprop state after several steps)."""
    g = Golden("chunk_ligru_run_nn")
    m = g.meta
    cfg = configparser.ConfigParser()
    cfg.read_string(m["cfgs"]["ck1"].replace("{OUT}", "/tmp"))
    arch_dict = m["arch_dict"]
    for sec in ("architecture1", "architecture2", "architecture3"):
        cfg[sec]["arch_library"] = "pytorch-kaldi_amd.nn"
    model = cfg["model"]["model"].split("\n")
    inp_out_dict = {"fmllr": m["fea_dict"]["fmllr"][5:]}
    nns, costs = ref_utils.model_init(inp_out_dict, model, cfg, arch_dict, False, False, "train")
    opts = ref_utils.optimizer_init(nns, cfg, arch_dict)
    for name, net in nns.items():
        sec = arch_dict[name][0]
        model_par = g.group("ck0/%s/model_par/" % sec)
        state = {}
        for key, v in g.group("ck0/%s/opt/" % sec).items():
            idx, field = key.split("/")
            state.setdefault(int(idx), {})[field] = v
        optimizer_par = {"state": state, "param_groups": m["param_groups"]["ck0/%s" % sec]}
        net.load_state_dict(model_par)                      # core.py:531
        opts[name].load_state_dict(optimizer_par)           # core.py:532
        opts[name].param_groups[0]["lr"] = float(cfg[sec]["arch_lr"])  # core.py:533-535
        for k, v in net.state_dict().items():
            assert torch.equal(v, model_par[k]), (name, k)
        got = opts[name].state_dict()["state"]
        assert sorted(got) == sorted(state)
        params = list(net.parameters())
        for idx, ent in state.items():
            assert got[idx]["square_avg"].shape == params[idx].shape
            assert torch.equal(got[idx]["square_avg"], ent["square_avg"])
