"""In-memory versions of the reference recipes BASELINE.json names (section values are the
strings of the shipped cfg files), plus the synthetic chunk generator of SURVEY.md 8(d).
Used by bench.py, smoke() and the tests; real runs keep reading the .cfg files through the
reference's own run_exp.py."""
import configparser

import torch

_COMMON = {"arch_halving_factor": "0.5", "arch_improvement_threshold": "0.001", "arch_freeze": "False",
           "arch_library": "pytorch-kaldi_amd.nn", "arch_pretrain_file": "none"}


def _rms(lr="0.0004"):
    d = {"arch_lr": str(lr), "arch_opt": "rmsprop", "opt_momentum": "0.0", "opt_alpha": "0.95", "opt_eps": "1e-8",
         "opt_centered": "False", "opt_weight_decay": "0.0"}
    d.update(_COMMON)
    return d


def _sgd(lr="0.08"):
    d = {"arch_lr": str(lr), "arch_opt": "sgd", "opt_momentum": "0.0", "opt_weight_decay": "0.0",
         "opt_dampening": "0.0", "opt_nesterov": "False"}
    d.update(_COMMON)
    return d


def _rep(v, n):
    return ",".join([str(v)] * n)


def _head(n_out, opt=None):
    d = {"arch_class": "MLP", "arch_seq_model": "False", "dnn_lay": str(n_out), "dnn_drop": "0.0",
         "dnn_use_laynorm_inp": "False", "dnn_use_batchnorm_inp": "False", "dnn_use_batchnorm": "False",
         "dnn_use_laynorm": "False", "dnn_act": "softmax"}
    d.update(opt or _rms())
    return d


def _mlp(n, opt):
    d = {"arch_class": "MLP", "arch_seq_model": "False", "dnn_lay": _rep(1024, n), "dnn_drop": _rep(0.15, n),
         "dnn_use_laynorm_inp": "False", "dnn_use_batchnorm_inp": "False", "dnn_use_batchnorm": _rep(True, n),
         "dnn_use_laynorm": _rep(False, n), "dnn_act": _rep("relu", n)}
    d.update(opt)
    return d


def _rec(kind, pre, n_lay, H, act, lr):
    d = {"arch_class": kind, "arch_seq_model": "True", pre + "_lay": _rep(H, n_lay), pre + "_drop": _rep(0.2, n_lay),
         pre + "_use_laynorm_inp": "False", pre + "_use_batchnorm_inp": "False",
         pre + "_use_laynorm": _rep(False, n_lay), pre + "_use_batchnorm": _rep(True, n_lay), pre + "_bidir": "True",
         pre + "_act": _rep(act, n_lay), pre + "_orthinit": "True"}
    d.update(_rms(lr))
    return d


TWO_HEAD_MODEL = ["out_dnn1=compute(%s,fea)", "out_dnn2=compute(MLP_layers,out_dnn1)",
                  "out_dnn3=compute(MLP_layers2,out_dnn1)", "loss_mono=cost_nll(out_dnn3,lab_mono)",
                  "loss_mono_w=mult_constant(loss_mono,1.0)", "loss_cd=cost_nll(out_dnn2,lab_cd)",
                  "loss_final=sum(loss_cd,loss_mono_w)", "err_final=cost_err(out_dnn2,lab_cd)"]
ONE_HEAD_MODEL = ["out_dnn1=compute(%s,fea)", "out_dnn2=compute(MLP_layers,out_dnn1)",
                  "loss_final=cost_nll(out_dnn2,lab_cd)", "err_final=cost_err(out_dnn2,lab_cd)"]


def recipe(name, n_lay=None, H=550):
    """Returns dict(cfg, model, fea_dict, lab_dict, arch_dict, nfea, n_cd, n_mono, seq, first, trunk, head_cd,
    head_mono); the last three name the cfg sections of the MLP trunk (SincNet recipe only) and of the two heads.
    Every section value is pinned to the shipped file by tests/test_recipes_cfg.py (tests/golden/cfg_recipes.json).

    timit_ligru  cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg:131-217   (BASELINE configs[1])
    timit_lstm   cfg/TIMIT_baselines/TIMIT_LSTM_fmllr.cfg:131-217    (configs[2])
    libri_gru    cfg/Librispeech_baselines/libri_GRU_fmllr.cfg:76-146 (configs[3])
    timit_mlp    cfg/TIMIT_baselines/TIMIT_MLP_fmllr.cfg:131-214     (configs[0])
    timit_sincnet cfg/TIMIT_baselines/TIMIT_SincNet_raw.cfg:96-211   (configs[4])
    """
    cfg = configparser.ConfigParser()
    cfg["exp"] = {"to_do": "train", "use_cuda": "True"}
    nfea, n_cd, n_mono, seq = 40, 1938, 48, True
    head_cd, head_mono = _rms(), _rms()
    if name == "timit_ligru":
        first = "liGRU_layers"
        cfg["architecture1"] = _rec("liGRU", "ligru", n_lay or 5, H, "relu", "0.0004")
        model = TWO_HEAD_MODEL
    elif name == "timit_lstm":
        first = "LSTM_layers"
        cfg["architecture1"] = _rec("LSTM", "lstm", n_lay or 4, H, "tanh", "0.0016")
        model, head_cd = TWO_HEAD_MODEL, _rms("0.0016")   # the shipped cfg halves only from different rates (:173, :198)
    elif name == "libri_gru":
        first = "GRU_layers"
        cfg["architecture1"] = _rec("GRU", "gru", n_lay or 5, H, "tanh", "0.0004")
        model, n_cd, n_mono = ONE_HEAD_MODEL, 3400, 0
    elif name == "timit_mlp":
        first, nfea, seq = "MLP_layers1", 440, False
        cfg["architecture1"] = _mlp(n_lay or 5, _sgd("0.08"))
        model = TWO_HEAD_MODEL
    elif name == "timit_sincnet":
        # cfg/TIMIT_baselines/TIMIT_SincNet_raw.cfg:87-211: 200 ms raw-waveform frames (3200 samples) -> SincNet
        # (rmsprop 0.0008) -> MLP 5x1024 (sgd 0.08) -> heads (sgd 0.08).  Four architectures, numbered as shipped.
        first, nfea, seq = "SincNet_layers", 3200, False
        d = {"arch_class": "SincNet", "arch_seq_model": "False", "sinc_N_filt": "128,60,60,60", "sinc_len_filt": "129,5,5,3",
             "sinc_max_pool_len": "3,3,3,2", "sinc_use_laynorm_inp": "True", "sinc_use_batchnorm_inp": "False",
             "sinc_use_laynorm": "True,True,True,True", "sinc_use_batchnorm": "False,False,False,False",
             "sinc_act": "relu,relu,relu,relu", "sinc_drop": "0.15,0.15,0.15,0.15", "sinc_sample_rate": "16000",
             "sinc_min_low_hz": "50", "sinc_min_band_hz": "50"}
        d.update(_rms("0.0008"))
        cfg["architecture1"] = d
        cfg["architecture2"] = _mlp(n_lay or 5, _sgd("0.08"))
        cfg["architecture3"] = _head(n_cd, _sgd("0.08"))
        cfg["architecture4"] = _head(n_mono, _sgd("0.08"))
        model = ["out_dnn1=compute(SincNet_layers,fea)", "out_dnn2=compute(MLP_layers,out_dnn1)",
                 "out_dnn3=compute(MLP_soft1,out_dnn2)", "out_dnn4=compute(MLP_soft2,out_dnn2)",
                 "loss_mono=cost_nll(out_dnn4,lab_mono)", "loss_mono_w=mult_constant(loss_mono,1.0)",
                 "loss_cd=cost_nll(out_dnn3,lab_cd)", "loss_final=sum(loss_cd,loss_mono_w)",
                 "err_final=cost_err(out_dnn3,lab_cd)"]
    else:
        raise ValueError("unknown recipe " + name)
    lab_dict = {"lab_cd": ["lab_cd", "f", "o", nfea]}
    if n_mono:
        lab_dict["lab_mono"] = ["lab_mono", "f", "o", nfea + 1]
    trunk, sec_cd, sec_mono = None, "architecture2", "architecture3" if n_mono else None
    if name == "timit_sincnet":
        trunk, sec_cd, sec_mono = "architecture2", "architecture3", "architecture4"
        arch_dict = {first: ["architecture1", first, False], "MLP_layers": ["architecture2", "MLP_layers", False],
                     "MLP_soft1": ["architecture3", "MLP_soft1", False], "MLP_soft2": ["architecture4", "MLP_soft2", False]}
    else:
        cfg["architecture2"] = _head(n_cd, head_cd)
        arch_dict = {first: ["architecture1", first, seq], "MLP_layers": ["architecture2", "MLP_layers", False]}
        if n_mono:
            cfg["architecture3"] = _head(n_mono, head_mono)
            arch_dict["MLP_layers2"] = ["architecture3", "MLP_layers2", False]
    model = [m % first if "%s" in m else m for m in model]
    fea_dict = {"fea": ["fea", "lst", "opts", "0", "0", 0, nfea, nfea]}
    return {"cfg": cfg, "model": model, "fea_dict": fea_dict, "lab_dict": lab_dict, "arch_dict": arch_dict,
            "nfea": nfea, "n_cd": n_cd, "n_mono": n_mono, "seq": seq, "first": first,
            "trunk": trunk, "head_cd": sec_cd, "head_mono": sec_mono}


def synthetic_batch(rcp, T, B, seed, device="cpu"):
    """One (T, B, nfea + n_lab) batch [(N, .) for non-sequence recipes]: N(0,1) features (chunks are
    mean/var normalised, data_io.py:263) and uniform integer labels stored as float columns."""
    g = torch.Generator().manual_seed(seed)
    nlab = 2 if rcp["n_mono"] else 1
    shape = (T, B, rcp["nfea"] + nlab) if rcp["seq"] else (B, rcp["nfea"] + nlab)
    inp = torch.randn(*shape, generator=g)
    inp[..., rcp["nfea"]] = torch.randint(0, rcp["n_cd"], shape[:-1], generator=g).float()
    if rcp["n_mono"]:
        inp[..., rcp["nfea"] + 1] = torch.randint(0, rcp["n_mono"], shape[:-1], generator=g).float()
    return inp.to(device)
