"""Generate golden vectors from the REAL reference (runs only in the build container).

TEST INFRASTRUCTURE ONLY.  Imports ``/root/reference/neural_networks.py`` and
``utils.py`` unmodified (read-only, no bytecode written), runs the reference
classes on CPU fp32 on seeded inputs and stores inputs, parameters, drop masks,
outputs and gradients as small ``.npz`` fixtures under ``tests/golden/``.
``/root/reference`` does not exist on the GPU box, so these committed fixtures
are what pins both the oracle (oracle/pk_oracle.py) and the HIP engine to the
reference.

    python oracle/make_golden.py            # regenerates tests/golden/*.npz
"""

import configparser
import json
import os
import sys

sys.dont_write_bytecode = True
REF = os.environ.get("PK_REFERENCE", "/root/reference")
sys.path.insert(0, REF)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import neural_networks as ref_nn  # noqa: E402  (the reference itself)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class _MaskTap:
    """Records every torch.bernoulli() result the reference draws in forward."""

    def __init__(self):
        self.masks = []
        self._orig = torch.bernoulli

    def __enter__(self):
        def tapped(*a, **k):
            m = self._orig(*a, **k)
            self.masks.append(m.clone())
            return m

        torch.bernoulli = tapped
        return self

    def __exit__(self, *exc):
        torch.bernoulli = self._orig


def _perturb(net, seed):
    """Make BN/LN affine parameters, biases and running stats non-trivial."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("gamma") or (".weight" in name and name.split(".")[0].startswith("bn")):
                p.add_(0.2 * torch.randn(p.shape, generator=g))
            elif name.endswith("beta") or name.endswith(".bias"):
                p.add_(0.1 * torch.randn(p.shape, generator=g))
        for name, b in net.named_buffers():
            if name.endswith("running_mean"):
                b.add_(0.1 * torch.randn(b.shape, generator=g))
            elif name.endswith("running_var"):
                b.mul_(1.0 + 0.3 * torch.rand(b.shape, generator=g))


def _save(name, meta, arrays):
    os.makedirs(OUT, exist_ok=True)
    meta = dict(meta)
    meta["torch"] = torch.__version__
    meta["threads"] = torch.get_num_threads()
    arrays = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    arrays["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %s (%d bytes)" % (path, os.path.getsize(path)))


def module_case(name, arch_class, options, inp_dim, x_shape, seed, to_do="train", training=True,
                x_scale=1.0, noncontig=False):
    options = dict(options)
    options["use_cuda"] = "False"
    options["to_do"] = to_do
    torch.manual_seed(seed)
    net = getattr(ref_nn, arch_class)(options, inp_dim)
    init_sd = {k: v.clone() for k, v in net.state_dict().items()}  # what seed -> init gives
    _perturb(net, seed + 1)
    net.train() if training else net.eval()
    sd0 = {k: v.clone() for k, v in net.state_dict().items()}
    g = torch.Generator().manual_seed(seed + 2)
    x = (torch.randn(*x_shape, generator=g) * x_scale)
    if noncontig:  # forward_model hands column slices of a wider tensor (utils.py:2321)
        wide = torch.randn(*x_shape[:-1], x_shape[-1] + 2, generator=g)
        wide[..., : x_shape[-1]] = x
        x = wide[..., : x_shape[-1]]
    x = x.clone().requires_grad_(True)
    torch.manual_seed(seed + 3)  # RNG state the drop masks are drawn from
    with _MaskTap() as tap:
        y = net(x)
    cot = torch.randn(y.shape, generator=g)
    arrays = {"x": x, "y": y, "cot": cot}
    if to_do == "train" or training:
        loss = (y * cot).sum()
        loss.backward()
        arrays["dx"] = x.grad
        for k, p in net.named_parameters():
            if p.grad is not None:
                arrays["grad/" + k] = p.grad
    for k, v in sd0.items():
        arrays["sd/" + k] = v
    for k, v in init_sd.items():
        arrays["init/" + k] = v
    for k, v in net.state_dict().items():
        if "running" in k or "num_batches" in k:
            arrays["sd_after/" + k] = v
    for i, m in enumerate(tap.masks):
        arrays["mask/%d" % i] = m
    meta = {"arch_class": arch_class, "options": options, "inp_dim": inp_dim, "to_do": to_do,
            "training": training, "seed": seed, "n_masks": len(tap.masks)}
    _save(name, meta, arrays)


def rec_opts(pre, lay, act, bn=True, ln=False, bidir=True, drop=0.2, orth=True, ln_inp=False, bn_inp=False):
    n = len(lay)
    j = lambda v: ",".join([str(v)] * n)  # noqa: E731
    return {
        pre + "_lay": ",".join(map(str, lay)),
        pre + "_drop": j(drop),
        pre + "_use_laynorm_inp": str(ln_inp),
        pre + "_use_batchnorm_inp": str(bn_inp),
        pre + "_use_laynorm": j(ln),
        pre + "_use_batchnorm": j(bn),
        pre + "_bidir": str(bidir),
        pre + "_act": j(act),
        pre + "_orthinit": str(orth),
    }


def e2e_case(name, seed, T, B, H, n_cd, n_mono):
    """Drive the reference one level up: utils.model_init / forward_model on the
    shipped Li-GRU recipe scaled down."""
    import utils as ref_utils

    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REF, "cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg"))
    cfg["exp"]["to_do"] = "train"
    cfg["exp"]["use_cuda"] = "False"
    cfg["architecture1"]["ligru_lay"] = "%d,%d" % (H, H)
    for k in ("ligru_drop", "ligru_use_laynorm", "ligru_use_batchnorm", "ligru_act"):
        cfg["architecture1"][k] = ",".join(cfg["architecture1"][k].split(",")[:2])
    cfg["architecture2"]["dnn_lay"] = str(n_cd)
    cfg["architecture3"]["dnn_lay"] = str(n_mono)
    nfea = 11
    fea_dict = {"fmllr": ["fmllr", "lst", "opts", "0", "0", 0, nfea, nfea]}
    lab_dict = {"lab_cd": ["lab_cd", "f", "o", nfea], "lab_mono": ["lab_mono", "f", "o", nfea + 1]}
    arch_dict = {
        "liGRU_layers": ["architecture1", "liGRU_layers", True],
        "MLP_layers": ["architecture2", "MLP_layers", False],
        "MLP_layers2": ["architecture3", "MLP_layers2", False],
    }
    model = cfg["model"]["model"].split("\n")
    inp_out_dict = {"fmllr": fea_dict["fmllr"][5:]}  # as utils.dict_fea_lab_arch builds it ([start,end,dim])
    torch.manual_seed(seed)
    nns, costs = ref_utils.model_init(inp_out_dict, model, cfg, arch_dict, False, False, "train")
    for k, net in nns.items():
        _perturb(net, seed + len(k))
    g = torch.Generator().manual_seed(seed + 2)
    inp = torch.randn(T, B, nfea + 2, generator=g)
    inp[:, :, nfea] = torch.randint(0, n_cd, (T, B), generator=g).float()
    inp[:, :, nfea + 1] = torch.randint(0, n_mono, (T, B), generator=g).float()
    sd0 = {n: {k: v.clone() for k, v in net.state_dict().items()} for n, net in nns.items()}
    torch.manual_seed(seed + 3)
    with _MaskTap() as tap:
        outs = ref_utils.forward_model(fea_dict, lab_dict, arch_dict, model, nns, costs, inp, inp_out_dict,
                                       T, B, "train", [])
    outs["loss_final"].backward()
    arrays = {"inp": inp, "loss_final": outs["loss_final"], "err_final": outs["err_final"],
              "out_dnn1": outs["out_dnn1"], "out_dnn2": outs["out_dnn2"], "out_dnn3": outs["out_dnn3"]}
    for n, net in nns.items():
        for k, v in sd0[n].items():
            arrays["sd/%s/%s" % (n, k)] = v
        for k, p in net.named_parameters():
            if p.grad is not None:
                arrays["grad/%s/%s" % (n, k)] = p.grad
    for i, m in enumerate(tap.masks):
        arrays["mask/%d" % i] = m
    opts = {sec: dict(cfg[sec]) for sec in ("architecture1", "architecture2", "architecture3")}
    meta = {"options": opts, "model": model, "nfea": nfea, "T": T, "B": B, "seed": seed,
            "n_cd": n_cd, "n_mono": n_mono, "n_masks": len(tap.masks)}
    _save(name, meta, arrays)


def _recipe_cfg(H_lay=None, n_cd=1938, n_mono=48, drop=None):
    """The shipped Li-GRU recipe as an in-memory ConfigParser (cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg), CPU,
    N_out placeholders resolved (utils.py:707-722 does that with hmm-info); optionally scaled down."""
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REF, "cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg"))
    cfg["exp"]["to_do"] = "train"
    cfg["exp"]["use_cuda"] = "False"
    a1 = cfg["architecture1"]
    if H_lay is not None:
        n = len(H_lay)
        a1["ligru_lay"] = ",".join(map(str, H_lay))
        for k in ("ligru_drop", "ligru_use_laynorm", "ligru_use_batchnorm", "ligru_act"):
            a1[k] = ",".join(a1[k].split(",")[:n])
    if drop is not None:
        a1["ligru_drop"] = ",".join([str(drop)] * len(a1["ligru_lay"].split(",")))
    cfg["architecture2"]["dnn_lay"] = str(n_cd)
    cfg["architecture3"]["dnn_lay"] = str(n_mono)
    return cfg


def _recipe_dicts(nfea):
    fea_dict = {"fmllr": ["fmllr", "lst", "opts", "0", "0", 0, nfea, nfea]}
    lab_dict = {"lab_cd": ["lab_cd", "f", "o", nfea], "lab_mono": ["lab_mono", "f", "o", nfea + 1]}
    arch_dict = {"liGRU_layers": ["architecture1", "liGRU_layers", True],
                 "MLP_layers": ["architecture2", "MLP_layers", False],
                 "MLP_layers2": ["architecture3", "MLP_layers2", False]}
    return fea_dict, lab_dict, arch_dict


def train_case(name, seed, T=20, B=4, H=(32, 32), n_cd=23, n_mono=7, n_batches=5, n_steps=30, lr=None):
    """CE-loss trajectory of the reference: the shipped Li-GRU recipe (scaled down) trained for `n_steps` optimizer
    steps exactly as the chunk loop does it (core.py:616-642: forward_model, zero_grad, loss_final.backward(),
    optimizers[opt].step()) with the reference's own utils.model_init / optimizer_init (RMSprop lr 4e-4, alpha .95,
    eps 1e-8 per architecture).  `n_batches` synthetic batches are cycled, so the loss actually falls (the labels
    can be memorised).  Stored: initial parameters, every batch, the drop masks of every step, loss_final / err_final
    at every step, and the final parameters + RMSprop state."""
    import utils as ref_utils

    cfg = _recipe_cfg(list(H), n_cd, n_mono)
    if lr is not None:  # the recipe's 4e-4 barely moves a 32-unit network in 30 steps; a larger rate makes the
        for sec in ("architecture1", "architecture2", "architecture3"):  # trajectory sensitive to gradient errors
            cfg[sec]["arch_lr"] = str(lr)
    nfea = 11
    fea_dict, lab_dict, arch_dict = _recipe_dicts(nfea)
    model = cfg["model"]["model"].split("\n")
    inp_out_dict = {"fmllr": fea_dict["fmllr"][5:]}
    torch.manual_seed(seed)
    nns, costs = ref_utils.model_init(inp_out_dict, model, cfg, arch_dict, False, False, "train")
    optimizers = ref_utils.optimizer_init(nns, cfg, arch_dict)
    sd0 = {n: {k: v.clone() for k, v in net.state_dict().items()} for n, net in nns.items()}
    g = torch.Generator().manual_seed(seed + 2)
    batches = []
    for _ in range(n_batches):
        inp = torch.randn(T, B, nfea + 2, generator=g)
        inp[:, :, nfea] = torch.randint(0, n_cd, (T, B), generator=g).float()
        inp[:, :, nfea + 1] = torch.randint(0, n_mono, (T, B), generator=g).float()
        batches.append(inp)
    torch.manual_seed(seed + 3)
    losses, errs = [], []
    with _MaskTap() as tap:
        for step in range(n_steps):
            inp = batches[step % n_batches]
            outs = ref_utils.forward_model(fea_dict, lab_dict, arch_dict, model, nns, costs, inp, inp_out_dict,
                                           T, B, "train", [])
            for opt in optimizers:
                optimizers[opt].zero_grad()
            outs["loss_final"].backward()
            for opt in optimizers:
                optimizers[opt].step()
            losses.append(float(outs["loss_final"]))
            errs.append(float(outs["err_final"]))
    arrays = {"batches": torch.stack(batches), "loss": np.array(losses, dtype=np.float64),
              "err": np.array(errs, dtype=np.float64)}
    for n, net in nns.items():
        for k, v in sd0[n].items():
            arrays["sd/%s/%s" % (n, k)] = v
        for k, v in net.state_dict().items():
            arrays["sd_final/%s/%s" % (n, k)] = v
    for i, m in enumerate(tap.masks):
        arrays["mask/%d" % i] = m
    opts = {sec: dict(cfg[sec]) for sec in ("architecture1", "architecture2", "architecture3")}
    meta = {"options": opts, "model": model, "nfea": nfea, "T": T, "B": B, "seed": seed, "n_cd": n_cd,
            "n_mono": n_mono, "n_masks": len(tap.masks), "n_batches": n_batches, "n_steps": n_steps,
            "n_lay": len(H)}
    _save(name, meta, arrays)
    print("  loss %.4f -> %.4f, err %.3f -> %.3f" % (losses[0], losses[-1], errs[0], errs[-1]))


def _rows_sample(t, cap=8192):
    """Row subsample of a gradient / output tensor that keeps a fixture small: every `stride`-th row of the 2-D view."""
    t2 = t.reshape(t.shape[0], -1) if t.dim() > 1 else t.reshape(1, -1)
    stride = max(1, -(-t2.numel() // cap))
    if t.dim() <= 1:
        return t2[0, ::stride], stride
    stride = min(stride, t2.shape[0])
    return t2[::stride], stride


def _projections(t, seed, k=4):
    """k projections of the flattened tensor on seeded Rademacher directions (numpy RandomState: bit-stable across
    platforms): a whole-tensor checksum that, unlike a norm, also sees sign / position errors."""
    v = t.detach().double().reshape(-1).numpy()
    rs = np.random.RandomState(seed)
    return np.array([float(np.dot(v, rs.randint(0, 2, v.size) * 2.0 - 1.0)) for _ in range(k)])


def scale_case(name, seed, T=500, B=4):
    """Config-scale golden: the UNSCALED recipe - liGRU 5 x 550 bidirectional,
    BatchNorm, relu, drop 0.2, 1938 + 48 softmax heads - through the reference's utils.model_init / forward_model at the
    metric's sequence length T = 500 (B = 4 keeps the reference's O(T^2) autograd to minutes).  The 10.1 M parameters
    are NOT stored: they are what `torch.manual_seed(seed)` + model_init gives (the engine's classes reproduce the
    reference's initialisation; per-tensor checksums are stored so that a mismatch is reported as such).  Stored:
    input, drop masks, the reference's ReLU kink pattern (a_t > 0, bit-packed) for the kink-forced gradient check,
    loss / err, row samples + norms + projections of the three outputs and of every parameter gradient."""
    import utils as ref_utils

    torch.set_num_threads(8)
    cfg = _recipe_cfg()
    nfea = 40
    fea_dict, lab_dict, arch_dict = _recipe_dicts(nfea)
    model = cfg["model"]["model"].split("\n")
    inp_out_dict = {"fmllr": fea_dict["fmllr"][5:]}
    torch.manual_seed(seed)
    nns, costs = ref_utils.model_init(inp_out_dict, model, cfg, arch_dict, False, False, "train")
    g = torch.Generator().manual_seed(seed + 2)
    inp = torch.randn(T, B, nfea + 2, generator=g)
    inp[:, :, nfea] = torch.randint(0, 1938, (T, B), generator=g).float()
    inp[:, :, nfea + 1] = torch.randint(0, 48, (T, B), generator=g).float()
    arrays, meta_ck = {"inp": inp}, {}
    for n, net in nns.items():
        for k, v in net.state_dict().items():
            if v.is_floating_point():
                arrays["init_ck/%s/%s" % (n, k)] = np.concatenate(([float(v.double().norm())], _projections(v, 7)))
    rec = nns["liGRU_layers"]
    kinks = [[] for _ in rec.act]
    hooks = [a.register_forward_hook(lambda m, i, o, lst=kinks[li]: lst.append(i[0].detach() > 0))
             for li, a in enumerate(rec.act)]
    torch.manual_seed(seed + 3)
    with _MaskTap() as tap:
        outs = ref_utils.forward_model(fea_dict, lab_dict, arch_dict, model, nns, costs, inp, inp_out_dict,
                                       T, B, "train", [])
    for h in hooks:
        h.remove()
    outs["loss_final"].backward()
    arrays["loss_final"], arrays["err_final"] = outs["loss_final"], outs["err_final"]
    for k in ("out_dnn1", "out_dnn2", "out_dnn3"):
        o = outs[k].detach().reshape(T * B, -1)
        smp, stride = _rows_sample(o, 65536)
        arrays["out/%s/rows" % k], meta_ck["out/%s/stride" % k] = smp, stride
        arrays["out/%s/ck" % k] = np.concatenate(([float(o.double().norm())], _projections(o, 11)))
    for li, lst in enumerate(kinks):  # (T, 2B, H) booleans in the reference's row order, bit-packed
        k = torch.stack(lst).numpy()
        assert k.shape == (T, 2 * B, 550)
        arrays["kink/%d" % li] = np.packbits(k.reshape(-1))
    for n, net in nns.items():
        for k, p in net.named_parameters():
            if p.grad is None:
                continue
            smp, stride = _rows_sample(p.grad, 8192)
            arrays["grad/%s/%s/rows" % (n, k)], meta_ck["grad/%s/%s/stride" % (n, k)] = smp, stride
            arrays["grad/%s/%s/ck" % (n, k)] = np.concatenate(([float(p.grad.double().norm())], _projections(p.grad, 13)))
    for i, m in enumerate(tap.masks):
        arrays["mask/%d" % i] = m.to(torch.uint8)
    opts = {sec: dict(cfg[sec]) for sec in ("architecture1", "architecture2", "architecture3")}
    meta = {"options": opts, "model": model, "nfea": nfea, "T": T, "B": B, "seed": seed, "n_masks": len(tap.masks),
            "strides": meta_ck, "H": 550, "n_lay": 5}
    _save(name, meta, arrays)
    print("  loss_final %.6f err_final %.4f" % (float(outs["loss_final"]), float(outs["err_final"])))


class _DropoutTap:
    """Recovers the masks nn.Dropout drew in the (unmodified) reference: forward hooks on every nn.Dropout module with
    p > 0 compare input and output (y = x * m / (1 - p)): m = (y != 0) wherever x != 0; where x == 0 the mask can be
    seen neither in the output nor - behind a ReLU, the only place the shipped recipes put dropout - in any gradient,
    and is recorded as 1."""

    def __init__(self, nets):
        self.masks, self.tags, self._hooks, self._nets = [], [], [], nets

    def __enter__(self):
        for an, net in self._nets.items():
            for mn, mod in net.named_modules():
                if isinstance(mod, torch.nn.Dropout) and mod.p > 0.0:
                    def hook(m, inp, out, tag="%s/%s" % (an, mn)):
                        if m.training:
                            self.masks.append(torch.where(inp[0] != 0, out != 0, torch.ones_like(out, dtype=torch.bool)))
                            self.tags.append(tag)
                    self._hooks.append(mod.register_forward_hook(hook))
        return self

    def __exit__(self, *exc):
        for h in self._hooks:
            h.remove()


class _DecisionTap:
    """The discrete decisions of the feed-forward stacks in the (unmodified) reference run, recorded by forward hooks:
    ReLU patterns (input > 0) of every nn.ReLU inside MLP / CNN / SincNet modules, and the arg-max offsets of the
    max-pools of the conv stacks (F.max_pool1d is a functional call in the reference: the hook sits on the conv module
    and repeats the pooling on the tensor the reference pooled, with return_indices).  A second implementation given
    these differentiates the same piecewise-linear function (extended to pooling)."""

    def __init__(self, nets):
        self.relu, self.pool, self._hooks, self._nets = [], [], [], nets

    def __enter__(self):
        for an, net in self._nets.items():
            cls = type(net).__name__
            if cls not in ("MLP", "CNN", "SincNet"):
                continue
            for i, mod in enumerate(net.act):
                if isinstance(mod, torch.nn.ReLU):
                    self._hooks.append(mod.register_forward_hook(
                        lambda m, inp, out, tag="%s/act.%d" % (an, i): self.relu.append((tag, (inp[0].detach() > 0)))))
            if cls in ("CNN", "SincNet"):
                pools = net.sinc_max_pool_len if cls == "SincNet" else net.cnn_max_pool_len
                for i, mod in enumerate(net.conv):
                    def hook(m, inp, out, tag="%s/conv.%d" % (an, i), pool=int(pools[i])):
                        _, idx = torch.nn.functional.max_pool1d(out.detach(), pool, return_indices=True)
                        base = torch.arange(idx.shape[-1]) * pool
                        self.pool.append((tag, pool, (idx - base).to(torch.uint8)))
                    self._hooks.append(mod.register_forward_hook(hook))
        return self

    def __exit__(self, *exc):
        for h in self._hooks:
            h.remove()


def recipe_scale_case(name, cfg_rel, seed, T, B, nfea, fea_name, n_cd, n_mono, x_scale=1.0, out_cap=65536):
    """Config-scale goldens of the other BASELINE configurations: a SHIPPED cfg file, unscaled, through the reference's
    own utils.model_init / forward_model + loss_final.backward() (what core.run_nn does per batch, core.py:616-634).

      timit_lstm     cfg/TIMIT_baselines/TIMIT_LSTM_fmllr.cfg:131-217      LSTM 4 x 550 bidirectional, T = 500
      libri_gru      cfg/Librispeech_baselines/libri_GRU_fmllr.cfg:76-146  GRU 5 x 550 + 3400-way head, T = 500
      timit_sincnet  cfg/TIMIT_baselines/TIMIT_SincNet_raw.cfg:87-211      SincNet [128,60,60,60] on 3200 samples -> MLP
      timit_mlp      cfg/TIMIT_baselines/TIMIT_MLP_fmllr.cfg:131-214       MLP 440 -> 1024 x 5 -> 1938 / 48

    Like scale_case the parameters are not stored (they are what torch.manual_seed(seed) + model_init gives; per-tensor
    checksums are).  Stored: the input batch, the recurrent drop masks (torch.bernoulli tap) and the nn.Dropout masks
    (_DropoutTap), loss / err, row samples + norm + projections of every out_* tensor and of every parameter gradient."""
    import utils as ref_utils

    torch.set_num_threads(8)
    cfg = configparser.ConfigParser()
    cfg.read(os.path.join(REF, cfg_rel))
    cfg["exp"]["to_do"] = "train"
    cfg["exp"]["use_cuda"] = "False"
    secs = [s for s in cfg.sections() if s.startswith("architecture")]
    for s in secs:  # utils.py:707-722 resolves the placeholders with hmm-info; the counts are BASELINE's
        if cfg[s].get("dnn_lay") == "N_out_lab_cd":
            cfg[s]["dnn_lay"] = str(n_cd)
        if cfg[s].get("dnn_lay") == "N_out_lab_mono":
            cfg[s]["dnn_lay"] = str(n_mono)
    model = [m.strip() for m in cfg["model"]["model"].split("\n")]
    seq = T is not None
    fea_dict = {fea_name: [fea_name, "lst", "opts", "0", "0", 0, nfea, nfea]}
    lab_dict = {"lab_cd": ["lab_cd", "f", "o", nfea]}
    uses_mono = any("lab_mono" in m for m in model)
    if uses_mono:
        lab_dict["lab_mono"] = ["lab_mono", "f", "o", nfea + 1]
    arch_dict = {cfg[s]["arch_name"]: [s, cfg[s]["arch_name"], cfg[s]["arch_seq_model"].strip() == "True"] for s in secs}
    inp_out_dict = {fea_name: fea_dict[fea_name][5:]}
    torch.manual_seed(seed)
    nns, costs = ref_utils.model_init(inp_out_dict, model, cfg, arch_dict, False, False, "train")
    g = torch.Generator().manual_seed(seed + 2)
    shape = (T, B) if seq else (B,)
    inp = torch.randn(*shape, nfea + (2 if uses_mono else 1), generator=g)
    inp[..., :nfea] *= x_scale
    inp[..., nfea] = torch.randint(0, n_cd, shape, generator=g).float()
    if uses_mono:
        inp[..., nfea + 1] = torch.randint(0, n_mono, shape, generator=g).float()
    arrays, meta_ck = {"inp": inp}, {}
    for n, net in nns.items():
        for k, v in net.state_dict().items():
            if v.is_floating_point():
                arrays["init_ck/%s/%s" % (n, k)] = np.concatenate(([float(v.double().norm())], _projections(v, 7)))
    torch.manual_seed(seed + 3)
    Tm, Bm = (T, B) if seq else (B, 1)  # forward_model's max_len / batch_size only reshape (utils.py:2323-2337)
    with _MaskTap() as tap, _DropoutTap(nns) as dtap, _DecisionTap(nns) as dec:
        outs = ref_utils.forward_model(fea_dict, lab_dict, arch_dict, model, nns, costs, inp, inp_out_dict,
                                       Tm, Bm, "train", [])
    outs["loss_final"].backward()
    arrays["loss_final"], arrays["err_final"] = outs["loss_final"], outs["err_final"]
    out_keys = sorted(k for k in outs if k.startswith("out_"))
    for k in out_keys:
        o = outs[k].detach()
        o = o.reshape(-1, o.shape[-1])
        smp, stride = _rows_sample(o, out_cap)
        arrays["out/%s/rows" % k], meta_ck["out/%s/stride" % k] = smp, stride
        arrays["out/%s/ck" % k] = np.concatenate(([float(o.double().norm())], _projections(o, 11)))
    for n, net in nns.items():
        for k, p in net.named_parameters():
            if p.grad is None:
                continue
            smp, stride = _rows_sample(p.grad, 8192)
            arrays["grad/%s/%s/rows" % (n, k)], meta_ck["grad/%s/%s/stride" % (n, k)] = smp, stride
            arrays["grad/%s/%s/ck" % (n, k)] = np.concatenate(([float(p.grad.double().norm())], _projections(p.grad, 13)))
    for i, m in enumerate(tap.masks):
        arrays["mask/%d" % i] = m.to(torch.uint8)
    dshapes = []
    for i, m in enumerate(dtap.masks):  # bit-packed, in call order; shapes and owners in the meta block
        arrays["dmask/%d" % i] = np.packbits(m.numpy().reshape(-1))
        dshapes.append([dtap.tags[i], list(m.shape)])
    relus, pools = [], []
    for i, (tag, pat) in enumerate(dec.relu):  # call order; bit-packed
        arrays["relu/%d" % i] = np.packbits(pat.numpy().reshape(-1))
        relus.append([tag, list(pat.shape)])
    for i, (tag, pool, off) in enumerate(dec.pool):  # offset of the arg-max inside its pooling window
        arrays["pool/%d" % i] = off.numpy()
        pools.append([tag, pool, list(off.shape)])
    opts = {sec: dict(cfg[sec]) for sec in secs}
    meta = {"options": opts, "model": model, "nfea": nfea, "T": T, "B": B, "seed": seed, "n_masks": len(tap.masks),
            "n_dmasks": len(dtap.masks), "dmasks": dshapes, "relus": relus, "pools": pools, "strides": meta_ck, "fea_dict": fea_dict, "lab_dict": lab_dict,
            "arch_dict": arch_dict, "n_cd": n_cd, "n_mono": n_mono if uses_mono else 0, "out_keys": out_keys,
            "cfg_file": cfg_rel, "seq": seq}
    _save(name, meta, arrays)
    print("  loss_final %.6f err_final %.4f" % (float(outs["loss_final"]), float(outs["err_final"])))


def config_scale_cases():
    recipe_scale_case("scale_lstm_T500", "cfg/TIMIT_baselines/TIMIT_LSTM_fmllr.cfg", 5234, 500, 4, 40, "fmllr", 1938, 48)
    recipe_scale_case("scale_gru_libri_T500", "cfg/Librispeech_baselines/libri_GRU_fmllr.cfg", 1234, 500, 4, 40, "fmllr",
                      3400, 0)
    recipe_scale_case("scale_sincnet_3200", "cfg/TIMIT_baselines/TIMIT_SincNet_raw.cfg", 2234, None, 32, 3200, "raw",
                      1938, 48, x_scale=0.05)
    recipe_scale_case("scale_mlp_440", "cfg/TIMIT_baselines/TIMIT_MLP_fmllr.cfg", 3234, None, 128, 440, "fmllr", 1938, 48)


def chunk_case(name, seed):
    """Drive the reference's own chunk loop, core.run_nn (core.py:439-753), on an in-memory synthetic chunk: train a
    tiny Li-GRU recipe for one chunk from scratch (checkpoint ck0), continue for a second chunk from ck0 (-> ck1 +
    loss/err), then validate and forward with ck1.  Kaldi is not installed, so the chunk reader thread is stubbed;
    everything else (padding with random left zeros, forward_model, RMSprop, checkpoint / info / ark writers) is the
    reference.  The fixture pins pytorch-kaldi_amd/core.py::run_nn_dp."""
    import tempfile

    import core as ref_core

    nfea, n_cd, n_mono, H, B = 11, 13, 5, 16, 4
    g = np.random.RandomState(seed)
    lens = g.randint(5, 13, size=10)
    end = np.cumsum(lens)
    N = int(end[-1])
    data = g.randn(N, nfea + 2).astype(np.float32)
    data[:, nfea] = g.randint(0, n_cd, N)
    data[:, nfea + 1] = g.randint(0, n_mono, N)
    data_name = ["utt%02d" % i for i in range(10)]
    fea_dict = {"fmllr": ["fmllr", "lst", "opts", "0", "0", 0, nfea, nfea]}
    lab_dict = {"lab_cd": ["lab_cd", "f", "o", nfea], "lab_mono": ["lab_mono", "f", "o", nfea + 1]}
    arch_dict = {"liGRU_layers": ["architecture1", "liGRU_layers", True],
                 "MLP_layers": ["architecture2", "MLP_layers", False],
                 "MLP_layers2": ["architecture3", "MLP_layers2", False]}
    tmp = tempfile.mkdtemp()
    counts = g.randint(1, 50, n_cd)
    with open(os.path.join(tmp, "counts"), "w") as f:
        f.write("[ " + " ".join(str(int(c)) for c in counts) + " ]\n")

    def write_cfg(tag, to_do, chunk_seed, pretrain):
        cfg = configparser.ConfigParser()
        cfg.read(os.path.join(REF, "cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg"))
        for sec in [x for x in cfg.sections() if x.startswith("dataset") or x in ("data_use", "decoding", "cfg_proto")]:
            cfg.remove_section(sec)
        e = cfg["exp"]
        e["to_do"], e["seed"], e["use_cuda"], e["multi_gpu"], e["save_gpumem"] = to_do, str(chunk_seed), "False", "False", "False"
        e["production"] = "False"
        e["out_folder"] = "{OUT}"
        e["out_info"] = "{OUT}/%s.info" % tag
        cfg["batches"]["batch_size_train"] = str(B)
        cfg["batches"]["batch_size_valid"] = str(B)
        a1 = cfg["architecture1"]
        a1["ligru_lay"] = "%d,%d" % (H, H)
        for k in ("ligru_drop", "ligru_use_laynorm", "ligru_use_batchnorm", "ligru_act"):
            a1[k] = ",".join(a1[k].split(",")[:2])
        a1["ligru_drop"] = "0.0,0.0"
        cfg["architecture2"]["dnn_lay"] = str(n_cd)
        cfg["architecture3"]["dnn_lay"] = str(n_mono)
        for i in (1, 2, 3):
            cfg["architecture%d" % i]["arch_pretrain_file"] = "none" if pretrain is None else "{OUT}/%s_architecture%d.pkl" % (pretrain, i)
        fw = cfg["forward"]
        fw["forward_out"], fw["normalize_posteriors"], fw["require_decoding"] = "out_dnn2", "True", "True"
        fw["normalize_with_counts_from"] = "{OUT}/counts"
        import io
        buf = io.StringIO()
        cfg.write(buf)
        return buf.getvalue()

    cfgs = {"ck0": write_cfg("ck0", "train", seed, None), "ck1": write_cfg("ck1", "train", seed + 1, "ck0"),
            "valid": write_cfg("valid", "valid", seed + 2, "ck1"), "forward": write_cfg("forward", "forward", seed + 3, "ck1")}
    ref_core.read_lab_fea = lambda cfg_file, is_production, shared_list, output_folder: shared_list.extend(
        [data_name, end, fea_dict, lab_dict, arch_dict, data])
    ref_core.progress = lambda *a, **k: None
    for tag in ("ck0", "ck1", "valid", "forward"):
        path = os.path.join(tmp, tag + ".cfg")
        with open(path, "w") as f:
            f.write(cfgs[tag].replace("{OUT}", tmp))
        ref_core.run_nn(data_name, torch.from_numpy(data).float(), end, {k: list(v) for k, v in fea_dict.items()},
                        lab_dict, arch_dict, path, False, path)
    arrays = {"data_set": data, "data_end_index": end, "counts": counts.astype(np.float32)}
    meta = {"cfgs": cfgs, "data_name": data_name, "fea_dict": fea_dict, "lab_dict": lab_dict, "arch_dict": arch_dict,
            "seed": seed, "param_groups": {}, "info": {}}
    for ck in ("ck0", "ck1"):
        for i in (1, 2, 3):
            c = torch.load(os.path.join(tmp, "%s_architecture%d.pkl" % (ck, i)), weights_only=False)
            for k, v in c["model_par"].items():
                arrays["%s/architecture%d/model_par/%s" % (ck, i, k)] = v
            for idx, ent in c["optimizer_par"]["state"].items():
                for k, v in ent.items():
                    if v is not None:
                        arrays["%s/architecture%d/opt/%d/%s" % (ck, i, idx, k)] = torch.as_tensor(v)
            meta["param_groups"]["%s/architecture%d" % (ck, i)] = c["optimizer_par"]["param_groups"]
    for tag in ("ck1", "valid"):
        info = configparser.ConfigParser()
        info.read(os.path.join(tmp, tag + ".info"))
        meta["info"][tag] = {"loss": float(info["results"]["loss"]), "err": float(info["results"]["err"])}
    ark = open(os.path.join(tmp, "forward_out_dnn2_to_decode.ark"), "rb").read()
    arrays["ark"] = np.frombuffer(ark, dtype=np.uint8)
    _save(name, meta, arrays)


def io_case(name, seed):
    """Kaldi tables through the reference's own reader (data_io.read_mat_ark / read_mat, data_io.py:1039-1198) and its
    chunk transforms (context_window :228-241; the normalisation / label lines of load_chunk :253-274, which cannot be
    called directly because load_dataset shells out to Kaldi).  Pins pytorch-kaldi_amd/data_io.py + csrc/pk_io.hip."""
    import struct
    import tempfile

    import data_io as ref_io

    g = np.random.RandomState(seed)
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "t.ark")
    mats = {"uttA": g.randn(7, 5).astype(np.float32), "uttB": g.randn(3, 4).astype(np.float64)}
    src = (g.randn(20, 6) * [1, 5, 0.1, 2, 1, 3] + [0, 1, -2, 0, 4, 0]).astype(np.float32)
    # a compressed ('CM ') record built by hand after Kaldi's compressed-matrix.h: global (min, range, rows, cols),
    # per-column percentile headers (uint16), then one byte per element, column-major
    mn, mx = float(src.min()), float(src.max())
    rng = mx - mn
    to16 = lambda v: int(round((v - mn) / rng * 65535.0))  # noqa: E731
    hdr, cols_bytes = [], []
    for c in range(src.shape[1]):
        col = np.sort(src[:, c])
        q = [to16(col[0]), to16(col[len(col) // 4]), to16(col[3 * len(col) // 4]), to16(col[-1])]
        q[1] = max(q[1], q[0] + 1); q[2] = max(q[2], q[1] + 1); q[3] = max(q[3], q[2] + 1)
        p = [mn + rng * v / 65535.0 for v in q]
        b = []
        for v in src[:, c]:
            if v < p[1]:
                b.append(int(np.clip(round((v - p[0]) / (p[1] - p[0]) * 64), 0, 64)))
            elif v < p[2]:
                b.append(int(np.clip(round(64 + (v - p[1]) / (p[2] - p[1]) * 128), 64, 192)))
            else:
                b.append(int(np.clip(round(192 + (v - p[2]) / (p[3] - p[2]) * 63), 192, 255)))
        hdr.append(struct.pack("<4H", *q))
        cols_bytes.append(bytes(b))
    offsets = {}
    with open(path, "wb") as f:
        for k, m in mats.items():
            f.write((k + " ").encode())
            offsets[k] = f.tell()
            ref_io.write_mat(tmp, f, m)
        f.write(b"uttC ")
        offsets["uttC"] = f.tell()
        f.write(b"\0BCM " + struct.pack("<ffii", mn, rng, src.shape[0], src.shape[1]) + b"".join(hdr) + b"".join(cols_bytes))
    arrays = {"ark": np.frombuffer(open(path, "rb").read(), dtype=np.uint8)}
    keys = []
    for k, m in ref_io.read_mat_ark(path, tmp):
        keys.append(k)
        arrays["mat/" + k] = np.asarray(m)
    for k, off in offsets.items():  # the scp route: "file:offset"
        arrays["scp/" + k] = np.asarray(ref_io.read_mat("%s:%d" % (path, off), tmp))
    # integer vectors (alignments), written and read back by the reference
    ipath = os.path.join(tmp, "ali.ark")
    vecs = {"uttA": g.randint(0, 1938, 17).astype(np.int32), "uttE": np.array([], dtype=np.int32),
            "uttB": g.randint(-5, 5, 4).astype(np.int32)}
    with open(ipath, "wb") as f:
        for k, v in vecs.items():
            ref_io.write_vec_int(f, tmp, v, key=k)
    arrays["ali_ark"] = np.frombuffer(open(ipath, "rb").read(), dtype=np.uint8)
    ikeys = []
    for k, v in ref_io.read_vec_int_ark(ipath, tmp):
        ikeys.append(k)
        arrays["ali/" + k] = np.asarray(v, dtype=np.int32)
    # chunk transforms
    fea = g.randn(31, 4).astype(np.float32)
    lab = g.randint(3, 9, 31)
    left, right = 2, 3
    end_index = np.array([10, 19, 31])
    cw = ref_io.context_window(fea, left, right)
    arrays.update({"cw/fea": fea, "cw/out": cw, "chunk/lab": lab, "chunk/end_index": end_index})
    data_set = (cw - np.mean(cw, axis=0)) / np.std(cw, axis=0)          # data_io.py:263
    data_lab = lab - lab.min()                                           # :266
    data_lab = data_lab[left:-right]                                     # :267-270
    arrays["chunk/data_set"] = np.column_stack((data_set, data_lab))     # :272
    e = end_index - left                                                 # :259-260
    e[-1] = e[-1] - right
    arrays["chunk/end_index_out"] = e
    _save(name, {"keys": keys, "ali_keys": ikeys, "offsets": offsets, "left": left, "right": right, "seed": seed}, arrays)


def loader_case(name, seed):
    """The reference's chunk loader ITSELF - data_io.load_chunk -> load_dataset (data_io.py:16-225, 244-274): key
    filtering, splitting of long sentences, the two length sorts, concatenation, end indices, context window,
    normalisation, label shift - on tables on disk.  Its two Kaldi pipes are satisfied by stand-in programs put on PATH
    for this run only: `copy-feats scp:X ark:-` re-emits the scp's records through the reference's own read_mat /
    write_mat, `ali-to-pdf MODEL ark:- ark:-` is a pass-through (the stored alignments already are pdf-ids).
    Pins pytorch-kaldi_amd/data_io.py::load_dataset / load_chunk."""
    import gzip
    import stat
    import tempfile

    import data_io as ref_io

    g = np.random.RandomState(seed)
    tmp = tempfile.mkdtemp()
    bindir = os.path.join(tmp, "bin")
    os.makedirs(bindir)
    stubs = {
        "copy-feats": "#!%s\nimport sys\nsys.path.insert(0, %r)\nimport data_io\nout = sys.stdout.buffer\n"
                      "for line in open(sys.argv[1].split(':', 1)[1]):\n    key, rx = line.strip().split(' ', 1)\n"
                      "    data_io.write_mat(%r, out, data_io.read_mat(rx, %r), key=key)\nout.flush()\n"
                      % (sys.executable, REF, tmp, tmp),
        "ali-to-pdf": "#!%s\nimport sys\nsys.stdout.buffer.write(sys.stdin.buffer.read())\n" % sys.executable,
    }
    for prog, text in stubs.items():
        path = os.path.join(bindir, prog)
        with open(path, "w") as f:
            f.write(text)
        os.chmod(path, os.stat(path).st_mode | stat.S_IEXEC)
    D = 6
    lengths = {"spk1_a": 30, "spk1_b": 50, "spk2_a": 55, "spk2_b": 70, "spk3_a": 130, "spk3_b": 30, "spk4_a": 12,
               "spk4_b": 63, "no_ali": 21}
    feats = {k: (g.randn(n, D) * g.uniform(0.5, 2.0, D) + g.uniform(-1, 1, D)).astype(np.float32) for k, n in lengths.items()}
    labs = {k: g.randint(2, 40, n).astype(np.int32) for k, n in lengths.items() if k != "no_ali"}
    labs["no_fea"] = g.randint(2, 40, 17).astype(np.int32)
    ark = os.path.join(tmp, "fea.ark")
    offsets = {}
    with open(ark, "wb") as f:
        for k in sorted(feats, reverse=True):  # file order differs from the sorted order the loader imposes
            f.write((k + " ").encode())
            offsets[k] = f.tell()
            ref_io.write_mat(tmp, f, feats[k])
    scp = os.path.join(tmp, "fea.scp")
    with open(scp, "w") as f:
        for k in sorted(feats, reverse=True):
            f.write("%s %s:%d\n" % (k, ark, offsets[k]))
    labdir = os.path.join(tmp, "ali")
    os.makedirs(labdir)
    ali_plain = os.path.join(tmp, "ali.ark")
    with open(ali_plain, "wb") as f:
        for k in sorted(labs):
            ref_io.write_vec_int(f, tmp, labs[k], key=k)
    with open(ali_plain, "rb") as f, gzip.open(os.path.join(labdir, "ali.1.gz"), "wb") as z:
        z.write(f.read())
    open(os.path.join(labdir, "final.mdl"), "w").close()
    arrays = {"fea_ark": np.frombuffer(open(ark, "rb").read(), dtype=np.uint8),
              "ali_ark": np.frombuffer(open(ali_plain, "rb").read(), dtype=np.uint8)}
    runs = {"seq_msl50": (0, 0, 50, False), "cw_3_2_msl1000": (3, 2, 1000, False), "forward_cw_2_2": (2, 2, -1, True),
            "dict_msl40": (0, 0, {"chunk_size_fea": 40, "chunk_step_fea": 40, "chunk_size_lab": 40, "chunk_step_lab": 40,
                                  "window_shift": 1, "window_size": 1}, False)}
    meta_runs = {}
    old_path = os.environ["PATH"]
    os.environ["PATH"] = bindir + os.pathsep + old_path
    try:
        for rname, (left, right, msl, fea_only) in runs.items():
            names, data_set, end_index = ref_io.load_chunk(scp, "", None if fea_only else labdir,
                                                            None if fea_only else "ali-to-pdf", left, right, msl, tmp, fea_only)
            arrays[rname + "/data_set"] = np.asarray(data_set)
            arrays[rname + "/end_index"] = np.asarray(end_index)
            meta_runs[rname] = {"left": left, "right": right, "max_sequence_length": msl, "fea_only": fea_only,
                                "names": list(names), "dtype": str(np.asarray(data_set).dtype)}
    finally:
        os.environ["PATH"] = old_path
    _save(name, {"offsets": offsets, "scp_order": sorted(feats, reverse=True), "runs": meta_runs, "seed": seed}, arrays)


def reader_case(name, seed):
    """The reference's chunk reader, data_io.read_lab_fea (data_io.py:536-655: what core.run_nn starts in its reader
    thread), on chunk cfg files that name two feature streams with different context windows and two label sets.  Kaldi
    pipes are satisfied as in loader_case.  Pins pytorch-kaldi_amd/data_io.py::read_lab_fea / dict_fea_lab_arch."""
    import gzip
    import stat
    import tempfile

    import data_io as ref_io

    g = np.random.RandomState(seed)
    tmp = tempfile.mkdtemp()
    bindir = os.path.join(tmp, "bin")
    os.makedirs(bindir)
    stubs = {
        "copy-feats": "#!%s\nimport sys\nsys.path.insert(0, %r)\nimport data_io\nout = sys.stdout.buffer\n"
                      "for line in open(sys.argv[1].split(':', 1)[1]):\n    key, rx = line.strip().split(' ', 1)\n"
                      "    data_io.write_mat(%r, out, data_io.read_mat(rx, %r), key=key)\nout.flush()\n"
                      % (sys.executable, REF, tmp, tmp),
        "ali-to-pdf": "#!%s\nimport sys\nsys.stdout.buffer.write(sys.stdin.buffer.read())\n" % sys.executable,
    }
    for prog, text in stubs.items():
        path = os.path.join(bindir, prog)
        with open(path, "w") as f:
            f.write(text)
        os.chmod(path, os.stat(path).st_mode | stat.S_IEXEC)
    lengths = {"a1": 14, "a2": 9, "b1": 23, "b2": 14, "c1": 31, "c2": 18}
    dims = {"fbank": 5, "mfcc": 3}
    arrays, offsets = {}, {}
    for stream, D in dims.items():
        ark = os.path.join(tmp, stream + ".ark")
        offsets[stream] = {}
        with open(ark, "wb") as f:
            for k in sorted(lengths):
                f.write((k + " ").encode())
                offsets[stream][k] = f.tell()
                ref_io.write_mat(tmp, f, (g.randn(lengths[k], D) * 2 + 1).astype(np.float32))
        with open(os.path.join(tmp, stream + ".scp"), "w") as f:
            for k in sorted(lengths):
                f.write("%s %s:%d\n" % (k, ark, offsets[stream][k]))
        arrays[stream + "_ark"] = np.frombuffer(open(ark, "rb").read(), dtype=np.uint8)
    for labname, hi in (("lab_cd", 37), ("lab_mono", 11)):
        d = os.path.join(tmp, labname)
        os.makedirs(d)
        plain = os.path.join(tmp, labname + ".ark")
        with open(plain, "wb") as f:
            for k in sorted(lengths):
                ref_io.write_vec_int(f, tmp, g.randint(1, hi, lengths[k]).astype(np.int32), key=k)
        with open(plain, "rb") as f, gzip.open(os.path.join(d, "ali.1.gz"), "wb") as z:
            z.write(f.read())
        open(os.path.join(d, "final.mdl"), "w").close()
        arrays[labname + "_ark"] = np.frombuffer(open(plain, "rb").read(), dtype=np.uint8)

    def cfg_text(to_do, streams, seq, msl):
        fea = "\n\n".join("fea_name=%s\n\tfea_lst={TMP}/%s.scp\n\tfea_opts=\n\tcw_left=%d\n\tcw_right=%d" % (n, n, l, r)
                          for n, l, r in streams)
        lab = "\n\n".join("lab_name=%s\n\tlab_folder={TMP}/%s\n\tlab_opts=ali-to-pdf" % (n, n) for n in ("lab_cd", "lab_mono"))
        if len(streams) == 2:
            model = ["conc1=concatenate(%s,%s)" % (streams[0][0], streams[1][0]), "out_dnn1=compute(net1,conc1)"]
        else:
            model = ["out_dnn1=compute(net1,%s)" % streams[0][0]]
        model += ["out_dnn2=compute(head_cd,out_dnn1)", "out_dnn3=compute(head_mono,out_dnn1)",
                  "loss_mono=cost_nll(out_dnn3,lab_mono)", "loss_mono_w=mult_constant(loss_mono,1.0)",
                  "loss_cd=cost_nll(out_dnn2,lab_cd)", "loss_final=sum(loss_cd,loss_mono_w)", "err_final=cost_err(out_dnn2,lab_cd)"]
        txt = "[exp]\nto_do = %s\n\n[batches]\nmax_seq_length_train = %d\nmax_seq_length_valid = %d\n\n" % (to_do, msl, msl)
        txt += "[data_chunk]\nfea = " + fea.replace("\n", "\n\t").replace("\t\t", "\t") + "\nlab = " + lab.replace("\n", "\n\t").replace("\t\t", "\t") + "\n\n"
        txt += "[architecture1]\narch_name = net1\narch_seq_model = %s\n\n[architecture2]\narch_name = head_cd\narch_seq_model = False\n\n" % seq
        txt += "[architecture3]\narch_name = head_mono\narch_seq_model = False\n\n[model]\nmodel = " + "\n\t".join(model) + "\n"
        return txt

    runs = {"train_mlp_two_streams": ("train", [("fbank", 2, 1), ("mfcc", 0, 0)], False, 1000, False),
            "train_seq_split": ("train", [("fbank", 0, 0)], True, 20, False),
            "valid_mlp_one_stream": ("valid", [("mfcc", 1, 1)], False, 1000, False),
            "forward_production": ("forward", [("fbank", 2, 1), ("mfcc", 1, 2)], False, 1000, True)}
    meta_runs = {}
    old_path = os.environ["PATH"]
    os.environ["PATH"] = bindir + os.pathsep + old_path
    try:
        for i, (rname, (to_do, streams, seq, msl, fea_only)) in enumerate(runs.items()):
            text = cfg_text(to_do, streams, seq, msl)
            path = os.path.join(tmp, rname + ".cfg")
            with open(path, "w") as f:
                f.write(text.replace("{TMP}", tmp))
            np.random.seed(seed + i)  # the reader shuffles non-sequence training chunks with the global numpy RNG
            shared = []
            ref_io.read_lab_fea(path, fea_only, shared, tmp)
            data_name, end_index, fea_dict, lab_dict, arch_dict, data_set = shared
            arrays[rname + "/data_set"] = np.asarray(data_set)
            arrays[rname + "/end_index"] = np.asarray(end_index)
            meta_runs[rname] = {"cfg": text, "fea_only": fea_only, "np_seed": seed + i, "names": list(data_name),
                                "fea_dict": {k: [v if isinstance(v, str) else int(v) for v in vals] for k, vals in fea_dict.items()},
                                "lab_dict": {k: ([v if isinstance(v, str) else int(v) for v in vals] if isinstance(vals, list) else vals)
                                             for k, vals in lab_dict.items()},
                                "arch_dict": {k: [v[0], v[1], bool(v[2])] for k, v in arch_dict.items()}}
    finally:
        os.environ["PATH"] = old_path
    _save(name, {"offsets": offsets, "runs": meta_runs, "seed": seed, "tmp": tmp}, arrays)


def oracle_extra_cases():
    """More option combinations of the same classes, used to pin the ORACLE only (prefix "ora_": tests/test_oracle_golden.py
    takes them, the GPU module-case lists do not until the engine has been run against them on hardware): per-step
    LayerNorm in the gated cells, input normalisations, eval mode of the other cells, BatchNorm in SincNet / CNN."""
    module_case("ora_lstm_bidir_ln_lninp", "LSTM", rec_opts("lstm", [16, 12], "tanh", bn=False, ln=True, drop=0.1, ln_inp=True),
                7, (8, 3, 7), 900)
    module_case("ora_lstm_eval", "LSTM", rec_opts("lstm", [14], "tanh"), 6, (7, 4, 6), 905, to_do="valid", training=False)
    module_case("ora_gru_bidir_ln_bninp", "GRU", rec_opts("gru", [12, 10], "tanh", bn=False, ln=True, bn_inp=True),
                5, (9, 4, 5), 910)
    module_case("ora_gru_eval", "GRU", rec_opts("gru", [11], "relu", bidir=False), 5, (6, 3, 5), 915, to_do="valid",
                training=False)
    module_case("ora_mingru_uni_plain_relu", "minimalGRU", rec_opts("minimalgru", [13], "relu", bn=False, bidir=False, drop=0.0),
                6, (7, 5, 6), 920)
    module_case("ora_mingru_bidir_ln", "minimalGRU", rec_opts("minimalgru", [10, 10], "tanh", bn=False, ln=True),
                4, (6, 3, 4), 925)
    module_case("ora_rnn_eval", "RNN", rec_opts("rnn", [12, 9], "relu"), 5, (8, 3, 5), 930, to_do="valid", training=False)
    sinc = {"sinc_N_filt": "6,5", "sinc_len_filt": "21,5", "sinc_max_pool_len": "3,2",
            "sinc_use_laynorm_inp": "False", "sinc_use_batchnorm_inp": "True",
            "sinc_use_laynorm": "False,False", "sinc_use_batchnorm": "True,True",
            "sinc_act": "relu,tanh", "sinc_drop": "0.0,0.0",
            "sinc_sample_rate": "16000", "sinc_min_low_hz": "50", "sinc_min_band_hz": "50"}
    module_case("ora_sincnet_bn_bninp", "SincNet", sinc, 160, (6, 160), 940, x_scale=0.1)
    cnn = {"cnn_N_filt": "7,5", "cnn_len_filt": "7,3", "cnn_max_pool_len": "2,2",
           "cnn_use_laynorm_inp": "True", "cnn_use_batchnorm_inp": "False",
           "cnn_use_laynorm": "True,False", "cnn_use_batchnorm": "False,True",
           "cnn_act": "leaky_relu,relu", "cnn_drop": "0.0,0.0"}
    module_case("ora_cnn_lninp_eval", "CNN", cnn, 90, (5, 90), 950, x_scale=0.5, to_do="valid", training=False)


def model_lang_case(name, seed):
    """The [model] mini-language one level up (utils.model_init / forward_model, utils.py:2031-2103, 2296-2420) on a
    graph that uses every operation the e2e recipe does not: two input streams, concatenate, avg, mult, sum,
    sum_constant, mult_constant, mse - as a training step on a (T, B, .) batch fed to non-sequence networks (the
    (T*B, .) views of :2323-2337) and as a forward pass that stops at forward_outs[-1] (:2341).  Networks are the
    reference's MLP; the fixture pins pytorch-kaldi_amd/utils.py::model_init / forward_model."""
    import utils as ref_utils

    n_cd, n_mono, T, B = 9, 4, 5, 3
    mlp = {"arch_library": "neural_networks", "arch_class": "MLP", "arch_pretrain_file": "none", "arch_freeze": "False",
           "arch_seq_model": "False", "dnn_drop": "0.0,0.0", "dnn_use_laynorm_inp": "False", "dnn_use_batchnorm_inp": "False",
           "dnn_use_batchnorm": "False,False", "dnn_use_laynorm": "False,False", "dnn_act": "tanh,relu"}
    head = dict(mlp, dnn_drop="0.0", dnn_use_batchnorm="False", dnn_use_laynorm="False", dnn_act="softmax")
    cfg = configparser.ConfigParser()
    cfg["exp"] = {"to_do": "train", "use_cuda": "False"}
    cfg["architecture1"] = dict(mlp, arch_name="net1", dnn_lay="12,7")
    cfg["architecture2"] = dict(mlp, arch_name="net2", dnn_lay="10,7")
    cfg["architecture3"] = dict(head, arch_name="head_cd", dnn_lay=str(n_cd))
    cfg["architecture4"] = dict(head, arch_name="head_mono", dnn_lay=str(n_mono))
    model = ["conc1=concatenate(fbank,mfcc)", "out_dnn1=compute(net1,conc1)", "out_dnn2=compute(net2,mfcc)",
             "avg1=avg(out_dnn1,out_dnn2)", "mul1=mult(out_dnn1,out_dnn2)", "s1=sum(avg1,mul1)", "s2=sum_constant(s1,0.5)",
             "out_dnn3=compute(head_cd,s2)", "out_dnn4=compute(head_mono,out_dnn1)", "loss_mono=cost_nll(out_dnn4,lab_mono)",
             "loss_mono_w=mult_constant(loss_mono,0.3)", "loss_cd=cost_nll(out_dnn3,lab_cd)", "loss_mse=mse(out_dnn1,out_dnn2)",
             "loss_a=sum(loss_cd,loss_mono_w)", "loss_final=sum(loss_a,loss_mse)", "err_final=cost_err(out_dnn3,lab_cd)"]
    fea_dict = {"fbank": ["fbank", "lst", "", "0", "0", 0, 5, 5], "mfcc": ["mfcc", "lst", "", "0", "0", 5, 8, 3]}
    lab_dict = {"lab_mono": ["lab_mono", "f", "o", 8], "lab_cd": ["lab_cd", "f", "o", 9]}
    arch_dict = {"net1": ["architecture1", "net1", False], "net2": ["architecture2", "net2", False],
                 "head_cd": ["architecture3", "head_cd", False], "head_mono": ["architecture4", "head_mono", False]}
    inp_out_dict = {k: list(v) for k, v in fea_dict.items()}
    torch.manual_seed(seed)
    nns, costs = ref_utils.model_init(inp_out_dict, model, cfg, arch_dict, False, False, "train")
    for k, net in nns.items():
        _perturb(net, seed + len(k))
    g = torch.Generator().manual_seed(seed + 2)
    inp = torch.randn(T, B, 10, generator=g)
    inp[:, :, 8] = torch.randint(0, n_mono, (T, B), generator=g).float()
    inp[:, :, 9] = torch.randint(0, n_cd, (T, B), generator=g).float()
    arrays = {"inp": inp}
    for n, net in nns.items():
        for k, v in net.state_dict().items():
            arrays["sd/%s/%s" % (n, k)] = v.clone()
    outs = ref_utils.forward_model(fea_dict, lab_dict, arch_dict, model, nns, costs, inp, inp_out_dict, T, B, "train", [])
    outs["loss_final"].backward()
    for k, v in outs.items():
        arrays["train/" + k] = v.detach()
    for n, net in nns.items():
        for k, q in net.named_parameters():
            if q.grad is not None:
                arrays["grad/%s/%s" % (n, k)] = q.grad
    for net in nns.values():
        net.eval()
    fwd = ref_utils.forward_model(fea_dict, lab_dict, arch_dict, model, nns, costs, inp[:, 0, :], inp_out_dict, T, 1,
                                  "forward", ["out_dnn3"])
    for k, v in fwd.items():
        arrays["forward/" + k] = v.detach()
    meta = {"options": {sec: dict(cfg[sec]) for sec in cfg.sections() if sec.startswith("architecture")}, "model": model,
            "fea_dict": fea_dict, "lab_dict": lab_dict, "arch_dict": arch_dict, "inp_out_dict": inp_out_dict, "T": T, "B": B,
            "seed": seed, "train_keys": sorted(outs), "forward_keys": sorted(fwd)}
    _save(name, meta, arrays)


def cfg_case(name):
    """The architecture / model / batch sections of the shipped cfg files BASELINE.json names, as parsed by
    configparser: pins pytorch-kaldi_amd/recipes.py (what bench.py builds) to the reference's recipes."""
    files = {"timit_ligru": "cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg", "timit_lstm": "cfg/TIMIT_baselines/TIMIT_LSTM_fmllr.cfg",
             "libri_gru": "cfg/Librispeech_baselines/libri_GRU_fmllr.cfg", "timit_mlp": "cfg/TIMIT_baselines/TIMIT_MLP_fmllr.cfg",
             "timit_sincnet": "cfg/TIMIT_baselines/TIMIT_SincNet_raw.cfg"}
    out = {}
    for key, rel in files.items():
        cfg = configparser.ConfigParser()
        cfg.read(os.path.join(REF, rel))
        out[key] = {"file": rel, "model": cfg["model"]["model"].split("\n"),
                    "sections": {sec: dict(cfg[sec]) for sec in cfg.sections() if sec.startswith("architecture")}}
    path = os.path.join(OUT, name + ".json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print("wrote %s (%d bytes)" % (path, os.path.getsize(path)))


def main():
    torch.set_num_threads(1)  # bit-stable fixtures
    if os.environ.get("PK_GOLDEN_ONLY") == "chunk":  # regenerate only the chunk-loop fixture
        chunk_case("chunk_ligru_run_nn", 1234)
        return
    if os.environ.get("PK_GOLDEN_ONLY") == "cfg":
        cfg_case("cfg_recipes")
        return
    if os.environ.get("PK_GOLDEN_ONLY") == "io":
        io_case("io_kaldi_tables", 77)
        return
    if os.environ.get("PK_GOLDEN_ONLY") == "loader":
        loader_case("io_chunk_loader", 91)
        return
    if os.environ.get("PK_GOLDEN_ONLY") == "reader":
        reader_case("io_chunk_reader", 57)
        return
    if os.environ.get("PK_GOLDEN_ONLY") == "model_lang":
        model_lang_case("e2e_model_language", 333)
        return
    if os.environ.get("PK_GOLDEN_ONLY") == "train":
        train_case("train_ligru_30steps", 4100, lr=0.004)
        return
    if os.environ.get("PK_GOLDEN_ONLY") == "scale":
        scale_case("scale_ligru_T500", 4234)
        return
    if os.environ.get("PK_GOLDEN_ONLY") == "scale_b32":
        scale_case("scale_ligru_T500_B32", 4334, B=32)
        return
    if os.environ.get("PK_GOLDEN_ONLY") == "config_scale":
        config_scale_cases()
        return
    if os.environ.get("PK_GOLDEN_ONLY") == "oracle_extra":
        oracle_extra_cases()
        return
    # --- recurrent family -----------------------------------------------------
    module_case("ligru_bidir_bn", "liGRU", rec_opts("ligru", [24, 16], "relu"), 7, (9, 3, 7), 100)
    module_case("ligru_uni_ln_bias", "liGRU", rec_opts("ligru", [20, 12], "relu", bn=False, ln=True, bidir=False,
                                                       drop=0.1, ln_inp=True), 6, (8, 4, 6), 110)
    module_case("ligru_plain_tanh_bninp", "liGRU", rec_opts("ligru", [18], "tanh", bn=False, ln=False, bidir=True,
                                                            drop=0.0, bn_inp=True), 5, (7, 2, 5), 120)
    module_case("ligru_bidir_bn_wide", "liGRU", rec_opts("ligru", [40, 40, 40], "relu"), 13, (16, 8, 13), 130,
                noncontig=True)
    module_case("ligru_eval", "liGRU", rec_opts("ligru", [24, 16], "relu"), 7, (9, 3, 7), 140,
                to_do="valid", training=False)
    module_case("lstm_bidir_bn", "LSTM", rec_opts("lstm", [20, 14], "tanh"), 7, (9, 3, 7), 200)
    module_case("lstm_uni_nobn", "LSTM", rec_opts("lstm", [16], "tanh", bn=False, bidir=False, drop=0.3), 5,
                (6, 4, 5), 210)
    module_case("gru_bidir_bn", "GRU", rec_opts("gru", [20, 14], "tanh"), 7, (9, 3, 7), 300)
    module_case("gru_uni_relu", "GRU", rec_opts("gru", [12], "relu", bn=False, bidir=False, drop=0.1), 4,
                (7, 3, 4), 310)
    module_case("mingru_bidir_bn", "minimalGRU", rec_opts("minimalgru", [20, 14], "relu"), 7, (9, 3, 7), 400)
    module_case("rnn_bidir_bn", "RNN", rec_opts("rnn", [20, 14], "relu"), 7, (9, 3, 7), 500)
    module_case("rnn_uni_ln_tanh", "RNN", rec_opts("rnn", [14], "tanh", bn=False, ln=True, bidir=False, drop=0.0), 5,
                (6, 3, 5), 510)

    # --- MLP --------------------------------------------------------------------
    mlp = {"dnn_lay": "32,24,17", "dnn_drop": "0.0,0.0,0.0", "dnn_use_laynorm_inp": "False",
           "dnn_use_batchnorm_inp": "False", "dnn_use_batchnorm": "True,True,False",
           "dnn_use_laynorm": "False,False,False", "dnn_act": "relu,relu,softmax"}
    module_case("mlp_bn_relu_softmax", "MLP", mlp, 20, (16, 20), 600)
    mlp2 = dict(mlp, dnn_use_batchnorm="False,True,False", dnn_use_laynorm="True,True,False",
                dnn_use_laynorm_inp="True", dnn_use_batchnorm_inp="True", dnn_act="tanh,leaky_relu,linear")
    module_case("mlp_ln_bn_mixed", "MLP", mlp2, 20, (16, 20), 610)
    mlp3 = dict(mlp, dnn_lay="16,9", dnn_drop="0.0,0.0", dnn_use_batchnorm="False,False",
                dnn_use_laynorm="False,False", dnn_act="sigmoid,elu")
    module_case("mlp_plain_sigmoid_elu", "MLP", mlp3, 10, (8, 10), 620)
    module_case("mlp_eval", "MLP", mlp, 20, (16, 20), 630, to_do="valid", training=False)

    # --- CNN / SincNet ----------------------------------------------------------
    sinc = {"sinc_N_filt": "8,6,5", "sinc_len_filt": "33,5,3", "sinc_max_pool_len": "3,2,2",
            "sinc_use_laynorm_inp": "True", "sinc_use_batchnorm_inp": "False",
            "sinc_use_laynorm": "True,True,True", "sinc_use_batchnorm": "False,False,False",
            "sinc_act": "relu,relu,leaky_relu", "sinc_drop": "0.0,0.0,0.0",
            "sinc_sample_rate": "16000", "sinc_min_low_hz": "50", "sinc_min_band_hz": "50"}
    module_case("sincnet_ln", "SincNet", sinc, 200, (4, 200), 700, x_scale=0.1)
    cnn = {"cnn_N_filt": "8,6", "cnn_len_filt": "9,5", "cnn_max_pool_len": "3,2",
           "cnn_use_laynorm_inp": "False", "cnn_use_batchnorm_inp": "False",
           "cnn_use_laynorm": "False,True", "cnn_use_batchnorm": "True,False",
           "cnn_act": "relu,tanh", "cnn_drop": "0.0,0.0"}
    module_case("cnn_bn_ln", "CNN", cnn, 120, (5, 120), 710, x_scale=0.5)

    oracle_extra_cases()

    # --- one level up: the shipped recipe through utils.forward_model ------------
    e2e_case("e2e_ligru_two_heads", 800, T=10, B=4, H=16, n_cd=23, n_mono=7)
    model_lang_case("e2e_model_language", 333)
    train_case("train_ligru_30steps", 4100, lr=0.004)   # CE-loss trajectory over 30 optimizer steps
    scale_case("scale_ligru_T500", 4234)      # the unscaled recipe at T = 500 (minutes of CPU time)
    scale_case("scale_ligru_T500_B32", 4334, B=32)  # the same above toy batch: 64 rows = 4 clusters of the persistent launch
    config_scale_cases()                      # the other four BASELINE configurations, unscaled

    # --- two levels up: the chunk loop core.run_nn (train from scratch, continue, validate, forward) ---
    chunk_case("chunk_ligru_run_nn", 1234)

    # --- either side of the path: Kaldi tables and the chunk transforms of data_io.load_chunk ---
    io_case("io_kaldi_tables", 77)
    loader_case("io_chunk_loader", 91)
    reader_case("io_chunk_reader", 57)
    cfg_case("cfg_recipes")


if __name__ == "__main__":
    main()
