"""Host-side mirror of the reference's model-graph interpreter for this path.

PyTorch-Kaldi keeps ``utils.model_init / optimizer_init / forward_model``
(utils.py:2031-2164, 2296-2420) as the caller of the arch classes; recipes that
switch ``arch_library`` to ``pytorch-kaldi_amd.nn`` keep using the reference's own
copies.  The GPU box has no reference checkout, so ``bench.py``, ``smoke()`` and the
parity tests drive the engine through this re-statement, which keeps the same
names, argument order and semantics (including the ``[model]`` mini-language:
compute / cost_nll / cost_err / concatenate / sum / mult / mult_constant /
sum_constant / avg / mse) so a test written against it reads like a test written
against the reference.
"""
import importlib
import re

import torch
import torch.nn as nn
import torch.optim as optim

F_ = importlib.import_module(__package__ + ".functional")

_LINE = re.compile(r"(.*)=(.*)\((.*),(.*)\)")


def strtobool(s):
    s = str(s).strip().lower()
    if s in ("y", "yes", "t", "true", "on", "1"):
        return 1
    if s in ("n", "no", "f", "false", "off", "0"):
        return 0
    raise ValueError("invalid truth value %r" % (s,))


def _parse(line):
    m = _LINE.findall(line)
    if not m:
        raise ValueError("cannot parse model line %r" % (line,))
    return [s.strip() for s in m[0]]


_ELEMENTWISE = ("mult", "sum", "mult_constant", "sum_constant", "avg", "mse")


def model_init(inp_out_dict, model, config, arch_dict, use_cuda, multi_gpu, to_do):
    """utils.py:2031-2103 - instantiate ``arch_class(options, inp_dim)`` for every
    ``compute`` line (injecting use_cuda / to_do into the section) and one
    ``nn.NLLLoss`` per ``cost_nll`` line.  ``multi_gpu`` is ignored here: data
    parallelism is one process per GPU (pytorch-kaldi_amd/dp.py), not DataParallel."""
    nns, costs = {}, {}
    for line in model:
        out_name, operation, inp1, inp2 = _parse(line)
        if operation == "compute":
            section = arch_dict[inp1][0]
            inp_dim = inp_out_dict[inp2][-1]
            module = importlib.import_module(config[section]["arch_library"])
            nn_class = getattr(module, config[section]["arch_class"])
            config.set(section, "use_cuda", config["exp"]["use_cuda"])
            config.set(section, "to_do", config["exp"]["to_do"])
            net = nn_class(config[section], inp_dim)
            if use_cuda:
                net.cuda()
            if to_do == "train" and not strtobool(config[section]["arch_freeze"]):
                net.train()
            else:
                net.eval()
            nns[arch_dict[inp1][1]] = net
            inp_out_dict[out_name] = [net.out_dim]
        elif operation == "concatenate":
            inp_out_dict[out_name] = [inp_out_dict[inp1][-1] + inp_out_dict[inp2][-1]]
        elif operation == "cost_nll":
            costs[out_name] = nn.NLLLoss()
            inp_out_dict[out_name] = [1]
        elif operation == "cost_err":
            inp_out_dict[out_name] = [1]
        elif operation in _ELEMENTWISE:
            inp_out_dict[out_name] = inp_out_dict[inp1]
    return [nns, costs]


def optimizer_init(nns, config, arch_dict):
    """utils.py:2106-2164 - one torch optimizer per architecture."""
    optimizers = {}
    for net in nns.keys():
        sec = config[arch_dict[net][0]]
        lr = float(sec["arch_lr"])
        kind = sec["arch_opt"]
        if kind == "sgd":
            optimizers[net] = optim.SGD(nns[net].parameters(), lr=lr, momentum=float(sec["opt_momentum"]),
                                        weight_decay=float(sec["opt_weight_decay"]),
                                        dampening=float(sec["opt_dampening"]),
                                        nesterov=bool(strtobool(sec["opt_nesterov"])))
        elif kind == "adam":
            betas = list(map(float, sec["opt_betas"].split(",")))
            optimizers[net] = optim.Adam(nns[net].parameters(), lr=lr, betas=betas, eps=float(sec["opt_eps"]),
                                         weight_decay=float(sec["opt_weight_decay"]),
                                         amsgrad=bool(strtobool(sec["opt_amsgrad"])))
        elif kind == "rmsprop":
            optimizers[net] = optim.RMSprop(nns[net].parameters(), lr=lr, momentum=float(sec["opt_momentum"]),
                                            alpha=float(sec["opt_alpha"]), eps=float(sec["opt_eps"]),
                                            centered=bool(strtobool(sec["opt_centered"])),
                                            weight_decay=float(sec["opt_weight_decay"]))
    return optimizers


def forward_model(fea_dict, lab_dict, arch_dict, model, nns, costs, inp, inp_out_dict, max_len, batch_size, to_do,
                  forward_outs):
    """utils.py:2296-2420 - evaluate the [model] lines on one batch.  ``inp`` is
    (T, B, feat+labels) for sequence chunks or (N, feat+labels); label columns are
    float and cast with .long() (utils.py:2352)."""
    outs = {}
    seq_inp = inp.dim() == 3
    for fea, spec in fea_dict.items():
        if len(spec) > 1:
            outs[fea] = inp[..., spec[5]:spec[6]]

    def labels(name):
        return inp[..., lab_dict[name][3]].reshape(-1).long()

    def flat(t):
        return t.reshape(max_len * batch_size, -1) if t.dim() == 3 else t

    for line in model:
        out_name, op, inp1, inp2 = _parse(line)
        last = to_do == "forward" and forward_outs and out_name == forward_outs[-1]
        if op == "compute":
            seq_arch = bool(arch_dict[inp1][2])
            if len(inp_out_dict[inp2]) > 1:  # an input feature stream
                x = inp[..., inp_out_dict[inp2][-3]:inp_out_dict[inp2][-2]]
                if seq_inp and not seq_arch:
                    x = x.reshape(max_len * batch_size, -1)
                if not seq_inp and seq_arch:
                    x = x.reshape(max_len, batch_size, -1)
            else:
                x = outs[inp2]
                if not seq_arch and x.dim() == 3:
                    twin = getattr(x, "_pk_twin", None)  # (the rows of the bf16 copy are already the flattened rows)
                    x = outs[inp2] = x.reshape(max_len * batch_size, -1)
                    if twin is not None:
                        x._pk_twin = (twin[0], twin[1], x._version)
                if seq_arch and x.dim() == 2:
                    x = outs[inp2] = x.reshape(max_len, batch_size, -1)
            outs[out_name] = nns[inp1](x)
        elif op == "cost_nll":
            if to_do != "forward":
                y, lab, cost = flat(outs[inp1]), labels(inp2), costs[out_name]
                fused = None
                if (F_.settings.fused_cost and type(cost) is nn.NLLLoss and cost.weight is None
                        and cost.reduction == "mean" and getattr(y, "_pk_head", None) is not None):
                    # perf-mode output layer: the cost goes straight behind the head's inputs (functional.HeadNllFn);
                    # the same pass counts the frame errors a cost_err line on the same pair asks for
                    fused = F_.head_nll(y, lab, cost.ignore_index)
                if fused is not None:
                    outs[out_name], stats = fused
                    y._pk_nll_stats = (inp2, stats)
                    F_.note_label_check(stats)
                else:
                    outs[out_name] = cost(y, lab)
        elif op == "cost_err":
            if to_do != "forward":
                y = flat(outs[inp1])
                stats = getattr(y, "_pk_nll_stats", None)
                if stats is not None and stats[0] == inp2:
                    outs[out_name] = stats[1][1]
                else:
                    pred = torch.max(y, dim=1)[1]
                    outs[out_name] = torch.mean((pred != labels(inp2)).float())
        elif op == "concatenate":
            outs[out_name] = torch.cat((outs[inp1], outs[inp2]), outs[inp1].dim() - 1)
        elif op == "mult":
            outs[out_name] = outs[inp1] * outs[inp2]
        elif op == "sum":
            outs[out_name] = outs[inp1] + outs[inp2]
        elif op == "mult_constant":
            # (x * 1.0 is x, value and gradient: every shipped recipe weights its monophone cost with 1.0 - two launches
            # per step, forward and backward, that a 0.3 ms MLP step can do without)
            outs[out_name] = outs[inp1] if float(inp2) == 1.0 else outs[inp1] * float(inp2)
        elif op == "sum_constant":
            outs[out_name] = outs[inp1] + float(inp2)
        elif op == "avg":
            outs[out_name] = (outs[inp1] + outs[inp2]) / 2
        elif op == "mse":
            outs[out_name] = torch.mean((outs[inp1] - outs[inp2]) ** 2)
        if last and op != "cost_nll" and op != "cost_err":
            break
    return outs
