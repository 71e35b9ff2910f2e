#!/usr/bin/env python
"""Training-throughput bench of the hot path (frames/sec), BASELINE.json contract.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one synthetic batch: forward_model
(Li-GRU 5x550 bidirectional + the 1938- and 48-way heads, the [model] section of
cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg), loss_final.backward(), gradient
all-reduce (N > 1) and the optimizer step - what run_nn times as
elapsed_time_chunk (core.py:567-701).  Inputs are resident in HBM before the
timed region.  N > 1: one process per GPU under torch.distributed.run, the batch
axis is sharded (weak scaling: B per GPU fixed), gradients all-reduced over RCCL.

Prints ONE JSON line on rank 0 (metric, roofline of the dominant kernel measured
live with HIP events, CPU baseline = the oracle timed on this host's cores).
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

PEAK = {"bf16": 2500.0, "fp32": 157.3}  # dense MFMA TFLOP/s, MI355X_MICROARCH.md
HBM_PEAK = 8000.0                      # GB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)   # 100 x ~20 ms: a timed region of ~2 s
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--recipe", default="timit_ligru")
    ap.add_argument("--T", type=int, default=500)
    ap.add_argument("--B", type=int, default=128, help="sequences per GPU (weak scaling)")
    ap.add_argument("--prec", default=os.environ.get("PK_PRECISION", "bf16"), choices=["bf16", "fp32"])
    ap.add_argument("--algo", default="auto", choices=["auto", "stepwise", "persistent"])
    ap.add_argument("--layers", type=int, default=None)
    ap.add_argument("--mask-rng", default=os.environ.get("PK_MASK_RNG", "device"), choices=["device", "reference", "reference_host"],
                    help="recurrent drop masks: the GPU RNG (the library's default), the reference's CPU torch.bernoulli "
                         "stream drawn on the device, or the reference's own call on the host")
    ap.add_argument("--no-overlap", dest="overlap", action="store_false",
                    help="N > 1: reduce every gradient bucket after backward instead of behind the layer that produced it")
    ap.set_defaults(overlap=True)
    ap.add_argument("--torch-optim", action="store_true", help="torch.optim instead of the fused flat optimizers")
    ap.add_argument("--graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the step as one HIP graph (auto: the launch-bound non-sequence recipes on one GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the parity_mode (fp32) and other_configs sub-records of the default single-GPU run")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--cpu-T", type=int, default=500, help="sequence length of the CPU baseline sample (the metric's: 500)")
    ap.add_argument("--cpu-B", type=int, default=16, help="batch of the CPU baseline sample (16 of the metric's 128 sequences)")
    ap.add_argument("--cpu-full-in-run", action="store_true", default=os.environ.get("PK_BENCH_CPU_FULL", "0") == "1",
                    help="additionally time ONE step of the CPU port at the metric's FULL shape inside this run (~4 minutes of "
                         "host time on the GPU box's 16 usable cores): cpu_baseline.full_shape is then measured_in_run = true. "
                         "Off by default - the default run's CPU leg stays a bounded sample of the workload; the round's "
                         "evidence pass (tools/gpu_evidence.sh) runs it and commits the line under profiles/")
    ap.add_argument("--cpu-full", action="store_true",
                    help="CPU baseline only: ONE step at the metric's full shape (T, B of --T / --B; minutes of CPU time), "
                         "printed as JSON - the source of profiles/r03_cpu_full_shape.json")
    ap.add_argument("--repeats", type=int, default=1, help="timed regions of --steps steps each; the median is reported")
    ap.add_argument("--unfused-cost", action="store_true",
                    help="cost lines through torch's NLLLoss on the log-posteriors (what the reference's own forward_model does)")
    ap.add_argument("--sync-every-step", action="store_true", help="read the loss back after every step (core.py:689)")
    ap.add_argument("--force-reducer", action="store_true",
                    help="one GPU: run the whole data-parallel path anyway - a ONE-rank RCCL communicator, the bucketed "
                         "reducer forced on (a one-rank sum is the identity: losses must equal the plain run bit for bit)")
    ap.add_argument("--dump-losses", default=None, help="write the loss of every timed step to this file (JSON list)")
    ap.add_argument("--only-forward-mode", action="store_true", help="print only the forward_mode sub-record (eval-mode forward)")
    ap.add_argument("--prewarm-s", type=float, default=float(os.environ.get("PK_BENCH_PREWARM_S", "-1")),
                    help="UNTIMED time-based pre-warm in front of the counted warm-up: steps of the same workload are enqueued "
                         "back to back (no host sync in between) until this many seconds have passed - clocks, caching "
                         "allocator, side streams and the host's run-ahead reach their steady state whatever --warmup is. "
                         "-1 = default (1.5 s); 0 = none.  Reported in the line as config.prewarm_s")
    ap.add_argument("--step-trace", action="store_true", default=bool(os.environ.get("PK_BENCH_STEP_TRACE")),
                    help="report every timed step's GPU time (HIP event per step) and host enqueue time, not only their summary")
    ap.add_argument("--launch-check", action="store_true",
                    help="rendezvous only: every rank joins, one all-reduce, rank 0 prints {\"n_gpus\": world} (tests)")
    return ap.parse_args()


class Trainer:
    """The run_nn inner loop (core.py:616-642) on the engine."""

    def __init__(self, args, rank, world):
        self.U = importlib.import_module("pytorch-kaldi_amd.utils")
        self.R = importlib.import_module("pytorch-kaldi_amd.recipes")
        self.F = importlib.import_module("pytorch-kaldi_amd.functional")
        self.DP = importlib.import_module("pytorch-kaldi_amd.dp")
        self.OPT = importlib.import_module("pytorch-kaldi_amd.optim")
        self.CORE = importlib.import_module("pytorch-kaldi_amd.core")
        F_ = importlib.import_module("pytorch-kaldi_amd.functional")
        F_.set_precision(args.prec)
        F_.set_rec_algo(args.algo)
        F_.set_mask_rng(args.mask_rng)
        F_.settings.fused_cost = not args.unfused_cost
        self.args, self.rank, self.world = args, rank, world
        self.rcp = rcp = self.R.recipe(args.recipe, n_lay=args.layers)
        self.inp_out_dict = {"fea": rcp["fea_dict"]["fea"][5:]}
        torch.manual_seed(2234)
        self.nns, self.costs = self.U.model_init(self.inp_out_dict, rcp["model"], rcp["cfg"], rcp["arch_dict"], True,
                                                 False, "train")
        if args.torch_optim:
            self.opts = self.U.optimizer_init(self.nns, rcp["cfg"], rcp["arch_dict"])
            flats = None
        else:
            self.opts = self.OPT.fused_optimizer_init(self.nns, rcp["cfg"], rcp["arch_dict"])
            for o in self.opts.values():
                o.zero_in_step = True  # (as core.run_nn_dp sets it: zero_grad() runs in front of every backward pass)
            flats = {k: o.flat for k, o in self.opts.items()}
        # 8 MB buckets: the recurrent stack's 31.7 MB of gradients leave in 4 pieces while BPTT of the lower layers runs;
        # the launch-bound recipes (a step shorter than its own exchange): 4 MB buckets and the bf16 wire, as core.make_reducer
        self.overlap = bool(args.overlap)
        self.reducer = self.DP.GradReducer(self.nns, flats=flats, bucket_bytes=(8 << 20) if rcp["seq"] else (4 << 20),
                                           overlap=self.overlap, force=bool(getattr(args, "force_reducer", False)),
                                           wire=self.CORE.default_wire(not rcp["seq"]))
        # one resident synthetic batch per rank (different seeds per rank = different shards)
        self.T, self.B = (args.T, args.B) if rcp["seq"] else (1, args.B)
        self.batches = [self.R.synthetic_batch(rcp, self.T, self.B, 4234 + 17 * rank + i, "cuda") for i in range(2)]
        self.n_params = sum(p.numel() for n in self.nns.values() for p in n.parameters())
        self.graphed = None
        self.fence = self.F.StepFence()  # as core.run_nn_dp: eager steps, at most PK_STEPS_IN_FLIGHT ahead of the GPU

    def step(self, i):
        inp = self.batches[i % len(self.batches)]
        if self.graphed is not None:
            return self.graphed(inp)
        out = self.step_on(inp)
        self.fence()
        return out

    def enable_graph(self):
        """Launch-bound recipes: replay the whole step as one HIP graph (pytorch-kaldi_amd/graphs.py).  Call after
        a few eager steps."""
        G = importlib.import_module("pytorch-kaldi_amd.graphs")
        self.graphed = G.GraphedStep(self.step_on, list(self.opts.values())).capture(self.batches[0])

    def step_on(self, inp):
        rcp = self.rcp
        with self.F.accumulating_backward():  # (a training step, as in core.run_nn: kernels may add to the flat .grad themselves)
            outs = self.U.forward_model(rcp["fea_dict"], rcp["lab_dict"], rcp["arch_dict"], rcp["model"], self.nns,
                                        self.costs, inp, self.inp_out_dict, self.T, self.B, "train", [])
            for o in self.opts.values():
                o.zero_grad()
            outs["loss_final"].backward()
        self.reducer.finish()
        for o in self.opts.values():
            o.step()
        if self.args.sync_every_step:  # the reference's progress bar reads the running loss once per batch (core.py:689)
            return float(outs["loss_final"].detach().cpu())
        return outs["loss_final"].detach()


def algorithmic_flops(rcp, T, B):
    """fwd+bwd FLOPs per step from the layer shapes: fwd + dX + dW for every
    GEMM, no dX for the first layer's input projection."""
    cfg = rcp["cfg"]
    a1 = cfg["architecture1"]
    frames = T * B
    total = 0.0
    rec_flops = 0.0
    if rcp["seq"]:
        pre = {"liGRU": "ligru", "LSTM": "lstm", "GRU": "gru"}[a1["arch_class"]]
        G = {"liGRU": 2, "LSTM": 4, "GRU": 3}[a1["arch_class"]]
        lay = [int(v) for v in a1[pre + "_lay"].split(",")]
        din = rcp["nfea"]
        for i, H in enumerate(lay):
            proj = 2.0 * frames * din * G * H            # x2 directions share weights: rows are not duplicated
            rec = 2.0 * (2 * frames) * H * G * H         # both directions
            total += proj * (2 if i == 0 else 3) + rec * 3
            rec_flops += rec * 3
            din = 2 * H
        feat = din
    elif a1["arch_class"] == "SincNet":
        # conv FLOPs of the four blocks (194.26 MFLOP/frame fwd; conv1 has no dX) + the MLP trunk
        total += frames * (101.45e6 * 2 + (78.34e6 + 12.10e6 + 2.38e6) * 3)
        din = 3300
        for H in [int(v) for v in rcp["cfg"][rcp["trunk"]]["dnn_lay"].split(",")]:
            total += 2.0 * frames * din * H * 3
            din = H
        feat = din
    else:
        lay = [int(v) for v in a1["dnn_lay"].split(",")]
        din = rcp["nfea"]
        for i, H in enumerate(lay):
            total += 2.0 * frames * din * H * (2 if i == 0 else 3)
            din = H
        feat = din
    heads = rcp["n_cd"] + rcp["n_mono"]
    total += 2.0 * frames * feat * heads * 3
    return total, rec_flops


def rec_launch_bytes(rcp, T, B, entry):
    """Algorithmic HBM bytes of one persistent recurrent launch (bidirectional, fp32 tensors, bf16 exchange).  The two-phase
    cells (GRU / minimalGRU, pk_rec2p_*: pk_rec_persist2_gru.hip) save their G gate tensors and publish a second bf16
    exchange buffer (Xb: r*h / z*h, the k-major operand of the dU_h GEMM) in the forward pass."""
    a1 = rcp["cfg"]["architecture1"]
    kind = a1["arch_class"]
    two_phase = kind in ("GRU", "minimalGRU")
    if kind not in ("liGRU", "LSTM", "RNN", "GRU", "minimalGRU"):
        return None
    if entry not in (("pk_rec2p_fwd_bf16", "pk_rec2p_bwd_bf16") if two_phase else ("pk_rec_fwd_bf16", "pk_rec_bwd_bf16")):
        return None
    G = {"liGRU": 2, "LSTM": 4, "RNN": 1, "GRU": 3, "minimalGRU": 2}[kind]
    NS = {"liGRU": 2, "LSTM": 5, "RNN": 1, "GRU": 3, "minimalGRU": 2}[kind]  # fp32 tensors saved per direction
    H = int(a1[{"liGRU": "ligru", "LSTM": "lstm", "RNN": "rnn", "GRU": "gru", "minimalGRU": "minimalgru"}[kind] + "_lay"].split(",")[0])
    Hp = (H + 7) // 8 * 8
    rows = T * B
    if "fwd" in entry:  # read P; write Y, S (both directions) and the bf16 copy Yb (+ Xb)
        return rows * G * H * 4 + rows * 2 * H * 4 + 2 * rows * NS * H * 4 + rows * 2 * Hp * 2 * (2 if two_phase else 1)
    # read S, Y (h_{t-1}) and dY (+ the bf16 r*h / z*h of the two-phase cells); write the bf16 gate gradients of both directions
    return 2 * rows * NS * H * 4 + 2 * rows * 2 * H * 4 + 2 * rows * G * Hp * 2 + (rows * 2 * Hp * 2 if two_phase else 0)


def profile_entry_points(tr, steps=2):
    """HIP-event timing of every C-ABI call of a few steps (events recorded on the stream the
    kernels are launched on = torch's current stream)."""
    _lib = importlib.import_module("pytorch-kaldi_amd._lib")
    F_ = importlib.import_module("pytorch-kaldi_amd.functional")
    prof = _lib.Profiler()
    side = F_.settings.wgrad_side
    F_.settings.wgrad_side = False  # one stream: a launch's events then bracket that launch alone (no co-running GEMMs)
    try:
        with prof:
            for i in range(steps):
                tr.step_on(tr.batches[i % len(tr.batches)])  # eager even when the timed region replays a HIP graph
            torch.cuda.synchronize()
    finally:
        F_.settings.wgrad_side = side
    return prof.summary(steps)


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def log(msg):
    if os.environ.get("PK_BENCH_VERBOSE"):
        print("[bench %.1fs] %s" % (time.time() - T_START, msg), file=sys.stderr, flush=True)


T_START = time.time()


def reference_root():
    """The PyTorch-Kaldi checkout when this host has one (PK_REFERENCE, /root/reference): the build container does, the
    GPU boxes do not."""
    for cand in (os.environ.get("PK_REFERENCE"), "/root/reference"):
        if cand and os.path.isfile(os.path.join(cand, "neural_networks.py")):
            return cand
    return None


def _cpu_model(rcp, kind_of_baseline):
    """The recipe's networks on the CPU: `reference` = the reference's own classes (neural_networks.py, imported
    unmodified), `port` = this repository's torch-CPU oracle on parameter dicts.  -> (forward(x) -> log-posterior list,
    leaves per cfg section)."""
    cfg = rcp["cfg"]
    a1 = cfg["architecture1"]
    kind = a1["arch_class"]
    trunk, s_cd, s_mono = rcp["trunk"], rcp["head_cd"], rcp["head_mono"]
    torch.manual_seed(2234)
    if kind_of_baseline == "reference":
        sys.path.insert(0, reference_root())
        import neural_networks as NN  # the reference's module, as run_exp.py imports it

        def make(sec, cls, din):
            # (the SectionProxy itself, with the two fields utils.model_init injects, utils.py:2051-2052: its lookups
            # are case-insensitive - `options["sinc_N_filt"]` - which a plain dict's are not)
            cfg[sec]["use_cuda"], cfg[sec]["to_do"] = "False", "train"
            return getattr(NN, cls)(cfg[sec], din)
    else:
        nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")

        def make(sec, cls, din):
            return getattr(nn_amd, cls)(dict(cfg[sec], use_cuda="False", to_do="train"), din)
    nets = {"architecture1": make("architecture1", kind, rcp["nfea"])}
    feat = nets["architecture1"].out_dim
    if trunk:  # SincNet recipe: an MLP trunk between the front-end and the heads
        nets[trunk] = make(trunk, "MLP", feat)
        feat = nets[trunk].out_dim
    nets[s_cd] = make(s_cd, "MLP", feat)
    if s_mono:
        nets[s_mono] = make(s_mono, "MLP", feat)
    if kind_of_baseline == "reference":
        for n in nets.values():
            n.train()
        leaves = {k: [p for p in net.parameters() if p.requires_grad] for k, net in nets.items()}

        def forward(x):
            h = nets["architecture1"](x)
            if rcp["seq"]:
                h = h.view(h.shape[0] * h.shape[1], -1)  # utils.py:2334
            if trunk:
                h = nets[trunk](h)
            return [nets[s_cd](h)] + ([nets[s_mono](h)] if s_mono else [])
        return forward, leaves
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pk_oracle as O

    sds = {k: {n: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in n)
               for n, v in net.state_dict().items()} for k, net in nets.items()}
    leaves = {k: [v for v in sd.values() if v.requires_grad] for k, sd in sds.items()}

    def forward(x):
        if rcp["seq"]:
            # index_like_reference: the projections are indexed inside the time loop as the reference does it
            # (neural_networks.py:1133-1134) - that is what makes its backward O(T^2), and with it the port runs
            # within 4-11 % of the reference's own speed (profiles/r02_cpu_port_vs_reference.json)
            h = O.recurrent_forward(kind, dict(a1), sds["architecture1"], x, index_like_reference=True)
            h = h.reshape(h.shape[0] * h.shape[1], -1)
        else:
            h = O.arch_forward(kind, dict(a1), sds["architecture1"], x)
            if trunk:
                h = O.mlp_forward(dict(cfg[trunk]), sds[trunk], h)
        return [O.mlp_forward(dict(cfg[s_cd]), sds[s_cd], h)] + ([O.mlp_forward(dict(cfg[s_mono]), sds[s_mono], h)] if s_mono else [])
    return forward, leaves


def cpu_baseline(args, rcp_name, full=False, budget_s=None, shape=None):
    """The reference path on THIS host's cores, in the same run (north_star): fwd + bwd + the cfg's optimizers of the same
    network on a bounded sample of the same workload.  kind = "reference": the reference's own classes
    (neural_networks.py imported unmodified - wherever a checkout exists: PK_REFERENCE, /root/reference); kind = "port":
    this repository's torch-CPU oracle, the stand-in on the GPU boxes (they carry no checkout; the port runs 1.04-1.11 x
    FASTER than the reference's classes, profiles/r02_cpu_port_vs_reference.json).

    The sample.  The reference's autograd cost per frame GROWS with T (its backward zero-fills a (T, 2B, H) tensor per
    step and gate; the port indexes the projections inside the time loop the same way), so a short
    sequence flatters the CPU (round 5: 4.7 x at T = 100).  The default sample of a sequence recipe therefore keeps the
    metric's sequence length T and cuts the BATCH (16 of 128 sequences: the zero-fill and the GEMMs both scale with the
    rows, so frames/s moves little); `full_shape` is one step at the metric's own (T, B) - minutes of CPU time -
    measured in the run with --cpu-full-in-run, otherwise quoted from the round's record with measured_in_run = false."""
    R = importlib.import_module("pytorch-kaldi_amd.recipes")
    rcp = R.recipe(rcp_name, n_lay=args.layers)
    cfg = rcp["cfg"]
    cores = usable_cores()
    torch.set_num_threads(cores)
    kind_of_baseline = "reference" if reference_root() and os.environ.get("PK_CPU_BASELINE", "") != "port" else "port"
    forward, leaves = _cpu_model(rcp, kind_of_baseline)
    opts = []
    for k, lv in leaves.items():  # one optimizer per architecture, as the shipped cfg sets them (run_nn: optimizer_init)
        if cfg[k]["arch_opt"] == "sgd":
            opts.append(torch.optim.SGD(lv, lr=float(cfg[k]["arch_lr"])))
        else:
            opts.append(torch.optim.RMSprop(lv, lr=float(cfg[k]["arch_lr"]), alpha=0.95, eps=1e-8))
    if full:
        T, B = (args.T, args.B) if rcp["seq"] else (1, args.B)
    elif shape is not None:
        T, B = shape
    else:
        T, B = (args.cpu_T, args.cpu_B) if rcp["seq"] else (1, 128)
    budget_s = args.cpu_budget_s if budget_s is None else budget_s

    def one_step(seed, T_=None):
        Ts = T_ or T
        inp = R.synthetic_batch(rcp, Ts, B, seed)
        x = inp[..., :rcp["nfea"]]
        outs = forward(x)
        loss = torch.nn.functional.nll_loss(outs[0], inp[..., rcp["nfea"]].reshape(-1).long())
        if len(outs) > 1:
            loss = loss + torch.nn.functional.nll_loss(outs[1], inp[..., rcp["nfea"] + 1].reshape(-1).long())
        for opt in opts:
            opt.zero_grad()
        loss.backward()
        for opt in opts:
            opt.step()

    one_step(1, 8 if rcp["seq"] else None)  # warm-up (thread pool, allocator) on a short sequence
    t0 = time.time()
    n = 0
    while True:
        one_step(2 + n)
        n += 1
        if full or time.time() - t0 > budget_s or n >= 50:
            break
    dt = time.time() - t0
    model = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    what = ("the reference's own classes (neural_networks.py, unmodified)" if kind_of_baseline == "reference" else
            "torch-CPU port of the reference path (oracle/pk_oracle.py, projections indexed in the time loop like the "
            "reference; no PyTorch-Kaldi checkout on this host - the reference's classes run 1.04-1.11 x slower than the "
            "port, profiles/r02_cpu_port_vs_reference.json)")
    rec = {"value": round(n * T * B / dt, 2), "unit": "frames/s", "cores": cores, "kind": kind_of_baseline,
           "sample": "%d step(s) of the same network (fwd+bwd+the cfg optimizers) at T=%d, B=%d of the metric's T=%d, B=%d on "
                     "%s, fp32, %d threads: %s" % (n, T, B, args.T if rcp["seq"] else 1, args.B, model or "host CPU", cores, what),
           "T": T, "B": B, "seconds": round(dt, 2)}
    if not full and shape is None and getattr(args, "cpu_full_in_run", False) and rcp["seq"]:
        fsr = cpu_baseline(args, rcp_name, full=True)  # ONE step at the metric's full (T, B): minutes
        rec["full_shape"] = {k: fsr[k] for k in ("value", "unit", "cores", "kind", "T", "B", "seconds", "sample") if k in fsr}
        rec["full_shape"]["measured_in_run"] = True
    elif not full and shape is None:
        for fsrc in ("r06_cpu_full_shape.json", "r05_cpu_full_shape.json", "r03_cpu_full_shape.json"):
            try:  # the same path at the metric's FULL shape, measured once per round on a GPU box's host (bench.py --cpu-full)
                fs = json.load(open(os.path.join(ROOT, "profiles", fsrc)))
            except (OSError, ValueError):
                continue
            if fs.get("recipe", "timit_ligru") == rcp_name:
                rec["full_shape"] = {k: fs[k] for k in ("value", "unit", "cores", "kind", "T", "B", "seconds", "sample") if k in fs}
                rec["full_shape"]["measured_in_run"] = False  # a constant quoted from the file below, NOT timed by this run
                rec["full_shape"]["source"] = "profiles/%s (one step at the full shape on a GPU box's host)" % fsrc
                break
    return rec


def roofline_of(tr, args, summ, prec):
    """Roofline record of the dominant entry point (HIP events, profile_entry_points)."""
    total_flops, rec_flops = algorithmic_flops(tr.rcp, tr.T, tr.B)
    dom = max(summ, key=lambda k: summ[k]["ms_per_step"]) if summ else None
    roof = {"bound": "mfma", "achieved": None, "peak": PEAK[prec], "unit": "TFLOP/s", "frac": None, "traffic": None}
    if dom is None:
        return roof, total_flops
    d = summ[dom]
    if dom in ("pk_rec_fwd", "pk_rec_bwd", "pk_rec_fwd_bf16", "pk_rec_bwd_bf16", "pk_rec2p_fwd_bf16", "pk_rec2p_bwd_bf16"):
        # recurrent launches: fwd step GEMM = 1/3 of rec_flops, bwd carry GEMM = 1/3, deferred dU = 1/3
        # (inside pk_rec_bwd in the fp32 library path, a separate pk_gemm_bf16 in the perf pipeline)
        share = 2.0 if dom == "pk_rec_bwd" else 1.0
        fl = rec_flops * share / 3.0 / d["calls_per_step"]
    elif dom in ("pk_gemm", "pk_gemm_bf16"):
        conv = tr.T * tr.B * (101.45e6 * 2 + 92.82e6 * 3) if tr.rcp["nfea"] == 3200 else 0.0
        dU = rec_flops / 3.0 if dom == "pk_gemm_bf16" else 0.0   # deferred dU GEMMs of the perf pipeline
        fl = (total_flops - rec_flops - conv + dU) / d["calls_per_step"]
    elif dom in ("pk_conv1d_pool_fwd", "pk_conv1d_pool_bwd"):
        # fp32 direct convolution = packed-FMA VALU work (the fp32 MFMA rate is the same 157 TFLOP/s)
        per_frame = 194.27e6 if dom.endswith("fwd") else 194.27e6 + 92.82e6 + 101.45e6
        fl = tr.T * tr.B * per_frame / d["calls_per_step"]
        roof.update({"bound": "valu-fp32", "peak": 157.3})
    else:
        fl = 0.0
    ach = fl / (d["avg_ms"] * 1e-3) / 1e12 if d["avg_ms"] > 0 else 0.0
    roof.update({"kernel": dom, "achieved": round(ach, 3), "frac": round(ach / roof["peak"], 5),
                 "avg_launch_ms": round(d["avg_ms"], 4), "launches_per_step": d["calls_per_step"], "flops_per_launch": fl})
    # algorithmic HBM bytes of one recurrent launch: which roof is the NEARER one follows from the
    # launch's arithmetic intensity against the ridge (peak FLOP/s / peak B/s = 312 FLOP/B in bf16)
    rb = rec_launch_bytes(tr.rcp, tr.T, tr.B, dom)
    if rb:
        gbps = rb / (d["avg_ms"] * 1e-3) / 1e9
        roof["algorithmic_bytes_per_launch"] = rb
        roof["arithmetic_intensity"] = round(fl / rb, 1)
        roof["ridge"] = round(roof["peak"] * 1e12 / (HBM_PEAK * 1e9), 1)
        roof["mfma"] = {"achieved": round(ach, 3), "peak": roof["peak"], "unit": "TFLOP/s", "frac": round(ach / roof["peak"], 5)}
        roof["hbm"] = {"achieved": round(gbps, 1), "peak": HBM_PEAK, "unit": "GB/s", "frac": round(gbps / HBM_PEAK, 5)}
        if fl / rb < roof["ridge"]:  # below the ridge: the HBM roof is the one this launch would hit first
            roof.update({"bound": "hbm", "achieved": round(gbps, 1), "peak": HBM_PEAK, "unit": "GB/s",
                         "frac": round(gbps / HBM_PEAK, 5)})
    if "rec" in dom and tr.rcp["seq"]:
        # A recurrent launch is T strictly dependent steps: neither roofline binds it, the per-step latency does.
        # `hop_floor_us` = one cross-CU hand-off (publish of the slowest member -> data in registers), measured with this
        # library's own traced kernels (profiles/r0N_rec_step_floor.json, tools/trace_rec2.py; the chip's handoff-1to1
        # price is 0.8-1.0 us).  (Rounds 2-5 also printed a `structure_frac` against an "empty step" taken from the TRACED
        # kernel, whose s_memtime stamps cost ~11 %: the production step beat that floor - it is gone.)
        lat = {"bound": "latency", "dependent_steps_per_launch": tr.T, "us_per_step": round(d["avg_ms"] * 1e3 / tr.T, 3)}
        hop, src_ = 0.9, None
        for cand in ("r06_rec_step_floor.json", "r05_rec_step_floor.json", "r04_rec_step_floor.json", "r03_rec_step_floor.json", "r02_rec_step_floor.json"):
            try:
                fl_ = json.load(open(os.path.join(ROOT, "profiles", cand)))
                hop, src_ = float(fl_["hop_us"]["bwd" if "bwd" in dom else "fwd"]), cand
                break
            except (OSError, ValueError, KeyError):
                pass
        lat["hop_floor_us"] = hop
        lat["frac"] = round(hop / lat["us_per_step"], 4)  # 1.0 = every step costs exactly one cross-CU hop
        if src_:
            lat["hop_source"] = "profiles/" + src_
        roof["latency"] = lat
        # (kept at the top level too: earlier rounds' readers look for them there)
        roof["dependent_steps_per_launch"], roof["us_per_step"] = tr.T, lat["us_per_step"]
        roof["hop_floor_us"], roof["latency_frac"] = hop, lat["frac"]
    # HBM bytes per launch from the PMC passes of the round's evidence run (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in
    # separate passes, FETCH doubled per MI355X_MICROARCH.md; tools/pmc_summaries.py): headline file, or <round>_pmc_traffic_<recipe>.json
    names = ["r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json",
             "r01_pmc_traffic.json"] if args.recipe == "timit_ligru" else ["r06_pmc_traffic_%s.json" % args.recipe]
    for src in names:
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", src)))
            if dom in pm and (tr.T, tr.B) == ((500, 128) if tr.rcp["seq"] else (1, 128)) and args.layers is None and prec == "bf16":
                roof["traffic"] = pm[dom]["traffic_bytes"]
                roof["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)" % src
                break
        except (OSError, ValueError):
            pass
    return roof, total_flops


def workload_name(recipe, tr):
    return ("%s: %s, T=%d, B=%d per GPU, %s, %d+%d senone/phone heads, fwd+bwd+optimizer"
            % (recipe, tr.rcp["cfg"]["architecture1"]["arch_class"], tr.T, tr.B,
               "3200-sample raw waveform chunks" if tr.rcp["nfea"] == 3200 else "%d-dim features" % tr.rcp["nfea"],
               tr.rcp["n_cd"], tr.rcp["n_mono"]))


def measure(args, rank, world, steps, warmup):
    """One configuration: build, warm up, time exactly `steps` steps between barriers, profile two more steps.
    Returns (record, Trainer)."""
    import torch.distributed as dist

    tr = Trainer(args, rank, world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    log("%s/%s built" % (args.recipe, args.prec))
    for i in range(warmup):
        tr.step(i)
        torch.cuda.synchronize()
    # (several ranks: the bucketed all-reduces are captured with the step - the reducer has placed them during the eager
    # warm-up steps; RCCL kernels are capturable)
    use_graph = args.graph == "on" or (args.graph == "auto" and not tr.rcp["seq"] and not args.torch_optim)
    if use_graph and tr.reducer.active and warmup < 2:
        for i in range(2 - warmup):  # the reducer learns which gradients arrive per step in its first step
            tr.step(i)
    if use_graph:
        if warmup == 0:
            tr.step(0)  # lazy one-time initialisation must not land inside the capture
        tr.enable_graph()
        tr.step(0)      # first replay (graph upload) outside the timed region
    # Untimed, time-based pre-warm, behind the counted warm-up and the graph capture: the driver's command is a FRESH process on a fresh box with
    # --warmup 5, i.e. 0.1 s of GPU work before the clock starts.  The steps are enqueued without host syncs so that
    # the device sees the same back-to-back stream of launches the timed region produces.
    prewarm_s = getattr(args, "prewarm_s", -1.0)
    prewarm_s = 1.5 if prewarm_s < 0 else prewarm_s
    n_pre = 0
    if prewarm_s > 0:
        t_pre = time.perf_counter()
        go = True
        while go:
            for _ in range(4):
                tr.step(n_pre)
                n_pre += 1
            torch.cuda.synchronize()  # bounds the host's run-ahead (each step holds GBs of activations alive)
            el = time.perf_counter() - t_pre
            if world > 1:  # every rank must run the same number of steps (each one holds collectives)
                t = torch.tensor([el], device="cuda", dtype=torch.float64)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                el = float(t)
            go = el < prewarm_s
        log("pre-warm: %d steps in %.2f s" % (n_pre, time.perf_counter() - t_pre))
    regions = []
    loss = None
    losses = [] if getattr(args, "dump_losses", None) else None
    step_events, host_marks = [], []
    for _ in range(max(1, getattr(args, "repeats", 1))):  # every region: EXACTLY `steps` steps between two barriers
        barrier()
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
        evs, marks = [ev0], [time.perf_counter()]
        t0 = time.perf_counter()
        for i in range(steps):
            loss = tr.step(i)
            if losses is not None:
                losses.append(loss)
            if steps <= 512:  # one event record per step (~2 us of host time): where inside the region the time goes
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
                marks.append(time.perf_counter())
        barrier()
        dt = time.perf_counter() - t0
        step_events.append(evs)
        host_marks.append(marks)
        if world > 1:
            t = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        regions.append(dt)
    dt = sorted(regions)[len(regions) // 2]  # the median region (one region: that one)
    log("%s/%s timed region(s) done: %s s" % (args.recipe, args.prec, ["%.3f" % r for r in regions]))
    _lib = importlib.import_module("pytorch-kaldi_amd._lib")
    _lib.raise_if_persist_failed()
    frames = steps * tr.T * tr.B * world
    ms_per_step = 1e3 * dt / steps
    out = {
        "metric": "train_frames_per_sec", "value": round(frames / dt, 1), "unit": "frames/s", "n_gpus": world,
        "steps": steps, "warmup": warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.prec, "data": "synthetic",
        "config": {"workload": workload_name(args.recipe, tr),
                   "global_batch": tr.B * world, "seq_len": tr.T, "parallelism": "dp%d" % world,
                   "rec_algo": args.algo, "mask_rng": args.mask_rng,
                   "allreduce": "overlapped" if tr.overlap else "after-backward",
                   "optimizer": "torch" if args.torch_optim else "fused-flat", "hip_graph": bool(use_graph),
                   "params": tr.n_params},
        "loss_final": round(float(loss), 5),
    }
    out["config"]["prewarm_s"] = prewarm_s
    out["config"]["prewarm_steps"] = n_pre
    mid = sorted(range(len(regions)), key=lambda k: regions[k])[len(regions) // 2]
    if len(step_events[mid]) > 1:
        evs, marks = step_events[mid], host_marks[mid]
        gpu = [evs[k].elapsed_time(evs[k + 1]) for k in range(len(evs) - 1)]       # ms between the steps' last launches
        host = [1e3 * (marks[k + 1] - marks[k]) for k in range(len(marks) - 1)]    # ms the host took to ENQUEUE each step
        sg = sorted(gpu)
        out["step_ms"] = {"first": round(gpu[0], 3), "median": round(sg[len(sg) // 2], 3), "max": round(sg[-1], 3),
                          "min": round(sg[0], 3), "host_enqueue_median": round(sorted(host)[len(host) // 2], 3),
                          "note": "per-step GPU time from one HIP event per step on the launch stream; host_enqueue = wall "
                                  "time the host needed to enqueue one step"}
        if getattr(args, "step_trace", False):
            out["step_ms"]["gpu"] = [round(v, 3) for v in gpu]
            out["step_ms"]["host_enqueue"] = [round(v, 3) for v in host]
    if losses is not None and rank == 0:
        with open(args.dump_losses, "w") as f:
            json.dump([float(v) for v in losses], f)
    if tr.reducer.active:
        out["config"]["reducer"] = "%d buckets, %s wire, %s%s" % (len(tr.reducer.buckets), tr.reducer.wire,
                                                                 "forced on one rank" if world == 1 else "RCCL",
                                                                 ", inside the step's HIP graph" if tr.graphed is not None else "")
        if tr.reducer.trace and rank == 0:
            out["allreduce_timeline"] = tr.reducer.timeline()[-3:]
    if len(regions) > 1:
        out["regions_ms_per_step"] = [round(1e3 * r / steps, 3) for r in regions]
        out["config"]["timing"] = "median of %d regions of %d steps" % (len(regions), steps)
        # (a median hides a region that ran five times slower - round 5's host-lead stalls sat in these lists unnoticed)
        out["regions_max_over_min"] = round(max(regions) / max(min(regions), 1e-9), 3)
    try:  # the allocator's own account of the run: retries / device frees during training steps mean the pool hit the HBM's end
        ms_ = torch.cuda.memory_stats()
        out["allocator"] = {"alloc_retries": int(ms_.get("num_alloc_retries", 0)), "device_frees": int(ms_.get("num_device_free", 0)),
                            "reserved_gb_peak": round(ms_.get("reserved_bytes.all.peak", 0) / 2**30, 2),
                            "steps_in_flight": tr.fence.depth}
    except Exception:  # noqa: BLE001 (no GPU: the CPU legs of the contract tests)
        pass
    if args.unfused_cost or args.sync_every_step:
        out["config"]["cost"] = "nn.NLLLoss on the log-posteriors" if args.unfused_cost else "fused"
        out["config"]["host_sync"] = "every step" if args.sync_every_step else "end of region"
    if tr.rcp["cfg"]["architecture1"]["arch_class"] == "LSTM":
        out["config"]["lstm_waves"] = int(_lib.load().pk_persist2_get_lstm_waves())  # DESIGN.md 6.1 (PK_EXPERIMENT lstm_waves)
    # roofline of the dominant kernel class, measured live with HIP events (every rank runs the two extra steps:
    # they contain the gradient all-reduce)
    summ = profile_entry_points(tr)
    if rank == 0:
        roof, total_flops = roofline_of(tr, args, summ, args.prec)
        out["roofline"] = roof
        out["entry_points_ms_per_step"] = {k: round(v["ms_per_step"], 3) for k, v in
                                           sorted(summ.items(), key=lambda kv: -kv[1]["ms_per_step"])}
        out["whole_step_tflops"] = round(total_flops / (ms_per_step * 1e-3) / 1e12, 3)
    return out, tr


def release(tr):
    """Drop a configuration's tensors before the next one is built (7-9 GB of activations per sequence recipe)."""
    tr.nns = tr.opts = tr.batches = tr.graphed = tr.reducer = None
    import gc
    gc.collect()
    torch.cuda.empty_cache()


# the other BASELINE.json configurations (parity-test cases, reported beside the headline): recipe, steps, warmup
# (each: median of 3 timed regions of `steps` steps)
OTHER_CONFIGS = [("timit_mlp", 400, 5), ("timit_lstm", 50, 5), ("libri_gru", 50, 5), ("timit_sincnet", 100, 5)]


def through_run_nn(args, n_batches=12, reps=3, ragged=False):
    """The same workload THROUGH the chunk loop (`elapsed_time_chunk`, core.py:567-701): a synthetic
    chunk of n_batches x B sentences of T frames resident in HBM, handed to pytorch-kaldi_amd.core.run_nn_dp with a
    chunk cfg on disk - batch assembly (the zero-padding gather), forward_model, backward, fused optimizers, ONE host
    sync per chunk, checkpoint + .info written afterwards.  `reps` chunks: the first warms up, the MEDIAN of the others is
    reported, each from its own .info file (`elapsed_time_chunk`, which brackets the batch loop exactly as the reference's
    does; every chunk builds its networks, flat buffers and optimizers anew, as the reference's chunk function does)."""
    import configparser
    import tempfile

    import numpy as np

    core = importlib.import_module("pytorch-kaldi_amd.core")
    R = importlib.import_module("pytorch-kaldi_amd.recipes")
    F_ = importlib.import_module("pytorch-kaldi_amd.functional")
    F_.set_precision(args.prec)
    rcp = R.recipe(args.recipe, n_lay=args.layers)
    T, B = args.T, args.B
    n_snt = n_batches * B
    g = torch.Generator().manual_seed(99)
    nlab = 2 if rcp["n_mono"] else 1
    if ragged:
        # ragged=True: sentence lengths U[T/2, T], sorted ascending like the reference's chunk lists (data_io sorts a chunk
        # by length, so a batch's sentences are close in length); every batch is padded to ITS longest sentence, each
        # sentence behind a random number of leading zeros (core.py:581-598)
        lens = np.sort(np.random.RandomState(7).randint(T // 2, T + 1, size=n_snt)).astype(np.int64)
    else:
        lens = np.full(n_snt, T, dtype=np.int64)  # equal-length sentences: the metric's (T, B) batch every time
    n_rows = int(lens.sum())
    data = torch.randn(n_rows, rcp["nfea"] + nlab, generator=g)
    data[:, rcp["nfea"]] = torch.randint(0, rcp["n_cd"], (n_rows,), generator=g).float()
    if rcp["n_mono"]:
        data[:, rcp["nfea"] + 1] = torch.randint(0, rcp["n_mono"], (n_rows,), generator=g).float()
    end = np.cumsum(lens)
    names = ["utt%05d" % i for i in range(n_snt)]
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
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

        def reader(cfg_file, is_production, shared_list, output_folder):
            return None  # no next chunk

        fea_dict = {k: list(v) for k, v in rcp["fea_dict"].items()}
        times = []
        for rep in range(max(2, reps)):
            core.run_nn_dp(names, data.cuda(), end, {k: list(v) for k, v in fea_dict.items()}, rcp["lab_dict"],
                           rcp["arch_dict"], path, False, path, reader=reader)
            info = configparser.ConfigParser()
            info.read(os.path.join(tmp, "chunk.info"))
            times.append(float(info["results"]["elapsed_time_chunk"]))
        dt = sorted(times[1:])[(len(times) - 1) // 2]
        padded = int(sum(int(lens[i * B:(i + 1) * B].max()) * B for i in range(n_batches)))  # rows the kernels process
        out = {"value": round(n_rows / dt, 1), "unit": "frames/s", "ms_per_step": round(1e3 * dt / n_batches, 3),
               "batches": n_batches, "elapsed_time_chunk_s": round(dt, 4), "warmup_chunk_s": round(times[0], 4),
               "chunks_s": [round(t, 4) for t in times],
               "note": "pytorch-kaldi_amd.core.run_nn_dp on a resident synthetic chunk (%d sentences, %s): batch "
                       "assembly + forward_model + backward + fused optimizers, one host sync per chunk; the time is the "
                       ".info file's elapsed_time_chunk (core.py:567, 701); value counts the sentences' own frames"
                       % (n_snt, "lengths U[%d, %d] sorted, zero-padded per batch with random left offsets" % (T // 2, T)
                          if ragged else "%d frames each" % T)}
        if ragged:
            out["padded_frames_per_s"] = round(padded / dt, 1)
            out["padding_share"] = round(1.0 - n_rows / padded, 4)
    return out


def self_launch(args):
    """`python bench.py --gpus N` with no rendezvous in the environment: start N ranks of this very command under
    torch.distributed.run (one process per GPU, RCCL) and pass their exit code on.  Under the driver's own
    `python -m torch.distributed.run ... bench.py --gpus N` WORLD_SIZE is set and this is skipped."""
    import socket
    import subprocess

    if torch.cuda.is_available() and torch.cuda.device_count() < args.gpus and not args.launch_check:
        sys.stderr.write("bench.py: --gpus %d asked for, %d GPU(s) visible\n" % (args.gpus, torch.cuda.device_count()))
        sys.exit(2)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


def launch_check(rank, world):
    """Rendezvous test (CPU: gloo): every rank contributes its rank + 1 to one all-reduce."""
    import torch.distributed as dist

    t = torch.tensor([float(rank + 1)], device="cuda" if torch.cuda.is_available() else "cpu")
    if world > 1:
        dist.all_reduce(t)
    if rank == 0:
        print(json.dumps({"n_gpus": world, "launch_check": True, "sum_of_ranks_plus_1": float(t)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def forward_mode(args, steps=30, warmup=3):
    """The validation / forward chunks of the same recipe (core.py:616-642 with to_do = valid): module.eval(),
    torch.no_grad(), forward_model on the same resident batch - running BatchNorm statistics folded into the projection
    scale / shift, the (1 - p) dropout scalar, no saved tensors, loss and error computed, nothing differentiated."""
    tr = Trainer(args, 0, 1)
    for n in tr.nns.values():
        n.eval()
    rcp = tr.rcp

    def step(i):
        inp = tr.batches[i % len(tr.batches)]
        with torch.no_grad():
            outs = tr.U.forward_model(rcp["fea_dict"], rcp["lab_dict"], rcp["arch_dict"], rcp["model"], tr.nns, tr.costs, inp,
                                      tr.inp_out_dict, tr.T, tr.B, "valid", [])
        return outs["loss_final"]

    for i in range(warmup):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    last = None
    for i in range(steps):
        last = step(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    frames = tr.T * tr.B if rcp["seq"] else tr.B
    rec = {"dtype": args.prec, "ms_per_step": round(dt / steps * 1e3, 3), "value": round(frames * steps / dt, 1),
           "unit": "frames/s", "steps": steps, "warmup": warmup, "loss_final": round(float(last), 5),
           "note": "to_do = valid on the same batch: eval-mode modules under torch.no_grad(), forward_model with its cost "
                   "lines, no backward, no optimizer (the validation / forward chunks of a recipe)"}
    release(tr)
    return rec


def sub_record(rec, extra=()):
    keep = ("dtype", "ms_per_step", "value", "unit", "steps", "warmup", "roofline", "entry_points_ms_per_step",
            "whole_step_tflops", "regions_ms_per_step") + tuple(extra)
    return {k: rec[k] for k in keep if k in rec}


def main():
    import copy

    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        self_launch(args)
    if args.cpu_full:
        rec = cpu_baseline(args, args.recipe, full=True)
        rec["recipe"] = args.recipe
        print(json.dumps(rec), flush=True)
        return
    DP = importlib.import_module("pytorch-kaldi_amd.dp")
    if args.force_reducer and "WORLD_SIZE" not in os.environ:
        import socket
        import torch.distributed as dist
        s_ = socket.socket()
        s_.bind(("127.0.0.1", 0))
        os.environ.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s_.getsockname()[1]))
        s_.close()
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    rank, world, _ = DP.init_from_env()
    if args.launch_check:
        launch_check(rank, world)
        return
    if args.only_forward_mode:
        print(json.dumps(forward_mode(args)), flush=True)
        return
    if world != args.gpus and world > 1:
        args.gpus = world
    out, tr = measure(args, rank, world, args.steps, args.warmup)
    headline = args.recipe == "timit_ligru" and args.prec == "bf16" and (args.T, args.B) == (500, 128) and args.layers is None
    if rank == 0:
        out["entry_points_note"] = ("HIP events around every C-ABI call of two extra eager steps on ONE stream; the timed "
                                    "region overlaps the weight-gradient GEMMs with the recurrences on a second stream")
    release(tr)
    if rank == 0 and world == 1 and headline and not args.no_extras:
        F_ = importlib.import_module("pytorch-kaldi_amd.functional")
        # (1) the 1e-4-grade mode of the SAME workload: exact-fp32 MFMA, what tests/test_gpu_parity.py grades
        a2 = copy.copy(args)
        a2.prec, a2.repeats = "fp32", 1
        rec, tr2 = measure(a2, rank, world, 3, 1)
        out["parity_mode"] = sub_record(rec)
        out["parity_mode"]["note"] = ("same network, batch and sequence length in the engine's exact-fp32 mode (fp32 MFMA, "
                                      "157.3 TFLOP/s peak): the mode the 1e-4 parity tests run in")
        release(tr2)
        # (2) the same workload through the chunk loop, and as the reference's own caller would drive these classes
        try:
            out["through_run_nn"] = through_run_nn(args)
        except Exception as e:
            out["through_run_nn"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:  # SURVEY.md 8(d)'s secondary variant: sentence lengths U[250, 500], zero-padded with a random left offset (core.py:588-595)
            out["padded_batches"] = through_run_nn(args, n_batches=8, reps=4, ragged=True)  # (the median of three timed chunks: one in three fresh processes shows a single 60 ms hiccup in its second chunk)
        except Exception as e:
            out["padded_batches"] = {"error": "%s: %s" % (type(e).__name__, e)}
        try:
            a4 = copy.copy(args)
            a4.torch_optim, a4.unfused_cost, a4.sync_every_step, a4.repeats = True, True, True, 1
            rec, tr4 = measure(a4, rank, world, 30, 3)
            out["reference_caller"] = sub_record(rec, ("config",))
            out["reference_caller"]["note"] = ("what a recipe gets that ONLY switches arch_library: torch.optim optimizers "
                                               "(utils.py:2106-2164), the cost through nn.NLLLoss on the log-posteriors "
                                               "(utils.py:2361), one loss read-back per batch (core.py:689)")
            release(tr4)
        except Exception as e:
            out["reference_caller"] = {"error": "%s: %s" % (type(e).__name__, e)}
        finally:
            F_.settings.fused_cost = True
        try:  # (2b) the forward-only chunks of the same recipe (validation / decoding passes)
            out["forward_mode"] = forward_mode(args)
        except Exception as e:
            out["forward_mode"] = {"error": "%s: %s" % (type(e).__name__, e)}
        # (3) the other BASELINE configurations, batch 128 per GPU: perf mode (median of 3 regions) and parity mode
        out["other_configs"] = []
        for recipe, steps, warmup in OTHER_CONFIGS:
            a3 = copy.copy(args)
            a3.recipe, a3.prec, a3.repeats = recipe, "bf16", 3
            try:
                rec, tr3 = measure(a3, rank, world, steps, warmup)
                ent = {"recipe": recipe, "config": rec["config"], "dtype": rec["dtype"],
                       "ms_per_step": rec["ms_per_step"], "value": rec["value"], "unit":
                       "frames/s" if tr3.rcp["seq"] or recipe == "timit_mlp" else "chunks/s",
                       "steps": steps, "warmup": warmup, "regions_ms_per_step": rec.get("regions_ms_per_step"),
                       "regions_max_over_min": rec.get("regions_max_over_min"), "allocator": rec.get("allocator"),
                       "roofline": rec["roofline"],
                       "entry_points_ms_per_step": dict(list(rec["entry_points_ms_per_step"].items())[:4])}
                release(tr3)
                # the same configuration in the exact-fp32 mode (BASELINE names bf16 for config 2 only: this is the row the
                # 1e-4 parity tests grade), graph-replayed where the bf16 row is
                a5 = copy.copy(a3)
                a5.prec, a5.repeats = "fp32", 1
                seq3 = recipe in ("timit_lstm", "libri_gru")
                rec5, tr5 = measure(a5, rank, world, 3 if seq3 else steps, 1 if seq3 else warmup)
                ent["parity_mode"] = {k: rec5[k] for k in ("dtype", "ms_per_step", "value", "steps", "warmup")}
                ent["parity_mode"]["roofline"] = {k: rec5["roofline"].get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac")}
                ent["parity_mode"]["entry_points_ms_per_step"] = dict(list(rec5["entry_points_ms_per_step"].items())[:3])
                release(tr5)
                if not args.no_cpu_baseline:
                    # the reference path of THIS configuration on this host's cores (bounded: ~8 s; sequence recipes at the
                    # metric's T with 8 of the 128 sequences)
                    ent["cpu_baseline"] = cpu_baseline(a3, recipe, budget_s=8.0, shape=(args.T, 8) if seq3 else (1, 128))
                out["other_configs"].append(ent)
            except Exception as e:  # a recipe that fails must not take the headline line with it
                out["other_configs"].append({"recipe": recipe, "error": "%s: %s" % (type(e).__name__, e)})
        F_.set_precision(args.prec)
    if rank == 0:
        log("roofline leg done")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.recipe)
        print(json.dumps(out), flush=True)
    import torch.distributed as dist
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
