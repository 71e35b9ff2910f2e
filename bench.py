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
    ap.add_argument("--mask-rng", default="device", choices=["device", "reference"],
                    help="recurrent drop masks: GPU RNG (default) or the reference's CPU torch.bernoulli stream")
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
    ap.add_argument("--cpu-T", type=int, default=500, help="sequence length of the CPU baseline sample (batch 8)")
    return ap.parse_args()


class Trainer:
    """The run_nn inner loop (core.py:616-642) on the engine."""

    def __init__(self, args, rank, world):
        self.U = importlib.import_module("pytorch-kaldi_amd.utils")
        self.R = importlib.import_module("pytorch-kaldi_amd.recipes")
        self.DP = importlib.import_module("pytorch-kaldi_amd.dp")
        self.OPT = importlib.import_module("pytorch-kaldi_amd.optim")
        F_ = importlib.import_module("pytorch-kaldi_amd.functional")
        F_.set_precision(args.prec)
        F_.set_rec_algo(args.algo)
        F_.set_mask_rng(args.mask_rng)
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
            flats = {k: o.flat for k, o in self.opts.items()}
        # 8 MB buckets: the recurrent stack's 31.7 MB of gradients leave in 4 pieces while BPTT of the lower layers runs
        self.reducer = self.DP.GradReducer(self.nns, flats=flats, bucket_bytes=8 << 20, overlap=args.overlap)
        # one resident synthetic batch per rank (different seeds per rank = different shards)
        self.T, self.B = (args.T, args.B) if rcp["seq"] else (1, args.B)
        self.batches = [self.R.synthetic_batch(rcp, self.T, self.B, 4234 + 17 * rank + i, "cuda") for i in range(2)]
        self.n_params = sum(p.numel() for n in self.nns.values() for p in n.parameters())
        self.graphed = None

    def step(self, i):
        inp = self.batches[i % len(self.batches)]
        if self.graphed is not None:
            return self.graphed(inp)
        return self.step_on(inp)

    def enable_graph(self):
        """Launch-bound recipes: replay the whole step as one HIP graph (pytorch-kaldi_amd/graphs.py).  Call after
        a few eager steps."""
        G = importlib.import_module("pytorch-kaldi_amd.graphs")
        self.graphed = G.GraphedStep(self.step_on, list(self.opts.values())).capture(self.batches[0])

    def step_on(self, inp):
        rcp = self.rcp
        outs = self.U.forward_model(rcp["fea_dict"], rcp["lab_dict"], rcp["arch_dict"], rcp["model"], self.nns,
                                    self.costs, inp, self.inp_out_dict, self.T, self.B, "train", [])
        for o in self.opts.values():
            o.zero_grad()
        outs["loss_final"].backward()
        self.reducer.finish()
        for o in self.opts.values():
            o.step()
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
    """Algorithmic HBM bytes of one persistent recurrent launch (bidirectional, fp32 tensors, bf16 exchange)."""
    a1 = rcp["cfg"]["architecture1"]
    kind = a1["arch_class"]
    if kind not in ("liGRU", "LSTM", "RNN") or entry not in ("pk_rec_fwd_bf16", "pk_rec_bwd_bf16"):
        return None
    G = {"liGRU": 2, "LSTM": 4, "RNN": 1}[kind]
    NS = {"liGRU": 2, "LSTM": 5, "RNN": 1}[kind]
    H = int(a1[{"liGRU": "ligru", "LSTM": "lstm", "RNN": "rnn"}[kind] + "_lay"].split(",")[0])
    Hp = (H + 7) // 8 * 8
    rows = T * B
    if "fwd" in entry:  # read P; write Y, S (both directions) and the bf16 copy Yb
        return rows * G * H * 4 + rows * 2 * H * 4 + 2 * rows * NS * H * 4 + rows * 2 * Hp * 2
    # read S, Y (h_{t-1}) and dY; write the bf16 gate gradients of both directions
    return 2 * rows * NS * H * 4 + 2 * rows * 2 * H * 4 + 2 * rows * G * Hp * 2


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


def cpu_baseline(args, rcp_name):
    """The CPU oracle (a torch-CPU port of the reference path; kind = "port") timed on this host's
    cores on a bounded sample of the same workload: same network, shorter/narrower batch."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pk_oracle as O

    R = importlib.import_module("pytorch-kaldi_amd.recipes")
    rcp = R.recipe(rcp_name, n_lay=args.layers)
    cfg = rcp["cfg"]
    cores = usable_cores()
    torch.set_num_threads(cores)
    a1 = cfg["architecture1"]
    kind = a1["arch_class"]
    nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
    torch.manual_seed(2234)
    nets = {"architecture1": getattr(nn_amd, kind)(dict(a1, use_cuda="False", to_do="train"), rcp["nfea"])}
    feat = nets["architecture1"].out_dim
    trunk, s_cd, s_mono = rcp["trunk"], rcp["head_cd"], rcp["head_mono"]
    if trunk:  # SincNet recipe: an MLP trunk between the front-end and the heads
        nets[trunk] = nn_amd.MLP(dict(cfg[trunk]), feat)
        feat = nets[trunk].out_dim
    nets[s_cd] = nn_amd.MLP(dict(cfg[s_cd]), feat)
    if s_mono:
        nets[s_mono] = nn_amd.MLP(dict(cfg[s_mono]), feat)
    sds = {k: {n: v.detach().clone().requires_grad_(v.is_floating_point() and "running" not in n)
               for n, v in net.state_dict().items()} for k, net in nets.items()}
    opts = []
    for k, sd in sds.items():  # one optimizer per architecture, as the shipped cfg sets them (run_nn: optimizer_init)
        leaves = [v for v in sd.values() if v.requires_grad]
        if cfg[k]["arch_opt"] == "sgd":
            opts.append(torch.optim.SGD(leaves, lr=float(cfg[k]["arch_lr"])))
        else:
            opts.append(torch.optim.RMSprop(leaves, lr=float(cfg[k]["arch_lr"]), alpha=0.95, eps=1e-8))
    # sequence recipes: the metric's sequence length (the reference's autograd cost per frame GROWS with T, SURVEY.md
    # 3.3), a batch of 8 so that a step is seconds of CPU work
    T, B = (args.cpu_T, 8) if rcp["seq"] else (1, 128)

    def one_step(seed):
        inp = R.synthetic_batch(rcp, T, B, seed)
        x = inp[..., :rcp["nfea"]]
        lab_cd = inp[..., rcp["nfea"]].reshape(-1).long()
        if rcp["seq"]:
            # index_like_reference: the projections are indexed inside the time loop as the reference does it
            # (neural_networks.py:1133-1134) - that is what makes its backward O(T^2), and with it the port runs
            # within 4-11 % of the reference's own speed (profiles/r02_cpu_port_vs_reference.json)
            h = O.recurrent_forward(kind, dict(a1), sds["architecture1"], x, index_like_reference=True)
            h = h.reshape(T * B, -1)
        else:
            h = O.arch_forward(kind, dict(a1), sds["architecture1"], x)
            if trunk:
                h = O.mlp_forward(dict(cfg[trunk]), sds[trunk], h)
        loss = torch.nn.functional.nll_loss(O.mlp_forward(dict(cfg[s_cd]), sds[s_cd], h), lab_cd)
        if s_mono:
            lab_m = inp[..., rcp["nfea"] + 1].reshape(-1).long()
            loss = loss + torch.nn.functional.nll_loss(O.mlp_forward(dict(cfg[s_mono]), sds[s_mono], h), lab_m)
        for opt in opts:
            opt.zero_grad()
        loss.backward()
        for opt in opts:
            opt.step()

    one_step(1)  # warm-up
    t0 = time.time()
    n = 0
    while True:
        one_step(2 + n)
        n += 1
        if time.time() - t0 > args.cpu_budget_s or n >= 50:
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
    return {"value": round(n * T * B / dt, 2), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": "%d steps of the same network (fwd+bwd+the cfg optimizers) at T=%d, B=%d on %s, torch-CPU oracle fp32, "
                      "projections indexed in the time loop like the reference; the reference's own classes run 1.04x "
                      "(T=500) / 1.11x (T=50) slower than this port on the build host (profiles/r02_cpu_port_vs_reference.json)"
                      % (n, T, B, model or "host CPU")}


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
    if "rec" in dom and tr.rcp["seq"]:
        # A recurrent launch is T strictly dependent steps: neither roofline binds it, the per-step latency does.  Two
        # floors, both measured with this library's own traced kernels (profiles/r02_rec_step_floor.json,
        # tools/trace_rec2.py): `hop_floor_us` = one cross-CU hand-off (publish of the slowest member -> data in
        # registers; the chip's handoff-1to1 price, 0.8-1.0 us) and `step_floor_us` = a step of THIS kernel structure with
        # its MFMA block and gate math removed (EMPTY=1: poll + barrier + flush / prefetch issue + patches + publish).
        roof["dependent_steps_per_launch"] = tr.T
        roof["us_per_step"] = round(d["avg_ms"] * 1e3 / tr.T, 3)
        hop, floor = 0.9, None
        try:
            fl_ = json.load(open(os.path.join(ROOT, "profiles", "r02_rec_step_floor.json")))
            side_ = "bwd" if "bwd" in dom else "fwd"
            hop, floor = float(fl_["hop_us"][side_]), float(fl_["floor_us_" + side_])
        except (OSError, ValueError, KeyError):
            pass
        roof["hop_floor_us"] = hop
        roof["latency_frac"] = round(hop / roof["us_per_step"], 4)
        if floor is not None and dom in ("pk_rec_fwd_bf16", "pk_rec_bwd_bf16") and tr.rcp["cfg"]["architecture1"]["arch_class"] == "liGRU":
            roof["step_floor_us"] = floor
            roof["structure_frac"] = round(floor / roof["us_per_step"], 4)
    # algorithmic HBM bytes of one recurrent launch and the PMC-measured traffic
    # of the same launch at this geometry (null for any other geometry)
    rb = rec_launch_bytes(tr.rcp, tr.T, tr.B, dom)
    if rb:
        roof["algorithmic_bytes_per_launch"] = rb
        roof["hbm_gbps"] = round(rb / (d["avg_ms"] * 1e-3) / 1e9, 1)
        roof["hbm_frac"] = round(rb / (d["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK, 5)
    for src in ("r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", src)))
            if dom in pm and args.recipe == "timit_ligru" and (tr.T, tr.B) == (500, 128) and args.layers is None:
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
    use_graph = args.graph == "on" or (args.graph == "auto" and not tr.rcp["seq"] and world == 1 and not args.torch_optim)
    if use_graph:
        if warmup == 0:
            tr.step(0)  # lazy one-time initialisation must not land inside the capture
        tr.enable_graph()
        tr.step(0)      # first replay (graph upload) outside the timed region
    barrier()
    t0 = time.perf_counter()
    loss = None
    for i in range(steps):
        loss = tr.step(i)
    barrier()
    dt = time.perf_counter() - t0
    log("%s/%s timed region done: %.3f s" % (args.recipe, args.prec, dt))
    if world > 1:
        t = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
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
                   "allreduce": "overlapped" if args.overlap else "after-backward",
                   "optimizer": "torch" if args.torch_optim else "fused-flat", "hip_graph": bool(use_graph),
                   "params": tr.n_params},
        "loss_final": round(float(loss), 5),
    }
    if tr.rcp["cfg"]["architecture1"]["arch_class"] == "LSTM":
        out["config"]["lstm_waves"] = int(_lib.load().pk_persist2_get_lstm_waves())  # DESIGN.md 6.1 (PK_LSTM_WAVES)
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
OTHER_CONFIGS = [("timit_mlp", 400, 5), ("timit_lstm", 10, 2), ("libri_gru", 10, 2), ("timit_sincnet", 100, 5)]


def main():
    import copy

    args = parse()
    DP = importlib.import_module("pytorch-kaldi_amd.dp")
    rank, world, _ = DP.init_from_env()
    if world != args.gpus and world > 1:
        args.gpus = world
    out, tr = measure(args, rank, world, args.steps, args.warmup)
    headline = args.recipe == "timit_ligru" and args.prec == "bf16" and (args.T, args.B) == (500, 128) and args.layers is None
    if rank == 0:
        out["entry_points_note"] = ("HIP events around every C-ABI call of two extra eager steps on ONE stream; the timed "
                                    "region overlaps the weight-gradient GEMMs with the recurrences on a second stream")
    release(tr)
    if rank == 0 and world == 1 and headline and not args.no_extras:
        # (1) the 1e-4-grade mode of the SAME workload: exact-fp32 MFMA, what tests/test_gpu_parity.py grades
        a2 = copy.copy(args)
        a2.prec = "fp32"
        rec, tr2 = measure(a2, rank, world, 3, 1)
        out["parity_mode"] = {k: rec[k] for k in ("dtype", "ms_per_step", "value", "unit", "steps", "warmup", "roofline",
                                                  "entry_points_ms_per_step", "whole_step_tflops")}
        out["parity_mode"]["note"] = ("same network, batch and sequence length in the engine's exact-fp32 mode (fp32 MFMA, "
                                      "157.3 TFLOP/s peak): the mode the 1e-4 parity tests run in")
        release(tr2)
        # (2) the other BASELINE configurations, batch 128 per GPU
        out["other_configs"] = []
        for recipe, steps, warmup in OTHER_CONFIGS:
            a3 = copy.copy(args)
            a3.recipe, a3.prec = recipe, "bf16"
            try:
                rec, tr3 = measure(a3, rank, world, steps, warmup)
                out["other_configs"].append({"recipe": recipe, "config": rec["config"], "dtype": rec["dtype"],
                                             "ms_per_step": rec["ms_per_step"], "value": rec["value"], "unit":
                                             "frames/s" if tr3.rcp["seq"] or recipe == "timit_mlp" else "chunks/s",
                                             "steps": steps, "warmup": warmup, "roofline": rec["roofline"],
                                             "entry_points_ms_per_step": dict(list(rec["entry_points_ms_per_step"].items())[:4])})
                release(tr3)
            except Exception as e:  # a recipe that fails must not take the headline line with it
                out["other_configs"].append({"recipe": recipe, "error": "%s: %s" % (type(e).__name__, e)})
        F_ = importlib.import_module("pytorch-kaldi_amd.functional")
        F_.set_precision(args.prec)
    if rank == 0:
        log("roofline leg done")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, args.recipe)
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
