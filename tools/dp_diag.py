"""Diagnostics for the bf16 two-rank difference: is the single-process reference run-to-run deterministic, and does the
difference follow the side-stream weight gradients?"""
import os
import subprocess
import sys

import torch

W = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "dp_two_ranks_gpu.py")


def run(args, env=None, dp=False, port=29611):
    e = dict(os.environ, **(env or {}))
    if dp:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), W] + args
        e["PK_DP_BACKEND"] = "gloo"
    else:
        cmd = [sys.executable, W, "--reference"] + args
    r = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        print(r.stdout[-2000:])
    return r.returncode


def err(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


run(["--out", "/tmp/r1.pt", "--prec", "bf16", "--steps", "1"])
# (round 3 ran the next two with the exchange forced to its placement-independent flavour; that switch is an API call now:
# pk_persist2_set_mode)
run(["--out", "/tmp/r5.pt", "--prec", "bf16", "--steps", "1"], env={})
run(["--out", "/tmp/d1.pt", "--prec", "bf16", "--steps", "1"], dp=True)
run(["--out", "/tmp/d3.pt", "--prec", "bf16", "--steps", "1"], env={}, dp=True, port=29613)
run(["--out", "/tmp/r6.pt", "--prec", "bf16", "--steps", "1"], env={"OMP_NUM_THREADS": "1"})
r6 = torch.load("/tmp/r6.pt")
r1, r5 = torch.load("/tmp/r1.pt"), torch.load("/tmp/r5.pt")
print("reference with OMP_NUM_THREADS=1 vs default:", {k: "%.2e" % err(r6["grad0"][k], r1["grad0"][k]) for k in r1["grad0"]})
for rank in (0, 1):
    g = torch.load("/tmp/d1.pt.raw%d" % rank)
    print("d1 rank", rank, "vs OMP=1 reference shard", {k: "%.2e" % err(g[k], r6["raw"][rank][k]) for k in g})
print("reference fast vs safe exchange:", {k: "%.2e" % err(r5["grad0"][k], r1["grad0"][k]) for k in r1["grad0"]})
for tag in ("d1", "d3"):
    for rank in (0, 1):
        g = torch.load("/tmp/%s.pt.raw%d" % (tag, rank))
        print(tag, "rank", rank, "raw gradient vs reference shard", rank, {k: "%.2e" % err(g[k], r1["raw"][rank][k]) for k in g},
              "vs safe reference", {k: "%.2e" % err(g[k], r5["raw"][rank][k]) for k in g})
