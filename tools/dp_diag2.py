"""Which parameters differ between the overlapped and the after-backward reduction (2 ranks, gloo, bf16, first step)?"""
import os
import subprocess
import sys

import torch

W = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "dp_two_ranks_gpu.py")
for ov, port in ((1, 29622),):
    e = dict(os.environ, PK_DP_BACKEND="gloo", PK_DP_DEBUG="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), W, "--out", "/tmp/ov%d.pt" % ov, "--prec", "bf16", "--steps", "1", "--overlap", str(ov)]
    r = subprocess.run(cmd, env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    print("overlap", ov, "rc", r.returncode)
    print("\n".join(l for l in r.stdout.split("\n") if "DPDBG" in l and "[default0]" not in l)[:9000])
