"""core.run_nn_dp on EIGHT ranks of one GPU, strong-scaling shapes (global batch 128 -> 16 sequences = 32 rows per rank:
two-cluster persistent launches, eight processes' kernels sharing the chip), against a one-process replay of the same chunk
shard by shard with averaged gradients (the N-GPU parity definition of SURVEY.md 8e; the reference's own multi-GPU hook is
core.py:103-104, 537-538).  Transport: gloo over device tensors (RCCL refuses several ranks per device); every kernel,
stream, bucket and the chunk loop's own plumbing - column assignment, reducer over the fused optimizers' flat buckets, rank
0's checkpoint and .info - is the production path."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
WORKER = os.path.join(HERE, "dp_run_nn_gpu.py")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,prec", [(8, "bf16"), (4, "fp32")])
def test_run_nn_dp_on_n_ranks_of_one_gpu_equals_the_shard_average(tmp_path, world, prec):
    ref_out, dp_out = str(tmp_path / "ref.pt"), str(tmp_path / "dp.pt")
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")  # (same thread count as torchrun's workers: LAPACK QR of the orthogonal init)
    r = subprocess.run([sys.executable, WORKER, "--reference", "--out", ref_out, "--tmp", str(tmp_path), "--prec", prec,
                        "--world", str(world)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-3000:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), WORKER, "--out", dp_out, "--tmp", str(tmp_path), "--prec", prec, "--world", str(world)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:]
    ref, got = torch.load(ref_out), torch.load(dp_out)
    assert abs(got["loss"] - ref["loss"]) < 1e-5 * abs(ref["loss"]), (got["loss"], ref["loss"])
    # One SGD step: parameter - initial value = -lr x the averaged gradient.  The transport adds the N shares in another
    # order than the replay (1e-7 relative per element); everything else is the same kernels on the same shards.
    worst = ("", 0.0)
    for arch, sd in ref["sd"].items():
        for k, v in sd.items():
            if not v.is_floating_point():
                assert int(got["sd"][arch][k]) == int(v), (arch, k)
                continue
            if "running" in k:  # (rank 0's replica keeps its statistics: the replay keeps shard 0's)
                assert float((got["sd"][arch][k].double() - v.double()).abs().max()) <= 1e-6 * float(v.double().abs().max() + 1e-12), (arch, k)
                continue
            upd_ref = v.double() - ref["init"][arch][k].double()
            upd = got["sd"][arch][k].double() - ref["init"][arch][k].double()
            if float(upd_ref.norm()) == 0.0:
                assert float(upd.norm()) == 0.0, (arch, k)
                continue
            e = float((upd - upd_ref).norm()) / float(upd_ref.norm())
            if e > worst[1]:
                worst = (arch + "/" + k, e)
    print("run_nn_dp on %d ranks (%s) vs the shard average, one SGD step: loss %.6f vs %.6f, worst update %s"
          % (world, prec, got["loss"], ref["loss"], worst))
    # (shard gradients cancel in the average: the 1e-7 of another summation order is relative to the shards, not to their mean)
    assert worst[1] < 1e-4, worst
