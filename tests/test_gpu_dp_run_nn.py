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
    # The transport adds the N shares in another order than the replay: 1e-7 relative on a gradient element.  RMSprop's
    # first steps move a parameter by +-lr / sqrt(1 - alpha) for ANY non-zero gradient, so an element whose gradient is
    # rounding noise around zero (a BatchNorm shift of a unit that is almost never active) may land an lr-sized step apart:
    # a tensor is held to 2e-4 outside at most 0.2 % of its elements, and those may differ by a few learning-rate steps only.
    LR_STEPS = 3 * 4e-4 / (1 - 0.95) ** 0.5 * 1.5
    worst, outliers = ("", 0.0), 0
    for arch, sd in ref["sd"].items():
        for k, v in sd.items():
            if not v.is_floating_point():
                assert int(got["sd"][arch][k]) == int(v), (arch, k)
                continue
            a, b = got["sd"][arch][k].double().reshape(-1), v.double().reshape(-1)
            d = (a - b).abs()
            far = d > 1e-5 + 1e-3 * b.abs()
            n_far = int(far.sum())
            assert n_far <= max(1, int(2e-3 * b.numel())), (arch, k, n_far, b.numel())
            assert float(d.max()) < LR_STEPS, (arch, k, float(d.max()))
            outliers += n_far
            e = float(d[~far].norm()) / max(float(b.norm()), 1e-12)
            if e > worst[1]:
                worst = (arch + "/" + k, e)
    print("run_nn_dp on %d ranks (%s) vs the shard average after 3 batches: loss %.6f vs %.6f, worst parameter %s, %d element(s) an "
          "lr-sized step apart" % (world, prec, got["loss"], ref["loss"], worst, outliers))
    assert worst[1] < 2e-4, worst
