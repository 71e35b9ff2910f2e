"""The headline shape, value for value, inside the GPU suite (round-4 review: "the full-shape parity run is a tool, not a
test"): ONE bf16 training step of Li-GRU 5 x 550 + heads at T = 500, B = 128 - 16 clusters x 500 steps per launch -
against the oracle's bf16-operand model run on the box's host cores (neural_networks.py:1130-1141, utils.py:2296-2420),
kink-forced.  ~20-40 s of host time.  TEST INFRASTRUCTURE: the tool imports oracle/."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_full_headline_shape_value_for_value(tmp_path):
    out = tmp_path / "full_shape.json"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "full_shape_parity.py"), "--out", str(out)],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    res = json.loads(out.read_text())
    keep = os.environ.get("PK_FULL_SHAPE_JSON")  # (the evidence pass keeps the record for profiles/)
    if keep:
        with open(keep, "w") as f:
            json.dump(res, f, indent=1)
    print("full shape: loss %.3e  outputs %s  worst gradient %s  by layer %s" % (
        res["loss_rel_diff"], [round(res["out_rel_err/out_dnn%d" % i], 6) for i in (1, 2, 3)], res["grad_rel_err_worst"],
        res["grad_rel_err_by_layer"]["wz"]))
    assert res["pass"], res
    assert res["loss_rel_diff"] < 1e-5, res["loss_rel_diff"]
    # limits of step (A) of tests/test_gpu_reference_pins.py are 5e-3 / 2e-2; the review asks for the worst gradient
    # (wz.4.weight, 1.04e-2 in round 4) to stay under 1.5e-2
    assert res["grad_rel_err_worst"]["err"] < 1.5e-2, res["grad_rel_err_worst"]
