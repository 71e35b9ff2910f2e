#!/usr/bin/env python
"""Why did PK_CONV_BF16=1 miss ONE tensor of its own grading for two rounds (MLP_layers wx.4.weight: 3.2e-2 engine vs the
bf16-operand model against max(2e-2, 2 x a one-sample noise floor))?  Host-only diagnosis (no GPU): the bf16-operand
model of the SincNet recipe (oracle/pk_oracle.py:198-203 - the same rule as the engine: layers with >= 8 input channels
take bf16 operands in forward, data gradient and filter gradient) run several times with a 1e-6 relative perturbation of
the waveform and DIFFERENT perturbation seeds, both sides taking their own ReLU / max-pool decisions as the GPU test does.
If the model's distance from ITSELF varies from seed to seed by more than the factor 2 the test allowed, the miss was the
limit (a one-sample estimate of a chaotic quantity), not an engine / model disagreement.
    python tools/diag_conv_bf16_floor.py [seeds] [out.json]
TEST INFRASTRUCTURE (imports oracle/)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pk_oracle as O  # noqa: E402
import scale_util as SU  # noqa: E402
from golden_util import Golden, grad_err  # noqa: E402

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.set_num_threads(min(len(os.sched_getaffinity(0)), 16))
g = Golden("scale_sincnet_3200")
U, cfg, iod, nns, costs = SU.build(g, False)
gtot = SU.grad_total(g)
res = {}
for mode, conv in (("PK_CONV_BF16=1", True), ("PK_CONV_BF16=2", "all")):
    base = SU.oracle_params(nns)
    SU.oracle_run(O, g, base, emulate=True, forced=False, force_pool=False, conv_bf16=conv)
    per = {}
    for s in range(nseeds):
        noisy = SU.oracle_params(nns)
        SU.oracle_run(O, g, noisy, emulate=True, forced=False, force_pool=False, conv_bf16=conv, inp_noise=1e-6, noise_seed=99 + s)
        for name in base:
            for k, v in base[name].items():
                if v.grad is None or float(v.grad.norm()) < 1e-6 * gtot:
                    continue
                e = grad_err(noisy[name][k].grad, v.grad, gtot * float(v.grad.norm()) / gtot)
                per.setdefault("%s/%s" % (name, k), []).append(float(e))
        print(mode, "seed", s, "done", flush=True)
    spread = {k: {"errs": [round(e, 5) for e in v], "max_over_min": round(max(v) / max(min(v), 1e-12), 2)} for k, v in per.items()}
    worst_ratio = max(spread.items(), key=lambda kv: kv[1]["max_over_min"])
    res[mode] = {"MLP_layers/wx.4.weight": spread.get("MLP_layers/wx.4.weight"),
                 "largest_seed_to_seed_ratio": {"tensor": worst_ratio[0], **worst_ratio[1]},
                 "tensors_whose_ratio_exceeds_2": sorted(k for k, v in spread.items() if v["max_over_min"] > 2.0),
                 "all": spread}
    print(mode, json.dumps({k: v for k, v in res[mode].items() if k != "all"}, indent=1), flush=True)
if len(sys.argv) > 2:
    json.dump(res, open(sys.argv[2], "w"), indent=1)
