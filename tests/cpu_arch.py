"""CPU stand-in for an `arch_library`: the reference's plug-in contract (README.md:588-636 - `Class(options, inp_dim)`,
`.out_dim`, `forward(x)`) served by the CPU oracle, so that the host-side model-graph interpreter
(pytorch-kaldi_amd/utils.py) can be exercised without a GPU.  Test infrastructure only."""
import torch

import pk_oracle as O


class MLP(torch.nn.Module):
    def __init__(self, options, inp_dim):
        super().__init__()
        self.options = dict(options)
        self.inp_dim = inp_dim
        self.out_dim = int(self.options["dnn_lay"].split(",")[-1])
        self.sd = None  # {name: tensor}, assigned by the test from a fixture

    def forward(self, x):
        return O.mlp_forward(self.options, self.sd, x, training=self.training)
