"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every
symbol include/pk_amd.h declares, and the product path refuses to run without a GPU
(no CPU fallback, no route through oracle/)."""
import importlib
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pk = importlib.import_module("pytorch-kaldi_amd")
_lib = importlib.import_module("pytorch-kaldi_amd._lib")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        importlib.import_module("pytorch-kaldi_amd.build").build()
    return _lib.load()


def test_header_symbols_exported(lib):
    names = _lib.declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libpk_amd.so does not export " + n
    # the ctypes table mirrors the header one to one
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_string(lib):
    assert lib.pk_version() >= 100
    assert isinstance(lib.pk_last_error(), bytes)
    assert lib.pk_rec_num_gates(0) == 2 and lib.pk_rec_num_gates(2) == 4 and lib.pk_rec_num_gates(3) == 3
    assert lib.pk_rec_work_floats(0, 10, 4, 1, 16) > 0
    assert lib.pk_bn_partial_floats(1000, 64) > 0


def test_argument_errors_do_not_throw(lib):
    # bad strides are rejected before any device call (no GPU needed)
    rc = lib.pk_gemm(None, 0, 4, 4, 4, 1.0, None, 3, 2, None, 1, 4, 0.0, None, 4, None, 1, None)
    assert rc != 0 and b"contiguous" in lib.pk_last_error()


def test_no_torch_types_in_header():
    text = open(os.path.join(ROOT, "include", "pk_amd.h")).read()
    assert "at::" not in text and "torch" not in re.sub(r"/\*.*?\*/", "", text, flags=re.S)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "pytorch-kaldi_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "pk_oracle" not in src and "import oracle" not in src, f


def test_cpu_tensors_are_refused():
    nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
    opts = {"dnn_lay": "8", "dnn_drop": "0.0", "dnn_use_laynorm_inp": "False", "dnn_use_batchnorm_inp": "False",
            "dnn_use_batchnorm": "False", "dnn_use_laynorm": "False", "dnn_act": "relu"}
    net = nn_amd.MLP(opts, 4)
    with pytest.raises(_lib.PkError):
        net(torch.randn(2, 4))


def test_state_dict_names_match_reference_layout():
    """Checkpoint interoperability: same keys / shapes as the golden fixtures that
    were dumped from the reference classes."""
    from golden_util import Golden, list_cases

    nn_amd = importlib.import_module("pytorch-kaldi_amd.nn")
    for case in list_cases():
        if case.startswith(("e2e_", "chunk_", "io_", "train_", "scale_")):
            continue
        g = Golden(case)
        opts = dict(g.meta["options"])
        torch.manual_seed(g.meta["seed"])
        net = getattr(nn_amd, g.meta["arch_class"])(opts, g.meta["inp_dim"])
        ref = g.group("sd/")
        sd = net.state_dict()
        # same names in the same ORDER: parameters() order indexes torch optimizer state (optimizer_par of a .pkl)
        assert list(sd) == list(ref), case
        for k in ref:
            assert tuple(sd[k].shape) == tuple(ref[k].shape), (case, k)
        # same seed -> same initial weights as the reference constructor
        init = g.group("init/")
        for k, v in init.items():
            if v.is_floating_point():
                # orthogonal_ goes through LAPACK QR: thread count may change the last bits
                assert torch.allclose(sd[k], v, rtol=1e-5, atol=1e-6), (case, k)
        assert net.out_dim == g.t("y").shape[-1]


def test_library_binds_to_torchs_hip_runtime_by_construction():
    """SURVEY.md 7.2 / round-5 review: not load-order luck.  A fresh interpreter that loads the library through _lib.load()
    ends up with exactly ONE libamdhip64 mapped, and it is the copy inside torch/lib (the file torch itself uses) - although
    the library's RUNPATH names /opt/rocm."""
    import subprocess
    import sys

    code = ("import importlib, os, sys; sys.path.insert(0, %r); L = importlib.import_module('pytorch-kaldi_amd._lib'); "
            "L.load(); import torch; r = L.hip_runtimes_mapped(); print(sorted(r)); "
            "assert len(r) == 1 and os.path.dirname(next(iter(r))) == os.path.join(os.path.dirname(torch.__file__), 'lib'), r"
            % ROOT)
    r = subprocess.run([sys.executable, "-W", "ignore", "-c", code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:]
