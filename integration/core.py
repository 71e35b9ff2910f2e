"""Shim named ``core``: put THIS directory in front of a PyTorch-Kaldi checkout on ``sys.path`` (PYTHONPATH) and an
unmodified ``run_exp.py`` trains on the MI355X engine.

``run_exp.py:129-131`` resolves the chunk function with ``importlib.import_module("core")`` + ``getattr(module,
run_nn_script)``, and ``run_exp.py:37`` imports three helpers (``read_next_chunk_into_shared_list_with_subprocess``,
``extract_data_from_shared_list``, ``convert_numpy_to_torch``) from the same module.  This file therefore

  1. loads the checkout's own ``core.py`` under another module name and re-exports everything it defines (so line 37's
     imports, ``run_nn_refac01`` and anything else a cfg may name keep working), and
  2. replaces ``run_nn`` by the engine's data-parallel chunk function ``run_nn_dp`` (same signature, return value, cfg
     fields and output files: core.py:439-753) and exports it under both names, so ``run_nn_script = run_nn`` - what all
     38 shipped cfg files say - picks it up, and ``run_nn_script = run_nn_dp`` works too.

The checkout is found in ``PK_KALDI_ROOT`` or, failing that, as the first ``sys.path`` entry that holds a ``core.py``
other than this one together with ``run_exp.py``.  ``PK_CORE_ENGINE=0`` leaves the reference's ``run_nn`` in place (the
shim is then a transparent pass-through: an A/B switch that needs no PYTHONPATH change).

    PYTHONPATH=/path/to/graft/integration:/path/to/graft python -m torch.distributed.run --nproc-per-node 8 \
        run_exp.py cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg        # cwd = the PyTorch-Kaldi checkout
"""
import importlib
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))


def _find_reference_core():
    root = os.environ.get("PK_KALDI_ROOT")
    cands = [root] if root else []
    cands += [p or os.getcwd() for p in sys.path]
    for d in cands:
        d = os.path.abspath(d)
        f = os.path.join(d, "core.py")
        if d != _HERE and os.path.isfile(f) and os.path.isfile(os.path.join(d, "run_exp.py")):
            return f
    raise ImportError("core shim: no PyTorch-Kaldi checkout found (set PK_KALDI_ROOT to the directory that holds "
                      "run_exp.py and core.py, or put it on sys.path behind this directory)")


_ref_file = _find_reference_core()
_ref_dir = os.path.dirname(_ref_file)
if _ref_dir not in sys.path:  # the reference's core.py imports its siblings (data_io, utils) by bare name
    sys.path.append(_ref_dir)
_spec = importlib.util.spec_from_file_location("_pk_reference_core", _ref_file)
reference_core = importlib.util.module_from_spec(_spec)
sys.modules["_pk_reference_core"] = reference_core
_spec.loader.exec_module(reference_core)
globals().update({k: v for k, v in vars(reference_core).items() if not k.startswith("__")})

if os.environ.get("PK_CORE_ENGINE", "1") != "0":
    _root = os.path.dirname(_HERE)  # the graft checkout (holds the package directory `pytorch-kaldi_amd`)
    if _root not in sys.path:
        sys.path.append(_root)
    _engine = importlib.import_module("pytorch-kaldi_amd.core")
    run_nn_dp = _engine.run_nn_dp
    run_nn = _engine.run_nn_dp
