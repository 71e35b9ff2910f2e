#!/usr/bin/env python
"""Run an UNMODIFIED PyTorch-Kaldi ``run_exp.py`` on the MI355X engine.

    cd /path/to/pytorch-kaldi
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        /path/to/graft/integration/run_exp_mi355x.py cfg/TIMIT_baselines/TIMIT_liGRU_fmllr.cfg

Why a launcher and not just PYTHONPATH: ``run_exp.py`` resolves the chunk function from a module that must be called
``core`` (run_exp.py:129-131) and imports helpers from it (run_exp.py:37) - and python puts the SCRIPT's directory, i.e.
the checkout with its own core.py, in front of every PYTHONPATH entry.  This file puts its own directory (which holds
the shim ``core.py``) first, the checkout second, the graft third, and then executes run_exp.py as ``__main__`` with
the remaining command line.  Nothing of the reference is modified or copied; ``PK_CORE_ENGINE=0`` turns the shim into
a pass-through (the reference's own ``run_nn``).

The checkout is ``PK_KALDI_ROOT`` or the current directory.  ``PK_RUN_EXP_SCRIPT`` names another script to execute in
that environment (tests/test_host_logic_round6.py runs a probe holding run_exp.py's own lookup statements).
"""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GRAFT = os.path.dirname(HERE)


def prepare(root):
    """sys.path = [shim dir, checkout, graft, ...what was there]."""
    root = os.path.abspath(root)
    if not os.path.isfile(os.path.join(root, "run_exp.py")) or not os.path.isfile(os.path.join(root, "core.py")):
        sys.stderr.write("ERROR: %s is not a PyTorch-Kaldi checkout (no run_exp.py / core.py): cd there or set "
                         "PK_KALDI_ROOT\n" % root)
        sys.exit(1)
    os.environ["PK_KALDI_ROOT"] = root
    rest = [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in (HERE, root, GRAFT)]
    sys.path[:] = [HERE, root, GRAFT] + rest
    return root


def main():
    root = prepare(os.environ.get("PK_KALDI_ROOT") or os.getcwd())
    script = os.environ.get("PK_RUN_EXP_SCRIPT") or os.path.join(root, "run_exp.py")
    sys.argv = [script] + sys.argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
