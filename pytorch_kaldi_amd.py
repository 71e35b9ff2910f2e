"""Alias so that ``import pytorch_kaldi_amd`` works: the real package directory is
``pytorch-kaldi_amd/`` (hyphen, importable through importlib / cfg arch_library only)."""
import importlib
import sys

_pkg = importlib.import_module("pytorch-kaldi_amd")
sys.modules[__name__] = _pkg
for _sub in ("_lib", "functional", "nn", "build", "utils", "optim", "dp", "core", "recipes", "graphs", "data_io"):
    sys.modules[__name__ + "." + _sub] = importlib.import_module("pytorch-kaldi_amd." + _sub)
