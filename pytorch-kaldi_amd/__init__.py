"""pytorch-kaldi_amd - MI355X-native engine for PyTorch-Kaldi's neural_networks.py hot path.

Import as ``importlib.import_module("pytorch-kaldi_amd")`` (the directory name has a
hyphen, like the upstream project) or through the ``pytorch_kaldi_amd`` alias module at
the repository root.  ``arch_library = pytorch-kaldi_amd.nn`` in a cfg selects the engine.
"""
__version__ = "0.1.0"
