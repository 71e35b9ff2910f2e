import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu through gpurun)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible."""
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The suite needs the in-tree library (ABI tests load it even without a GPU): build it if a fresh checkout has
    none (hipcc cross-compiles gfx950 on a CPU-only host in under a minute).  The product itself never builds or
    falls back on its own - a missing library is an error there."""
    import importlib

    lib = os.path.join(ROOT, "pytorch-kaldi_amd", "lib", "libpk_amd.so")
    if not os.path.exists(lib):
        importlib.import_module("pytorch-kaldi_amd.build").build()
    yield
