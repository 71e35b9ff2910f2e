"""Host-side logic added in round 6 (runs without a GPU): the shipped ``core`` shim resolved the way run_exp.py resolves
it, the gradient wire's default, the reducer's reset after a step that did not complete."""
import importlib
import os
import subprocess
import sys
import textwrap

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("PK_REFERENCE", "/root/reference")
core = importlib.import_module("pytorch-kaldi_amd.core")
dp = importlib.import_module("pytorch-kaldi_amd.dp")
F_ = importlib.import_module("pytorch-kaldi_amd.functional")

needs_reference = pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "run_exp.py")),
                                     reason="no PyTorch-Kaldi checkout here (the GPU box has none)")

RUN_EXP_LINES = """
import importlib
from core import read_next_chunk_into_shared_list_with_subprocess, extract_data_from_shared_list, convert_numpy_to_torch
run_nn_script = "SCRIPT".split(".py")[0]
module = importlib.import_module("core")
run_nn = getattr(module, run_nn_script)
"""


def _as_run_exp_does(script, body, tmp_path, **env_extra):
    """integration/run_exp_mi355x.py started in the checkout, executing - in place of run_exp.py, which needs Kaldi data -
    a probe that holds the statements of run_exp.py that touch `core` (line 37 and lines 129-131), then `body`."""
    probe = tmp_path / "probe_run_exp.py"
    probe.write_text(RUN_EXP_LINES.replace("SCRIPT", script) + textwrap.dedent(body))
    env = dict(os.environ, PK_RUN_EXP_SCRIPT=str(probe))
    env.pop("PK_KALDI_ROOT", None)
    env.pop("PYTHONPATH", None)
    env.update(env_extra)
    r = subprocess.run([sys.executable, "-W", "ignore", os.path.join(ROOT, "integration", "run_exp_mi355x.py"), "cfg/x.cfg"],
                       cwd=REF, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-3000:]
    return r.stdout


@needs_reference
@pytest.mark.parametrize("script", ["run_nn", "run_nn.py", "run_nn_dp"])
def test_core_shim_resolves_like_run_exp(script, tmp_path):
    """run_exp.py:129-131 (importlib.import_module("core"), getattr(module, run_nn_script)) and :37 (three helpers imported
    from core) against integration/core.py: the chunk function is the engine's, with the reference's parameter list, and
    the helpers are the reference's own objects."""
    out = _as_run_exp_does(script, """
        import inspect
        ref = module.reference_core
        assert run_nn is module.run_nn_dp and run_nn.__module__ == "pytorch-kaldi_amd.core", run_nn.__module__
        assert ref.run_nn is not run_nn and ref.__name__ == "_pk_reference_core"
        assert list(inspect.signature(run_nn).parameters)[:9] == list(inspect.signature(ref.run_nn).parameters)
        assert read_next_chunk_into_shared_list_with_subprocess is ref.read_next_chunk_into_shared_list_with_subprocess
        assert extract_data_from_shared_list is ref.extract_data_from_shared_list
        assert convert_numpy_to_torch is ref.convert_numpy_to_torch
        assert module.run_nn_refac01 is ref.run_nn_refac01      # anything else a cfg may name passes through
        import sys
        assert sys.argv[1:] == ["cfg/x.cfg"]                    # the command line reaches run_exp.py unchanged
        print("resolved", run_nn.__name__, ref.__file__)
    """, tmp_path)
    assert "resolved run_nn_dp " + os.path.join(REF, "core.py") in out


@needs_reference
def test_core_shim_passes_through_when_switched_off(tmp_path):
    out = _as_run_exp_does("run_nn", """
        assert run_nn is module.reference_core.run_nn and not hasattr(module, "run_nn_dp")
        print("reference", run_nn.__module__)
    """, tmp_path, PK_CORE_ENGINE="0")
    assert "reference _pk_reference_core" in out


def test_launcher_says_what_it_needs_without_a_checkout(tmp_path):
    env = dict(os.environ)
    env.pop("PK_KALDI_ROOT", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "integration", "run_exp_mi355x.py"), "x.cfg"], cwd=str(tmp_path),
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode != 0 and "PK_KALDI_ROOT" in r.stdout


@needs_reference
def test_pythonpath_alone_does_not_reach_the_shim():
    """Why there is a launcher: with the shim only on PYTHONPATH, `python run_exp.py` (script directory first on
    sys.path) still imports the checkout's own core.py."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "integration"), ROOT]))
    r = subprocess.run([sys.executable, "-W", "ignore", "-c", "import core; print(core.__file__)"], cwd=REF, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith(os.path.join(REF, "core.py")), r.stdout[-2000:]


def test_gradient_wire_default_follows_the_precision(monkeypatch):
    """advisor, round 5: in the parity mode (PK_PRECISION=fp32) several ranks must equal the shard average the way the
    reference's DataParallel sum does (core.py:103-104): the wire stays fp32 there; bf16 only when the engine computes
    with bf16 operands AND the recipe is launch-bound; PK_DP_WIRE always decides."""
    monkeypatch.delenv("PK_DP_WIRE", raising=False)
    F_.set_precision("fp32")
    assert core.default_wire(True) == "fp32" and core.default_wire(False) == "fp32"
    F_.set_precision("bf16")
    try:
        assert core.default_wire(True) == "bf16" and core.default_wire(False) == "fp32"
        monkeypatch.setenv("PK_DP_WIRE", "fp32")
        assert core.default_wire(True) == "fp32"
    finally:
        F_.set_precision("fp32")
    monkeypatch.setenv("PK_DP_WIRE", "bf16")
    assert core.default_wire(False) == "bf16"


def test_reducer_reset_rearms_after_a_step_that_did_not_complete():
    """GradReducer.reset(): a step that died between bucket hand-overs (a failed HIP-graph capture, core.capture_on_every_rank)
    leaves fired buckets, half-counted signals, handles and wire copies; after reset() the next step runs as if the broken
    one had never started, and what step 1 taught (`expect`) is kept."""
    lin = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    r = dp.GradReducer({"net": lin}, bucket_bytes=64, force=True, overlap=True)
    assert r.active and len(r.buckets) >= 2
    calls = []
    r._reduce = lambda b, buf: calls.append(b["idx"]) or _Done()
    x = torch.randn(4, 6)

    def step():
        lin.zero_grad()
        lin(x).sum().backward()
        r.finish()

    step()                                   # step 1: learns the pattern, everything reduced in finish()
    assert all(b["expect"] for b in r.buckets)
    n1 = len(calls)
    step()                                   # step 2: buckets leave from the hooks
    assert len(calls) == 2 * n1
    # a broken step: backward ran (hooks fired, buckets handed over), finish() never did
    lin.zero_grad()
    lin(x).sum().backward()
    assert any(b["fired"] for b in r.buckets) and r.handles
    with pytest.raises(RuntimeError, match="arrived after its bucket"):
        lin(x).sum().backward()              # without a reset the next backward trips over the dirty state
    r.reset()
    assert not r.handles and not any(b["fired"] or b["got"] for b in r.buckets)
    assert all(b["pending"] == len(b["expect"]) for b in r.buckets)
    calls.clear()
    step()
    assert sorted(calls) == list(range(len(r.buckets)))


class _Done:
    def wait(self):
        return True


def test_experiment_keys_are_parsed_once_per_value_and_trimmed(monkeypatch):
    """PK_EXPERIMENT="key=value, key = value": one parse per distinct string (the lookup sits on per-layer, per-step paths),
    blanks around keys and values ignored on both sides (pk_lib.hip::pk_experiment trims the same way), first occurrence
    wins, a changed environment is seen at the next call (advisor, round 5)."""
    _lib = importlib.import_module("pytorch-kaldi_amd._lib")
    monkeypatch.setenv("PK_EXPERIMENT", " mlp_fused = 0 ,direct_grads=1,mlp_fused=1, f32_wgrad_side=0")
    assert _lib.experiment("mlp_fused", "1") == "0"
    assert _lib.experiment("direct_grads") == "1"
    assert _lib.experiment("f32_wgrad_side", "1") == "0"
    assert _lib.experiment("absent", "dflt") == "dflt"
    parsed = _lib._EXP_CACHE[1]
    assert _lib.experiment("direct_grads") == "1" and _lib._EXP_CACHE[1] is parsed   # no second parse
    monkeypatch.setenv("PK_EXPERIMENT", "direct_grads=0")
    assert _lib.experiment("direct_grads") == "0" and _lib.experiment("mlp_fused", "1") == "1"
    monkeypatch.delenv("PK_EXPERIMENT")
    assert _lib.experiment("direct_grads", "1") == "1"


def test_stray_pk_variables_are_reported_once(monkeypatch):
    """The A/B levers of earlier rounds moved behind PK_EXPERIMENT; their old names (PK_MLP_FUSED, ...) are not read any
    more.  Loading the library with one of them in the environment says so once - an A/B script would otherwise compare
    two identical configurations (advisor, round 5) - while documented switches and the harness's own variables pass."""
    import warnings

    _lib = importlib.import_module("pytorch-kaldi_amd._lib")
    monkeypatch.setenv("PK_MLP_FUSED", "0")
    monkeypatch.setenv("PK_PRECISION", "fp32")
    monkeypatch.setenv("PK_BENCH_VERBOSE", "1")
    monkeypatch.setattr(_lib, "_warned_unknown", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        _lib.warn_unknown_switches()
        _lib.warn_unknown_switches()
    msgs = [str(x.message) for x in w if "not switches of this version" in str(x.message)]
    assert len(msgs) == 1 and "PK_MLP_FUSED" in msgs[0] and "PK_PRECISION" not in msgs[0] and "PK_BENCH_VERBOSE" not in msgs[0]


def test_small_batch_fp32_products_split_only_long_reductions(monkeypatch):
    """functional._small_m_splitk: an exact-fp32 product of a small batch is split along K over the chip (an MLP layer at
    128 frames: 16 slices), but reductions shorter than 512 keep ONE fmaf chain per output element - every small fixture
    and the 30-step trajectory fixture (H = 32), whose bits are pinned, stay on the arithmetic they were recorded with."""
    monkeypatch.delenv("PK_EXPERIMENT", raising=False)
    assert F_._small_m_splitk(128, 1024, 1024) == 16
    assert F_._small_m_splitk(128, 1938, 1024) == 16
    assert F_._small_m_splitk(128, 1024, 440) == 1       # (the first MLP layer: 440 inputs)
    assert F_._small_m_splitk(80, 23, 64) == 1           # (the trajectory fixture's head)
    assert F_._small_m_splitk(64000, 1100, 1100) == 1    # (row-streaming products: plenty of tiles)
    monkeypatch.setenv("PK_EXPERIMENT", "f32_small_splitk=0")
    assert F_._small_m_splitk(128, 1024, 1024) == 1
