"""Host-side logic added in round 5 (runs without a GPU): the PK_EXPERIMENT parser on both sides of the ABI, the bookkeeping of
the heads' shared input-gradient operand, the library's new entry points."""
import ctypes
import importlib

import torch

F_ = importlib.import_module("pytorch-kaldi_amd.functional")
_lib = importlib.import_module("pytorch-kaldi_amd._lib")


def test_experiment_keys_are_parsed_from_one_variable(monkeypatch):
    monkeypatch.delenv("PK_EXPERIMENT", raising=False)
    assert _lib.experiment("side_late", "1") == "1" and _lib.experiment("rec_gen") is None
    monkeypatch.setenv("PK_EXPERIMENT", "side_late=0, rec_gen_bwd=3,helper_lead=4:2:3")
    assert _lib.experiment("side_late", "1") == "0"
    assert _lib.experiment("rec_gen_bwd") == "3" and _lib.experiment("rec_gen") is None   # (a key, not a prefix)
    assert _lib.experiment("helper_lead") == "4:2:3"


def test_library_reads_its_experiment_keys_from_the_same_string(monkeypatch):
    """pk_experiment (pk_lib.hip) is not exported; its effect is: the helper mode stays the per-cell default (-1) whatever
    PK_EXPERIMENT holds, and an explicit mode round-trips through the setter (bits 0-2; -1 = back to the default)."""
    lib = _lib.load()
    monkeypatch.setenv("PK_EXPERIMENT", "helper_wgs=2,helper_lead=4:2:3")
    old = lib.pk_rec_helper_get_mode()
    try:
        lib.pk_rec_helper_set_mode(5 | (4 << 8) | (6 << 20))
        assert lib.pk_rec_helper_get_mode() == 5
        lib.pk_rec_helper_set_mode(-1)
        assert lib.pk_rec_helper_get_mode() == -1
    finally:
        lib.pk_rec_helper_set_mode(old)


class _Ctx:
    def __init__(self, key):
        self.dx_share = key


def test_heads_on_one_input_share_one_operand(monkeypatch):
    """functional._cat_register / _cat_slot: heads registered in one forward pass get column ranges of ONE bf16 operand
    (widths rounded up to 64), small batches and lone heads keep the per-head path, a head registered before another
    head was BUILT (interleaved model lines) falls back as well."""
    monkeypatch.delenv("PK_EXPERIMENT", raising=False)
    monkeypatch.setattr(F_._DxShare, "cat", {})
    monkeypatch.setattr(F_._DxShare, "epoch_fwd", 7)
    key = (1234, 0, (64000, 1100), (1100, 1))
    w1, w2 = torch.zeros(1938, 1152, dtype=torch.bfloat16), torch.zeros(48, 1152, dtype=torch.bfloat16)
    a, b = _Ctx(key), _Ctx(key)
    F_._cat_register(key, id(a), 1938, w1, 64000)
    assert F_._cat_slot(a, 8) is None                      # one head: nothing to share
    F_._cat_register(key, id(b), 48, w2, 64000)
    c, mine = F_._cat_slot(b, 8)
    assert mine["off"] == 1984 and c["width"] == 2048 and tuple(c["dz"].shape) == (8, 2048)
    c2, first = F_._cat_slot(a, 8)
    assert c2 is c and first["off"] == 0
    c["done"].add(id(a))
    assert F_._cat_slot(a, 8) is None                      # a second backward pass over a retained graph: per-head path
    # small batch (the graph-replayed recipes): never registered
    F_._DxShare.cat.clear()
    F_._cat_register(key, id(a), 1938, w1, 128)
    assert F_._DxShare.cat == {}
    # interleaved: a head built (epoch_fwd moves) between two cost lines
    F_._cat_register(key, id(a), 1938, w1, 64000)
    monkeypatch.setattr(F_._DxShare, "epoch_fwd", 8)
    F_._cat_register(key, id(b), 48, w2, 64000)
    assert F_._cat_slot(a, 8) is None and F_._cat_slot(b, 8) is None
    # different weight pitches cannot be rows of one matrix
    F_._DxShare.cat.clear()
    F_._cat_register(key, id(a), 1938, w1, 64000)
    F_._cat_register(key, id(b), 48, torch.zeros(48, 1216, dtype=torch.bfloat16), 64000)
    assert F_._cat_slot(a, 8) is None
    monkeypatch.setenv("PK_EXPERIMENT", "head_dx_cat=0")
    F_._DxShare.cat.clear()
    F_._cat_register(key, id(a), 1938, w1, 64000)
    assert F_._DxShare.cat == {}


def test_round5_entry_points_are_exported():
    lib = _lib.load()
    for name in ("pk_sinc_bank_fwd", "pk_sinc_bank_bwd", "pk_ln_last_act_drop_fwd", "pk_ln_last_act_drop_bwd",
                 "pk_logsoftmax_bwd_bf16_p", "pk_nll_logsoftmax_bwd_bf16_p", "pk_rec_helper_set_mode", "pk_rec_helper_get_mode"):
        assert isinstance(getattr(lib, name), ctypes._CFuncPtr), name


def test_conv_bf16_default_is_shared_by_engine_and_model(monkeypatch):
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import golden_util as G

    monkeypatch.delenv("PK_CONV_BF16", raising=False)
    assert F_.conv_bf16_mode() == F_.CONV_BF16_DEFAULT == "1" and G.conv_bf16_on() is True
    monkeypatch.setenv("PK_CONV_BF16", "0")
    assert G.conv_bf16_on() is False
    monkeypatch.setenv("PK_CONV_BF16", "2")
    assert G.conv_bf16_on() == "all"


def test_step_fence_waits_for_the_step_depth_back(monkeypatch):
    """functional.StepFence: one event per step, the host waits for the event `depth` steps back and for nothing else
    (an unbounded lead of the host filled the HBM with record_stream'd activations: DESIGN.md 12.8)."""
    import importlib

    import torch

    F_ = importlib.import_module("pytorch-kaldi_amd.functional")
    log = []

    class Ev:
        n = 0

        def __init__(self):
            self.k = Ev.n
            Ev.n += 1

        def record(self):
            log.append(("record", self.k))

        def synchronize(self):
            log.append(("wait", self.k))

    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "Event", Ev)
    monkeypatch.delenv("PK_STEPS_IN_FLIGHT", raising=False)
    f = F_.StepFence()
    assert f.depth == 4
    for _ in range(7):
        f()
    assert [e for e in log if e[0] == "wait"] == [("wait", 0), ("wait", 1), ("wait", 2)]
    assert log.index(("wait", 0)) > log.index(("record", 4))  # (step 5 waits for step 1: four steps stay in flight)
    log.clear()
    monkeypatch.setenv("PK_STEPS_IN_FLIGHT", "0")
    g = F_.StepFence()
    for _ in range(5):
        g()
    assert log == []
    h = F_.StepFence(depth=1)
    h(); h(); h()
    assert [e for e in log if e[0] == "wait"] == [("wait", Ev.n - 3), ("wait", Ev.n - 2)]


def test_fp32_weight_gradient_split_fills_four_workgroups_per_cu(monkeypatch):
    """functional._splitk (exact-fp32 GEMMs with a long reduction): the split fills 1024 workgroup slots - four 128 x 128
    workgroups fit a CU - capped at 16 and at K / 1024; PK_EXPERIMENT f32_splitk_slots=256 gives rounds 1-4's rule back
    (profiles/r05_fp32_gemm.json: the 1100 x 1104 x 64000 product ran as 243 workgroups at 66 TFLOP/s)."""
    import importlib

    F_ = importlib.import_module("pytorch-kaldi_amd.functional")
    monkeypatch.delenv("PK_EXPERIMENT", raising=False)
    assert F_._splitk(F_._tiles(1100, 1104), 64000) == 12      # 81 tiles -> 972 workgroups
    assert F_._splitk(F_._tiles(550, 550), 64000) == 16        # 25 tiles: the cap
    assert F_._splitk(F_._tiles(1938, 1100), 64000) == 7       # 144 tiles
    assert F_._splitk(F_._tiles(1100, 1104), 4000) == 1        # short reductions are not split
    assert F_._splitk(F_._tiles(128, 128), 8192) == 8          # K / 1024 bounds it
    monkeypatch.setenv("PK_EXPERIMENT", "f32_splitk_slots=256")
    assert F_._splitk(F_._tiles(1100, 1104), 64000) == 3
