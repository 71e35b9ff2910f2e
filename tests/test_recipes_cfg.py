"""recipes.py (what bench.py, smoke() and the GPU tests build) against the reference's shipped cfg files.

tests/golden/cfg_recipes.json holds the [architectureN] sections and the [model] lines of the five cfg files
BASELINE.json's configs name, parsed from /root/reference by oracle/make_golden.py (cfg_case).  Every value
recipes.recipe() sets must equal the shipped one; the only allowed differences are the documented ones:

  arch_library        the drop-in switch itself
  arch_name           a label; the [model] lines are compared after mapping names to their section
  head dnn_lay        the shipped files carry the N_out_lab_* placeholders run_exp.py resolves from the alignments;
                      the recipes carry the resolved TIMIT / Librispeech counts
"""
import importlib
import json
import os
import re

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "cfg_recipes.json")))
R = importlib.import_module("pytorch-kaldi_amd.recipes")

RESOLVED = {"N_out_lab_cd": "n_cd", "N_out_lab_mono": "n_mono"}


def _norm(v):
    return ",".join(t.strip() for t in str(v).split(","))


@pytest.mark.parametrize("name", sorted(GOLD))
def test_sections_match_shipped_cfg(name):
    rcp = R.recipe(name)
    ref = GOLD[name]["sections"]
    mine = {s: dict(rcp["cfg"][s]) for s in rcp["cfg"].sections() if s.startswith("architecture")}
    assert sorted(mine) == sorted(ref), "same architecture sections as %s" % GOLD[name]["file"]
    for sec, kv in mine.items():
        checked = 0
        for k, v in kv.items():
            if k == "arch_library":
                assert v == "pytorch-kaldi_amd.nn"
                continue
            assert k in ref[sec], "%s/%s: recipes.py sets %s, which %s does not" % (name, sec, k, GOLD[name]["file"])
            want = ref[sec][k]
            if want in RESOLVED:
                want = str(rcp[RESOLVED[want]])
            assert _norm(v) == _norm(want), "%s [%s] %s" % (name, sec, k)
            checked += 1
        # nothing the shipped section sets for the network or the optimizer is left out
        skip = {"arch_name", "arch_proto", "arch_library"}
        missing = [k for k in ref[sec] if k not in kv and k not in skip]
        assert not missing, "%s [%s] lacks %s" % (name, sec, missing)
        assert checked >= 10


def _sectioned(lines, name_to_sec, feature_names):
    out = []
    for ln in lines:
        ln = ln.replace(" ", "")
        for nm in sorted(name_to_sec, key=len, reverse=True):
            ln = re.sub(r"\(%s," % re.escape(nm), "(%s," % name_to_sec[nm], ln)
        for f in feature_names:
            ln = re.sub(r",%s\)" % re.escape(f), ",FEA)", ln)
        out.append(ln)
    return out


@pytest.mark.parametrize("name", sorted(GOLD))
def test_model_lines_match_shipped_cfg(name):
    rcp = R.recipe(name)
    ref = GOLD[name]
    ref_names = {kv["arch_name"]: sec for sec, kv in ref["sections"].items()}
    my_names = {nm: v[0] for nm, v in rcp["arch_dict"].items()}
    a = _sectioned(ref["model"], ref_names, ["fmllr", "raw", "mfcc", "fbank"])
    b = _sectioned(rcp["model"], my_names, ["fea"])
    assert a == b


def test_section_roles():
    for name in GOLD:
        rcp = R.recipe(name)
        cfg = rcp["cfg"]
        assert cfg[rcp["head_cd"]]["dnn_lay"] == str(rcp["n_cd"])
        if rcp["head_mono"]:
            assert cfg[rcp["head_mono"]]["dnn_lay"] == str(rcp["n_mono"])
        if rcp["trunk"]:
            assert cfg[rcp["trunk"]]["arch_class"] == "MLP"
