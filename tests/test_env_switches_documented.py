"""Every PK_* environment switch the library reads is listed in INTEGRATION.md and every switch listed there is read
somewhere (the table is what a recipe author sees)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_environment_switches_match_the_integration_table():
    doc = set(re.findall(r"`(PK_[A-Z0-9_]+)`", open(os.path.join(ROOT, "INTEGRATION.md")).read()))
    src = set()
    files = (glob.glob(os.path.join(ROOT, "pytorch-kaldi_amd", "*.py")) + glob.glob(os.path.join(ROOT, "pytorch-kaldi_amd", "csrc", "*.hip"))
             + glob.glob(os.path.join(ROOT, "pytorch-kaldi_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "integration", "*.py")))
    for f in files:
        t = open(f).read()
        src |= set(re.findall(r'"(PK_[A-Z0-9_]+)"', t))
    src = {s for s in src if not s.startswith(("PK_ACT_", "PK_CELL_", "PK_REC_ALGO_"))}  # enum names quoted in messages
    assert src - doc == set(), "read but not documented: %s" % sorted(src - doc)
    assert doc - src == set(), "documented but never read: %s" % sorted(doc - src)
