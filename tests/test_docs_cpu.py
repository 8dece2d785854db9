"""The documents list what the code reads: every MICI_AMD_* environment switch of the library, the package and bench.py
has a row in INTEGRATION.md ("Environment switches")."""
import glob
import os
import re

from conftest import ROOT


def _switches():
    names = set()
    for path in glob.glob(os.path.join(ROOT, "mici_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "mici_amd", "csrc", "*.h")):
        names |= set(re.findall(r'getenv\("(MICI_AMD_[A-Z0-9_]+)"\)', open(path).read()))
    for path in glob.glob(os.path.join(ROOT, "mici_amd", "*.py")) + [os.path.join(ROOT, "bench.py")]:
        names |= set(re.findall(r'environ(?:\.get)?\(?\[?"(MICI_AMD_[A-Z0-9_]+)"', open(path).read()))
    return names


def test_every_environment_switch_is_documented():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = _switches()
    assert len(names) >= 15, names
    missing = sorted(n for n in names if n not in doc)
    assert not missing, f"INTEGRATION.md has no row for {missing}"


def test_fixture_count_quoted_in_the_documents():
    n = len(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz")))
    for name in ("README.md", "DESIGN.md"):
        text = open(os.path.join(ROOT, name)).read()
        assert f"{n} fixtures" in text or f"{n} cases" in text, f"{name} does not quote the {n} fixtures of tests/golden/"
