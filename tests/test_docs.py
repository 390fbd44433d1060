"""The documents only name tests that exist (DESIGN.md cited a GPU test that a commit had deleted)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "tools/README.md", "include/figh.h"]
MODULES = {os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(ROOT, "tests", "test_*.py"))}


def _defined_tests():
    names = set()
    for path in glob.glob(os.path.join(ROOT, "tests", "test_*.py")):
        with open(path) as f:
            names.update(re.findall(r"^def (test_\w+)", f.read(), flags=re.M))
    return names


def test_documents_name_only_tests_that_exist():
    defined = _defined_tests()
    assert len(defined) > 100
    missing = []
    for doc in DOCS:
        path = os.path.join(ROOT, doc)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            text = f.read()
        for name in sorted(set(re.findall(r"\b(test_[a-z0-9_]+)\b", text))):
            if name.endswith("_") or name in MODULES:
                continue  # module names
            if name not in defined:
                missing.append((doc, name))
    assert not missing, "documents cite tests that do not exist: %s" % missing


def test_profiles_gputest_summaries_name_only_tests_that_exist():
    defined = _defined_tests()
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_gputest_summary*.txt"))):
        with open(path) as f:
            for name in set(re.findall(r"\b(test_[a-z0-9_]+)\b", f.read())):
                assert name in defined or name in MODULES, (path, name)
