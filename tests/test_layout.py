"""Repository rules: the product never touches oracle/ or the reference tree."""
import os
import re

from conftest import ROOT

PRODUCT_DIRS = ["sleqp_amd", "shim", "include"]


def _files(d):
    for base, _, names in os.walk(os.path.join(ROOT, d)):
        for n in names:
            if n.endswith((".py", ".c", ".h", ".cpp", ".hip", ".inc", ".cmake", "Makefile")):
                yield os.path.join(base, n)


def test_product_does_not_use_oracle_or_reference():
    bad = []
    for d in PRODUCT_DIRS:
        for f in _files(d):
            text = open(f, errors="replace").read()
            code = "\n".join(l for l in text.splitlines() if not l.lstrip().startswith(("#", "//", "*", "/*", '"""')))
            if re.search(r"(liboracle|import oracle|from oracle|oracle/|/root/reference)", code):
                # documentation strings may name the oracle as the place where parity is defined
                hits = [l for l in code.splitlines() if re.search(r"(liboracle|import oracle|from oracle|/root/reference)", l)]
                if hits:
                    bad.append((f, hits[:2]))
    assert not bad, bad


def test_required_layout():
    for p in ["bench.py", "__graft_entry__.py", "DESIGN.md", "INTEGRATION.md", "include/hipfact.h", "oracle/kkt_oracle.c",
              "tests/golden/kkt_cases.npz", "tests/golden/make_golden.py", "shim/fact_hipfact.c", "profiles"]:
        assert os.path.exists(os.path.join(ROOT, p)), p


def test_option_table_of_the_header_is_generated_from_the_source():
    """include/hipfact.h documents exactly the options hipfact_set_option knows (scripts/gen_option_table.py), and
    there are at most 50 of them: an option that was measured slower for two rounds goes, with its code."""
    import subprocess
    import sys

    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "gen_option_table.py"), "--check"])
    src = open(os.path.join(ROOT, "sleqp_amd", "csrc", "abi_options.inc")).read().split("int hipfact_debug_copy")[0]
    names = re.findall(r'!strcmp\(name, "(\w+)"\)', src)
    assert len(names) == len(set(names)) <= 50, len(names)
    # no compile-time experiment switches in the device sources
    for f in _files("sleqp_amd"):
        if f.endswith((".inc", ".hip", ".h")) and "_timeline_build" not in f and "_exp_" not in f:
            text = open(f, errors="replace").read()
            assert not re.search(r"#\s*if(n?def)?\s+!?\s*(defined\()?HIPFACT_(?!TRACE|H\b|STANDALONE)", text), f
