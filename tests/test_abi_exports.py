"""CPU-side checks of the C-ABI: the library loads and exports every symbol include/*.h declares."""
import ctypes
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bwt_amd", "lib", "libbwt_hip.so")


def declared_symbols():
    names = set()
    for h in glob.glob(os.path.join(ROOT, "include", "bwt_hip*.h")):
        src = open(h).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(bwt_hip_[a-zA-Z0-9_]+)\s*\(", src))
    return sorted(names)


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert "bwt_hip_pmbwt" in syms and "bwt_hip_leaf" in syms and "bwt_hip_merge" in syms
    assert len(syms) >= 30


@pytest.mark.skipif(not os.path.exists(LIB), reason="libbwt_hip.so not built (python -c 'import __graft_entry__ as g; g.build()')")
def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(LIB)
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing


@pytest.mark.skipif(not os.path.exists(LIB), reason="libbwt_hip.so not built")
def test_python_binding_signatures_resolve():
    import bwt_amd
    L = bwt_amd.lib()
    assert L.bwt_hip_last_error() is not None


def test_no_oracle_in_product_path():
    # the product (bwt_amd/, include/) must never import, link or call the CPU checkers
    bad = []
    for pat in ("bwt_amd/**/*.py", "bwt_amd/csrc/*", "include/*"):
        for f in glob.glob(os.path.join(ROOT, pat), recursive=True):
            if os.path.isfile(f) and not f.endswith(".so"):
                txt = open(f, errors="replace").read()
                if re.search(r"\boracle\b|liboracle|libbwt_ref|pmbwt_oracle", txt):
                    bad.append(f)
    assert not bad, bad
