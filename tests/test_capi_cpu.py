"""CPU tests of the boundary: the C-ABI library loads, exports every symbol the
header declares, and fails loudly without a GPU (no compute calls here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols(header="wfmash_hip.h", prefix="wfm_"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(" + prefix + r"[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from wfmash_amd import capi
    L = capi.load()
    decl = _declared_symbols()
    assert len(decl) >= 10
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/wfmash_hip.h but not exported"
    assert sorted(capi.EXPORTS) == decl
    host = _declared_symbols("wfmash_host.h", "wfmh_")
    assert sorted(capi.HOST_EXPORTS) == host
    for name in host:
        assert hasattr(L, name), f"{name} declared in include/wfmash_host.h but not exported"


def test_no_cpu_fallback():
    import torch
    from wfmash_amd import capi
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.WfmError):
        capi.Handle(0)


def test_struct_sizes_match_header():
    import ctypes as C
    from wfmash_amd import capi
    assert C.sizeof(capi.Penalties) == 20
    assert C.sizeof(capi.Result) == 32
    assert C.sizeof(capi.Minmer) == 32  # skch::MinmerInfo, base_types.hpp:28-35
    assert C.sizeof(capi.Problem) == 56


def test_product_never_imports_oracle():
    """The product path must not route through oracle/ (only tests, smoke and bench may)."""
    pkg = os.path.join(ROOT, "wfmash_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".hpp", ".c")):
                src = open(os.path.join(dp, f), errors="ignore").read()
                bad = re.findall(r"#include[^\n]*oracle|import\s+oracle|from\s+oracle|liboracle|oracle/_ref|pyoracle", src)
                assert not bad, (os.path.join(dp, f), bad)


def test_shard_records_balanced_and_deterministic():
    from wfmash_amd.dist import shard_records
    w = [((i * 37) % 11 + 1) ** 2 for i in range(100)]
    a = shard_records(w, 8)
    b = shard_records(w, 8)
    assert a == b
    assert sorted(i for s in a for i in s) == list(range(100))
    loads = [sum(w[i] for i in s) for s in a]
    assert max(loads) - min(loads) <= max(w)
