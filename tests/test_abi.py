"""The C-ABI library loads and exports every symbol include/sqe.h declares (no compute
calls: this runs without a GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "sqe.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sqe_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    so = os.path.join(ROOT, "semantic_query_engine_amd", "libsqe.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    from semantic_query_engine_amd import _native
    return _native.load()


def test_header_symbols_are_exported_and_bound(lib):
    from semantic_query_engine_amd import _native
    declared = _declared_symbols()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in sqe.h but not exported"
    assert sorted(_native.SIGNATURES) == declared      # the ctypes table covers the header exactly


def test_version_and_error_string(lib):
    assert lib.sqe_version() == 100
    assert isinstance(lib.sqe_last_error(), bytes)


def test_struct_layouts_match_header():
    from semantic_query_engine_amd import _native
    assert ctypes.sizeof(_native.BertCfg) == 32
    assert ctypes.sizeof(_native.Stats) == 6 * 8 + 6 * 8 + 4 * 8      # + sample_ms and the three int8 counters (r03)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import importlib
    from semantic_query_engine_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(ImportError):
        _native.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "semantic_query_engine_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), encoding="utf-8").read()
                assert "oracle" not in text.replace("no CPU fallback", ""), f"{f} mentions the oracle"
