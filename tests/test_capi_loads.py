"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads without a GPU and
exports every symbol that include/pgw_hip.h declares (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'pgw_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(pgw_[a-z0-9_]+)\s*\(', txt)))


def test_header_symbols_are_bound_by_ctypes_layer():
    from pgw4era5_amd import _lib
    assert sorted(_lib.SIGNATURES) == _declared_symbols()


def test_library_loads_and_exports_every_symbol():
    from pgw4era5_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert b'gfx950' in lib.pgw_version()


def test_no_cpu_fallback_and_loud_failure(monkeypatch, tmp_path):
    from pgw4era5_amd import _lib
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'missing.so'))
    with pytest.raises(ImportError) as e:
        _lib.load()
    assert 'no CPU fallback' in str(e.value)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, 'pgw4era5_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
                assert 'pgw_oracle' not in src, f


def test_status_codes_map_to_reference_exceptions():
    from pgw4era5_amd import _lib

    class FakeLib:
        def pgw_last_error(self, h):
            return b'Extrapolation deactivated but data out of bounds.'

        def pgw_error_column(self, h):
            return 7
    old = _lib._lib
    _lib._lib = FakeLib()
    try:
        with pytest.raises(ValueError) as e:
            _lib.check(None, 12)
        assert str(e.value) == 'Extrapolation deactivated but data out of bounds.'
        with pytest.raises(KeyError):
            _lib.check(None, 14)
        with pytest.raises(ValueError) as e:
            _lib.check(None, 15)
        assert str(e.value) == ''
        with pytest.raises(_lib.PGWHipError):
            _lib.check(None, 1)
    finally:
        _lib._lib = old
