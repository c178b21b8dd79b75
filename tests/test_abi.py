'''CPU-side checks of the drop-in boundary: the shared library loads and exports
every symbol include/nutils_hip.h declares (no compute calls without a GPU), and
the ctypes signature table covers exactly that set.'''
import ctypes
import os
import re
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared():
    src = open(os.path.join(ROOT, 'include', 'nutils_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(nh_[a-z0-9_]+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from nutils_amd import _lib
    assert os.path.exists(_lib.LIBPATH), 'run __graft_entry__.build() first'
    lib = ctypes.CDLL(_lib.LIBPATH)
    names = declared()
    assert len(names) >= 20
    for name in names:
        assert hasattr(lib, name), name
    assert lib.nh_abi_version() == 1


def test_ctypes_table_matches_header():
    from nutils_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared()


def test_error_reporting_without_gpu():
    from nutils_amd import _lib
    lib = _lib.load()
    # an invalid argument is reported through the status code + nh_last_error, never a crash
    rc = lib.nh_poly_tabulate(None, 1, 7, None, 1, 3, None, None)
    assert rc == -1
    assert b'not a valid coefficient count' in lib.nh_last_error()
    with pytest.raises(_lib.NutilsHipError):
        _lib.check(rc)
