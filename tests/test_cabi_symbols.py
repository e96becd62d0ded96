"""The C-ABI library loads on a CPU-only machine, exports every symbol the header
declares, and the product path refuses to compute without a GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT
from cellcomm_amd import _lib


def header_symbols():
    text = open(os.path.join(ROOT, 'include', 'cellcomm_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(cc_[a-z0-9_]+)\s*\(', text)))


def test_library_is_built():
    assert os.path.exists(_lib.lib_path()), 'run __graft_entry__.build() first'


def test_every_declared_symbol_is_exported_and_bound():
    handle = ctypes.CDLL(_lib.lib_path())
    names = header_symbols()
    assert len(names) >= 25
    for name in names:
        assert hasattr(handle, name), f'{name} declared in include/cellcomm_hip.h but not exported'
        assert name in _lib.SIGNATURES, f'{name} has no ctypes signature in cellcomm_amd/_lib.py'
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_string():
    handle = _lib.lib()
    assert handle.cc_version() >= 100
    assert isinstance(handle.cc_last_error(), bytes)


def test_no_cpu_fallback_without_gpu():
    handle = _lib.lib()
    if handle.cc_device_count() > 0:
        pytest.skip('a GPU is visible here')
    import cellcomm_amd as cc
    with pytest.raises(cc.HipUnavailable):
        cc.HipEncoder(3, 5)
    with pytest.raises(cc.HipUnavailable):
        cc.knn_graph([[0.0, 1.0], [1.0, 0.0], [2.0, 2.0]], 1)
    with pytest.raises(cc.HipUnavailable):
        cc.CellMatrix.from_dense([[1.0, 0.0]]).device_handle()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'cellcomm_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h')):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, flags=re.M), f
