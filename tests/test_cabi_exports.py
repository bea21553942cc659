"""CPU: the C-ABI library loads and exports every symbol include/devqa.h declares (no compute)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    so = os.path.join(ROOT, "de-vqa_amd", "csrc", "libdevqa_hip.so")
    if not os.path.exists(so):
        g.build()
    return so


def test_header_symbols_exported(built):
    hdr = open(os.path.join(ROOT, "include", "devqa.h")).read()
    names = set(re.findall(r"\b(devqa_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 15
    lib = ctypes.CDLL(built)
    for n in sorted(names):
        assert hasattr(lib, n), "missing export %s" % n
    lib.devqa_abi_version.restype = ctypes.c_int
    assert lib.devqa_abi_version() == 1


def test_binding_covers_header(built):
    import devqa_amd  # noqa: F401
    from devqa_amd import lib
    hdr = open(os.path.join(ROOT, "include", "devqa.h")).read()
    names = set(re.findall(r"\b(devqa_[a-z0-9_]+)\s*\(", hdr))
    assert names == set(lib.EXPORTS)
    lib.load()


def test_argument_validation_without_gpu(built):
    """Host-side shape checks run before any launch, so they are testable on CPU."""
    lib = ctypes.CDLL(built)
    lib.devqa_last_error.restype = ctypes.c_char_p
    rc = lib.devqa_gemm_bf16(None, 0, None, 0, None, 1, 1, 8, ctypes.c_float(1.0), 0, None, None, None, 0, None)
    assert rc == -1 and b"null" in lib.devqa_last_error()
