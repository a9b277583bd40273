"""The C-ABI library loads without a GPU and exports every symbol include/ccsm.h declares (no compute calls)."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def libpath():
    path = os.path.join(ROOT, "ccsmeth_amd", "lib", "libccsm.so")
    if not os.path.exists(path):
        subprocess.check_call(["python", "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)
    return path


def test_exports_match_header(libpath):
    header = open(os.path.join(ROOT, "include", "ccsm.h")).read()
    declared = set(re.findall(r"\b(ccsm_[a-z0-9_]+)\s*\(", header))
    declared -= {"ccsm_status"}
    lib = ctypes.CDLL(libpath)
    missing = [name for name in sorted(declared) if not hasattr(lib, name)]
    assert not missing, missing
    from ccsmeth_amd import _lib
    assert set(_lib.EXPORTS) == declared


def test_errors_without_touching_the_gpu(libpath):
    lib = ctypes.CDLL(libpath)
    lib.ccsm_last_error.restype = ctypes.c_char_p
    lib.ccsm_version.restype = ctypes.c_char_p
    assert b"libccsm" in lib.ccsm_version()
    out = ctypes.c_void_p()
    assert lib.ccsm_create(None, None, 0, ctypes.byref(out)) == 1          # CCSM_ERR_INVALID_ARG
    assert b"non-NULL" in lib.ccsm_last_error()
    from ccsmeth_amd import _lib
    cfg = _lib.Config(21, 3, 2, 256, 1, 0, 0, 0, b"transencoder2s", 0)
    w = _lib.Weights()
    assert lib.ccsm_create(ctypes.byref(cfg), ctypes.byref(w), 0, ctypes.byref(out)) == 2   # CCSM_ERR_UNSUPPORTED
    assert b"model_type" in lib.ccsm_last_error()
    assert lib.ccsm_group_pending(None) == 0 and lib.ccsm_debug_rows_padded(2048) == 4224   # padded to 192 strand rows: whole workgroups of 96, 64 and 32


def test_fp8_e4m3_host_encoder():
    """The weight packer's fp8 (OCP e4m3fn) encoder: every finite code round-trips, halfway cases round to even, values
    beyond the format saturate at +-448 (no NaN codes), tiny values flush to zero.  Host arithmetic only."""
    import numpy as np
    from ccsmeth_amd import _lib
    lib = _lib.load()

    def dec(c):
        s, e, m = c >> 7, (c >> 3) & 15, c & 7
        v = m * 2.0 ** -9 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7)
        return -v if s else v
    vals = {}
    for c in range(256):
        if (c & 0x7f) == 0x7f:
            continue                                        # NaN codes
        v = dec(c)
        got = lib.ccsm_debug_fp8_e4m3(v)
        assert dec(got) == v and (got == c or v == 0.0)
        vals[v] = c
    pos = sorted(v for v in vals if v >= 0)
    for lo, hi in zip(pos[:-1], pos[1:]):
        mid = 0.5 * (lo + hi)
        c = lib.ccsm_debug_fp8_e4m3(mid)
        assert dec(c) in (lo, hi) and (c & 1) == 0          # ties to even mantissa
        assert dec(lib.ccsm_debug_fp8_e4m3(np.nextafter(np.float32(mid), np.float32(hi)))) == hi
        assert dec(lib.ccsm_debug_fp8_e4m3(np.nextafter(np.float32(mid), np.float32(lo)))) == lo
    assert dec(lib.ccsm_debug_fp8_e4m3(1e9)) == 448.0 and dec(lib.ccsm_debug_fp8_e4m3(-500.0)) == -448.0
    assert dec(lib.ccsm_debug_fp8_e4m3(1e-6)) == 0.0


def test_bam_header_symbols_exported():
    """Every function include/ccsm_bam.h declares is exported by libccsm_bam.so (and listed in bamnative.EXPORTS)."""
    import ctypes
    import re
    from conftest import ROOT
    from ccsmeth_amd import bamnative
    hdr = open(os.path.join(ROOT, "include", "ccsm_bam.h")).read()
    declared = set(re.findall(r"\b(ccsm_bam_\w+)\s*\(", hdr))
    assert declared == set(bamnative.EXPORTS)
    lib = ctypes.CDLL(bamnative.LIB_PATH)
    for name in declared:
        getattr(lib, name)


def test_train_library_exports_match_header_and_reject_bad_arguments():
    """libccsm_train (include/ccsm_train.h): loads without a GPU, exports every declared symbol, NULL arguments are errors."""
    path = os.path.join(ROOT, "ccsmeth_amd", "lib", "libccsm_train.so")
    if not os.path.exists(path):
        subprocess.check_call(["python", "-c", "import __graft_entry__ as g; g.build()"], cwd=ROOT)
    header = open(os.path.join(ROOT, "include", "ccsm_train.h")).read()
    declared = set(re.findall(r"\b(ccsm_train_[a-z0-9_]+)\s*\(", header))
    from ccsmeth_amd import train
    assert set(train.EXPORTS) == declared
    lib = train.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.ccsm_train_num_params() == 3043114
    off = train.param_offsets()
    assert off[0] == 0 and off[1] == 40 and off[-1] == 3043114 and len(off) == 31 and off == sorted(off)
    sizes = [int(__import__("numpy").prod(train.PARAM_SHAPES[k])) for k in train.PARAM_NAMES]
    assert [b - a for a, b in zip(off[:-1], off[1:])] == sizes
    out = ctypes.c_void_p()
    assert lib.ccsm_train_create(None, 0, 16, None, ctypes.byref(out)) == 1          # CCSM_ERR_INVALID_ARG
    assert b"non-NULL" in lib.ccsm_train_last_error()
    assert lib.ccsm_train_step(None, 1e-3, 0.9, 0.999, 1e-8, 0.5, None) == 1
