"""ctypes binding of lib/libmgbsetup.so (csrc/setup_host.cpp): the sequential loops of the hierarchy construction in C++.
`available()` is False when the library has not been built; callers then run their pure-Python twins (same results, bit for bit:
tests/test_setup.py).  MGB_SETUP_PYTHON=1 forces the twins."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_LIB = None
_TRIED = False


def _lib():
    global _LIB, _TRIED
    if _TRIED:
        return _LIB
    _TRIED = True
    if os.environ.get("MGB_SETUP_PYTHON") == "1":
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmgbsetup.so")
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    ip, dp, bp = C.POINTER(C.c_int32), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    lib.mgbsetup_rs_cf_splitting.argtypes = [C.c_int64, ip, ip, ip, ip, C.c_int32, bp]
    lib.mgbsetup_csr_sort_rows.argtypes = [C.c_int64, ip, ip, dp]
    _LIB = lib
    return lib


def available() -> bool:
    return _lib() is not None


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def rs_cf_splitting(Sp, Sj, Tp, Tj, diag_quirk: bool) -> np.ndarray:
    """C/F splitting of amg_prolongators._rs_cf_splitting; S = (Sp, Sj) sorted rows, T = S'.  Returns a bool array."""
    lib = _lib()
    n = len(Sp) - 1
    Sp, Sj, Tp, Tj = _i32(Sp), _i32(Sj), _i32(Tp), _i32(Tj)
    out = np.zeros(max(n, 1), dtype=np.uint8)
    ip, bp = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
    rc = lib.mgbsetup_rs_cf_splitting(n, Sp.ctypes.data_as(ip), Sj.ctypes.data_as(ip), Tp.ctypes.data_as(ip),
                                      Tj.ctypes.data_as(ip), int(bool(diag_quirk)), out.ctypes.data_as(bp))
    if rc != 0:
        raise RuntimeError("mgbsetup_rs_cf_splitting: bad arguments")
    return out[:n].astype(bool)


def csr_sort_rows(M) -> bool:
    """Sort the rows of a scipy CSR matrix in place (int32 indices, float64 data, C-contiguous); False if the layout does
    not qualify and nothing was done."""
    lib = _lib()
    if lib is None or M.indices.dtype != np.int32 or M.indptr.dtype != np.int32 or M.data.dtype != np.float64:
        return False
    if not (M.indices.flags.c_contiguous and M.data.flags.c_contiguous and M.indptr.flags.c_contiguous
            and M.indices.flags.writeable and M.data.flags.writeable):
        return False
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    rc = lib.mgbsetup_csr_sort_rows(M.shape[0], M.indptr.ctypes.data_as(ip), M.indices.ctypes.data_as(ip), M.data.ctypes.data_as(dp))
    if rc != 0:
        raise RuntimeError("mgbsetup_csr_sort_rows: bad arguments")
    M.has_sorted_indices = True
    return True
