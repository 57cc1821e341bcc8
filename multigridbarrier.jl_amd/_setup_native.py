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


def _csr_ok(M) -> bool:
    import scipy.sparse as sp
    return (sp.issparse(M) and M.format == "csr" and M.dtype == np.float64 and M.indices.dtype == np.int32
            and M.indptr.dtype == np.int32 and max(M.shape) < 2**31 - 1)


def compose_chain(A0, factors):
    """[A0 @ B1, (A0 @ B1) @ B2, ...] with sorted rows, every product formed by scipy's own `csr_matmat` loop restated in
    C++ on the previous product in ITS storage order (see csrc/setup_host.cpp) -- entry for entry what
    `C = C @ B; D = C.copy(); D.sort_indices()` gives.  Returns None when the library or the operands do not qualify."""
    import scipy.sparse as sp
    lib = _lib()
    if lib is None or not _csr_ok(A0) or not all(_csr_ok(B) for B in factors):
        return None
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    lib.mgbsetup_chain_create.restype = C.c_void_p
    lib.mgbsetup_chain_create.argtypes = [C.c_int64, C.c_int64, ip, ip, dp]
    lib.mgbsetup_chain_destroy.argtypes = [C.c_void_p]
    lib.mgbsetup_chain_multiply.restype = C.c_int64
    lib.mgbsetup_chain_multiply.argtypes = [C.c_void_p, C.c_int64, C.c_int64, ip, ip, dp]
    lib.mgbsetup_chain_emit_sorted.argtypes = [C.c_void_p, ip, ip, dp]
    rows = A0.shape[0]
    idx0 = A0.indices if A0.nnz else np.zeros(1, dtype=np.int32)
    val0 = A0.data if A0.nnz else np.zeros(1)
    h = lib.mgbsetup_chain_create(rows, A0.shape[1], A0.indptr.ctypes.data_as(ip), idx0.ctypes.data_as(ip), val0.ctypes.data_as(dp))
    if not h:
        return None
    out = []
    try:
        for B in factors:
            bj = B.indices if B.nnz else np.zeros(1, dtype=np.int32)
            bx = B.data if B.nnz else np.zeros(1)
            nnz = lib.mgbsetup_chain_multiply(h, B.shape[0], B.shape[1], B.indptr.ctypes.data_as(ip), bj.ctypes.data_as(ip), bx.ctypes.data_as(dp))
            if nnz < 0:
                return None
            indptr = np.empty(rows + 1, dtype=np.int32)
            indices = np.empty(max(nnz, 1), dtype=np.int32)
            data = np.empty(max(nnz, 1), dtype=np.float64)
            if lib.mgbsetup_chain_emit_sorted(h, indptr.ctypes.data_as(ip), indices.ctypes.data_as(ip), data.ctypes.data_as(dp)) != 0:
                return None
            M = sp.csr_matrix((data[:nnz], indices[:nnz], indptr), shape=(rows, B.shape[1]), copy=False)
            M.has_sorted_indices = True
            out.append(M)
    finally:
        lib.mgbsetup_chain_destroy(h)
    return out
