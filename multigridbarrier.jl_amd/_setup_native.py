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


class _Csr(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("ptr", C.POINTER(C.c_int32)), ("idx", C.POINTER(C.c_int32)),
                ("val", C.POINTER(C.c_double))]


class _Out(C.Structure):
    _fields_ = [("nnz", C.c_int64), ("ptr", C.POINTER(C.c_int32)), ("idx", C.POINTER(C.c_int32)), ("val", C.POINTER(C.c_double))]


def _adopt(lib, ptr, count, ctype, dtype):
    """NumPy array over a malloc'ed buffer of the library; the buffer is released when the last view of it dies."""
    import weakref
    addr = C.cast(ptr, C.c_void_p).value
    buf = (ctype * max(count, 1)).from_address(addr)
    weakref.finalize(buf, lib.mgbsetup_free, C.c_void_p(addr))
    return np.frombuffer(buf, dtype=dtype)[:count]


def compose_chain(A0, factors):
    """[A0 @ B1, (A0 @ B1) @ B2, ...] with sorted rows, every product formed by scipy's own `csr_matmat` loop restated in
    C++ on the previous product in ITS storage order (see csrc/setup_host.cpp) -- entry for entry what
    `C = C @ B; D = C.copy(); D.sort_indices()` gives.  One library call: product k + 1 is formed while product k is copied
    out and sorted on a second thread.  Returns None when the library or the operands do not qualify."""
    import scipy.sparse as sp
    lib = _lib()
    if lib is None or not _csr_ok(A0) or not all(_csr_ok(B) for B in factors):
        return None
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    lib.mgbsetup_chain_create.restype = C.c_void_p
    lib.mgbsetup_chain_create.argtypes = [C.c_int64, C.c_int64, ip, ip, dp]
    lib.mgbsetup_chain_destroy.argtypes = [C.c_void_p]
    lib.mgbsetup_chain_run.restype = C.c_int64
    lib.mgbsetup_chain_run.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_Csr), C.POINTER(_Out)]
    lib.mgbsetup_free.argtypes = [C.c_void_p]
    lib.mgbsetup_free.restype = None
    rows = A0.shape[0]
    z32, z64 = np.zeros(1, dtype=np.int32), np.zeros(1)
    keep = []

    def arrs(M):
        idx = M.indices if M.nnz else z32
        val = M.data if M.nnz else z64
        keep.extend([M.indptr, idx, val])
        return M.indptr.ctypes.data_as(ip), idx.ctypes.data_as(ip), val.ctypes.data_as(dp)

    h = lib.mgbsetup_chain_create(rows, A0.shape[1], *arrs(A0))
    if not h:
        return None
    K = len(factors)
    fac = (_Csr * max(K, 1))()
    outs = (_Out * max(K, 1))()
    for k, B in enumerate(factors):
        fac[k].rows, fac[k].cols = B.shape
        fac[k].ptr, fac[k].idx, fac[k].val = arrs(B)
    try:
        rc = lib.mgbsetup_chain_run(h, K, fac, outs)
    finally:
        lib.mgbsetup_chain_destroy(h)
    mats = []
    for k in range(K):                                   # adopt every buffer that was handed out, also on failure
        if not outs[k].ptr:
            continue
        nnz = int(outs[k].nnz)
        indptr = _adopt(lib, outs[k].ptr, rows + 1, C.c_int32, np.int32)
        indices = _adopt(lib, outs[k].idx, nnz, C.c_int32, np.int32)
        data = _adopt(lib, outs[k].val, nnz, C.c_double, np.float64)
        M = sp.csr_matrix((data, indices, indptr), shape=(rows, factors[k].shape[1]), copy=False)
        M.has_sorted_indices = True
        M.has_canonical_format = True                      # sorted and duplicate-free by construction: spares scipy's scans
        mats.append(M)
    if rc != 0 or len(mats) != K:
        return None
    return mats


def csr_row_sums(M):
    """`np.asarray(M.sum(axis=1)).ravel()` of a CSR matrix with numpy's own association of the terms (see setup_host.cpp), or
    None when the library or the layout does not qualify."""
    lib = _lib()
    if lib is None or not _csr_ok(M):
        return None
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    lib.mgbsetup_csr_row_sums.argtypes = [C.c_int64, ip, dp, dp]
    out = np.empty(M.shape[0])
    val = M.data if M.nnz else np.zeros(1)
    if lib.mgbsetup_csr_row_sums(M.shape[0], M.indptr.ctypes.data_as(ip), val.ctypes.data_as(dp), out.ctypes.data_as(dp)) != 0:
        return None
    return out


def blockdiag(mats):
    """scipy CSR block diagonal of CSR blocks (sorted rows assumed by the caller), or None when the library or a block does not
    qualify.  The call runs without the interpreter lock: device.py builds the levels of an AMG on a thread pool."""
    import scipy.sparse as sp
    lib = _lib()
    if lib is None or not mats or not all(_csr_ok(M) for M in mats):
        return None
    ip, dp = C.POINTER(C.c_int32), C.POINTER(C.c_double)
    lib.mgbsetup_blockdiag.argtypes = [C.c_int32, C.POINTER(_Csr), ip, ip, dp]
    rows = sum(M.shape[0] for M in mats)
    cols = sum(M.shape[1] for M in mats)
    nnz = sum(int(M.indptr[-1]) for M in mats)
    if max(rows + 1, cols, nnz) >= 2**31 - 1:
        return None
    z32, z64 = np.zeros(1, dtype=np.int32), np.zeros(1)
    blk = (_Csr * len(mats))()
    keep = []
    for k, M in enumerate(mats):
        idx = M.indices if M.nnz else z32
        val = M.data if M.nnz else z64
        keep.extend([M.indptr, idx, val])
        blk[k].rows, blk[k].cols = M.shape
        blk[k].ptr, blk[k].idx, blk[k].val = M.indptr.ctypes.data_as(ip), idx.ctypes.data_as(ip), val.ctypes.data_as(dp)
    indptr = np.empty(rows + 1, dtype=np.int32)
    indices = np.empty(max(nnz, 1), dtype=np.int32)
    data = np.empty(max(nnz, 1), dtype=np.float64)
    if lib.mgbsetup_blockdiag(len(mats), blk, indptr.ctypes.data_as(ip), indices.ctypes.data_as(ip), data.ctypes.data_as(dp)) != 0:
        return None
    out = sp.csr_matrix((data[:nnz], indices[:nnz], indptr), shape=(rows, cols), copy=False)
    out.has_sorted_indices = True
    return out
