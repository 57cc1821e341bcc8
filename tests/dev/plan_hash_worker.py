"""Worker of tests/test_analysis_threads.py: hash of the multifrontal plan of a synthetic pattern (thread count from the environment)."""
import ctypes as C, os, sys, time, numpy as np, scipy.sparse as sp
lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "oracle", "_build", "libmf_host.so"))
lib.mf_host_plan_hash.restype = C.c_uint64
lib.mf_host_plan_hash.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
nx = int(sys.argv[1])
# 9-point stencil on an nx x nx grid plus a "slack" node hanging off every grid node (peelable leaves)
idx = np.arange(nx * nx).reshape(nx, nx)
rows, cols = [], []
for di in (-1, 0, 1):
    for dj in (-1, 0, 1):
        a = idx[max(0, -di):nx - max(0, di), max(0, -dj):nx - max(0, dj)]
        b = idx[max(0, di):nx - max(0, -di), max(0, dj):nx - max(0, -dj)]
        rows.append(a.ravel()); cols.append(b.ravel())
n0 = nx * nx
rows.append(np.arange(n0)); cols.append(n0 + np.arange(n0))
rows.append(n0 + np.arange(n0)); cols.append(np.arange(n0))
rows.append(n0 + np.arange(n0)); cols.append(n0 + np.arange(n0))
r = np.concatenate(rows); c = np.concatenate(cols)
A = sp.csr_matrix((np.ones(r.size), (r, c)), shape=(2 * n0, 2 * n0)); A.sum_duplicates(); A.sort_indices()
ip = A.indptr.astype(np.int32); ii = A.indices.astype(np.int32)
t = time.time()
h = lib.mf_host_plan_hash(A.shape[0], ip.ctypes.data, ii.ctypes.data, 1)
print(A.shape[0], hex(h), round(time.time() - t, 3))
