"""Worker of tests/test_analysis_threads.py: hash of the multifrontal plan of the fine-level Hessian pattern of fem2d_P2 at level L."""
import ctypes as C, os, sys, time, numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import mgb_amd as m
from oracle import mgb_oracle as O
from helpers import stacked
lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "libmf_host.so"))
lib.mf_host_plan_hash.restype = C.c_uint64
lib.mf_host_plan_hash.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
L = int(sys.argv[1])
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=1.0)
Mo = O.OracleAMG(prob.M[0]); B = O.Barrier(prob.Q)
R = Mo.R_fine[-1]
H = sp.csr_matrix(B.f2(np.zeros(R.shape[1]), Mo.w, 0.1 * prob.f, R, Mo.D_fine, stacked(prob.g)))
H.sum_duplicates(); H.sort_indices()
ip = H.indptr.astype(np.int32); ii = H.indices.astype(np.int32)
for leaves in (0, 1):
    t = time.time(); h = lib.mf_host_plan_hash(H.shape[0], ip.ctypes.data, ii.ctypes.data, leaves)
    print(L, H.shape[0], H.nnz, "flag", leaves, hex(h), round(time.time() - t, 3))
