"""Accuracy of the coarse-level sums: device f1 / f2 on every level against an 80-bit accumulation of the oracle's
per-node terms (development only).  python tests/dev/gpu_sum_accuracy.py fem3d|fem2d L p"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, scipy.sparse as sp
import mgb_amd as m
from mgb_amd.device import DeviceMGBProblem
from oracle import mgb_oracle as O
fam, L, p = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
geo = m.fem3d(k=1) if fam == "fem3d" else m.fem2d_P2()
prob = m.assemble(m.amg(m.subdivide(geo, L)), p=p)
if fam == "fem3d":
    prob.g[:, 1] = 1.0e4
D = DeviceMGBProblem(prob); P = D.main
Mo, B = O.OracleAMG(prob.M[0]), O.Barrier(prob.Q)
z0 = np.ascontiguousarray(prob.g.T).reshape(-1); c = 0.1 * prob.f
rng = np.random.default_rng(1)
# push one node towards the wall so that the terms span many orders of magnitude
for lev in range(len(P.level_sizes)):
    R = Mo.R_fine[lev]
    s = 1e-3 * rng.standard_normal(R.shape[1])
    g_d = P.f1(lev, s, c, z0)
    g_o = B.f1(s, Mo.w, c, R, Mo.D_fine, z0)
    # 80-bit reference of the restriction: ret from the oracle, R' ret accumulated in long double
    Dz = O.apply_D(Mo.D_fine, z0 + R @ s)
    G = O.convex_eval(prob.Q, Dz, 1)
    Y = G / Mo.w.size + Mo.w[:, None] * c
    ret = np.zeros(R.shape[0])
    for k in range(len(Mo.D_fine)):
        ret = ret + Mo.D_fine[k].T @ Y[:, k]
    Rc = sp.csc_matrix(R)
    g_x = np.array([np.sum((Rc.data[Rc.indptr[j]:Rc.indptr[j + 1]].astype(np.longdouble)) *
                           ret[Rc.indices[Rc.indptr[j]:Rc.indptr[j + 1]]].astype(np.longdouble)) for j in range(R.shape[1])], dtype=np.float64)
    sc = np.abs(g_x).max()
    print(f"level {lev:2d} m={R.shape[1]:7d}  |g_dev - g_80|/max|g| = {np.abs(g_d - g_x).max()/sc:.2e}   |g_oracle - g_80| = {np.abs(g_o - g_x).max()/sc:.2e}", flush=True)
D.close()
