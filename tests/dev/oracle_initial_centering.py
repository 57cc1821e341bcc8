"""Exact Newton counts of the oracle's initial centring (the first mgb_step, src/mgb.jl:124-125) on the
reference-default ladder; development only.  python tests/dev/oracle_initial_centering.py L p"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, math
import mgb_amd as m
from oracle import mgb_oracle as O
L, p = int(sys.argv[1]), float(sys.argv[2])
prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
M = O.OracleAMG(prob.M[0])
print("ladder", [R.shape[1] for R in M.R_fine], flush=True)
z = np.ascontiguousarray(prob.g.T).reshape(-1).copy()
t0 = time.time()
n = M.w.size
SOL = O.mgb_step(prob.Q, M, z, 0.1 * prob.f, maxit=10000, max_newton=int(math.ceil(math.log2(-math.log2(O.EPS)) + 2)),
                 line_search=O.linesearch_backtracking(), stopping_criterion=O.stopping_inexact(0.25 / math.sqrt(n), 0.9),
                 finalize=O.NoFinalize(), initial_step=True)
print("converged", SOL["converged"], "its per level", SOL["its"].tolist(), f"{time.time()-t0:.0f}s", flush=True)
