"""Row N1: evaluation of a smoother-based iterative linear solve for the Newton systems.

Not a product feature: this file is the committed evidence for the decision recorded in DESIGN.md (section "N1").
It builds, with SciPy on the CPU and the oracle's Hessians, exactly the scheme SURVEY.md section 7 "Stage B"
describes -- eliminate the slack unknowns exactly (H_ss is diagonal), precondition CG on the Schur complement
with a two-level cycle on the hierarchy's own subspaces (Jacobi smoothing + exact coarse solve) -- and pins
what was observed:

* p = 1.5: the PCG direction solves the system to 1e-8 and reproduces the Newton decrement to 1e-8 (it is a
  valid inexact-Newton direction), but it
  takes tens of iterations with the exact Galerkin coarse operator and ~100 with the coarse operator the
  device can form cheaply (Schur complement of the coarse-level Hessian); at >= 3 sparse products per
  iteration that is far above the cost of one device LDL' factor + solve (about 3 ms at L = 9).
* p = 1.0 (the headline configuration) near the end of the central path: hundreds of iterations already on
  this 1.2k-unknown mesh (259 when this was written) and no convergence within 500 one level finer (L = 5,
  measured while writing this file) -- the barrier Hessian's conditioning grows like 1/tol^2 there.  The
  reference has no iterative solver for the same reason; the direct solve stays.
"""
import numpy as np
import pytest
import scipy.sparse as sp
import scipy.sparse.linalg as spla

import mgb_amd as m
from oracle import mgb_oracle as O


def _central_point(p, L=4):
    prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
    M = O.OracleAMG(prob.M[0])
    B = O.Barrier(prob.Q)
    sol = O.mgb_solve(prob)
    z = np.ascontiguousarray(np.asarray(sol["z"]).T).reshape(-1)
    J = len(M.R_fine) - 1
    R = sp.csr_matrix(M.R_fine[J])
    s = np.zeros(R.shape[1])
    H = sp.csr_matrix(B.f2(s, M.w, prob.f, R, M.D_fine, z))
    g = B.f1(s, M.w, prob.f, R, M.D_fine, z)
    n = prob.g.shape[0]
    Rc = R.tocsc()
    is_u = np.array([Rc.indices[Rc.indptr[j]] < n for j in range(R.shape[1])])
    mu = int(is_u.sum())
    assert is_u[:mu].all()            # unknowns are ordered [u | s]
    return M, R, H, g, mu, J


def _two_level_pcg(M, R, H, g, mu, kc, galerkin, maxiter):
    """CG on S = H_uu - H_us H_ss^{-1} H_su with a symmetric two-level preconditioner on level kc."""
    Huu, Hus, Hss = H[:mu][:, :mu], H[:mu][:, mu:], H[mu:][:, mu:]
    assert abs(Hss - sp.diags(Hss.diagonal())).max() == 0.0          # the slack block is diagonal
    dss = Hss.diagonal()
    S = sp.csr_matrix(Huu - Hus @ sp.diags(1.0 / dss) @ Hus.T)
    gs = g[:mu] - Hus @ (g[mu:] / dss)
    D = np.asarray(R.multiply(R).sum(axis=0)).ravel()
    P = sp.csr_matrix(sp.diags(1.0 / D) @ (R.T @ sp.csr_matrix(M.R_fine[kc])))   # level kc -> fine unknowns (nested)
    assert abs(R @ P - sp.csr_matrix(M.R_fine[kc])).max() < 1e-12
    Pu = sp.csr_matrix(P[:mu])
    if galerkin:
        keep = np.nonzero(np.asarray(abs(Pu).sum(axis=0)).ravel() > 0)[0]
        Pg = Pu[:, keep]
        lu = spla.splu(sp.csc_matrix(Pg.T @ S @ Pg))
        coarse = lambda r: Pg @ lu.solve(Pg.T @ r)
    else:       # what the device has: the assembled coarse-level Hessian P' H P (both unknown groups), solved directly
        lu = spla.splu(sp.csc_matrix(P.T @ H @ P))
        def coarse(r):
            rhs = P.T @ np.concatenate([r, np.zeros(H.shape[0] - mu)])
            return (P @ lu.solve(rhs))[:mu]
    dinv = 0.6 / S.diagonal()

    def prec(r):
        x = dinv * r
        x = x + coarse(r - S @ x)
        return x + dinv * (r - S @ x)

    its = [0]
    x, _ = spla.cg(S, gs, rtol=1e-10, maxiter=maxiter, M=spla.LinearOperator(S.shape, matvec=prec),
                   callback=lambda xk: its.__setitem__(0, its[0] + 1))
    xs = (g[mu:] - Hus.T @ x) / dss
    return np.concatenate([x, xs]), its[0]


def test_pcg_direction_matches_direct_solve_p15_and_costs_tens_of_iterations():
    M, R, H, g, mu, J = _central_point(1.5)
    x_direct = spla.spsolve(sp.csc_matrix(H), g)
    x_gal, its_gal = _two_level_pcg(M, R, H, g, mu, J - 2, True, 400)
    x_dev, its_dev = _two_level_pcg(M, R, H, g, mu, J - 2, False, 400)
    lam2 = float(g @ x_direct)                      # the Newton decrement the stopping rule reads
    for x in (x_gal, x_dev):
        assert np.linalg.norm(H @ x - g) <= 1e-8 * np.linalg.norm(g)
        assert abs(float(g @ x) - lam2) <= 1e-8 * abs(lam2)
    print(f"N1 evaluation p=1.5: PCG iterations {its_gal} (Galerkin coarse), {its_dev} (coarse-level Hessian)")
    assert 5 <= its_gal <= 120 and its_gal <= its_dev <= 400       # tens of iterations: never competitive with 3 ms direct


def test_pcg_needs_hundreds_of_iterations_on_the_headline_p1_system():
    M, R, H, g, mu, J = _central_point(1.0)
    x, its = _two_level_pcg(M, R, H, g, mu, J - 2, True, 600)
    rel = np.linalg.norm(H @ x - g) / np.linalg.norm(g)
    print(f"N1 evaluation p=1.0: {its} iterations, relative residual {rel:.1e}")
    assert its >= 150          # the evidence: two orders of magnitude above what one direct solve costs
