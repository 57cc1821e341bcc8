"""Shared by tests/test_gpu_solver.py and its worker: the (problem, level, matrix) cases of the solver A/B test."""
import numpy as np
import scipy.sparse as sp

import mgb_amd as m

CASES = [("fem2d_P2", 3, 1.5, {}), ("fem2d_P2", 5, 3.5, {}), ("fem2d_P2", 5, 1.0, dict(max_coarse=40)),
         ("fem3d", 3, 1.5, {}),
         ("fem2d_P2", 7, 1.0, {})]       # large fronts: multi-workgroup path, LDS-sized fronts folded into it
GRADES = (0, 6, 12)           # cond(H) ~ 10^grade x the diagonally dominant core's


def build(fam, L, p, rs):
    geom = m.subdivide(m.fem2d_P2() if fam == "fem2d_P2" else m.fem3d(k=1), L)
    return m.assemble(m.amg(geom, prolongator=m.amg_ruge_stuben(**rs)) if rs else m.amg(geom), p=p)


def key_of(fam, L, p, rs, lev, grade):
    return f"{fam}_L{L}_p{p}_{'rs' if rs else 'def'}_lev{lev}_g{grade}"


def graded_spd(indptr, indices, rng, grade):
    """SPD matrix on a symmetric CSR pattern: a diagonally dominant core scaled by D = 10^(+-grade/2) on both sides."""
    mm = indptr.size - 1
    v = rng.standard_normal(indices.size)
    A = sp.csr_matrix((v, indices, indptr), shape=(mm, mm))
    A = (A + A.T) * 0.5
    A = A + sp.diags(np.asarray(abs(A).sum(axis=1)).ravel() + 1.0)
    d = 10.0 ** rng.uniform(-grade / 2, grade / 2, mm)
    A = sp.csr_matrix(sp.diags(d) @ A @ sp.diags(d))
    A.sort_indices()
    assert np.array_equal(A.indptr, indptr) and np.array_equal(A.indices, indices)
    return A


def matrices(P, seed=17):
    """Yield (level, grade, A, g) for every level of the device problem P, deterministically."""
    rng = np.random.default_rng(seed)
    for lev, msz in enumerate(P.level_sizes):
        indptr, indices = P.hessian_pattern(lev)
        for grade in GRADES:
            A = graded_spd(indptr, indices, rng, grade)
            yield lev, grade, A, A @ rng.standard_normal(msz)
