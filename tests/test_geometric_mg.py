"""`geometric_mg` hierarchies (reference: src/fem2d_P2.jl:468-596, src/TensorFEM.jl:888-954): the one
hierarchy the reference specifies completely in-repo, hence the place where prolongator INDEX MAPS can be
checked bit-exactly (BASELINE north_star).  The expected integer structure is derived here independently
of the package's sparse products: per fine element, follow the parent chain and multiply the boolean
patterns of the child-interpolation table."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

import mgb_amd as m
from mgb_amd import fem2d_p2 as fp
from mgb_amd import tensorfem as tf


def test_refine_table_is_the_reference_literal_and_nearly_the_exact_interpolant():
    T = fp.refine_table(True)
    assert T.shape == (28, 7) and np.allclose(T.sum(axis=1), 1.0, atol=1e-15)
    # exact evaluation of the P2+bubble basis at the child nodes (this package's rational-arithmetic basis)
    R = fp.reference_triangle(True)
    nodes = R["K"]                                             # barycentric rows
    lin = np.zeros((28, 3))                                    # barycentric coordinates of the child nodes in the parent
    for s_, cc in enumerate(fp._CHILD_CORNERS):
        lin[7 * s_:7 * s_ + 7] = nodes @ nodes[list(cc)]
    assert np.allclose(T @ nodes, lin, atol=1e-15)             # linears (hence straight-sided coordinates) are reproduced exactly
    l1, l2, l3 = lin.T
    b = 27 * l1 * l2 * l3
    P2 = np.stack([l1 * (2 * l1 - 1), 4 * l1 * l2, l2 * (2 * l2 - 1), 4 * l2 * l3, l3 * (2 * l3 - 1), 4 * l3 * l1], axis=1)
    cen = np.array([-1 / 9, 4 / 9, -1 / 9, 4 / 9, -1 / 9, 4 / 9])
    E = np.concatenate([P2 - b[:, None] * cen[None, :], b[:, None]], axis=1)
    diff = np.abs(T - E).max(axis=1)
    rounded = np.flatnonzero(diff > 1e-12)
    assert rounded.size == 6 and diff.max() < 4e-3             # the six rows with integer-rounded numerators
    assert np.allclose(T[np.setdiff1d(np.arange(28), rounded)], E[np.setdiff1d(np.arange(28), rounded)], atol=1e-15)
    src = "/root/reference/src/fem2d_P2.jl"
    if os.path.exists(src):                                     # this container only: the literal itself
        mm = re.search(r"refine = sparse\(\[([^\]]*)\], \[([^\]]*)\], T\[([^\]]*)\]\./648, 28, 7\)", open(src).read())
        Tr = np.zeros((28, 7))
        for r, c, v in zip(*(x.split(",") for x in mm.groups())):
            Tr[int(r) - 1, int(c) - 1] = float(v) / 648
        assert np.array_equal(Tr, T)
    P = fp.refine_table(False)
    assert P.shape == (24, 6) and np.allclose(P.sum(axis=1), 1.0)


def _expected_pattern_p2(geom0, L, level, table):
    """(rowptr, colidx) of R_dirichlet[level] from first principles: fine node r of fine element e depends on
    the nodes of e's level-`level` ancestor that a chain of child tables connects it to; those nodes map to
    continuous unknowns through the level's labels (boundary labels dropped)."""
    p = geom0.x.shape[0]
    t = geom0.t
    for _ in range(level):
        t = fp._refine_p2_connectivity(t)
    labels = t.T.reshape(-1)
    bd = fp._p2_boundary_dedup_set(labels, t.shape[1])
    interior = np.array(sorted(set(range(int(labels.max()) + 1)) - set(bd)))
    pos = -np.ones(int(labels.max()) + 1, dtype=np.int64)
    pos[interior] = np.arange(interior.size)
    B = (table != 0).reshape(4, p, p)
    Nl = t.shape[1]
    steps = L - 1 - level
    rows = []
    for e in range(Nl * 4 ** steps):
        ee = e
        chain = []                                              # child slots from the fine element up to the ancestor
        for _ in range(steps):
            chain.append(ee % 4)
            ee //= 4
        # boolean product from the ancestor down to the fine element: ancestor nodes -> ... -> fine nodes
        M = np.eye(p, dtype=bool)
        for ch in reversed(chain):
            M = (B[ch].astype(int) @ M.astype(int)) > 0
        anc = ee
        for r in range(p):
            cols = pos[t[np.flatnonzero(M[r]), anc]]
            rows.append(np.unique(cols[cols >= 0]))
    rowptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])])
    colidx = np.concatenate(rows) if rows else np.zeros(0, dtype=np.int64)
    return rowptr, colidx


@pytest.mark.parametrize("L", [2, 3])
def test_fem2d_P2_geometric_mg_index_maps_are_bit_exact(L):
    geom0 = m.fem2d_P2()
    mg = m.geometric_mg(geom0, L)
    fine = m.subdivide(geom0, L)
    assert np.array_equal(mg.geometry.t, fine.t)
    assert np.allclose(mg.geometry.x, fine.x, atol=1e-14) and np.allclose(mg.geometry.w, fine.w, atol=1e-15)
    n = fine.w.size
    assert len(mg.R["dirichlet"]) == L
    for level in range(L):
        R = sp.csr_matrix(mg.R["dirichlet"][level])
        R.sort_indices()
        rp, ci = _expected_pattern_p2(geom0, L, level, fp.refine_table(True))
        assert R.shape[0] == n and np.array_equal(R.indptr, rp) and np.array_equal(R.indices, ci)
        Rf = sp.csr_matrix(mg.R["full"][level])
        assert Rf.shape == (n, 7 * 2 * 4 ** level)
        assert np.allclose(np.asarray(sp.csr_matrix(mg.R["uniform"][level]).todense()), 1.0, atol=1e-13)
    fineR = sp.csr_matrix(mg.R["dirichlet"][L - 1])
    assert set(np.unique(fineR.data)) == {1.0} and fineR.getnnz(axis=1).max() == 1


def test_tensor_geometric_mg_shapes_and_nesting():
    # fem1d: 9 nodes, L = 3 (test/test_cuda.jl:35); fem3d k = 1, L = 2
    g1 = m.fem1d(nodes=np.linspace(-1.0, 1.0, 9))
    mg1 = m.geometric_mg(g1, 3)
    assert [R.shape for R in mg1.R["dirichlet"]] == [(64, 7), (64, 15), (64, 31)]
    assert [R.shape for R in mg1.R["full"]] == [(64, 16), (64, 32), (64, 64)]
    fine = m.subdivide(g1, 3)
    assert np.array_equal(mg1.geometry.t, fine.t) and np.allclose(mg1.geometry.x, fine.x)
    g3 = m.fem3d(k=1)
    mg3 = m.geometric_mg(g3, 2)
    assert mg3.R["dirichlet"][1].shape == (64, 1) and mg3.R["dirichlet"][0].shape[1] == 0    # one interior vertex after one refinement
    # nested spaces: every coarse continuous function is a fine continuous function
    for mg in (mg1, m.geometric_mg(m.fem2d_P2(), 3), m.geometric_mg(m.fem2d(k=2), 2)):
        Rd = [sp.csr_matrix(R) for R in mg.R["dirichlet"]]
        F = Rd[-1]
        for R in Rd[:-1]:
            if R.shape[1] == 0:
                continue
            coef = F.T @ R                                       # fine coefficients (F has orthogonal 0/1 columns up to counts)
            cnt = np.asarray(F.sum(axis=0)).ravel()
            lift = F @ sp.diags(1.0 / cnt) @ coef
            assert abs(lift - R).max() < 1e-12


def test_geometric_hierarchy_reproduces_the_goldens_through_the_oracle(golden):
    """The reference's goldens are hierarchy independent (test/test_algebraic.jl:24-31): the geometric ladder
    must reproduce them as well."""
    from helpers import gold_z
    from oracle import mgb_oracle as O
    c = golden["fem2d_P2_L2_p1"]
    prob = m.assemble(m.geometric_mg(m.fem2d_P2(), 2), p=1.0)
    assert np.linalg.norm(O.mgb_solve(prob)["z"] - gold_z(c)) < c["tol"]
