"""north_star: "bit-exact for prolongator index maps" -- on the DEVICE (-m gpu).

`geometric_mg` is the one hierarchy the reference specifies completely in-repo (src/fem2d_P2.jl:468-596); its level ->
fine transfers are 0/1 selections and products of the element-local child table.  tests/test_geometric_mg.py derives
their index maps independently on the CPU; here the maps are read back from what the device actually holds: every column
of every R_fine[l] is recovered through the C ABI (`mgbhip_prolong_add` with unit vectors on device-resident vectors:
z = R e_j) and must equal (i) the uploaded matrix bit for bit -- indices AND values -- and (ii) on the :dirichlet block
the independently derived (rowptr, colidx)."""
import numpy as np
import pytest
import scipy.sparse as sp

import mgb_amd as m
from mgb_amd import fem2d_p2 as fp
from test_geometric_mg import _expected_pattern_p2

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("L", [2, 3])
def test_device_prolongators_have_the_reference_index_maps_bit_for_bit(L):
    from mgb_amd.device import DeviceMGBProblem
    geom0 = m.fem2d_P2()
    mg = m.geometric_mg(geom0, L)
    prob = m.assemble(mg, p=1.5)
    D = DeviceMGBProblem(prob)
    try:
        P = D.main
        n = prob.M[0].w.size
        assert len(P.level_sizes) == L
        for level in range(L):
            Rh = sp.csc_matrix(prob.M[0].R_fine[level])
            mJ = P.level_sizes[level]
            assert Rh.shape == (2 * n, mJ)
            cols = []
            for j in range(mJ):
                s = np.zeros(mJ)
                s[j] = 1.0
                z = P.prolong_add(level, P.vec(s), P.vec(np.zeros(2 * n))).to_host()
                cols.append(sp.csc_matrix(z.reshape(-1, 1)))
            Rd = sp.hstack(cols, format="csc")
            Rd.sort_indices(); Rh.sort_indices(); Rh.eliminate_zeros()
            assert np.array_equal(Rd.indptr, Rh.indptr) and np.array_equal(Rd.indices, Rh.indices)      # index maps: integer equality
            assert np.array_equal(Rd.data, Rh.data)                                                     # entries: bitwise
            # the :dirichlet block (rows of u, the first columns) against the independent derivation
            nd = sp.csr_matrix(mg.R["dirichlet"][level]).shape[1]
            Rdir = sp.csr_matrix(Rd[:n, :nd])
            Rdir.sort_indices()
            rp, ci = _expected_pattern_p2(geom0, L, level, fp.refine_table(True))
            assert np.array_equal(Rdir.indptr, rp) and np.array_equal(Rdir.indices, ci)
        fine = sp.csr_matrix(prob.M[0].R_fine[L - 1])
        assert set(np.unique(fine.data)) == {1.0} and fine.getnnz(axis=1).max() == 1                     # the fine level is a selection
    finally:
        D.close()
