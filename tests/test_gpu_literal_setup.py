"""Device vs oracle vs the reference's golden vector on a problem whose SETUP IS WRITTEN OUT BY HAND (-m gpu).

Everywhere else in this suite the device and the oracle consume the same `multigridbarrier.jl_amd` setup layer
(`m.assemble(m.amg(...))`), so a setup bug would be common-mode and invisible ("what's weak" #2).  Here
nothing comes from that layer: the operator blocks, weights, hierarchy matrices and cone grids below are literals typed
from the reference's definitions --

  * fem1d, k = 1 on the nodes (-1, 0, 1): two elements, two nodes each, broken basis (src/TensorFEM.jl:199-219, :428-490);
    on an element of length h the k = 1 Clenshaw-Curtis weights are h/2, h/2 and d/dx is [[-1, 1], [-1, 1]] / h;
  * state_variables = [:u :dirichlet; :s :full], D = [:u :id; :u :dx; :s :id] (src/mgb.jl:595-607);
  * `R_fine[l] = blockdiag(R[:dirichlet][l], R[:full][l])` (src/multigrid.jl:474-512): the single interior vertex
    carries u; s lives on (constants) -> (continuous P1: 3 values) -> (broken: 4 values);
  * f = (0.5, 0, 1), g = (x, 2), Q = power cone s >= |u'|^p with idx = (2, 3) (src/mgb.jl:587-613).

The reference's own test holds the answer for p = 1 (test/runtests.jl:13-16, `tests/golden/golden.json`:
z = [-1 -1 -1 1; 0 0 2 2]'), so the literal problem is pinned end to end without either setup layer.  A second, non-uniform
literal mesh (-1, 0.25, 1), p = 1.5, has no golden: device vs oracle at north_star's 1e-10."""
import numpy as np
import pytest
import scipy.sparse as sp

from helpers import assert_z_close, literal_fem1d_problem as _literal_problem, record_observation, stacked
from oracle import mgb_oracle as O

pytestmark = pytest.mark.gpu

KERNEL_RTOL = 1e-10


def _primitives(prob):
    from mgb_amd.device import DeviceMGBProblem
    D = DeviceMGBProblem(prob)
    try:
        Mo, B = O.OracleAMG(prob.M[0]), O.Barrier(prob.Q)
        z0, c = stacked(prob.g), 0.1 * prob.f
        rng = np.random.default_rng(5)
        for J, R in enumerate(Mo.R_fine):
            s = 1e-2 * rng.standard_normal(R.shape[1])
            y_o, g_o = B.f0(s, Mo.w, c, R, Mo.D_fine, z0), B.f1(s, Mo.w, c, R, Mo.D_fine, z0)
            H_o = np.asarray(sp.csr_matrix(B.f2(s, Mo.w, c, R, Mo.D_fine, z0)).todense())
            assert abs(D.main.f0(J, s, c, z0) - y_o) <= KERNEL_RTOL * abs(y_o)
            g_d = D.main.f1(J, s, c, z0)
            assert np.linalg.norm(g_d - g_o) <= KERNEL_RTOL * np.linalg.norm(g_o)
            H_d = np.asarray(D.main.f2(J, s, c, z0).todense())
            assert np.abs(H_d - H_o).max() <= KERNEL_RTOL * np.abs(H_o).max()
            x_d = D.main.solve(J, g_d)
            assert np.linalg.norm(H_o @ x_d - g_o) <= 1e-12 * np.linalg.norm(g_o)
    finally:
        D.close()


def test_hand_built_fem1d_three_nodes_reproduces_the_reference_golden(golden):
    import mgb_amd as m
    prob = _literal_problem((-1.0, 0.0, 1.0), 1.0)
    _primitives(prob)
    sol = m.mgb_solve(prob)
    case = golden["fem1d_3nodes_p1"]                                   # test/runtests.jl:13-16
    z_gold = np.array(case["z_colmajor"]).reshape(case["ncols"], -1).T
    err = float(np.linalg.norm(sol.z - z_gold))
    record_observation(f"literal fem1d 3 nodes p=1 vs reference golden: |dz|_2 {err:.2e} (reference tolerance {case['tol']:.0e})")
    assert err < case["tol"]
    ref = O.mgb_solve(prob)
    assert_z_close(sol.z, ref["z"], "literal fem1d 3 nodes p=1")
    assert np.array_equal(np.asarray(sol.SOL_main["its"])[:, :-1], np.asarray(ref["SOL_main"]["its"])[:, :-1])


def test_hand_built_nonuniform_mesh_device_vs_oracle():
    import mgb_amd as m
    prob = _literal_problem((-1.0, 0.25, 1.0), 1.5)
    _primitives(prob)
    sol = m.mgb_solve(prob)
    ref = O.mgb_solve(prob)
    assert_z_close(sol.z, ref["z"], "literal non-uniform fem1d p=1.5")
    assert np.array_equal(np.asarray(sol.SOL_main["its"])[:, :-1], np.asarray(ref["SOL_main"]["its"])[:, :-1])


def test_hand_built_mesh_infeasible_start_runs_phase1_like_the_oracle():
    """Start with s = 0.5 < |u'| on the long element: the hand-written phase-I image (three states, six operator rows,
    src/multigrid.jl:515-538) is what the feasibility solve runs on."""
    import mgb_amd as m
    prob = _literal_problem((-1.0, 0.25, 1.0), 1.5)
    prob.g[:, 1] = 0.5
    sol = m.mgb_solve(prob)
    ref = O.mgb_solve(prob)
    assert sol.SOL_feasibility is not None and ref["SOL_feasibility"] is not None
    assert_z_close(sol.z, ref["z"], "literal non-uniform fem1d p=1.5, phase I")
