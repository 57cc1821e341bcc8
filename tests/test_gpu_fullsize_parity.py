"""Oracle-vs-device parity AT THE SIZES THE BENCH REPORTS (-m gpu).

VERDICT r3 "what's weak" #2: per-kernel parity used to run at L <= 3 only; BASELINE configs 2, 3 and 4 at full size
were covered by size-independent properties.  One oracle evaluation of f0 / f1 / f2 costs seconds even at 917 504
nodes, so here every level of the full-size hierarchies is compared with the oracle on identical seeded inputs:

  * f0, f1, the assembled H = R'H_blk R at north_star's 1e-10 relative (`KERNEL_RTOL`), on EVERY level -- this is the
    shipping instantiation `elem_f2_fast<4,7,SigDefault>` at 917 504 nodes, the chunked coarse gathers, the LDS
    accumulators and the projected slabs at the sizes they run at in the bench;
  * `mgbhip_newton_direction` called twice at the same point (the first call takes the assembled path, the second one
    the CONDENSE kernel + packed leaves exactly as the resident Newton loop does from its second iteration on): the
    direction must solve the ORACLE's system -- normwise backward error ||H_o x - g_o|| / (||H_o|| ||x|| + ||g_o||)
    <= 1e-10 (KERNEL_RTOL; observed ~1e-16) and lambda^2 = <g, x> equal to the oracle's <g_o, x> to 1e-10 --, which
    is a check of the 15-level elimination tree independent of the device's own H and g.

Reference: src/convex.jl:155-202 (f0/f1/f2), src/BlockMatrices.jl:506-555 (R'HR), src/newton.jl:252-255."""
import os
import time

import numpy as np
import pytest
import scipy.sparse as sp

import mgb_amd as m
from helpers import record_observation, stacked
from oracle import mgb_oracle as O

pytestmark = pytest.mark.gpu

KERNEL_RTOL = 1e-10


def rel(a, b):
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def _device(prob):
    from mgb_amd.device import DeviceMGBProblem
    return DeviceMGBProblem(prob)


def _norm_inf(H):
    return float(abs(H).sum(axis=1).max())


def _compare_levels(tag, P, Mo, Q, c, z0, scale, seed, expect_condensed_on_fine):
    """Every level of the hierarchy: f0 / f1 / H / Newton direction (assembled, then condensed) vs the oracle."""
    B = O.Barrier(Q)
    rng = np.random.default_rng(seed)
    Lmax = len(Mo.R_fine) - 1
    worst = dict(f0=0.0, f1=0.0, H=0.0, bwd=0.0, lam=0.0)
    for J in range(Lmax + 1):
        R = Mo.R_fine[J]
        s = scale * rng.standard_normal(R.shape[1])
        t0 = time.time()
        y_o = B.f0(s, Mo.w, c, R, Mo.D_fine, z0)
        g_o = B.f1(s, Mo.w, c, R, Mo.D_fine, z0)
        H_o = sp.csr_matrix(B.f2(s, Mo.w, c, R, Mo.D_fine, z0))
        t_or = time.time() - t0
        assert np.isfinite(y_o) and np.all(np.isfinite(g_o)), (tag, J)
        e0 = abs(P.f0(J, s, c, z0) - y_o) / abs(y_o)
        e1 = rel(P.f1(J, s, c, z0), g_o)
        H_d = P.f2(J, s, c, z0)
        eH = float(abs(H_d - H_o).max() / abs(H_o).max())
        del H_d
        hn, gn = _norm_inf(H_o), float(np.linalg.norm(g_o, np.inf))
        ebw, elam = 0.0, 0.0
        conds = []
        for call in range(2):
            x, lam, cond = P.newton_direction(J, s, c, z0)
            conds.append(cond)
            assert np.all(np.isfinite(x)) and lam > 0, (tag, J, call, lam)
            r = H_o @ x - g_o
            ebw = max(ebw, float(np.linalg.norm(r, np.inf) / (hn * np.linalg.norm(x, np.inf) + gn)))
            elam = max(elam, abs(lam - float(g_o @ x)) / abs(lam))
        if J == Lmax and expect_condensed_on_fine:
            assert conds == [False, True], conds          # first call assembled, second call the condensing f2 kernel
        record_observation(f"{tag} level {J} m={R.shape[1]}: f0 {e0:.1e} f1 {e1:.1e} H {eH:.1e} "
                           f"direction backward error {ebw:.1e} lambda^2 {elam:.1e} condensed {conds} oracle {t_or:.1f}s")
        for k, v in (("f0", e0), ("f1", e1), ("H", eH), ("bwd", ebw), ("lam", elam)):
            worst[k] = max(worst[k], v)
        assert e0 <= KERNEL_RTOL, (tag, J, e0)
        assert e1 <= KERNEL_RTOL, (tag, J, e1)
        assert eH <= KERNEL_RTOL, (tag, J, eH)
        assert ebw <= KERNEL_RTOL, (tag, J, ebw)
        assert elam <= KERNEL_RTOL, (tag, J, elam)
    record_observation(f"{tag} WORST over {Lmax + 1} levels: " + " ".join(f"{k} {v:.1e}" for k, v in worst.items()))


def test_config2_fem2d_P2_L7_p15_every_level_vs_oracle():
    """BASELINE configs[1]: fem2d_P2 p = 1.5, L = 7 (57 344 nodes) on the reference-default ladder."""
    prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 7)), p=1.5)
    assert prob.M[0].w.size == 57344
    D = _device(prob)
    try:
        _compare_levels("config2 fem2d_P2 L=7 p=1.5", D.main, O.OracleAMG(prob.M[0]), prob.Q, 0.1 * prob.f,
                        stacked(prob.g), 1e-4, 71, expect_condensed_on_fine=True)
    finally:
        D.close()


def test_config3_fem2d_P2_L9_p1_every_level_vs_oracle():
    """BASELINE configs[2], the headline workload exactly as bench.py runs it: fem2d_P2 p = 1.0, L = 9 (917 504 nodes,
    1 309 697 fine unknowns, 11 levels) on amg_ruge_stuben(max_coarse=2)."""
    prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 9)), p=1.0)
    assert prob.M[0].w.size == 917504 and len(prob.M[0].R_fine) == 11
    D = _device(prob)
    try:
        _compare_levels("config3 fem2d_P2 L=9 p=1.0", D.main, O.OracleAMG(prob.M[0]), prob.Q, 0.1 * prob.f,
                        stacked(prob.g), 1e-5, 93, expect_condensed_on_fine=True)
    finally:
        D.close()


def test_config3_north_star_p15_L9_every_level_vs_oracle():
    """The north_star target problem (p = 1.5 at L = 9) on the ladder its bench line uses (max_coarse=10)."""
    prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 9), prolongator=m.amg_ruge_stuben(max_coarse=10)), p=1.5)
    D = _device(prob)
    try:
        _compare_levels("north_star fem2d_P2 L=9 p=1.5 max_coarse=10", D.main, O.OracleAMG(prob.M[0]), prob.Q,
                        0.1 * prob.f, stacked(prob.g), 1e-5, 95, expect_condensed_on_fine=True)
    finally:
        D.close()


def test_config4_fem3d_L6_p4_every_level_vs_oracle():
    """BASELINE configs[3]: fem3d Q1 p = 4, L = 6 (262 144 nodes) on the ladder the full-size solve runs on
    (max_coarse=500).  The default start is infeasible there (phase I runs first), so the MAIN image is compared at a
    lifted slack (s = 1e4: s^(2/p) = 100 > |grad g|^2 <= 12) and the PHASE-I image (cobarrier + box, src/mgb.jl:217-287)
    at the default start with the slack the driver would choose."""
    prob = m.assemble(m.amg(m.subdivide(m.fem3d(k=1), 6), prolongator=m.amg_ruge_stuben(max_coarse=500)), p=4.0)
    n = prob.M[0].w.size
    assert n == 262144
    D = _device(prob)
    try:
        g_lift = prob.g.copy()
        g_lift[:, 1] = 1.0e4
        _compare_levels("config4 fem3d L=6 p=4 main", D.main, O.OracleAMG(prob.M[0]), prob.Q, 0.1 * prob.f,
                        stacked(g_lift), 1e-5, 41, expect_condensed_on_fine=False)
        feas = D.feasibility
        nD = len(prob.M[0].D_fine)
        z0 = stacked(prob.g)
        sl = D.main.node_slack(z0)
        assert sl.max() > 0                                     # the default start really is infeasible
        z1 = np.concatenate([z0, 2 * np.maximum(sl, 1.0)])      # slack_init (src/mgb.jl:437-440)
        b = 2 * max(1.0, float(z1[-n:].max()))
        Rbox = max(10.0, 10.0 * float(np.abs(z0).max()))
        feas.set_box(b, Rbox)
        c1 = np.zeros((n, nD + 1 + 2))
        c1[:, nD] = 0.1
        _compare_levels("config4 fem3d L=6 p=4 phase-I", feas, O.OracleAMG(prob.M[1]), O.FeasConvex(prob.Q, b, Rbox, nD + 1),
                        c1, z1, 1e-5, 43, expect_condensed_on_fine=False)
    finally:
        D.close()
