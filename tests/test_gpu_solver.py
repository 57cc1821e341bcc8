"""The device LDL' against arbitrary matrices on the levels' patterns (mgbhip_set_hessian): backward error of both
solve paths on graded (ill-conditioned) SPD matrices, and an A/B of the kernel families that are gated by system
size -- the one-wave-per-front kernel and the inverse-based large-front path -- against the blocked / substitution
kernels on the same plans and values (a size gate must not hide a kernel bug).

Reference behaviour: `solve(symmetric(H), g)` = CHOLMOD, backward stable (src/utils.jl:142-145)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from solver_cases import CASES, build, key_of, matrices

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "dev", "solver_ab_worker.py")
ETA_MAX = 4e-12        # componentwise backward error (Oettli-Prager) accepted from an LDL' in fp64


def _run(tmp_path, name, env):
    out = str(tmp_path / f"{name}.npz")
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, WORKER, out], check=True, env=e, cwd=ROOT, timeout=900)
    return dict(np.load(out))


def test_both_solve_paths_are_backward_stable_and_gated_kernels_agree(tmp_path):
    from mgb_amd.device import DeviceMGBProblem
    runs = {"default": _run(tmp_path, "default", {}),                                             # size gates as shipped
            "fast": _run(tmp_path, "fast", {"MGBHIP_WAVE_MIN_N": "0", "MGBHIP_INV_MIN_N": "0"}),   # gated kernels everywhere
            "slow": _run(tmp_path, "slow", {"MGBHIP_NO_WAVE_SMALL": "1", "MGBHIP_OLD_BIG": "1"})}  # nowhere
    assert runs["default"].keys() == runs["fast"].keys() == runs["slow"].keys()
    worst = {k: (0.0, "") for k in runs}
    for fam, L, p, rs in CASES:
        D = DeviceMGBProblem(build(fam, L, p, rs))
        try:
            for lev, grade, A, g in matrices(D.main):
                key = key_of(fam, L, p, rs, lev, grade)
                absA = abs(A)
                for tag, res in runs.items():
                    for suffix in ("_x", "_xn"):
                        x = res[key + suffix]
                        assert np.isfinite(x).all(), (tag, key, suffix)
                        eta = float(np.max(np.abs(A @ x - g) / (absA @ np.abs(x) + np.abs(g))))
                        if eta > worst[tag][0]:
                            worst[tag] = (eta, key + suffix)
                        assert eta <= ETA_MAX, (tag, key, suffix, eta)
                    assert res[key + "_lam"][1] == 0.0, (tag, key)                               # no pivot flagged
                    lam_ref = float(g @ res[key + "_x"])
                    assert abs(res[key + "_lam"][0] - lam_ref) <= 1e-9 * abs(lam_ref), (tag, key)
        finally:
            D.close()
    print("worst componentwise backward error per kernel selection:", {k: (f"{v[0]:.2e}", v[1]) for k, v in worst.items()})


@pytest.mark.parametrize("L,p", [(3, 1.5), (5, 1.0), (6, 1.5)])
def test_condensed_leaves_reproduce_the_assembled_newton_direction(L, p):
    """Fine level of fem2d_P2: from the second evaluation on the element kernel eliminates every element's slack and
    bubble unknowns itself and writes the leaf fronts of the factorization (no element block reaches HBM, no shared
    entry is summed).  The Newton direction must be the one of the assembled system: against the generic path
    (materialised H, forward + backward sweeps) and against SciPy on the device's own H."""
    import mgb_amd as m
    import scipy.sparse.linalg as spla
    from helpers import stacked
    from mgb_amd.device import DeviceMGBProblem
    prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
    D = DeviceMGBProblem(prob)
    try:
        P = D.main
        J = len(P.level_sizes) - 1
        rng = np.random.default_rng(5)
        z0, c = stacked(prob.g), 0.1 * prob.f
        for trial in range(3):
            s = 1e-3 * rng.standard_normal(P.level_sizes[J])
            x, lam, condensed = P.newton_direction(J, s, c, z0)
            assert condensed == (trial > 0)                       # the first call builds the plan and the leaf map
            g = P.f1(J, s, c, z0)
            H = P.f2(J, s, c, z0).tocsc()
            x_ref = spla.spsolve(H, g)
            assert np.linalg.norm(H @ x - g) <= 1e-10 * np.linalg.norm(g)
            assert np.linalg.norm(x - x_ref) <= 1e-9 * np.linalg.norm(x_ref)
            assert abs(lam - g @ x_ref) <= 1e-10 * abs(g @ x_ref)
            assert np.linalg.norm(P.solve(J, g) - x) <= 1e-10 * np.linalg.norm(x)
    finally:
        D.close()


def test_one_wave_mfma_ldlt32_matches_host_ldlt_and_flags_zero_pivots():
    """The 32 x 32 LDL' of the pivot chain (csrc/ldlt32.hpp: matrix tiles in MFMA accumulators, one wave, no barriers
    inside the block) and the inverse built alongside it, in the standalone harness tools/micro/ldlt32_mfma_test.hip:
    L and d against a host LDL' in double for nb = 32, 31, 17, 16, 5, 1 (identity padding), (I + L) W = I, a clean upper
    triangle, and the status flag on a zero pivot.  The binary is built by __graft_entry__.build()."""
    import subprocess
    exe = os.path.join(ROOT, "tools", "micro", "ldlt32_mfma_test")
    if not os.path.exists(exe):
        pytest.fail("tools/micro/ldlt32_mfma_test is not built: run python __graft_entry__.py")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [l for l in out.stdout.splitlines() if l.startswith("nb=") or l.startswith("zero pivot")]
    assert len(lines) == 7 and all(l.rstrip().endswith("ok") for l in lines), out.stdout


def test_zero_pivot_falls_back_to_the_pivoted_lu_like_the_reference():
    """`solve(symmetric(H), g)` is Julia's `Symmetric(H) \\ g`: Cholesky, then LDL', then LU when the symmetric
    factorizations meet a zero pivot (src/utils.jl:142-145).  The device LDL' is un-pivoted; an exactly zero pivot used to
    end the Newton attempt as "not converged" (missing #1).  Now systems small enough to be held densely take
    the reference's last resort on the device: dense LU with partial pivoting (csrc/dense.hip).  Indefinite-but-regular
    matrices on the levels' own patterns, each with a ZERO leading entry so that the un-pivoted factorization must fail:
    both solve paths return H^{-1} g to 1e-9; a singular matrix still reports MGBHIP_ERR_NOT_SPD; and with
    MGBHIP_NO_LU_FALLBACK=1 (worker process) the zero pivot is an error as before."""
    import mgb_amd as m
    import scipy.sparse as sp
    from mgb_amd import device as dev
    from mgb_amd.device import DeviceMGBProblem
    prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 4)), p=1.5)
    D = DeviceMGBProblem(prob)
    rng = np.random.default_rng(3)
    try:
        P = D.main
        tried = 0
        for lev, msz in enumerate(P.level_sizes):
            if msz > 2048:
                continue
            indptr, indices = P.hessian_pattern(lev)
            v = rng.standard_normal(indices.size)
            A = sp.csr_matrix((v, indices, indptr), shape=(msz, msz))
            A = sp.csr_matrix((A + A.T) * 0.5 + sp.diags(rng.choice([-3.0, 3.0], msz)))      # symmetric, indefinite
            A = sp.lil_matrix(A)
            A[0, 0] = 0.0                                                                      # the first pivot of the LDL' is exactly zero
            A = sp.csr_matrix(A)
            A.sort_indices()
            if msz > 1:
                assert abs(A[0, 1:]).sum() > 0                                                 # ... but the matrix is regular
            Ad = np.asarray(A.todense())
            if msz == 1 or abs(np.linalg.det(Ad / np.abs(Ad).max())) < 1e-12:
                continue
            vals = np.zeros(indices.size)
            vals[:] = np.asarray(A[np.repeat(np.arange(msz), np.diff(indptr)), indices]).ravel()      # explicit zeros keep their slot
            g = rng.standard_normal(msz)
            x_ref = np.linalg.solve(np.triu(Ad) + np.triu(Ad, 1).T, g)
            P.set_hessian(lev, vals)
            x = P.solve(lev, g)
            assert np.linalg.norm(x - x_ref) <= 1e-9 * np.linalg.norm(x_ref), lev
            P.set_hessian(lev, vals)
            xn, lam, status = P.solve_newton(lev, g)
            assert status == dev.OK and np.linalg.norm(xn - x_ref) <= 1e-9 * np.linalg.norm(x_ref), lev
            assert abs(lam - g @ x_ref) <= 1e-9 * max(abs(g @ x_ref), np.linalg.norm(g) * np.linalg.norm(x_ref) * 1e-3)
            tried += 1
        assert tried >= 3
        # a singular matrix stays an error: first row and column zero
        lev = 1
        msz = P.level_sizes[lev]
        indptr, indices = P.hessian_pattern(lev)
        rows = np.repeat(np.arange(msz), np.diff(indptr))
        vals = np.where(rows == indices, 2.0, 0.0)
        vals[(rows == 0) | (indices == 0)] = 0.0
        P.set_hessian(lev, vals)
        with pytest.raises(dev.MGBHipError) as ei:
            P.solve(lev, np.ones(msz))
        assert ei.value.status == dev.ERR_NOT_SPD
    finally:
        D.close()
    # the switch restores round 3's behaviour: the zero pivot is reported
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import numpy as np, mgb_amd as m\n"
            "from mgb_amd import device as dev\n"
            "from mgb_amd.device import DeviceMGBProblem\n"
            "D = DeviceMGBProblem(m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 2)), p=1.5)); P = D.main\n"
            "ip, ix = P.hessian_pattern(0); n = P.level_sizes[0]\n"
            "rows = np.repeat(np.arange(n), np.diff(ip)); v = np.where(rows == ix, 0.0, 1.0)\n"
            "P.set_hessian(0, v)\n"
            "try:\n    P.solve(0, np.ones(n)); print('SOLVED')\n"
            "except dev.MGBHipError as e:\n    print('STATUS', e.status)\n") % (ROOT, os.path.join(ROOT, "tests"))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300,
                         env=dict(os.environ, MGBHIP_NO_LU_FALLBACK="1"))
    assert "STATUS 3" in out.stdout, out.stdout + out.stderr
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ))
    assert "SOLVED" in out.stdout, out.stdout + out.stderr
