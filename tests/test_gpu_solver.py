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
