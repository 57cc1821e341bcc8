"""The A/B switches of the symbolic analysis / launch plan that have no other test (tools/README.md, switch table):
MGBHIP_NO_GEO (BFS level-set bisection instead of geometric nested dissection), MGBHIP_NO_MERGE_GROUPS (no folding of
straggler launch groups), MGBHIP_NO_PACKED_LEAVES (square leaf fronts).  Each changes the elimination order or the launch
grouping, never the mathematics: a complete solve under each switch must reproduce the default solve (z to 1e-10 relative,
Newton counts within +-3).  The switches are read once per process, hence the worker processes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = ("import sys, json; sys.path.insert(0, %r)\n"
        "import numpy as np, mgb_amd as m\n"
        "sol = m.mgb_solve(m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 6)), p=1.5))\n"
        "np.save(sys.argv[1], sol.z); print(json.dumps(int(sol.SOL_main['its'].sum())))\n") % ROOT


def _solve(tmp_path, tag, env):
    out = str(tmp_path / f"{tag}.npy")
    r = subprocess.run([sys.executable, "-c", CODE, out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(out), json.loads(r.stdout.strip().splitlines()[-1])


def test_plan_switches_reproduce_the_default_solve(tmp_path):
    z0, its0 = _solve(tmp_path, "default", {})
    for sw in ("MGBHIP_NO_GEO", "MGBHIP_NO_MERGE_GROUPS", "MGBHIP_NO_PACKED_LEAVES"):
        z, its = _solve(tmp_path, sw, {sw: "1"})
        assert np.abs(z - z0).max() <= 1e-10 * max(1.0, np.abs(z0).max()), sw
        assert abs(its - its0) <= 3, (sw, its, its0)


def test_upper_triangle_coarse_assembly_is_bitwise_the_full_one(tmp_path):
    """Inside the Newton loop the coarse levels project and gather only the upper triangle of H = R'H_blk R (the factorization
    and `symmetric(H)` read nothing else: src/newton.jl:253, csrc/mf_analysis.cpp).  The entries it forms are the same sums in
    the same order, so the whole solve is bit for bit the one with MGBHIP_FULL_COARSE_H=1 (both triangles, as in round 3)."""
    z0, its0 = _solve(tmp_path, "upper", {})
    z1, its1 = _solve(tmp_path, "full", {"MGBHIP_FULL_COARSE_H": "1"})
    assert np.array_equal(z0, z1) and its0 == its1


CODE3D = ("import sys, json; sys.path.insert(0, %r)\n"
          "import numpy as np, mgb_amd as m\n"
          "prob = m.assemble(m.amg(m.subdivide(m.fem3d(k=1), 4), prolongator=m.amg_ruge_stuben(max_coarse=40)), p=2.0)\n"
          "sol = m.mgb_solve(prob)\n"
          "np.save(sys.argv[1], sol.z); print(json.dumps(int(sol.SOL_main['its'].sum())))\n") % ROOT


def _solve3d(tmp_path, tag, env):
    out = str(tmp_path / f"{tag}.npy")
    r = subprocess.run([sys.executable, "-c", CODE3D, out], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(out), json.loads(r.stdout.strip().splitlines()[-1])


def test_coarse_assembly_layout_switches_are_bitwise_and_the_matrix_core_projection_agrees(tmp_path):
    """Round 4 coarse assembly.  The slab of a projected level is kept in the order of the contribution lists
    (MGBHIP_NO_SORTED_SLAB=1: element-major, gathered through the index lists) and the upper-triangle positions are summed in
    column-major order (MGBHIP_UPPER_ROW_MAJOR=1: CSR order): both only move data, every sum keeps its terms and their order,
    so complete solves agree bit for bit -- on a 2-D ladder (staged loop projection) and on a 3-D one (matrix-core projection,
    wide supports).  The matrix-core projection itself (MGBHIP_NO_MFMA_PROJECT=1 restores the loop kernels) rounds
    differently: z to 1e-10 relative, Newton counts within +-3."""
    for solve in (_solve, _solve3d):
        z0, its0 = solve(tmp_path, "default", {})
        for sw in ("MGBHIP_NO_SORTED_SLAB", "MGBHIP_UPPER_ROW_MAJOR"):
            z, its = solve(tmp_path, sw, {sw: "1"})
            assert np.array_equal(z, z0) and its == its0, sw
        z, its = solve(tmp_path, "loops", {"MGBHIP_NO_MFMA_PROJECT": "1"})
        assert np.abs(z - z0).max() <= 1e-10 * max(1.0, np.abs(z0).max())
        assert abs(its - its0) <= 3, (its, its0)


def test_device_transpose_of_the_prolongators_is_bitwise_the_host_loop(tmp_path):
    """Round 4 (time to first solution): the CSR of R' that the restriction R' v gathers through is built on the device from
    the uploaded R (one stable radix sort by column, csrc/plan_device.hip: transpose_csr_device) instead of by a host loop
    whose result crossed PCIe.  Same entries in the same order inside every row of R', so every restriction sums in the same
    order: complete solves agree bit for bit with MGBHIP_HOST_TRANSPOSE=1, on a 2-D and on a 3-D ladder."""
    for solve in (_solve, _solve3d):
        z0, its0 = solve(tmp_path, "device_T", {})
        z1, its1 = solve(tmp_path, "host_T", {"MGBHIP_HOST_TRANSPOSE": "1"})
        assert np.array_equal(z0, z1) and its0 == its1


def test_polled_read_backs_equal_the_synchronised_ones(tmp_path):
    """The Newton loop's two read-backs per iteration (direction statistics, line-search trial) are awaited by polling a sequence
    stamp that the finishing kernel stores behind its results in the pinned block (mgbhip_problem::wait_results; the runtime's
    hipStreamSynchronize reports the kernel a few microseconds later: 1 650 waits per solve at L = 9).  MGBHIP_NO_POLL=1 waits
    with hipStreamSynchronize: the same numbers, hence the same solve bit for bit."""
    z0, its0 = _solve(tmp_path, "poll", {})
    z1, its1 = _solve(tmp_path, "sync", {"MGBHIP_NO_POLL": "1"})
    assert np.array_equal(z0, z1) and its0 == its1


def test_fused_line_search_step_is_bitwise_the_step_launch(tmp_path):
    """On selection levels a backtracking trial is evaluated at x - s n formed on the fly by the element kernel (the same fused
    multiply-add the step kernel performs), and the step kernel that materialises the trial vector runs behind the evaluation:
    the first launch after the host's decision is the long one.  MGBHIP_NO_FUSED_STEP=1 launches the step first, as before:
    the same solve bit for bit."""
    for solve in (_solve, _solve3d):
        z0, its0 = solve(tmp_path, "fused", {})
        z1, its1 = solve(tmp_path, "step_first", {"MGBHIP_NO_FUSED_STEP": "1"})
        assert np.array_equal(z0, z1) and its0 == its1


def test_fused_restriction_of_a_trial_is_bitwise_the_separate_launches(tmp_path):
    """A line-search trial runs three launches behind the element kernel -- restriction R' v, the partial sums of |g|^2, the
    step kernel -- as ONE (`restrict_trial_kernel`: the same row gathers, the same grid-stride partial sums and LDS tree, the same
    fused multiply-add) on levels whose restriction is the row-parallel kernel.  MGBHIP_NO_FUSED_RESTRICT=1 keeps the separate
    launches: the same solve bit for bit, on a 2-D and on a 3-D ladder."""
    for solve in (_solve, _solve3d):
        z0, its0 = solve(tmp_path, "fused_restrict", {})
        z1, its1 = solve(tmp_path, "separate", {"MGBHIP_NO_FUSED_RESTRICT": "1"})
        assert np.array_equal(z0, z1) and its0 == its1
