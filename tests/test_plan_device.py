"""Row f4: the assembly plan (pattern of R'HR + contribution lists) is built on the device by one stable radix
sort (csrc/plan_device.hip).  The host builder it replaces stays reachable with MGBHIP_HOST_PLAN=1; the two
must give bitwise-identical Hessians on every level (same pattern, same summation order)."""
import json
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
TOOL = os.path.join(HERE, "..", "tools", "plan_digest.py")


def _run(env_extra, *args):
    env = dict(os.environ)
    env.update(env_extra)
    out = subprocess.run([sys.executable, TOOL, *args], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])


@pytest.mark.gpu
def test_device_plan_equals_host_plan_bitwise():
    dev = _run({"MGBHIP_HOST_PLAN": "0"})
    host = _run({"MGBHIP_HOST_PLAN": "1"})
    assert dev.keys() == host.keys() and len(dev) >= 12
    diff = [k for k in dev if dev[k] != host[k]]
    assert not diff, diff


@pytest.mark.gpu
def test_direct_values_equal_materialised_hessian_bitwise():
    """In the Newton loop the fine-level Hessian is never formed as a CSR value array: the factorization reads the
    single-contribution entries from the element-block slab and a small kernel sums the shared ones
    (problem.cpp: eval_f2(materialize=false), MfSolver::set_direct_map).  Same numbers, same order: complete solves
    must be bitwise those of the materialised path (MGBHIP_NO_DIRECT=1), iteration counts included."""
    direct = _run({"MGBHIP_NO_DIRECT": "0", "MGBHIP_NO_CONDENSE": "1"}, "solve")
    mat = _run({"MGBHIP_NO_DIRECT": "1"}, "solve")
    keys = [k for k in direct if k.startswith("solve/")]
    assert len(keys) == 2
    for k in keys:
        assert direct[k] == mat[k], k


@pytest.mark.gpu
def test_condensed_leaves_solve_equals_assembled_solve():
    """Round 3: on the fine level of fem2d_P2 the element kernel eliminates each element's slack and bubble unknowns
    itself (kernels.hpp: launch_elem_f2_condense), so the summation order of the leaf fronts differs from the assembled
    path (MGBHIP_NO_CONDENSE=1): Newton iteration counts equal up to +-1 in a level solve, z equal to rounding."""
    import numpy as np
    cond = _run({}, "solve")
    plain = _run({"MGBHIP_NO_CONDENSE": "1"}, "solve")
    for name in ("fem2d_L4_p15", "fem1d_L5_p1"):
        ia, ib = np.array(cond[f"solve/{name}"][1]), np.array(plain[f"solve/{name}"][1])
        # Newton counts: equal up to the +-1 of stopping rules that compare quantities at rounding level
        assert ia.shape == ib.shape and np.abs(ia - ib).max() <= 1 and abs(int(ia.sum()) - int(ib.sum())) <= 2, name
        a, b = np.array(cond[f"z/{name}"]), np.array(plain[f"z/{name}"])
        assert np.abs(a - b).max() <= 1e-10 * max(1.0, np.abs(b).max()), name
    assert cond["solve/fem1d_L5_p1"][0] == plain["solve/fem1d_L5_p1"][0]              # fem1d has no condensed leaves: bitwise


@pytest.mark.gpu
def test_plans_prepared_side_by_side_equal_the_lazily_built_ones_bitwise():
    """Round 4 (time to first solution): `mgb_core` builds every level's plan and symbolic analysis on host threads before
    the first Newton iteration; MGBHIP_LAZY_PLANS=1 keeps the level-by-level order of rounds 1-3.  Same plans, same
    factorization order: complete solves agree bit for bit (leaf condensation off in both runs -- lazily it would start at the
    second factorization, prepared at the first, and its summation order differs from the assembled path)."""
    eager = _run({"MGBHIP_NO_CONDENSE": "1"}, "solve")
    lazy = _run({"MGBHIP_NO_CONDENSE": "1", "MGBHIP_LAZY_PLANS": "1"}, "solve")
    keys = [k for k in eager if k.startswith("solve/") or k.startswith("z/")]
    assert len(keys) >= 4
    for k in keys:
        assert eager[k] == lazy[k], k
