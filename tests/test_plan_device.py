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
    direct = _run({"MGBHIP_NO_DIRECT": "0"}, "solve")
    mat = _run({"MGBHIP_NO_DIRECT": "1"}, "solve")
    keys = [k for k in direct if k.startswith("solve/")]
    assert len(keys) == 2
    for k in keys:
        assert direct[k] == mat[k], k
