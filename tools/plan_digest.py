"""Digest of the assembled Hessians (pattern + values) of every level of a few small problems.  Run once with
the device plan builder and once with MGBHIP_HOST_PLAN=1: identical digests mean identical patterns AND
identical contribution-list order (the sums are bitwise equal only then).  Used by tests/test_plan_device.py."""
import hashlib
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def digest(H):
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(H.indptr, dtype=np.int64).tobytes())
    h.update(np.ascontiguousarray(H.indices, dtype=np.int64).tobytes())
    h.update(np.ascontiguousarray(H.data, dtype=np.float64).tobytes())
    return h.hexdigest()


def main():
    import mgb_amd as m
    from mgb_amd.device import DeviceMGBProblem
    cases = {
        "fem2d_L3_amg": lambda: m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 3)), p=1.5),
        "fem2d_L2_geometric": lambda: m.assemble(m.geometric_mg(m.fem2d_P2(), 2), p=1.0),
        "fem1d_L4": lambda: m.assemble(m.amg(m.subdivide(m.fem1d(), 4)), p=1.0),
        "fem3d_L2": lambda: m.assemble(m.amg(m.subdivide(m.fem3d(k=1), 2)), p=1.0),
    }
    out = {}
    for name, make in cases.items():
        prob = make()
        D = DeviceMGBProblem(prob, device_id=0)
        z0 = np.ascontiguousarray(prob.g.T).reshape(-1)
        c = 0.1 * prob.f
        amg = prob.M[0]
        for l in range(len(amg.R_fine)):
            s = 1e-4 * np.random.default_rng(l).standard_normal(amg.R_fine[l].shape[1])
            out[f"{name}/{l}"] = digest(D.main.f2(l, s, c, z0))
        D.close()
    if len(sys.argv) > 1 and sys.argv[1] == "solve":
        # complete solves: the Newton loop keeps the fine Hessian in the slab (direct values) unless MGBHIP_NO_DIRECT=1
        for name, make in (("fem2d_L4_p15", lambda: m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 4)), p=1.5)),
                           ("fem1d_L5_p1", lambda: m.assemble(m.amg(m.subdivide(m.fem1d(), 5)), p=1.0))):
            sol = m.mgb_solve(make())
            h = hashlib.sha256(np.ascontiguousarray(sol.z, dtype=np.float64).tobytes()).hexdigest()
            out[f"solve/{name}"] = [h, [int(v) for v in np.asarray(sol.SOL_main["its"]).ravel()]]
            out[f"z/{name}"] = [float(v) for v in np.asarray(sol.z).ravel()]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
