"""Time-to-first-solution breakdown (VERDICT f4): host hierarchy/operator build, device upload, plan + symbolic
analysis (first solve) and the steady-state solve.  Usage: python tools/setup_timing.py [L] [p]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    p = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    import mgb_amd as m
    from mgb_amd.device import DeviceMGBProblem
    from mgb_amd.solve import mgb_driver
    out = {"L": L, "p": p}
    import torch
    t = time.perf_counter(); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
    out["hip_runtime_init_s_not_included"] = time.perf_counter() - t      # lazy device initialisation of the runtime, once per process
    t = time.perf_counter(); g = m.subdivide(m.fem2d_P2(), L); out["subdivide_s"] = time.perf_counter() - t
    t = time.perf_counter(); mg = m.amg(g); out["amg_s"] = time.perf_counter() - t
    t = time.perf_counter(); prob = m.assemble(mg, p=p); out["assemble_s"] = time.perf_counter() - t
    t = time.perf_counter(); D = DeviceMGBProblem(prob, device_id=0); out["upload_s"] = time.perf_counter() - t
    t = time.perf_counter(); mgb_driver(D); out["first_solve_s"] = time.perf_counter() - t
    t = time.perf_counter(); mgb_driver(D); out["second_solve_s"] = time.perf_counter() - t
    out["time_to_first_solution_s"] = sum(out[k] for k in ("subdivide_s", "amg_s", "assemble_s", "upload_s", "first_solve_s"))
    D.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
