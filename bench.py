#!/usr/bin/env python3
"""bench.py -- Newton iterations/sec + wall-clock to converge of the MI355X hot path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 the driver
launches one rank per GPU through torch.distributed.run.  A *step* is one complete
`mgb_solve` of the workload (phase probe + t-ramp to 1/tol with the reference's default
controls, reference: src/mgb.jl:95-101, :360-363) on data already resident in HBM (the
hierarchy, operators and plans are uploaded/built before the clock starts; only the
start iterate and the cost grid, ~45 MB at L=9, are handed over per solve).

The direct solve does not shard ("replicas only"), so at N > 1 every rank
solves its own replica of the workload and `value` is the aggregate rate (weak scaling).

Extra objects in the JSON line:
  roofline     -- the fused element Hessian kernel (the largest HBM stream of the path,
                  SURVEY.md section 8d) timed live with hipEvents on the library's stream;
  cpu_baseline -- the NumPy/SciPy oracle (+ host multifrontal Cholesky) on a bounded
                  sample of the same workload, rank 0 / N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
F2_BYTES_PER_NODE = 347.0        # compulsory bytes/node of f2 (fem2d_P2 default problem)


# Hierarchy variants tried in order.  The reference's default `amg_ruge_stuben(max_coarse=2)`
# coarsens down to 2-3 unknowns; at L=9 the Newton solves in those tiny coarse spaces creep
# along the barrier wall during the initial centring (the regime the reference's comments at
# src/mgb.jl:64-71 describe) and the t-ramp reports :stall, so the workload keeps the coarsest
# space at a few hundred unknowns (`max_coarse` is a documented knob of the reference's
# prolongator factory, src/multigrid.jl:304-306).  DESIGN.md section "Workload" has the details.
HIERARCHIES = [dict(max_coarse=300), dict(max_levels=6), dict(theta=0.5), dict()]


def build_problem(L, p, rs_kwargs):
    import mgb_amd as m
    geom = m.subdivide(m.fem2d_P2(), L)
    return m.assemble(m.amg(geom, prolongator=m.amg_ruge_stuben(**rs_kwargs)), p=p)


def cpu_baseline(prob, budget_s):
    """Oracle (kind 'port') Newton iterations/sec on the host, bounded by a time budget."""
    from oracle import mgb_oracle as O
    try:
        O.set_solver("mf")
    except Exception:
        O.set_solver("splu")
    st = {}
    t0 = time.perf_counter()
    st["deadline"] = t0 + budget_s
    done = True
    try:
        O.mgb_solve(prob, stats=st)
    except TimeoutError:
        done = False
    el = time.perf_counter() - t0
    its = st.get("newton_its", 0)
    return dict(value=its / el if el > 0 else 0.0, unit="newton_iters/s", cores=1, kind="port",
                sample=(f"{its} Newton iterations of the same workload from the default start "
                        f"({'complete solve' if done else f'stopped at the {budget_s:.0f} s budget'}); NumPy/SciPy "
                        "evaluate+assemble, single-thread host multifrontal Cholesky (oracle/csrc/mf_host.cpp)"),
                seconds=el, solve_seconds=st.get("solve_s", 0.0))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--L", type=int, default=int(os.environ.get("MGB_BENCH_L", "9")))
    ap.add_argument("--p", type=float, default=float(os.environ.get("MGB_BENCH_P", "1.0")))
    ap.add_argument("--cpu-budget", type=float, default=float(os.environ.get("MGB_BENCH_CPU_BUDGET", "25")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # MGB_BENCH_REHEARSAL=1 (development only): all ranks share GPU 0 and talk over gloo, so that the
    # multi-rank control flow can be exercised on a one-GPU box; the driver's runs use RCCL.
    rehearsal = os.environ.get("MGB_BENCH_REHEARSAL", "0") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if rehearsal:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_index))

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    import mgb_amd as m
    from mgb_amd.device import DeviceMGBProblem
    from mgb_amd.solve import mgb_driver

    from mgb_amd.solve import MGBConvergenceFailure
    prob = D = None
    used = None
    for rs_kwargs in (HIERARCHIES if args.L >= 9 else [dict()] + HIERARCHIES):
        t0 = time.perf_counter()
        prob = build_problem(args.L, args.p, rs_kwargs)
        t_setup = time.perf_counter() - t0
        t0 = time.perf_counter()
        D = DeviceMGBProblem(prob, device_id=dev_index)
        t_upload = time.perf_counter() - t0
        try:
            for _ in range(max(args.warmup, 1) if used is None else args.warmup):
                mgb_driver(D)          # untimed: builds plans + symbolic factorizations, proves convergence
            used = rs_kwargs
            break
        except MGBConvergenceFailure as e:
            if rank == 0:
                print(f"bench: hierarchy {rs_kwargs} failed ({e.code}); trying the next variant", file=sys.stderr)
            D.close()
            D = None
    if D is None:
        raise SystemExit("bench.py: no hierarchy variant converged")
    barrier()
    t0 = time.perf_counter()
    its_total = 0
    solve_s = 0.0
    core_s = 0.0
    last = None
    for _ in range(args.steps):
        SOL = mgb_driver(D)
        last = SOL
        for key in ("SOL_feasibility", "SOL_main"):
            if SOL[key] is not None:
                its_total += int(SOL[key]["its"].sum())
                solve_s += SOL[key]["solve_seconds"]
                core_s += SOL[key]["t_elapsed"]
    barrier()
    elapsed = time.perf_counter() - t0
    from mgb_amd.replicas import aggregate
    elapsed_max, its_all = aggregate(elapsed, its_total, dist, device="cpu" if rehearsal else "cuda")

    # ---- roofline of the dominant HBM kernel: fused element Hessian (f2), fine level ----
    # One more solve of the same workload, outside the timed region, with the library's hipEvent
    # stage timers switched on (events recorded on the library's own stream around every stage):
    # the fine-level f2 launches are timed exactly as they occur in the Newton loop.
    main = D.main
    fine = len(main.level_sizes) - 1
    n = prob.M[0].w.size
    main.reset_stage_timers(True)
    mgb_driver(D)
    f2_ms, f2_n = main.stage_ms("f2")
    asm_ms, asm_n = main.stage_ms("assemble")
    f0_ms, f0_n = main.stage_ms("f0")
    f1_ms, f1_n = main.stage_ms("f1")
    fac_ms, fac_n = main.stage_ms("factor")
    tri_ms, tri_n = main.stage_ms("trisolve")
    main.reset_stage_timers(False)
    f2_avg_s = (f2_ms / max(f2_n, 1)) * 1e-3
    bytes_per_launch = F2_BYTES_PER_NODE * n
    achieved = bytes_per_launch / f2_avg_s / 1e9 if f2_avg_s > 0 else 0.0
    # HBM traffic of the same launch from the committed rocprofv3 PMC passes (separate FETCH_SIZE /
    # WRITE_SIZE runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced
    # streams on gfx950; counters are in KiB).  Only valid for the workload it was collected on.
    traffic = None
    try:
        pmc = json.load(open(os.path.join(HERE, "profiles", "r01_pmc_traffic_L9.json")))
        if args.L == 9 and n == 917504:
            traffic = 1024.0 * (2.0 * pmc["elem_f2_fast"]["FETCH_SIZE"] + pmc["elem_f2_fast"]["WRITE_SIZE"])
    except Exception:
        traffic = None
    def _frac(bytes_per_node, ms, cnt):
        return (bytes_per_node * n / max(ms / max(cnt, 1) * 1e-3, 1e-12) / 1e9) / HBM_PEAK_GBS
    roofline = dict(bound="hbm", kernel="elem_f2_fast<4,7,SigDefault>: fused Dz + cone Hessian + element blocks, fine level "
                                        "(the f2 stage of the Newton loop; z0 + R*s is cached from the preceding f1)",
                    achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
                    traffic=traffic, bytes_per_launch=bytes_per_launch, avg_launch_us=f2_avg_s * 1e6, launches=int(f2_n),
                    assemble_avg_us=(asm_ms / max(asm_n, 1)) * 1e3, assemble_frac=_frac(488.0, asm_ms, asm_n),
                    f0_avg_us=(f0_ms / max(f0_n, 1)) * 1e3, f0_frac=_frac(219.0, f0_ms, f0_n),
                    f1_avg_us=(f1_ms / max(f1_n, 1)) * 1e3, f1_frac=_frac(231.0 - 11.4, f1_ms, f1_n),
                    factor_avg_us=(fac_ms / max(fac_n, 1)) * 1e3, trisolve_avg_us=(tri_ms / max(tri_n, 1)) * 1e3)
    stats = main.solver_stats(fine)

    out = None
    if rank == 0:
        sm = last["SOL_main"]
        out = {
            "metric": "Newton iters/sec + wall-clock to converge, 2D P2 p-Laplace L=%d" % args.L,
            "value": its_all / elapsed_max,
            "unit": "newton_iters/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed_max / max(args.steps, 1),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"fem2d_P2() p={args.p} L={args.L} default f,g (BASELINE.json configs[2] family), "
                                   f"AMG hierarchy, n={n} broken nodes, {main.level_sizes[fine]} fine unknowns",
                       "hierarchy": f"amg_ruge_stuben({used})",
                       "parallelism": "replicas" if world > 1 else "single",
                       "solver_controls": "reference defaults (tol=sqrt(eps), t=0.1, kappa=10, max_newton=8, backtracking)"},
            "wall_clock_to_converge_s": elapsed_max / max(args.steps, 1),
            "newton_iterations_per_solve": its_total / max(args.steps, 1),
            "linear_solve_fraction": solve_s / core_s if core_s > 0 else None,
            "t_steps": int(sm["k"]),
            "setup_s": {"host_setup": t_setup, "upload": t_upload},
            "factorization": {k: stats[k] for k in ("fronts", "max_front", "factor_flops", "tree_levels", "nnz", "unknowns")},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(prob, args.cpu_budget)
            if out["cpu_baseline"]["value"] > 0:
                out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    D.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
