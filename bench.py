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
# SURVEY.md section 8(d), compulsory bytes per broken node of the fem2d_P2 default problem
BYTES = dict(f0=219.0, f1=231.0, f2=347.0, assemble=488.0)          # total 1285 B / node / Newton iteration
# bytes the Newton loop's f2 kernel itself moves (condensed leaves): z0 + R*s arrives through
# the cached prolongation kernel -> 128 B in (operators, z, cone exponents) + 8 B (slack gradient) + 16 B / element of
# leaf descriptor; out: the packed leaf front, 120 doubles per 7-node element = 137 B per node
F2_MOVED_BYTES_PER_NODE = 128.0 + 8.0 + 16.0 / 7.0 + 120.0 * 8.0 / 7.0


# Hierarchy policy.  The workload runs the reference default `amg(geom)` = amg_ruge_stuben(max_coarse=2)
# (src/multigrid.jl:296) whenever it converges: p = 1.0 at L = 9 does, and since round 3 (compensated sums in the
# coarse assembly) p = 1.5 does at L <= 8.  At L = 9, p = 1.5 the initial centring still ends in
# the 2-unknown space with lambda^2 <= 0 (device and oracle, tests/dev/logs/); the p = 1.5 line then falls back to the
# smallest deviation that converges, max_coarse=10 -- a documented knob of the reference's factory
# (src/multigrid.jl:304-306) -- and says so in its `hierarchy` field.
def hierarchies(p, L):
    return [dict(), dict(max_coarse=10), dict(max_coarse=300)]


def build_problem(L, p, rs_kwargs):
    import mgb_amd as m
    geom = m.subdivide(m.fem2d_P2(), L)
    return m.assemble(m.amg(geom, prolongator=m.amg_ruge_stuben(**rs_kwargs)), p=p)


def cpu_baseline(prob, budget_s, its_per_level=None):
    """Oracle (kind 'port') Newton iterations/sec on the host, bounded by a time budget.

    Like for like with the device figure: ONE Newton iteration of the oracle is timed on every level the device solve
    iterated on -- gradient, Hessian + R'HR assembly, the direct solve, and one line-search trial (f0 + f1) at the default
    start -- and the levels are weighted with the device solve's own iteration counts (`its_per_level`): the value is
    total iterations / estimated CPU seconds for the same iteration mix.  (Round 3 timed the first few iterations of an
    oracle solve from the default start, all of them on one level.)  Without `its_per_level`: that older sample."""
    from oracle import mgb_oracle as O
    try:
        O.set_solver("mf")
    except Exception:
        O.set_solver("splu")
    import shutil
    common = dict(unit="newton_iters/s", cores=1, kind="port", host_cores=os.cpu_count(),
                  all_cores_note=("the port's evaluate/assemble (NumPy/SciPy sparse) and its host multifrontal Cholesky are "
                                  "single-threaded like the reference's Julia path (bench.md:81); no all-cores figure exists for it"),
                  julia=(shutil.which("julia") or "unavailable: reference Julia path cannot be timed on this box"))
    if its_per_level is not None and sum(its_per_level) > 0:
        Mo = O.OracleAMG(prob.M[0])
        B = O.Barrier(prob.Q)
        z0 = np.concatenate([prob.g[:, k] for k in range(prob.g.shape[1])])
        c = 0.1 * prob.f
        solve = O._SOLVER[0]
        t_begin = time.perf_counter()
        per_level = {}
        order = sorted((J for J, n in enumerate(its_per_level) if n > 0), key=lambda J: -J)     # finest first
        with np.errstate(all="ignore"):
            for J in order:
                if time.perf_counter() - t_begin > budget_s and per_level:
                    break
                R = Mo.R_fine[J]
                sv = np.zeros(R.shape[1])
                t0 = time.perf_counter()
                g = B.f1(sv, Mo.w, c, R, Mo.D_fine, z0)
                H = B.f2(sv, Mo.w, c, R, Mo.D_fine, z0)
                t_eval = time.perf_counter() - t0
                solve(H, g)                               # untimed: builds the symbolic analysis this pattern keeps for the whole solve
                t0 = time.perf_counter()
                x = solve(H, g)
                s2 = sv - x
                B.f0(s2, Mo.w, c, R, Mo.D_fine, z0)
                B.f1(s2, Mo.w, c, R, Mo.D_fine, z0)
                per_level[J] = t_eval + time.perf_counter() - t0
        sampled = sorted(per_level)
        est = 0.0
        for J, n in enumerate(its_per_level):
            if n > 0:
                est += n * per_level[min(sampled, key=lambda q: abs(q - J))]      # unsampled level: its nearest sampled neighbour
        total = int(sum(its_per_level))
        el = time.perf_counter() - t_begin
        return dict(value=total / est if est > 0 else 0.0, estimated_seconds_to_converge=est,
                    seconds_per_iteration_by_level={int(J): per_level[J] for J in sampled},
                    device_iterations_by_level=[int(n) for n in its_per_level],
                    sample=(f"one oracle Newton iteration (f1, f2 + R'HR, direct solve, one f0 + f1 trial) timed on {len(sampled)} of the "
                            f"{sum(1 for n in its_per_level if n > 0)} levels the device solve iterated on, weighted by the device's "
                            f"per-level iteration counts ({total} iterations); NumPy/SciPy evaluate + assemble, single-thread host "
                            "multifrontal Cholesky (oracle/csrc/mf_host.cpp)"),
                    seconds=el, **common)
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
    return dict(value=its / el if el > 0 else 0.0,
                sample=(f"{its} Newton iterations of the same workload from the default start "
                        f"({'complete solve' if done else f'stopped at the {budget_s:.0f} s budget'}); NumPy/SciPy "
                        "evaluate+assemble, single-thread host multifrontal Cholesky (oracle/csrc/mf_host.cpp)"),
                seconds=el, solve_seconds=st.get("solve_s", 0.0), **common)


def profile_children(L, p, rs_kwargs):
    """rocprofv3 child processes, started before this process touches the GPU, each running tools/gpu_kernels.py (the
    Newton loop's own fine-level sequence through mgbhip_newton_direction, 1 assembled + 3 condensed factorizations):
      * two --pmc passes (FETCH_SIZE and WRITE_SIZE need separate passes, MI355X_MICROARCH.md; FETCH_SIZE doubled for wide
        coalesced reads on gfx950, both in KiB) -> HBM bytes of one condensing f2 launch;
      * one --kernel-trace --stats pass -> average device duration of every kernel (the f2 roofline and the solver's
        roofline are computed from these, not from hipEvents).
    Returns (f2 bytes or None, note, {kernel name: (calls, avg_ns, total_ns)} or None)."""
    import csv, glob, shutil, subprocess, tempfile
    if shutil.which("rocprofv3") is None:
        return None, "rocprofv3 not found", None
    child = [sys.executable, os.path.join(HERE, "tools", "gpu_kernels.py"), str(L), str(p), "3", json.dumps(rs_kwargs)]
    env = dict(os.environ, TMPDIR="/tmp")
    kstats, knote = None, ""
    d = tempfile.mkdtemp(prefix="mgb_kt_", dir="/tmp")
    try:
        subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp",
                       env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
        f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
        kstats = {r["Name"]: (int(r["Calls"]), float(r["AverageNs"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(f))}
    except Exception as e:          # profiler unavailable / refused: report, never guess
        knote = f"kernel-trace pass failed: {type(e).__name__}"
    finally:
        shutil.rmtree(d, ignore_errors=True)
    vals = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="mgb_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child
        try:
            subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=240, check=True)
            f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
            tot, cnt = 0.0, 0
            for r in csv.DictReader(open(f)):
                # the Newton loop's f2 is the condensing instantiation elem_f2_fast<4, 7, SigDefault<4>, true>
                if r.get("Counter_Name") == counter and "elem_f2_fast" in r["Kernel_Name"] and ("true" in r["Kernel_Name"] or "(bool)1" in r["Kernel_Name"]):
                    tot += float(r["Counter_Value"]); cnt += 1
            if cnt == 0:
                return None, f"no elem_f2_fast rows in the {counter} pass", kstats
            vals[counter] = tot / cnt
        except Exception as e:
            return None, f"{counter} pass failed: {type(e).__name__} {knote}", kstats
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return (1024.0 * (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]),
            "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run (KiB; FETCH doubled on gfx950) " + knote, kstats)


FP64_MFMA_PEAK_TFLOPS = 78.6       # MI355X dense fp64 matrix peak (spec); tools/micro/mfma_f64_peak.hip sustains 49


def _kernel_group(kstats, prefixes, condensing=None):
    """(calls, total_ns, {short name: (calls, avg_us)}) of the kernels whose demangled name contains one of `prefixes`."""
    calls, total, detail = 0, 0.0, {}
    for name, (c, avg, tot) in (kstats or {}).items():
        for pre in prefixes:
            if pre in name:
                calls += c; total += tot
                key = pre + ("<...>" if "<" in name.split(pre, 1)[1][:2] else "")
                pc, pt = detail.get(key, (0, 0.0))
                detail[key] = (pc + c, pt + tot)
                break
    return calls, total, {k: dict(calls=c, avg_us=t / c / 1e3) for k, (c, t) in detail.items() if c}


FACTOR_KERNELS = ("mf_big_step", "mf_big_gather", "mf_big_schur", "mf_big_diag0", "mf_big_assemble", "mf_big_panel", "mf_big_update",
                  "mf_factor_small", "mf_factor_wave", "mf_factor_tiny", "border_tail_kernel")
BACKWARD_KERNELS = ("mf_bwd_inv", "mf_backward_small", "mf_backward_tiny", "mf_bwd_big")


def solver_roofline(kstats, chain, stats, hipevent_factor_us, hipevent_trisolve_us, factorizations_in_child=4):
    """`roofline_solver`: what bounds the sparse LDL' that takes most of a Newton iteration.  Neither HBM nor the matrix
    cores: the fraction of each is reported from the rocprofv3 kernel durations of the child pass, next to the latency
    chain that does bound it (sequential 32-column pivot blocks of the large fronts x device time per block)."""
    arena_bytes = 8.0 * chain["arena_doubles"]
    floor_bytes = 2.0 * arena_bytes                    # every frontal entry written once (assembly) and read once (update / parent)
    out = dict(bound="latency (pivot chain): neither hbm nor mfma",
               arena_bytes=arena_bytes, algorithmic_bytes=floor_bytes,
               algorithmic_bytes_note="2 x arena: every front is formed once and consumed once; the trailing updates of the "
                                      "32-column steps re-read and re-write part of it (extra_trailing_bytes)",
               extra_trailing_bytes=8.0 * chain["extra_trailing_doubles"],
               factor_flops=chain["factor_flops"], pivot_blocks_on_critical_path=chain["pivot_blocks_on_critical_path"],
               large_front_tree_levels=chain["large_front_tree_levels"], launches_per_factorization=chain["launches_per_factorization"],
               launches_per_backward_sweep=chain["launches_per_backward_sweep"], tree_levels=stats["tree_levels"],
               hipevent_factor_avg_us_all_levels=hipevent_factor_us, hipevent_trisolve_avg_us_all_levels=hipevent_trisolve_us)
    if kstats:
        fc, ft, fd = _kernel_group(kstats, FACTOR_KERNELS)
        bc, bt, bd = _kernel_group(kstats, BACKWARD_KERNELS)
        fac_us = ft / 1e3 / factorizations_in_child
        bwd_us = bt / 1e3 / factorizations_in_child
        step = fd.get("mf_big_step", {}).get("avg_us", 0.0)
        out.update(source="rocprofv3 --kernel-trace --stats child pass (tools/gpu_kernels.py: 4 fine-level Newton directions)",
                   factor_us=fac_us, backward_us=bwd_us, kernels=dict(factor=fd, backward=bd),
                   hbm=dict(achieved=floor_bytes / (fac_us * 1e-6) / 1e9 if fac_us else None, peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=(floor_bytes / (fac_us * 1e-6) / 1e9 / HBM_PEAK_GBS) if fac_us else None),
                   mfma=dict(achieved=chain["factor_flops"] / (fac_us * 1e-6) / 1e12 if fac_us else None, peak=FP64_MFMA_PEAK_TFLOPS,
                             unit="TFLOP/s", frac=(chain["factor_flops"] / (fac_us * 1e-6) / 1e12 / FP64_MFMA_PEAK_TFLOPS) if fac_us else None),
                   critical_path=dict(blocks=chain["pivot_blocks_on_critical_path"], mf_big_step_avg_us=step,
                                      model_us=chain["pivot_blocks_on_critical_path"] * step,
                                      note="device time of the step launches alone if every one of them sat on the chain; "
                                           "the rest of factor_us is assembly (gather) and the LDS-front levels"))
    else:
        out["source"] = "kernel-trace child pass unavailable: hipEvent stage timers only"
    return out


def run_workload(m, DeviceMGBProblem, mgb_driver, MGBConvergenceFailure, L, p, dev_index, warmup, rank):
    """Build the problem on the first hierarchy variant that converges; returns everything the timed loop needs."""
    for rs_kwargs in hierarchies(p, L):
        t0 = time.perf_counter()
        prob = build_problem(L, p, rs_kwargs)
        t_setup = time.perf_counter() - t0
        t0 = time.perf_counter()
        D = DeviceMGBProblem(prob, device_id=dev_index)
        t_upload = time.perf_counter() - t0
        try:
            t0 = time.perf_counter()
            mgb_driver(D)          # untimed: builds plans + symbolic factorizations, proves convergence
            t_first = time.perf_counter() - t0
            for _ in range(max(warmup - 1, 0)):
                mgb_driver(D)
            return prob, D, rs_kwargs, dict(host_setup=t_setup, upload=t_upload, first_solve=t_first,
                                            time_to_first_solution=t_setup + t_upload + t_first)
        except MGBConvergenceFailure as e:
            if rank == 0:
                print(f"bench: hierarchy {rs_kwargs} failed ({e.code}); trying the next variant", file=sys.stderr)
            D.close()
    raise SystemExit("bench.py: no hierarchy variant converged")


def timed_solves(mgb_driver, D, steps):
    its_total, solve_s, core_s, last = 0, 0.0, 0.0, None
    for _ in range(steps):
        SOL = mgb_driver(D)
        last = SOL
        for key in ("SOL_feasibility", "SOL_main"):
            if SOL[key] is not None:
                its_total += int(SOL[key]["its"].sum())
                solve_s += SOL[key]["solve_seconds"]
                core_s += SOL[key]["t_elapsed"]
    return its_total, solve_s, core_s, last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--L", type=int, default=int(os.environ.get("MGB_BENCH_L", "9")))
    ap.add_argument("--p", type=float, default=float(os.environ.get("MGB_BENCH_P", "1.0")))
    ap.add_argument("--cpu-budget", type=float, default=float(os.environ.get("MGB_BENCH_CPU_BUDGET", "20")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child passes")
    ap.add_argument("--no-p15", action="store_true", help="skip the north_star p=1.5 line")
    ap.add_argument("--sharded", action="store_true",
                    help="N > 1: ONE workload domain-decomposed over the ranks (strong scaling) instead of N replicas")
    ap.add_argument("--device-pointers", action="store_true", help="--sharded: pass device pointers to RCCL (no host staging)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # PMC passes first: child processes, before this process initialises the GPU (rank 0, N = 1 only)
    traffic, traffic_note, kstats = None, "not measured", None
    if world == 1 and rank == 0 and not args.no_traffic and os.environ.get("MGB_BENCH_TRAFFIC", "1") == "1":
        traffic, traffic_note, kstats = profile_children(args.L, args.p, hierarchies(args.p, args.L)[0])

    import torch
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
    from mgb_amd.solve import mgb_driver, MGBConvergenceFailure

    if args.sharded and world > 1:
        # Strong scaling: ONE workload, elements partitioned over the ranks, interior unknowns eliminated per rank, the
        # interface front of the factorization summed over ranks (mgb_amd/sharded.py).
        from mgb_amd.sharded import ShardedSolver
        prob = build_problem(args.L, args.p, {})
        S = ShardedSolver(prob, dist, device_id=dev_index, torch_device="cpu" if rehearsal else "cuda",
                          device_pointers=args.device_pointers and not rehearsal)
        for _ in range(max(args.warmup, 1)):
            S.solve_local()
        barrier()
        t0 = time.perf_counter()
        its_total = 0
        for _ in range(args.steps):
            SOL = S.solve_local()
            its_total += int(SOL["SOL_main"]["its"].sum())
        barrier()
        elapsed = time.perf_counter() - t0
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed_max = float(tmax.item())
        iface = [int(sh.iface.size) for sh in S.shards[0]]
        local = [int(sh.cols.size) for sh in S.shards[0]]
        S.close()
        if rank == 0:
            print(json.dumps({
                "metric": "Newton iters/sec + wall-clock to converge, 2D P2 p-Laplace L=%d" % args.L,
                "value": its_total / elapsed_max, "unit": "newton_iters/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": 1e3 * elapsed_max / max(args.steps, 1), "higher_is_better": True,
                "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                "config": {"workload": f"fem2d_P2() p={args.p} L={args.L}, reference-default hierarchy, ONE solve domain-decomposed over {world} ranks",
                           "parallelism": f"dd{world}: contiguous element ranges, interior elimination per rank, interface front all-reduced",
                           "interface_unknowns_per_level": iface, "local_unknowns_per_level_rank0": local,
                           "collectives": "RCCL device pointers" if (args.device_pointers and not rehearsal) else "host-staged all-reduce"},
                "wall_clock_to_converge_s": elapsed_max / max(args.steps, 1),
                "newton_iterations_per_solve": its_total / max(args.steps, 1)}))
        dist.barrier()
        dist.destroy_process_group()
        return

    # The HIP runtime initialises a device lazily at the first allocation (0.15 s on an MI355X box, once per process, nothing
    # of this package): done here so that `setup_s` times the package's own work; reported beside it.
    t0 = time.perf_counter()
    torch.zeros(1, device=f"cuda:{dev_index}")
    torch.cuda.synchronize()
    hip_runtime_init_s = time.perf_counter() - t0
    prob, D, used, setup = run_workload(m, DeviceMGBProblem, mgb_driver, MGBConvergenceFailure, args.L, args.p, dev_index,
                                        args.warmup, rank)
    setup["hip_runtime_init_not_included"] = hip_runtime_init_s
    if traffic is not None and used != hierarchies(args.p, args.L)[0]:
        # the PMC child passes profiled the first hierarchy variant; the timed workload fell through to another one
        traffic, traffic_note, kstats = None, f"PMC passes ran on hierarchy {hierarchies(args.p, args.L)[0]}, the workload on {used}: not comparable", None
    barrier()
    t0 = time.perf_counter()
    its_total, solve_s, core_s, last = timed_solves(mgb_driver, D, args.steps)
    its_levels = [int(v) for v in last["SOL_main"]["its"].sum(axis=1)] if last is not None else None      # per level, one solve
    barrier()
    elapsed = time.perf_counter() - t0
    from mgb_amd.replicas import aggregate
    elapsed_max, its_all = aggregate(elapsed, its_total, dist, device="cpu" if rehearsal else "cuda")

    # ---- roofline of the dominant HBM kernel: fused element Hessian (f2), fine level ----
    # One more solve of the same workload, outside the timed region, with the library's hipEvent
    # stage timers switched on (events recorded on the library's own stream around every stage):
    # the fine-level launches are timed exactly as they occur in the Newton loop.
    main = D.main
    fine = len(main.level_sizes) - 1
    n = prob.M[0].w.size
    main.reset_stage_timers(True)
    mgb_driver(D)
    st = {k: main.stage_ms(k) for k in ("f2", "assemble", "f0", "f1", "f01", "restrict", "prolong", "factor", "trisolve")}
    main.reset_stage_timers(False)
    avg_us = {k: (1e3 * ms / cnt if cnt else 0.0) for k, (ms, cnt) in st.items()}
    gbs = lambda nbytes, us: (nbytes / (us * 1e-6) / 1e9) if us > 0 else 0.0
    # the roofline's launch duration: rocprofv3's kernel-trace average of the condensing f2 instantiation (child pass);
    # the hipEvent stage timer of the live solve is kept beside it (it brackets the launch, a few us more)
    f2_hipevent_us = avg_us["f2"]
    f2_rocprof = [(c, a) for name, (c, a, t) in (kstats or {}).items()
                  if "elem_f2_fast" in name and ("true" in name or "(bool)1" in name)]
    f2_us = (sum(c * a for c, a in f2_rocprof) / sum(c for c, a in f2_rocprof) / 1e3) if f2_rocprof else f2_hipevent_us
    achieved = gbs(BYTES["f2"] * n, f2_us)
    # aggregate of SURVEY 8(d): one fine Newton iteration = f2 + assembly + f0 + f1 (prolongation and the
    # R' gather are part of those stages' stage timers) against 1285 B / node.  The line search evaluates
    # f0 and f1 of a trial in ONE pass over the operators (stage "f01"); the separate f0 / f1 stages only
    # run once per Newton solve (the start point) and are not part of an iteration.
    trial_us = avg_us["f01"] if st["f01"][1] else avg_us["f0"] + avg_us["f1"]
    agg_us = avg_us["f2"] + avg_us["assemble"] + trial_us + avg_us["restrict"]
    roofline = dict(
        bound="hbm",
        kernel="elem_f2_fast<4,7,SigDefault,CONDENSE>: fused Dz + cone Hessian + element blocks + static condensation of "
               "the element's slack and bubble unknowns (writes the leaf fronts of the factorization), fine level -- the f2 "
               "stage of the Newton loop; it replaces round 2's f2 + assembly + leaf-level factorization kernels",
        achieved=achieved, peak=HBM_PEAK_GBS, unit="GB/s", frac=achieved / HBM_PEAK_GBS,
        numerator="SURVEY 8(d) algorithmic bytes of f2 alone: 347 B/node x n (the kernel also does the assembly's and the leaf "
                  "level's work, which 8(d) prices at 488 B/node more; not claimed here)",
        traffic=traffic, traffic_source=traffic_note,
        bytes_per_launch=BYTES["f2"] * n, avg_launch_us=f2_us,
        avg_launch_source=("rocprofv3 --kernel-trace --stats child pass" if f2_rocprof else "hipEvent stage timers (no kernel trace)"),
        avg_launch_us_hipevents=f2_hipevent_us, launches=int(st["f2"][1]),
        moved_bytes_model=F2_MOVED_BYTES_PER_NODE * n,
        moved_gbs=gbs(F2_MOVED_BYTES_PER_NODE * n, f2_us),
        moved_frac=gbs(F2_MOVED_BYTES_PER_NODE * n, f2_us) / HBM_PEAK_GBS,
        measured_gbs=(gbs(traffic, f2_us) if traffic else None),
        measured_frac=(gbs(traffic, f2_us) / HBM_PEAK_GBS if traffic else None),
        aggregate=dict(stages_us={"f0+f1 (one pass)": trial_us, "restrict": avg_us["restrict"], "f2": avg_us["f2"],
                                  "assemble": avg_us["assemble"]}, total_us=agg_us,
                       note=("bytes are SURVEY 8(d)'s algorithmic 1285 B/node; the device moves fewer: the trial reads the operators "
                             "once for f0 and f1, and the fine level never forms H as a CSR value array (the factorization reads "
                             "single-contribution entries from the element-block slab, 'assemble' sums only the shared ones)"),
                       bytes=sum(BYTES.values()) * n, gbs=gbs(sum(BYTES.values()) * n, agg_us),
                       frac=gbs(sum(BYTES.values()) * n, agg_us) / HBM_PEAK_GBS),
        factor_avg_us=avg_us["factor"], trisolve_avg_us=avg_us["trisolve"],
        solver_note="factor carries the forward substitution (bordered LDL'); trisolve is the backward sweep only")
    stats = main.solver_stats(fine)
    roofline_solver = solver_roofline(kstats, main.solver_chain(fine), stats, avg_us["factor"], avg_us["trisolve"])

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
                                   f"amg(subdivide(fem2d_P2(), {args.L})), n={n} broken nodes, {main.level_sizes[fine]} fine unknowns",
                       "hierarchy": f"amg_ruge_stuben({used if used else 'max_coarse=2: the reference default'})",
                       "hierarchy_is_reference_default": not used,
                       "hierarchy_note": ("the ladder comes from this package's restatement of AlgebraicMultigrid.jl's Ruge-Stueben "
                                          "(the third-party package is not in the reference tree: P entries are parity-unpinned, "
                                          "DESIGN.md section 1); the reference's tests pin only the downstream z"),
                       "levels": [int(v) for v in main.level_sizes],
                       "parallelism": "replicas" if world > 1 else "single",
                       "solver_controls": "reference defaults (tol=sqrt(eps), t=0.1, kappa=10, max_newton=8, backtracking)"},
            "wall_clock_to_converge_s": elapsed_max / max(args.steps, 1),
            "newton_iterations_per_solve": its_total / max(args.steps, 1),
            "linear_solve_fraction": solve_s / core_s if core_s > 0 else None,
            "t_steps": int(sm["k"]),
            "setup_s": setup,
            "factorization": {k: stats[k] for k in ("fronts", "max_front", "factor_flops", "tree_levels", "nnz", "unknowns")},
            "roofline": roofline,
            "roofline_solver": roofline_solver,
        }
    D.close()
    # ---- north_star target line: fem2d_P2 p = 1.5 at the same L (BASELINE.json north_star), rank 0 / N = 1 ----
    if rank == 0 and world == 1 and not args.no_p15 and args.p != 1.5:
        prob15, D15, used15, setup15 = run_workload(m, DeviceMGBProblem, mgb_driver, MGBConvergenceFailure, args.L, 1.5,
                                                    dev_index, 1, rank)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        its15, ss15, cs15, last15 = timed_solves(mgb_driver, D15, args.steps)
        its15_levels = [int(v) for v in last15["SOL_main"]["its"].sum(axis=1)] if last15 is not None else None
        torch.cuda.synchronize()
        el15 = time.perf_counter() - t0
        out["north_star_p15"] = dict(workload=f"fem2d_P2() p=1.5 L={args.L}", value=its15 / el15, unit="newton_iters/s",
                                     wall_clock_to_converge_s=el15 / max(args.steps, 1),
                                     newton_iterations_per_solve=its15 / max(args.steps, 1),
                                     hierarchy=f"amg_ruge_stuben({used15 if used15 else 'max_coarse=2: the reference default'})",
                                     hierarchy_is_reference_default=not used15,
                                     levels=[int(v) for v in D15.main.level_sizes],
                                     linear_solve_fraction=ss15 / cs15 if cs15 > 0 else None, setup_s=setup15)
        if not args.no_cpu_baseline:        # the north_star target is ">= 10x host CPU at p = 1.5": its own CPU figure
            cb = cpu_baseline(prob15, min(args.cpu_budget, 12.0), its_per_level=its15_levels)
            out["north_star_p15"]["cpu_baseline"] = cb
            if cb["value"] > 0:
                out["north_star_p15"]["gpu_over_cpu"] = out["north_star_p15"]["value"] / cb["value"]
        D15.close()
        # ---- the one published number on a BASELINE mesh: fem2d_P2 p = 1.0 L = 7 (bench.md:21) ----
        if args.L != 7:
            prob7, D7, used7, setup7 = run_workload(m, DeviceMGBProblem, mgb_driver, MGBConvergenceFailure, 7, 1.0, dev_index, 1, rank)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            its7, _, _, _ = timed_solves(mgb_driver, D7, args.steps)
            torch.cuda.synchronize()
            el7 = time.perf_counter() - t0
            out["reference_anchor_L7_p1"] = dict(
                workload="fem2d_P2() p=1.0 L=7 (57 344 nodes), total mgb_solve wall-clock as in tools/bench_cuda_vs_native.jl",
                wall_clock_to_converge_s=el7 / max(args.steps, 1), value=its7 / el7, unit="newton_iters/s",
                newton_iterations_per_solve=its7 / max(args.steps, 1),
                hierarchy=f"amg_ruge_stuben({used7 if used7 else 'max_coarse=2: the reference default'})",
                published=dict(source="/root/reference/bench.md:21", cpu_s=66.509, gpu_s=5.122,
                               hardware="DMOG cluster CPU (unspecified) / 1x NVIDIA A40 + cuDSS, Julia 1.11.6",
                               note="different hardware and a different Ruge-Stueben implementation (iteration counts differ): an anchor, not a like-for-like ratio"))
            D7.close()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(prob, args.cpu_budget, its_per_level=its_levels)
        if out["cpu_baseline"]["value"] > 0:
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
