"""`mgb_solve`: the reference's top-level entry on the HIP backend.

Mirrors `mgb_solve` / `mgb_driver` (reference: src/mgb.jl:798-842, :332-584): move the
assembled problem to the device, probe feasibility of the start, run phase I with box
escalation when needed, hand over with `_matched_t`, run the main t-ramp, move the solution
back.  All loops below `mgb_core` (level sweep, Newton, line search) run inside
`libmgbhip.so`; this file only orchestrates the rarely-taken phase-I branch and packs the
diagnostics the reference returns in `MGBSOL`.
"""
from __future__ import annotations

import ctypes as C
import math
import time
from dataclasses import dataclass, field
from typing import Any, Dict, Optional

import numpy as np

from . import device as dev
from .device import ERR_CONVERGENCE, OK, DeviceMGBProblem, HIPDevice, native_to_device
from .problem import MGBProblem

EPS = float(np.finfo(np.float64).eps)


class MGBConvergenceFailure(Exception):
    """reference: src/utils.jl:178-184; `code` in {'infeasible', 'feasibility_Rmax', 'stall',
    'iteration_limit', 'failure'}."""

    def __init__(self, message: str, code: str = "failure"):
        super().__init__(message)
        self.message = message
        self.code = code


@dataclass
class MGBSOL:
    """reference: src/mgb.jl:637-643."""

    z: np.ndarray
    SOL_feasibility: Optional[Dict[str, Any]]
    SOL_main: Dict[str, Any]
    log: str
    geometry: Any


def _barrier_weights(w: np.ndarray, barrier_nodes):
    """reference: src/convex.jl:279-304."""
    if barrier_nodes is None or (isinstance(barrier_nodes, str) and barrier_nodes == "colon"):
        return None
    sel = np.asarray(barrier_nodes)
    if sel.dtype == bool:
        if sel.size != w.size:
            raise ValueError(f"barrier_nodes mask has length {sel.size} but the mesh has {w.size} nodes")
        nz = sel.astype(np.float64)
    else:
        if sel.size == 0:
            raise ValueError("barrier_nodes must select at least one node")
        if sel.min() < 0 or sel.max() >= w.size:
            raise ValueError(f"barrier_nodes indices must lie in 0:{w.size - 1}")
        nz = np.zeros(w.size)
        nz[sel] = 1.0
    m = nz.sum()
    if m <= 0:
        raise ValueError("barrier_nodes selects no nodes")
    if m == nz.size:
        return None
    return nz / m


def _options(P, tol, t, kappa, maxit, max_newton, line_search, stopping_criterion, finalize, early_stop,
             early_stop_fn=None, keep=None, n_nodes=None):
    """Reference keyword arguments -> `mgbhip_options`.  `stopping_criterion` is either a tagged tuple
    selecting a built-in rule or any callable with the reference's signature
    `stop(ymin, ynext, gmin, gnext, n, ndecmin, ndec) -> bool` (src/newton.jl:187,222-225): `gnext` arrives
    as a one-element array holding its norm (so `norm(gnext)` is right) and `n` as None -- the vectors
    stay on the device; `early_stop_fn` any callable
    `z -> bool` or `(z, t) -> bool` on the stacked iterate (src/mgb.jl:85-89).  Callables cross the C
    ABI as function pointers; `keep` collects the ctypes thunks so they outlive the call."""
    o = P.default_options(n_nodes)       # n_nodes: node count of the whole mesh when P holds one rank's slice (sharded.py)
    if tol is not None:
        o.tol = float(tol)
    o.t = float(t)
    if kappa is not None:
        o.kappa = float(kappa)
    if maxit is not None:
        o.maxit = int(maxit)
    if max_newton is not None:
        o.max_newton = int(max_newton)
    if line_search is not None:
        kind = line_search[0] if isinstance(line_search, (tuple, list)) else line_search
        args = list(line_search[1:]) if isinstance(line_search, (tuple, list)) else []
        if kind == "backtracking":
            o.line_search = 0
            if len(args) > 0:
                o.ls_beta = float(args[0])
            if len(args) > 1:
                o.ls_c1 = float(args[1])
        elif kind == "illinois":
            o.line_search = 1
            if len(args) > 0:
                o.ls_beta = float(args[0])
        else:
            raise ValueError("line_search must be ('backtracking', beta, c1) or ('illinois', beta)")
    if keep is None:
        keep = []
    o._keep = keep                     # the thunks live as long as the options object that points at them
    if callable(stopping_criterion):
        fn = stopping_criterion

        def _stop(ymin, ynext, gmin, gn, ndecmin, ndec, _u):
            # ctypes would print and swallow an exception raised in here and return 0 ("not converged"); in the
            # reference it propagates out of mgb_solve.  Keep the first one, end every Newton solve and the ramp
            # at once, and let _run_core re-raise it.
            if _pending(keep):
                return 1
            try:
                return 1 if fn(ymin, ynext, gmin, np.array([gn]), None, ndecmin, ndec) else 0
            except BaseException as e:          # noqa: BLE001 -- re-raised by _run_core
                keep.append(_CallbackError(e))
                return 1
        thunk = dev.STOP_FN(_stop)
        keep.append(thunk)
        o.stopping_criterion = C.cast(thunk, C.c_void_p)
    elif stopping_criterion is not None:
        kind = stopping_criterion[0]
        if kind == "inexact":
            o.stop_lambda_tol, o.stop_theta = float(stopping_criterion[1]), float(stopping_criterion[2])
        elif kind == "exact":
            o.stop_lambda_tol, o.stop_theta = -1.0, float(stopping_criterion[1])
        else:
            raise ValueError("stopping_criterion must be ('inexact', lambda_tol, theta) or ('exact', theta)")
    if finalize is False:
        o.finalize = 0
    elif finalize is not None and finalize is not True:
        o.finalize, o.finalize_theta = 1, float(finalize[1] if isinstance(finalize, (tuple, list)) else finalize)
    o.early_stop = int(early_stop)
    if early_stop_fn is not None:
        import inspect
        two = len(inspect.signature(early_stop_fn).parameters) >= 2
        zn = P.nu * P.n

        def _early(zp, t, _u):
            if _pending(keep):
                return 1
            try:
                z = np.ctypeslib.as_array(zp, shape=(zn,)).copy()
                return 1 if (early_stop_fn(z, t) if two else early_stop_fn(z)) else 0
            except BaseException as e:          # noqa: BLE001 -- re-raised by _run_core
                keep.append(_CallbackError(e))
                return 1
        thunk = dev.EARLY_FN(_early)
        keep.append(thunk)
        o.early_stop_fn = C.cast(thunk, C.c_void_p)
    if callable(stopping_criterion) and not o.early_stop_fn:
        # a raising stopping rule must also be able to end the t-ramp: a do-nothing early_stop that only reports it
        thunk = dev.EARLY_FN(lambda zp, t, _u: 1 if _pending(keep) else 0)
        keep.append(thunk)
        o.early_stop_fn = C.cast(thunk, C.c_void_p)
    return o


class _CallbackError:
    """An exception raised inside a user callable while the library was running it."""

    def __init__(self, exc):
        self.exc = exc


def _pending(keep):
    return any(isinstance(k, _CallbackError) for k in keep)


def _run_core(P, z, c, opt, what):
    status, znew, diag = P.mgb_core(z, c, opt, cap_steps=max(64, min(int(opt.maxit), 4096)))
    for k in getattr(opt, "_keep", None) or []:
        if isinstance(k, _CallbackError):          # a user callable raised: propagate like the reference would
            raise k.exc
    diag["z"] = znew
    if status == ERR_CONVERGENCE:
        code = "iteration_limit" if diag["failure_code"] == 2 else "stall"
        if diag["k"] == 1 and diag["failure_code"] == 1 and diag["t_final"] == opt.t:
            msg = f"Initial centering failed in mgb_solve at t={opt.t}, tol={opt.tol}, maxit={opt.maxit}."
        else:
            msg = (f"Convergence failure in mgb_solve ({what}) at t={diag['t_final']}, k={diag['k']}, "
                   f"tol={opt.tol}, maxit={opt.maxit}.")
        raise MGBConvergenceFailure(msg, code)
    return diag


# ---------------------------------------------------------------------------------------------------------------------
# A custom `line_search` closure (reference: src/mgb.jl:362, src/newton.jl:84-154) takes the objective closures
# themselves, which the resident ramp cannot call back into: such solves run the reference's loops here, on DEVICE vectors
# through the fine-grained entry points (mgbhip_f0_d / f1_d / f2_d / solve_d / prolong_add) --
# nothing but scalars crosses PCIe.  The Julia extension does the same with the reference's own `newton`.
# ---------------------------------------------------------------------------------------------------------------------

def _newton_on_device(F0, F1, F2solve, x, maxit, stop, line_search):
    """src/newton.jl:227-287 on DeviceVectors; F2solve(x, g) returns the Newton direction H(x)^{-1} g."""
    if not x.all_isfinite():
        raise FloatingPointError("newton: initial point has non-finite entries")
    y = F0(x)
    if not math.isfinite(y):
        raise FloatingPointError("newton: initial objective value is not finite")
    ymin, g, k, converged = y, F1(x), 0, False
    if not g.all_isfinite():
        raise FloatingPointError("newton: initial gradient has non-finite entries")
    gmin, incmin = g.norm(), math.inf
    while k < maxit and not converged:
        k += 1
        n = F2solve(x, g)
        if not n.all_isfinite():
            raise FloatingPointError("newton: Newton direction has non-finite entries")
        inc = g.dot(n)
        if inc <= 0:
            converged = abs(inc) <= EPS * max(abs(y), 1.0)
            break
        xn, yn, gn = line_search(x, y, g, n, F0, F1)
        if stop(ymin, yn, gmin, gn, n, math.sqrt(incmin), math.sqrt(inc)):
            converged = True
        x, y, g = xn, yn, gn
        gmin, ymin, incmin = min(gmin, g.norm()), min(ymin, y), min(inc, incmin)
    return x, k, converged


def _generic_core(P, z, c, opt, line_search, stopping_criterion=None, early_stop_fn=None):
    """mgb_step + mgb_core (src/mgb.jl:16-183) around `_newton_on_device`.  Returns the diagnostics dict of `_run_core`."""
    L = len(P.level_sizes)
    n = P.n
    sc = stopping_criterion
    if sc is None:
        lt, th = opt.stop_lambda_tol, opt.stop_theta
        sc = lambda ymin, yn, gmin, gn, nn, ndmin, nd: (lt >= 0 and nd < lt) or (yn >= ymin and gn.norm() >= th * gmin)
    fth = opt.finalize_theta
    fin = (lambda ymin, yn, gmin, gn, nn, ndmin, nd: yn >= ymin and gn.norm() >= fth * gmin) if opt.finalize else None
    zv = P.vec(np.asarray(z, dtype=np.float64))
    t0 = time.time()

    def step(cv, finalize_now, initial_step):
        nonlocal zv
        its = np.zeros(L, dtype=np.int64)
        zsave = zv.copy()

        def eta(j, J, crit, mi):
            lev = J - 1
            zJ = zv.copy()
            F0 = lambda s: P.f0_d(lev, s, cv, zJ)
            F1 = lambda s: P.f1_d(lev, s, cv, zJ)

            def F2solve(s, g):
                P.f2_d(lev, s, cv, zJ)
                return P.solve_d(lev, g)
            x, k, ok = _newton_on_device(F0, F1, F2solve, P.vec(length=P.level_sizes[lev]), mi, crit, line_search)
            its[lev] += k
            if ok:
                P.prolong_add(lev, x, zv)
            return ok

        def dac(j, J):
            if eta(j, J, sc, opt.maxit if (initial_step and J - j == 1) else opt.max_newton):
                return True
            mid = (j + J) // 2
            if mid == j or mid == J:
                return False
            return dac(j, mid) and dac(mid, J)
        ok = dac(0, L)
        if finalize_now and fin is not None:
            ok = eta(L - 1, L, fin, opt.maxit) and ok
        if not ok:
            zv = zsave                                   # z = SOL.z only on success (src/mgb.jl:150-157)
        return its, ok

    cflat = np.asfortranarray(c, dtype=np.float64).reshape(-1, order="F")
    cvec = lambda tt: P.vec(tt * cflat)
    tol, t, kappa, kappa0 = opt.tol, opt.t, opt.kappa, opt.kappa
    target = 1.0 / tol
    stop_early = (lambda zz, tt: False) if early_stop_fn is None else early_stop_fn
    its_all, ts, kappas = [], [], []
    its, ok = step(cvec(t), bool(opt.finalize) and t >= target, True)
    if not ok:
        raise MGBConvergenceFailure(f"Initial centering failed in mgb_solve at t={t}, tol={tol}, maxit={opt.maxit}.", "stall")
    its_all.append(its); ts.append(t); kappas.append(kappa)
    k = 1
    while t < target and kappa > 1 and k < opt.maxit and not stop_early(zv.to_host(), t):
        k += 1
        acc = np.zeros(L, dtype=np.int64)
        while kappa > 1:
            t1 = kappa * t
            its, ok = step(cvec(t1), bool(opt.finalize) and t1 >= target, False)
            acc += its
            if ok:
                if its.max() <= opt.max_newton * 0.5:
                    kappa = min(kappa0, kappa * kappa)
                t = t1
                break
            kappa = math.sqrt(kappa)
        its_all.append(acc); ts.append(t); kappas.append(kappa)
    if not (t >= target or stop_early(zv.to_host(), t)):
        raise MGBConvergenceFailure(f"Convergence failure in mgb_solve at t={t}, k={k}, kappa={kappa}, tol={tol}, maxit={opt.maxit}.",
                                    "stall" if kappa <= 1 else "iteration_limit")
    return dict(z=zv.to_host(), its=np.stack(its_all, axis=1), ts=np.array(ts), kappas=np.array(kappas), times=np.zeros(0),
                c_dot_Dz=np.zeros(0), t_elapsed=time.time() - t0, t_final=t, solve_seconds=0.0,
                newton_iterations=int(np.sum(its_all)), failure_code=0, k=k)


def mgb_driver(D: DeviceMGBProblem, t: float = 0.1, t_feasibility: Optional[float] = None,
               feasibility_Rmax: float = 1.0 / math.sqrt(EPS), tol=None, kappa=None, maxit=None, max_newton=None,
               stopping_criterion=None, line_search=None, finalize=None, barrier_nodes="default",
               printlog=lambda *a: None, early_stop=None, _shard=None):
    """reference: src/mgb.jl:332-584.  `_shard` (sharded.py): this process holds one rank's slice of a domain-decomposed
    problem -- maxima / feasibility flags are reduced over ranks and the barrier averages use the global node count."""
    rmax = (lambda x: _shard["reduce"].max(x)) if _shard else (lambda x: x)
    rall = (lambda b: _shard["reduce"].all(b)) if _shard else (lambda b: b)
    prob = D.prob
    keep: list = []          # ctypes thunks of user callables: alive until the solves return
    main = D.main
    M1 = prob.M[0]
    if t_feasibility is None:
        t_feasibility = t
    if isinstance(barrier_nodes, str) and barrier_nodes == "default":
        barrier_nodes = M1.w != 0
    bw_main = _shard["bw_main"] if _shard else _barrier_weights(M1.w, barrier_nodes)
    m = M1.w.size
    nD = len(M1.D_fine)
    c0, z0 = prob.f, prob.g
    ncomp = z0.shape[1]
    z2 = np.ascontiguousarray(z0.T).reshape(-1).copy()
    common = dict(tol=tol, kappa=kappa, maxit=maxit, max_newton=max_newton, line_search=line_search,
                  stopping_criterion=stopping_criterion, finalize=finalize, n_nodes=_shard["n_global"] if _shard else None)
    SOL_feasibility = None
    F, w_Dz = main.node_barrier(z2, want_Dz=True)
    if not rall(bool(np.all(np.isfinite(F)))):
        feas = D.feasibility
        if _shard:
            feas.set_barrier_weights(_shard["bw_feas"])
        sl = main.node_slack(z2)
        z1cols = np.concatenate([z0, (2 * np.maximum(sl, 1.0))[:, None]], axis=1)
        b = 2 * max(1.0, rmax(float(z1cols[:, -1].max())))
        c1 = np.zeros((m, nD + 1 + ncomp))
        c1[:, nD] = 1.0
        z1 = np.ascontiguousarray(z1cols.T).reshape(-1).copy()
        slack_of = lambda z: z[ncomp * m:(ncomp + 1) * m]
        feasible = lambda z: bool(rmax(float(slack_of(z).max())) < 0)
        Rbox = max(10.0, 10.0 * rmax(float(np.abs(z2).max())))
        Rmax = max(float(feasibility_Rmax), Rbox)
        while True:
            printlog("mgb_driver: feasibility phase with bounding box R=", Rbox)
            feas.set_box(float(b), Rbox)
            failure = None
            try:
                # (a callable line_search applies to the main phase; phase I runs the resident ramp with the default search)
                opt = _options(feas, t=t_feasibility, early_stop=1, keep=keep, **(dict(common, line_search=None) if callable(line_search) else common))
                SOL_feasibility = _run_core(feas, z1, c1, opt, "feasibility phase")
            except (MGBConvergenceFailure, dev.MGBHipError) as e2:   # each round is a probe (src/mgb.jl:505-515)
                failure = e2
            if failure is None:
                zf = SOL_feasibility["z"]
                if feasible(zf):
                    break
                vmax = rmax(max(float(np.abs(zf[k * m:(k + 1) * m]).max()) for k in range(ncomp)))
                smax = rmax(float(slack_of(zf).max()))
                if vmax <= Rbox / 2:
                    raise MGBConvergenceFailure(
                        "The problem appears to be infeasible: the feasibility subproblem converged to a minimizer "
                        f"with positive constraint violation (max slack ~ {smax}) strictly inside the bounding box "
                        f"(max |nodal value| ~ {vmax} <= R/2 with R = {Rbox}).", "infeasible")
                printlog("mgb_driver: phase-I minimizer presses the box (max |nodal value|=", vmax,
                         ", max slack=", smax, "); growing R")
            else:
                printlog("mgb_driver: feasibility solve failed at R=", Rbox, ": ", failure)
            Rnext = 10 * Rbox
            if Rnext > Rmax:
                reason = ("the phase-I minimizer still presses against the bounding box" if failure is None
                          else f"the last attempt failed with: {failure}")
                raise MGBConvergenceFailure(
                    f"Could not find a strictly feasible point with nodal values bounded by R = {Rbox} "
                    f"(cap feasibility_Rmax ~ {Rmax}); {reason}. The problem is infeasible, or its feasible points "
                    "have nodal values exceeding the cap (rescale the problem, or raise feasibility_Rmax).",
                    "feasibility_Rmax")
            Rbox = Rnext
        z2 = SOL_feasibility["z"][: z2.size].copy()
        main.set_barrier_weights(bw_main)
        tm = main.matched_t(z2, c0, t)
        printlog("_matched_t: starting main ramp at t=", tm)
        t = min(t, tm)
    main.set_barrier_weights(bw_main)
    if callable(line_search):
        # (x, y, g, n, F0, F1) -> (xnext, ynext, gnext) on device vectors: the generic loops of this module
        common_g = dict(common, line_search=None, stopping_criterion=None if callable(stopping_criterion) else stopping_criterion)
        opt = _options(main, t=t, early_stop=0, keep=keep, **common_g)
        two = early_stop is not None and len(__import__("inspect").signature(early_stop).parameters) >= 2
        es = None if early_stop is None else ((lambda zz, tt: bool(early_stop(zz, tt))) if two else (lambda zz, tt: bool(early_stop(zz))))
        SOL_main = _generic_core(main, z2, c0, opt, line_search,
                                 stopping_criterion=stopping_criterion if callable(stopping_criterion) else None, early_stop_fn=es)
    else:
        opt = _options(main, t=t, early_stop=0, early_stop_fn=early_stop, keep=keep, **common)
        SOL_main = _run_core(main, z2, c0, opt, "main phase")
    z = SOL_main["z"].reshape(ncomp, m).T.copy()
    return dict(z=z, SOL_feasibility=SOL_feasibility, SOL_main=SOL_main)


def mgb_solve(prob: MGBProblem, device=None, verbose: bool = False, logfile=None, device_id: int = 0,
              stream: Optional[int] = None, keep_device: bool = False, **rest) -> MGBSOL:
    """reference: `mgb_solve`, src/mgb.jl:798-842."""
    if device is None:
        device = dev.default_device()
    lines = []

    def printlog(*args):
        s = "".join(str(a) for a in args)
        lines.append(s)
        if logfile is not None:
            print(s, file=logfile)

    printlog("mgb_solve: device = ", getattr(device, "__name__", device))
    D = native_to_device(device, prob, device_id=device_id, stream=stream)
    try:
        SOL = mgb_driver(D, printlog=printlog, **rest)
    except BaseException:
        D.close()          # the throw path flushes plans / factorizations too (src/mgb.jl:832-839)
        raise
    sol = MGBSOL(SOL["z"], SOL["SOL_feasibility"], SOL["SOL_main"], "\n".join(lines), prob.geometry)
    if keep_device:
        sol.device = D
    else:
        D.close()
    return sol
