"""`parabolic_solve`: implicit-Euler time stepping of the p-Laplace flow on a resident device image.

Host-side mirror of the reference's caller loop (src/Parabolic.jl:126-173): every time step is
one `mgb_solve` of a problem that differs from the previous one only in its linear-term grid
`f_grid` (built from the previous state) and its boundary/start grid `g_grid`.  The reference
rebuilds the device image and flushes its plan / factorization caches on every step
(src/mgb.jl:805-841); here the `(AMG, Convex)` pair is uploaded once and the assembly plans and
the symbolic factorization of every level stay resident across the steps (SURVEY.md section 8f,
rank 3) -- only the two grids travel per step.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np

from .convex import Convex, convex_Euclidian_power, intersect
from .multigrid import MultiGrid, prepare_amg
from .problem import MGBProblem, assemble, _default_slack_space, _dim


def default_D_parabolic(dim: int):
    """reference: src/Parabolic.jl:3-18."""
    return [("u", "id")] + [("u", s) for s in ("dx", "dy", "dz")[:dim]] + [("s1", "id"), ("s2", "id")]


def default_f_parabolic(dim: int) -> Callable:
    """reference: src/Parabolic.jl:19-22: (f1, w1, w2) -> [f1, 0 (dim times), w1, w2]."""
    return lambda f1, w1, w2: np.array([f1] + [0.0] * dim + [w1, w2])


def default_g_parabolic(dim: int) -> Callable:
    """reference: src/Parabolic.jl:24-27."""
    if dim == 1:
        return lambda t, x: np.array([x[0], 0.0, 0.0])
    return lambda t, x: np.array([float(np.sum(np.asarray(x[:dim]) ** 2)), 0.0, 0.0])


def parabolic_idx1(dim: int):
    """1-based rows of D entering s1 >= u^2 (src/Parabolic.jl:31-34)."""
    return (1, 2 + dim)


def parabolic_idx2(dim: int):
    """1-based rows of D entering s2 >= |grad u|^p (src/Parabolic.jl:37-40)."""
    return tuple(range(2, 2 + dim)) + (3 + dim,)


@dataclass
class ParabolicSOL:
    """reference: `ParabolicSOL`, src/Parabolic.jl:56-63."""
    geometry: object
    ts: np.ndarray
    u: List[np.ndarray]          # one (n_nodes, n_components) matrix per time stamp
    steps: Optional[list] = None  # the per-step MGBSOL objects (diagnostics)


def parabolic_solve(mg: MultiGrid, state_variables=None, dim: Optional[int] = None, f1: Optional[Callable] = None,
                    f_default: Optional[Callable] = None, p: float = 1.0, h: float = 0.2, t0: float = 0.0,
                    t1: float = 1.0, ts: Optional[Sequence[float]] = None, f1_grid=None, f_grid: Optional[Callable] = None,
                    g: Optional[Callable] = None, g_grid: Optional[Callable] = None, D=None, Q: Optional[Convex] = None,
                    verbose: bool = False, solver: Optional[Callable] = None, **rest) -> ParabolicSOL:
    """reference: `parabolic_solve`, src/Parabolic.jl:126-173.

    `solver(prob, **rest)` defaults to the device `mgb_solve` with one resident image for all
    steps; the parity tests pass the CPU oracle's solve here to run the very same loop."""
    geom = mg.geometry
    if dim is None:
        dim = _dim(geom)
    if state_variables is None:
        sp = _default_slack_space(geom)
        state_variables = [("u", "dirichlet"), ("s1", sp), ("s2", sp)]
    if D is None:
        D = default_D_parabolic(dim)
    if ts is None:
        nsteps = int(np.floor((t1 - t0) / h + 1e-12))
        ts = t0 + h * np.arange(nsteps + 1)           # Julia range t0:h:t1
    ts = np.asarray(ts, dtype=np.float64)
    x = geom.xflat
    nx = x.shape[0]
    if f1 is None:
        f1 = lambda t, xx: 0.5
    if f_default is None:
        f_default = default_f_parabolic(dim)
    if f1_grid is None:
        f1_grid = np.array([[f1(tj, xi) for tj in ts] for xi in x], dtype=np.float64)      # (nodes, times)
    if f_grid is None:
        def f_grid(z, j):          # j is 0-based here; the reference's is 1-based
            dt = ts[j] - ts[j - 1]
            return np.stack([f_default(dt * f1_grid[i, j] - z[i, 0], 0.5, dt / p) for i in range(nx)], axis=0)
    if g is None:
        g = default_g_parabolic(dim)
    if g_grid is None:
        g_grid = lambda j: np.stack([np.asarray(g(ts[j], xi), dtype=np.float64) for xi in x], axis=0)
    if Q is None:
        Q = intersect(mg,
                      convex_Euclidian_power(mg, idx=parabolic_idx1(dim), p_grid=np.full(nx, 2.0)),
                      convex_Euclidian_power(mg, idx=parabolic_idx2(dim), p_grid=np.full(nx, float(p))))
    n = len(ts)
    U = [g_grid(k) for k in range(n)]
    M = prepare_amg(mg, state_variables, D)      # built once, reused by every step (src/Parabolic.jl:158)
    sols = []
    D_dev = None
    try:
        for k in range(n - 1):
            prob = assemble(mg, dim=dim, state_variables=state_variables, D=D, M=M, g_grid=U[k + 1],
                            f_grid=f_grid(U[k], k + 1), Q=Q)
            if solver is not None:
                sol = solver(prob, **rest)
                z = sol["z"] if isinstance(sol, dict) else sol.z
            else:
                from .device import native_to_device, default_device
                from .solve import mgb_driver, MGBSOL
                if D_dev is None:
                    D_dev = native_to_device(default_device(), prob)
                else:
                    D_dev.prob = prob                  # same (AMG, Convex): only the grids change
                lines: List[str] = []
                SOL = mgb_driver(D_dev, printlog=lambda *a: lines.append("".join(str(v) for v in a)), **rest)
                sol = MGBSOL(SOL["z"], SOL["SOL_feasibility"], SOL["SOL_main"], "\n".join(lines), geom)
                z = sol.z
            sols.append(sol)
            U[k + 1] = z
            if verbose:
                print(f"parabolic_solve: step {k + 1}/{n - 1} t={ts[k + 1]:g}", flush=True)
    finally:
        if D_dev is not None:
            D_dev.close()
    return ParabolicSOL(geom, ts, U, sols)
