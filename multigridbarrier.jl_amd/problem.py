"""`assemble`: lower a problem specification to a closure-free MGBProblem (host, CPU).

reference: src/mgb.jl:587-613 (defaults), :650-727 (`MGBProblem`, `assemble`).  As in
the reference, assembly always runs on the CPU and yields pure data; `mgb_solve`
moves that data to the device once.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, List, Optional, Sequence, Tuple

import numpy as np

from . import fem2d_p1, fem2d_p2, spectral, tensorfem
from .convex import Convex, convex_Euclidian_power
from .multigrid import AMG, Geometry, MultiGrid, prepare_amg


def _dim(geom: Geometry) -> int:
    return int(geom.discretization.dim)


def default_f(dim: int) -> Callable:
    """reference: src/mgb.jl:587-590."""
    return lambda x: np.array([0.5] + [0.0] * dim + [1.0])


def default_g(dim: int) -> Callable:
    """reference: src/mgb.jl:591-594."""
    if dim == 1:
        return lambda x: np.array([x[0], 2.0])
    return lambda x: np.array([float(np.dot(x[:dim], x[:dim])), 100.0])


def default_D(dim: int) -> List[Tuple[str, str]]:
    """reference: src/mgb.jl:595-607."""
    return [("u", "id")] + [("u", s) for s in ("dx", "dy", "dz")[:dim]] + [("s", "id")]


def default_idx(dim: int) -> Tuple[int, ...]:
    """1-based, like the reference (src/mgb.jl:610-613)."""
    return tuple(range(2, dim + 3))


def _default_slack_space(geom: Geometry) -> str:
    """reference: src/multigrid.jl:420, src/fem2d_P2.jl:70."""
    disc = geom.discretization
    if isinstance(disc, fem2d_p2.FEM2D_P2) and not disc.bubble:
        return "broken_P1"
    return "full"


@dataclass
class MGBProblem:
    """reference: src/mgb.jl:666-674."""

    M: Tuple[AMG, AMG]
    f: np.ndarray        # (n, nD) linear-term grid
    g: np.ndarray        # (n, nu) Dirichlet / initial-data grid
    Q: Convex
    geometry: Geometry


def assemble(mg: MultiGrid, dim: Optional[int] = None, state_variables=None, D=None, x=None,
             p: float = 1.0, g: Optional[Callable] = None, f: Optional[Callable] = None,
             g_grid=None, f_grid=None, Q: Optional[Convex] = None, M=None, **_ignored) -> MGBProblem:
    """reference: `assemble`, src/mgb.jl:711-727."""
    geom = mg.geometry
    if dim is None:
        dim = _dim(geom)
    if state_variables is None:
        state_variables = [("u", "dirichlet"), ("s", _default_slack_space(geom))]
    if D is None:
        D = default_D(dim)
    if x is None:
        x = geom.xflat
    nx = x.shape[0]
    if g_grid is None:
        if g is None:      # default_g, vectorised over the nodes
            g_grid = (np.stack([x[:, 0], np.full(nx, 2.0)], axis=1) if dim == 1 else
                      np.stack([np.sum(x[:, :dim] ** 2, axis=1), np.full(nx, 100.0)], axis=1))
        else:
            g_grid = np.stack([np.atleast_1d(np.asarray(g(xi), dtype=np.float64)) for xi in x], axis=0)
    if f_grid is None:
        if f is None:      # default_f
            f_grid = np.tile(np.array([0.5] + [0.0] * dim + [1.0]), (nx, 1))
        else:
            f_grid = np.stack([np.atleast_1d(np.asarray(f(xi), dtype=np.float64)) for xi in x], axis=0)
    if Q is None:
        pval = float(p)
        Q = convex_Euclidian_power(mg, idx=default_idx(dim), p_grid=np.full(x.shape[0], pval))
    if M is None:
        M = prepare_amg(mg, state_variables, D)
    Q.validate_inputs(len(M[0].D_fine))
    return MGBProblem(M, np.asarray(f_grid, dtype=np.float64), np.asarray(g_grid, dtype=np.float64), Q, geom)


# ---------------------------------------------------------------------------
# discretization-dispatched front doors (the reference dispatches on the
# Geometry's Discretization type parameter)
# ---------------------------------------------------------------------------

def amg(geom: Geometry, **kw) -> MultiGrid:
    disc = geom.discretization
    if isinstance(disc, fem2d_p1.FEM2D_P1):
        return fem2d_p1.amg(geom, **kw)
    if isinstance(disc, fem2d_p2.FEM2D_P2):
        return fem2d_p2.amg(geom, **kw)
    if isinstance(disc, tensorfem.TensorFEM):
        return tensorfem.amg(geom, **kw)
    if isinstance(disc, (spectral.SPECTRAL1D, spectral.SPECTRAL2D)):
        return spectral.amg(geom)
    raise TypeError(f"amg: unsupported discretization {type(disc).__name__}")


def subdivide(geom: Geometry, L: int) -> Geometry:
    disc = geom.discretization
    if isinstance(disc, fem2d_p1.FEM2D_P1):
        return fem2d_p1.subdivide(geom, L)
    if isinstance(disc, fem2d_p2.FEM2D_P2):
        return fem2d_p2.subdivide(geom, L)
    if isinstance(disc, tensorfem.TensorFEM):
        return tensorfem.subdivide(geom, L)
    if isinstance(disc, (spectral.SPECTRAL1D, spectral.SPECTRAL2D)):
        return geom      # no geometric subdivision for spectral (reference: src/multigrid.jl:428-430)
    raise TypeError(f"subdivide: unsupported discretization {type(disc).__name__}")


def geometric_mg(geom: Geometry, L: int) -> MultiGrid:
    """reference: `geometric_mg`, src/multigrid.jl:422-431 (spectral discretizations ignore L and return `amg`)."""
    disc = geom.discretization
    if isinstance(disc, fem2d_p1.FEM2D_P1):
        return fem2d_p1.geometric_mg(geom, L)
    if isinstance(disc, fem2d_p2.FEM2D_P2):
        return fem2d_p2.geometric_mg(geom, L)
    if isinstance(disc, tensorfem.TensorFEM):
        return tensorfem.geometric_mg(geom, L)
    if isinstance(disc, (spectral.SPECTRAL1D, spectral.SPECTRAL2D)):
        return spectral.amg(geom)
    raise TypeError(f"geometric_mg: unsupported discretization {type(disc).__name__}")


def find_boundary(geom: Geometry):
    disc = geom.discretization
    if isinstance(disc, fem2d_p1.FEM2D_P1):
        return fem2d_p1.find_boundary(geom)
    if isinstance(disc, fem2d_p2.FEM2D_P2):
        return fem2d_p2.find_boundary(geom)
    if isinstance(disc, tensorfem.TensorFEM):
        return tensorfem.find_boundary(geom)
    if isinstance(disc, spectral.SPECTRAL1D):     # informational only (reference: src/spectral1d.jl:119-130)
        return [(0, 0), (disc.n - 1, 0)]
    if isinstance(disc, spectral.SPECTRAL2D):     # perimeter of the tensor grid (src/spectral2d.jl:52-69), 0-based
        n = disc.n
        return [(j * n + i, 0) for j in range(n) for i in range(n) if i in (0, n - 1) or j in (0, n - 1)]
    raise TypeError(f"find_boundary: unsupported discretization {type(disc).__name__}")
