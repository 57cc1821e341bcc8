"""Closure-free convex-set descriptors.

The reference lowers every constraint closure to per-vertex grids at assembly and
keeps Julia functors that the GPU compiler specialises (reference: src/convex.jl:80-97,
src/convex_euclidian_power.jl:352-453, src/convex_linear.jl:78-223,
src/convex_piecewise.jl:114-182).  A C-ABI backend cannot compile arbitrary functors,
so the functor *families* are enumerated instead (SURVEY.md section 7 "closures -> closed
kernel set"): a `Convex` is a list of pieces, each an Euclidean-power cone or a linear
inequality block with its static index list and its grids, plus an optional
per-vertex select grid.  The HIP kernels (and the test oracle) interpret this
descriptor; no barrier arithmetic lives in this file.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .multigrid import MultiGrid

KIND_EP = 1       # Euclidean power cone
KIND_LINEAR = 2   # linear inequalities


@dataclass
class Piece:
    kind: int
    idx: Tuple[int, ...]          # 0-based positions into y (the D rows); Colon is resolved at assemble
    A: np.ndarray                 # EP: (n, nz*nz); LINEAR: (n, nc*ni)  (per-node matrix, column-major flattened)
    b: np.ndarray                 # EP: (n, nz);    LINEAR: (n, nc)
    p: Optional[np.ndarray] = None    # EP only, (n,)
    mu: Optional[np.ndarray] = None   # EP only, (n,)
    colon: bool = False           # idx was Colon(): validated against nD at assemble

    @property
    def ni(self) -> int:
        return len(self.idx)

    @property
    def nc(self) -> int:
        return self.b.shape[1]


@dataclass
class Convex:
    """reference: src/convex.jl:80-86 -- `args` are the grids carried by the pieces."""

    pieces: List[Piece]
    select: Optional[np.ndarray] = None   # (n, npieces) as float (non-zero = active); None = all active

    def validate_inputs(self, nD: int):
        """reference: src/convex.jl:54-69, :97."""
        for pc in self.pieces:
            if pc.colon:
                if pc.ni != nD:
                    raise ValueError(f"convex constraint with idx = Colon() expects exactly {pc.ni} D row(s), but D has {nD} row(s)")
            elif max(pc.idx) + 1 > nD:
                raise ValueError(f"convex constraint indexes input row {max(pc.idx) + 1}, but D has only {nD} row(s)")


def _grid(f: Callable, x: np.ndarray, width: Optional[int] = None) -> np.ndarray:
    """`map_rows(f, x)` on the host (reference: src/utils.jl:122-126)."""
    rows = [np.atleast_1d(np.asarray(f(xi), dtype=np.float64)).reshape(-1, order="F") for xi in x]
    out = np.stack(rows, axis=0)
    if width is not None and out.shape[1] != width:
        raise ValueError(f"grid has {out.shape[1]} columns per node, expected {width}")
    return out


def _resolve_idx(idx, n_default: Optional[int]):
    if idx is None:   # Colon()
        if n_default is None:
            raise ValueError("idx = Colon() with a UniformScaling A cannot determine the constraint dimension; pass an explicit idx")
        return tuple(range(n_default)), True
    idx = tuple(int(i) for i in idx)
    if len(idx) == 0:
        raise ValueError("idx must contain at least one input row")
    if any(i <= 0 for i in idx):
        raise ValueError(f"idx entries must be positive; got {list(idx)}")
    return tuple(i - 1 for i in idx), False      # the public API is 1-based like the reference


def convex_Euclidian_power(mg: MultiGrid, idx=None, A=None, b=None, p=None,
                           A_grid=None, b_grid=None, p_grid=None) -> Convex:
    """Power cone {y : s >= ||q||^p, [q; s] = A(x) y[idx] + b(x)} (reference:
    src/convex_euclidian_power.jl:352-453).  `idx` is 1-based like the reference,
    `None` means Colon()."""
    x = mg.geometry.xflat
    n = x.shape[0]
    if A_grid is None:
        if A is None:
            nz_guess = None if idx is None else len(idx)
            if nz_guess is None:
                raise ValueError("a UniformScaling A with idx = Colon() cannot determine the constraint dimension")
            A_grid = np.tile(np.eye(nz_guess).reshape(1, -1), (n, 1))
        else:
            A_grid = _grid(lambda xi: np.asarray(A(xi), dtype=np.float64), x)
    A_grid = np.asarray(A_grid, dtype=np.float64)
    nz = len(idx) if idx is not None else int(round(np.sqrt(A_grid.shape[1])))
    if nz * nz != A_grid.shape[1]:
        raise ValueError(f"A_grid has {A_grid.shape[1]} columns per node but nz = {nz} requires nz^2 = {nz * nz}")
    if b_grid is None:
        if b is None:
            b_grid = np.zeros((n, nz))
        else:
            def brow(xi):
                bx = b(xi)
                if np.isscalar(bx):
                    out = np.zeros(nz)
                    out[-1] = bx
                    return out
                return np.asarray(bx, dtype=np.float64)
            b_grid = _grid(brow, x)
    b_grid = np.asarray(b_grid, dtype=np.float64).reshape(n, -1)
    if b_grid.shape[1] != nz:
        raise ValueError(f"b_grid has {b_grid.shape[1]} value(s) per node but [q; s] has nz = {nz} components")
    if p_grid is None:
        p_grid = np.full(n, 2.0) if p is None else np.array([float(p(xi)) for xi in x])
    p_grid = np.asarray(p_grid, dtype=np.float64).reshape(n)
    # mu = 0 for p in {1, 2}, 1 for p < 2, 2 for p > 2 (reference: src/convex_euclidian_power.jl:380-381)
    mu_grid = np.where((p_grid == 2) | (p_grid == 1), 0.0, np.where(p_grid < 2, 1.0, 2.0))
    ridx, colon = _resolve_idx(idx, nz)
    return Convex([Piece(KIND_EP, ridx, A_grid, b_grid, p_grid, mu_grid, colon)])


def convex_linear(mg: MultiGrid, idx=None, A=None, b=None, A_grid=None, b_grid=None) -> Convex:
    """Linear inequalities A(x) y[idx] + b(x) > 0 (reference: src/convex_linear.jl:78-223)."""
    x = mg.geometry.xflat
    n = x.shape[0]
    if A_grid is None:
        if A is None:
            if idx is None:
                raise ValueError("a UniformScaling A with idx = Colon() cannot determine the constraint size")
            m = len(idx)
            A_grid = np.tile(np.eye(m).reshape(1, -1), (n, 1))
        else:
            A_grid = _grid(lambda xi: np.asarray(A(xi), dtype=np.float64), x)
    A_grid = np.asarray(A_grid, dtype=np.float64).reshape(n, -1)
    if b_grid is None:
        if b is None:
            if idx is None:
                raise ValueError("cannot determine the constraint count for the default b with idx = Colon()")
            b_grid = np.zeros((n, A_grid.shape[1] // len(idx)))
        else:
            sample = b(x[0])
            if np.isscalar(sample):
                if idx is None:
                    nc = np.asarray(A(x[0])).shape[0]
                else:
                    nc = A_grid.shape[1] // len(idx)
                b_grid = np.stack([np.full(nc, float(b(xi))) for xi in x], axis=0)
            else:
                b_grid = _grid(lambda xi: np.asarray(b(xi), dtype=np.float64), x)
    b_grid = np.asarray(b_grid, dtype=np.float64).reshape(n, -1)
    nca, ncb = A_grid.shape[1], b_grid.shape[1]
    if nca % ncb != 0:
        raise ValueError(f"A_grid has {nca} columns per node, not a multiple of the {ncb} constraint row(s)")
    ni = nca // ncb
    if idx is not None and len(idx) != ni:
        raise ValueError(f"A_grid has {nca} columns per node but b_grid implies nc = {ncb} constraint(s) on ni = {len(idx)} indexed components")
    ridx, colon = _resolve_idx(idx, ni)
    return Convex([Piece(KIND_LINEAR, ridx, A_grid, b_grid, None, None, colon)])


def convex_piecewise(mg: MultiGrid, Q: Sequence[Convex], select: Optional[Callable] = None,
                     select_grid: Optional[np.ndarray] = None) -> Convex:
    """Spatially selected sum of pieces (reference: src/convex_piecewise.jl:114-182).
    Nested piecewise sets are flattened: a nested piece is active where both its
    own and the outer select are non-zero."""
    x = mg.geometry.xflat
    n = x.shape[0]
    if select_grid is None:
        if select is None:
            select_grid = np.ones((n, len(Q)))
        else:
            select_grid = np.stack([np.asarray(select(xi), dtype=np.float64) for xi in x], axis=0)
    select_grid = np.asarray(select_grid, dtype=np.float64).reshape(n, len(Q))
    pieces, cols = [], []
    for k, q in enumerate(Q):
        for j, pc in enumerate(q.pieces):
            pieces.append(pc)
            inner = np.ones(n) if q.select is None else (q.select[:, j] != 0).astype(np.float64)
            cols.append((select_grid[:, k] != 0) * inner)
    sel = np.stack(cols, axis=1)
    return Convex(pieces, None if np.all(sel != 0) else sel)


def intersect(mg: MultiGrid, *Q: Convex) -> Convex:
    """reference: src/convex.jl:116-122."""
    return convex_piecewise(mg, Q)
