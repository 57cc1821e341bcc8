"""Chebyshev spectral discretizations (dense operators, dense polynomial hierarchy).

Setup-time data producer for BASELINE config 5 (reference: src/spectral1d.jl:63-109,
src/spectral2d.jl:15-42).  The Clenshaw-Curtis rule comes from the third-party
QuadratureRules.jl in the reference (`ClenshawCurtisQuadrature(T, n)`, nodes on
[0, 1], weights summing to 1); the published closed form is restated here and pinned
through the spectral golden vectors (tests/golden).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

from .multigrid import Geometry, MultiGrid, make_multigrid


@dataclass
class SPECTRAL1D:
    n: int
    dim = 1


@dataclass
class SPECTRAL2D:
    n: int
    dim = 2


def _clenshaw_curtis(n: int):
    """n-point Clenshaw-Curtis rule on [-1, 1], ascending nodes, weights summing to 2."""
    if n == 1:
        return np.zeros(1), np.array([2.0])
    N = n - 1
    i = np.arange(n)
    x = -np.cos(np.pi * i / N)
    x[0], x[-1] = -1.0, 1.0
    if N % 2 == 0:
        x[N // 2] = 0.0
    w = np.zeros(n)
    for ii in range(n):
        val = 1.0
        for j in range(1, N // 2 + 1):
            c = 1.0 if 2 * j == N else 2.0
            val += c / (1 - 4.0 * j * j) * np.cos(np.pi * (2 * j * ii) / N)
        w[ii] = val / N if (ii == 0 or ii == N) else 2 * val / N
    return x, w


def _chebyshev_values(x: float, n: int) -> np.ndarray:
    v = np.empty(n)
    v[0] = 1.0
    if n >= 2:
        v[1] = x
        for j in range(2, n):
            v[j] = 2 * x * v[j - 1] - v[j - 2]
    return v


def evaluation(xs: np.ndarray, n: int) -> np.ndarray:
    """reference: src/spectral1d.jl:55-61."""
    return np.stack([_chebyshev_values(float(x), n) for x in np.asarray(xs).reshape(-1)], axis=0)


def derivative(n: int) -> np.ndarray:
    """Chebyshev coefficient-space derivative (reference: src/spectral1d.jl:44-53)."""
    D = np.zeros((n, n))
    for j in range(n - 1):
        for k in range(j + 1, n, 2):
            D[j, k] = 2 * k
    D[0, :] /= 2
    return D


def _spectral1d_mg(n: int) -> MultiGrid:
    """reference: src/spectral1d.jl:63-109."""
    L = int(np.ceil(np.log2(n))) if n > 1 else 0
    L = max(L, 1)
    ls = [min(n, 2 ** k) for k in range(1, L + 1)]
    xs, dirichlet, full, uniform = [], [], [], []
    w = M = None
    for l in range(L):
        nodes, weights = _clenshaw_curtis(ls[l])
        w = weights.copy()
        x = nodes.reshape(-1, 1)
        M = evaluation(x, ls[l])
        CI = M[:, 2:].copy()
        for kk in range(0, CI.shape[1], 2):
            CI[:, kk] -= M[:, 0]
        for kk in range(1, CI.shape[1], 2):
            CI[:, kk] -= M[:, 1]
        xs.append(x)
        dirichlet.append(CI)
        full.append(M)
        uniform.append(np.ones((x.shape[0], 1)))
    D0 = derivative(ls[-1])
    dx = np.linalg.solve(M.T, (M @ D0).T).T          # M * D0 / M
    idm = np.eye(ls[-1])
    refine = [None] * L
    refine[L - 1] = idm
    for l in range(L - 1):
        E = evaluation(xs[l + 1], ls[l])
        refine[l] = np.linalg.solve(full[l].T, E.T).T  # E / full[l]
    subspaces = {"dirichlet": dirichlet, "full": full, "uniform": uniform}
    ops = {"id": idm, "dx": dx}
    x_fine = xs[-1].reshape(-1, 1, 1)
    t = np.arange(x_fine.shape[0]).reshape(-1, 1)
    geom = Geometry(SPECTRAL1D(n), t, x_fine, w, ops)
    return make_multigrid(geom, subspaces, refine)


def spectral1d(n: int = 16) -> Geometry:
    return _spectral1d_mg(n).geometry


def _spectral2d_mg(n: int) -> MultiGrid:
    """reference: src/spectral2d.jl:15-42 (Kronecker lift of the 1D hierarchy)."""
    M = _spectral1d_mg(n)
    w = M.geometry.w
    w2 = np.outer(w, w).reshape(-1, order="F")
    R = {X: [np.kron(Rl, Rl) for Rl in M.R[X]] for X in M.R}
    xl = M.geometry.xflat[:, 0]
    N1 = xl.size
    y = np.tile(xl, N1)            # x varies fastest
    z = np.repeat(xl, N1)
    x = np.stack([y, z], axis=1)
    ID, DX = M.geometry.operators["id"], M.geometry.operators["dx"]
    ops = {"id": np.kron(ID, ID), "dx": np.kron(DX, ID), "dy": np.kron(ID, DX)}
    x_fine = x.reshape(N1 * N1, 1, 2)
    t = np.arange(N1 * N1).reshape(-1, 1)
    geom = Geometry(SPECTRAL2D(n), t, x_fine, w2, ops)
    return MultiGrid(geom, R)


def spectral2d(n: int = 4) -> Geometry:
    return _spectral2d_mg(n).geometry


def amg(geom: Geometry) -> MultiGrid:
    disc = geom.discretization
    return _spectral1d_mg(disc.n) if isinstance(disc, SPECTRAL1D) else _spectral2d_mg(disc.n)
