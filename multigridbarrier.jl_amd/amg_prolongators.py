"""Pluggable prolongator factories for the auxiliary corner problem.

The reference delegates this step to the third-party AlgebraicMultigrid.jl
(`ruge_stuben(K; max_coarse=2).levels[i].P`, reference: src/amg_prolongators.jl:16-18),
which is not vendored and whose exact P entries are therefore *unpinned* (SURVEY.md
section 8c: the reference's own tests only pin the downstream `z`, which three different
prolongators reproduce).  This module restates the published classical
Ruge-Stueben algorithm (Ruge & Stueben 1987; the PyAMG/AlgebraicMultigrid.jl
variant: absolute-value classical strength with theta = 0.25, first-pass RS C/F
splitting, direct interpolation).  Setup-only; never on the per-iteration path.
"""
from __future__ import annotations

import heapq
from typing import Callable, List

import numpy as np
import scipy.sparse as sp


def _classical_strength(A: sp.csr_matrix, theta: float) -> sp.csr_matrix:
    """S[i, j] != 0 iff j strongly influences i: |a_ij| >= theta * max_{k != i} |a_ik|."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    indptr, indices, data = A.indptr, A.indices, A.data
    rows = np.repeat(np.arange(n), np.diff(indptr))
    offdiag = rows != indices
    absd = np.abs(data) * offdiag
    rowmax = np.zeros(n)
    np.maximum.at(rowmax, rows, absd)
    keep = offdiag & (absd >= theta * rowmax[rows]) & (absd > 0)
    S = sp.csr_matrix((np.ones(keep.sum()), (rows[keep], indices[keep])), shape=(n, n))
    return S


def _rs_cf_splitting(S: sp.csr_matrix) -> np.ndarray:
    """First-pass Ruge-Stueben C/F splitting.  Returns a bool array, True = C point."""
    n = S.shape[0]
    S = sp.csr_matrix(S)
    ST = sp.csr_matrix(S.T)
    Sp, Sj = S.indptr, S.indices
    Tp, Tj = ST.indptr, ST.indices
    lam = np.diff(Tp).astype(np.int64)        # number of points each node strongly influences
    U, C, F = 0, 1, 2
    state = np.zeros(n, dtype=np.int8)
    # isolated / uninfluential points: nodes with no strong connections at all become F
    # only if they influence nobody and depend on nobody -> make them C so P keeps them.
    heap = [(-int(lam[i]), i) for i in range(n)]
    heapq.heapify(heap)
    lam_l = lam.tolist()
    while heap:
        negl, i = heapq.heappop(heap)
        if state[i] != U or -negl != lam_l[i]:
            continue
        state[i] = C
        # everything that strongly depends on i becomes F
        for jj in range(Tp[i], Tp[i + 1]):
            j = Tj[jj]
            if state[j] != U:
                continue
            state[j] = F
            # points that influence the new F point become more attractive as C points
            for kk in range(Sp[j], Sp[j + 1]):
                k = Sj[kk]
                if state[k] == U:
                    lam_l[k] += 1
                    heapq.heappush(heap, (-lam_l[k], k))
        # points that i depends on lose one potential dependent
        for jj in range(Sp[i], Sp[i + 1]):
            j = Sj[jj]
            if state[j] == U:
                lam_l[j] -= 1
                heapq.heappush(heap, (-lam_l[j], j))
    return state == C


def _direct_interpolation(A: sp.csr_matrix, S: sp.csr_matrix, is_c: np.ndarray) -> sp.csr_matrix:
    n = A.shape[0]
    A = sp.csr_matrix(A)
    cidx = np.cumsum(is_c) - 1
    nc = int(is_c.sum())
    Ap, Aj, Ax = A.indptr, A.indices, A.data
    strong = sp.csr_matrix(S).astype(bool).tolil().rows
    rows, cols, vals = [], [], []
    for i in range(n):
        if is_c[i]:
            rows.append(i); cols.append(int(cidx[i])); vals.append(1.0)
            continue
        sset = set(strong[i])
        diag = 0.0
        sum_all_pos = sum_all_neg = sum_s_pos = sum_s_neg = 0.0
        for jj in range(Ap[i], Ap[i + 1]):
            j, v = Aj[jj], Ax[jj]
            if j == i:
                diag += v
                continue
            if v < 0:
                sum_all_neg += v
            else:
                sum_all_pos += v
            if is_c[j] and j in sset:
                if v < 0:
                    sum_s_neg += v
                else:
                    sum_s_pos += v
        alpha = sum_all_neg / sum_s_neg if sum_s_neg != 0 else 0.0
        if sum_s_pos == 0:
            diag += sum_all_pos
            beta = 0.0
        else:
            beta = sum_all_pos / sum_s_pos
        if diag == 0:
            continue
        neg_c, pos_c = -alpha / diag, -beta / diag
        for jj in range(Ap[i], Ap[i + 1]):
            j, v = Aj[jj], Ax[jj]
            if j != i and is_c[j] and j in sset:
                rows.append(i); cols.append(int(cidx[j])); vals.append((neg_c if v < 0 else pos_c) * v)
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, nc))


def ruge_stuben_prolongations(K: sp.spmatrix, max_coarse: int = 2, max_levels: int = 10,
                              theta: float = 0.25) -> List[sp.csr_matrix]:
    """Level prolongations, finest first (the `[lvl.P for lvl in ...levels]` of the reference)."""
    A = sp.csr_matrix(K).astype(np.float64)
    Ps: List[sp.csr_matrix] = []
    while len(Ps) + 1 < max_levels and A.shape[0] > max_coarse:
        S = _classical_strength(A, theta)
        is_c = _rs_cf_splitting(S)
        nc = int(is_c.sum())
        if nc == 0 or nc == A.shape[0]:
            break
        P = _direct_interpolation(A, S, is_c)
        Ps.append(P)
        A = sp.csr_matrix(P.T @ A @ P)
    return Ps


def amg_ruge_stuben(**kwargs) -> Callable[[sp.spmatrix], List[sp.csr_matrix]]:
    """Factory with the reference's calling convention (reference: src/amg_prolongators.jl:16-18)."""
    return lambda K: ruge_stuben_prolongations(K, **kwargs)


# ---------------------------------------------------------------------------
# smoothed aggregation (reference: src/amg_prolongators.jl:27-29 ->
# AlgebraicMultigrid.smoothed_aggregation; published algorithm: Vanek, Mandel, Brezina 1996,
# the defaults of AlgebraicMultigrid.jl / PyAMG: symmetric strength theta = 0, standard
# aggregation, constant near-nullspace candidate, Jacobi prolongation smoothing omega = 4/3)
# ---------------------------------------------------------------------------

def _symmetric_strength(A: sp.csr_matrix, theta: float) -> sp.csr_matrix:
    """C[i, j] != 0 iff |a_ij| >= theta * sqrt(|a_ii| |a_jj|), i != j."""
    A = sp.csr_matrix(A)
    n = A.shape[0]
    d = np.abs(A.diagonal())
    rows = np.repeat(np.arange(n), np.diff(A.indptr))
    off = rows != A.indices
    keep = off & (np.abs(A.data) >= theta * np.sqrt(d[rows] * d[A.indices])) & (A.data != 0)
    return sp.csr_matrix((np.abs(A.data[keep]), (rows[keep], A.indices[keep])), shape=(n, n))


def _standard_aggregation(C: sp.csr_matrix) -> np.ndarray:
    """Three-pass standard aggregation.  Returns the aggregate id per node (-1 = isolated)."""
    n = C.shape[0]
    Cp, Cj, Cx = C.indptr, C.indices, C.data
    agg = np.full(n, -1, dtype=np.int64)
    na = 0
    # pass 1: a node whose strong neighbourhood is entirely free seeds an aggregate
    for i in range(n):
        if agg[i] >= 0 or Cp[i] == Cp[i + 1]:
            continue
        nb = Cj[Cp[i]:Cp[i + 1]]
        if np.all(agg[nb] < 0):
            agg[i] = na
            agg[nb] = na
            na += 1
    # pass 2: remaining nodes join the aggregate of their strongest aggregated neighbour
    agg1 = agg.copy()
    for i in range(n):
        if agg[i] >= 0 or Cp[i] == Cp[i + 1]:
            continue
        nb = Cj[Cp[i]:Cp[i + 1]]
        w = Cx[Cp[i]:Cp[i + 1]]
        ok = agg1[nb] >= 0
        if ok.any():
            agg[i] = agg1[nb[ok][np.argmax(w[ok])]]
    # pass 3: whatever is left forms aggregates with its free neighbours
    for i in range(n):
        if agg[i] >= 0 or Cp[i] == Cp[i + 1]:
            continue
        nb = Cj[Cp[i]:Cp[i + 1]]
        free = nb[agg[nb] < 0]
        agg[i] = na
        agg[free] = na
        na += 1
    return agg


def _spectral_radius_DinvA(A: sp.csr_matrix, iters: int = 15) -> float:
    """Power iteration on D^{-1} A (symmetrically scaled), deterministic start vector."""
    d = A.diagonal()
    dis = 1.0 / np.sqrt(np.where(d != 0, np.abs(d), 1.0))
    x = np.cos(np.arange(A.shape[0]) * 0.7 + 0.3)
    x /= np.linalg.norm(x)
    lam = 1.0
    for _ in range(iters):
        y = dis * (A @ (dis * x))
        lam = float(np.linalg.norm(y))
        if lam == 0:
            return 1.0
        x = y / lam
    return lam


def smoothed_aggregation_prolongations(K: sp.spmatrix, max_coarse: int = 10, max_levels: int = 10,
                                       theta: float = 0.0, omega: float = 4.0 / 3.0) -> List[sp.csr_matrix]:
    """Level prolongations, finest first."""
    A = sp.csr_matrix(K).astype(np.float64)
    Ps: List[sp.csr_matrix] = []
    while len(Ps) + 1 < max_levels and A.shape[0] > max_coarse:
        n = A.shape[0]
        C = _symmetric_strength(A, theta)
        agg = _standard_aggregation(C)
        na = int(agg.max()) + 1 if n else 0
        if na == 0 or na >= n:
            break
        keep = agg >= 0
        counts = np.bincount(agg[keep], minlength=na).astype(np.float64)
        T = sp.csr_matrix((1.0 / np.sqrt(counts[agg[keep]]), (np.nonzero(keep)[0], agg[keep])), shape=(n, na))
        d = A.diagonal()
        dinv = np.where(d != 0, 1.0 / d, 0.0)
        rho = _spectral_radius_DinvA(A)
        P = sp.csr_matrix(T - (omega / rho) * (sp.diags(dinv) @ (A @ T)))
        P.eliminate_zeros()
        Ps.append(P)
        A = sp.csr_matrix(P.T @ A @ P)
    return Ps


def amg_smoothed_aggregation(**kwargs) -> Callable[[sp.spmatrix], List[sp.csr_matrix]]:
    """Factory with the reference's calling convention (reference: src/amg_prolongators.jl:27-29)."""
    return lambda K: smoothed_aggregation_prolongations(K, **kwargs)


def amg_prolongations(K_int: sp.spmatrix, prolongator) -> List[sp.csr_matrix]:
    """reference: src/amg_prolongators.jl:70-78."""
    if K_int.shape[0] == 0:
        return []
    return [sp.csr_matrix(P) for P in prolongator(sp.csr_matrix(K_int))]
