"""Pluggable prolongator factories for the auxiliary corner problem.

The reference delegates this step to the third-party AlgebraicMultigrid.jl
(`ruge_stuben(K; max_coarse=2).levels[i].P`, reference: src/amg_prolongators.jl:16-18),
which is not vendored and whose exact P entries are therefore *unpinned* (SURVEY.md
section 8c: the reference's own tests only pin the downstream `z`, which three different
prolongators reproduce).  This module restates that package's published algorithm
(Ruge & Stueben 1987 as implemented by AlgebraicMultigrid.jl 1.x, a port of PyAMG's
`ruge_stuben_solver`):

  * `Classical(0.25)` strength: j strongly influences i iff
    |a_ij| >= 0.25 * max_{k != i} |a_ik|  (absolute values, explicit zeros dropped);
  * `RS()` splitting: first pass only, bucket-sorted by lambda with the package's
    tie-breaking (see `_rs_cf_splitting`), lambda = 0 nodes are F points;
  * direct interpolation with separate negative/positive coefficient sums;
  * `while length(levels) + 1 < max_levels && size(A, 1) > max_coarse` with
    `max_levels = 10`, Galerkin `A <- P' A P`.

Setup-only; never on the per-iteration path.
"""
from __future__ import annotations

from typing import Callable, List

import numpy as np
import scipy.sparse as sp

from . import _setup_native


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
    S.strong_mask = keep           # the same information per stored entry of A (in A's storage order): _direct_interpolation reads it
    return S


U_NODE, C_NODE, F_NODE = 2, 1, 0


def _rs_cf_splitting(S: sp.csr_matrix, diag_quirk: bool = False) -> np.ndarray:
    """Ruge-Stueben first-pass C/F splitting with the bucket ("interval") ordering of
    AlgebraicMultigrid.jl's `RS_CF_splitting` (src/splitting.jl; itself the algorithm of
    PyAMG's `rs_cf_splitting`).  `S[i, j] != 0` iff i strongly depends on j (no diagonal).

    Nodes are kept sorted by lambda (number of points that strongly depend on the node):
    `index_to_node[interval_ptr[l] : interval_ptr[l] + interval_count[l]]` holds the nodes of
    measure l; the initial order inside an interval is by node index.  The node at the top
    (largest lambda, last in its interval) is taken as a C point; everything that depends on
    it becomes F, every still-undecided influence of a new F point moves to the END of the next
    interval (lambda + 1), every undecided influence of the new C point to the BEGINNING of the
    previous interval (lambda - 1).  Nodes nobody depends on are F from the start and never
    become C.  There is no second pass.  Returns a bool array, True = C point.

    `diag_quirk`: PyAMG's pre-filter also marks F the nodes with `lambda == 1` whose first
    stored strength entry is the diagonal; whether the Julia port (whose strength matrix keeps
    the diagonal, but whose lambda excludes it) applies the same test is not recoverable from
    the reference tree, so it is an option, off by default (on the BASELINE meshes it moves at most two
    unknowns between the two coarsest levels)."""
    n = S.shape[0]
    S = sp.csr_matrix(S)
    S.sort_indices()
    ST = sp.csr_matrix(S.T)
    ST.sort_indices()
    if _setup_native.available() and S.nnz < 2**31 - 1:   # the same loop in C++ (csrc/setup_host.cpp); tests compare the two
        return _setup_native.rs_cf_splitting(S.indptr, S.indices, ST.indptr, ST.indices, diag_quirk)
    return _rs_cf_splitting_loop(S, ST, diag_quirk)


def _rs_cf_splitting_loop(S: sp.csr_matrix, ST: sp.csr_matrix, diag_quirk: bool) -> np.ndarray:
    """The splitting loop itself in Python (S, S' with sorted rows): the readable statement and the test twin of the C++ loop."""
    n = S.shape[0]
    Sp, Sj = S.indptr.tolist(), S.indices.tolist()        # S row i: the nodes i depends on
    Tp, Tj = ST.indptr.tolist(), ST.indices.tolist()      # T row i: the nodes that depend on i
    lam = [Tp[i + 1] - Tp[i] for i in range(n)]
    interval_count = [0] * (n + 2)
    for i in range(n):
        interval_count[lam[i]] += 1
    interval_ptr = [0] * (n + 2)
    cs = 0
    for l in range(n + 1):
        interval_ptr[l] = cs
        cs += interval_count[l]
        interval_count[l] = 0
    index_to_node = [0] * n
    node_to_index = [0] * n
    for i in range(n):
        l = lam[i]
        idx = interval_ptr[l] + interval_count[l]
        index_to_node[idx] = i
        node_to_index[i] = idx
        interval_count[l] += 1
    split = [U_NODE] * n
    for i in range(n):
        if lam[i] == 0:
            split[i] = F_NODE
        elif diag_quirk and lam[i] == 1 and (Sp[i] == Sp[i + 1] or Sj[Sp[i]] > i):
            split[i] = F_NODE
    for top in range(n - 1, -1, -1):
        i = index_to_node[top]
        interval_count[lam[i]] -= 1
        if split[i] == F_NODE:
            continue
        split[i] = C_NODE
        for jj in range(Tp[i], Tp[i + 1]):              # nodes that depend on the new C point
            j = Tj[jj]
            if split[j] != U_NODE:
                continue
            split[j] = F_NODE
            for kk in range(Sp[j], Sp[j + 1]):          # what the new F point depends on
                k = Sj[kk]
                if split[k] != U_NODE or lam[k] >= n - 1:
                    continue
                lk = lam[k]
                old = node_to_index[k]
                new = interval_ptr[lk] + interval_count[lk] - 1      # end of its interval
                a, b = index_to_node[old], index_to_node[new]
                node_to_index[a], node_to_index[b] = new, old
                index_to_node[old], index_to_node[new] = b, a
                interval_count[lk] -= 1
                interval_count[lk + 1] += 1
                interval_ptr[lk + 1] = new
                lam[k] = lk + 1
        for jj in range(Sp[i], Sp[i + 1]):              # what the new C point depends on
            j = Sj[jj]
            if split[j] != U_NODE or lam[j] == 0:
                continue
            lj = lam[j]
            old = node_to_index[j]
            new = interval_ptr[lj]                       # beginning of its interval
            a, b = index_to_node[old], index_to_node[new]
            node_to_index[a], node_to_index[b] = new, old
            index_to_node[old], index_to_node[new] = b, a
            interval_count[lj] -= 1
            interval_count[lj - 1] += 1
            interval_ptr[lj] += 1
            interval_ptr[lj - 1] = interval_ptr[lj] - interval_count[lj - 1]
            lam[j] = lj - 1
    return np.array(split) == C_NODE


def _direct_interpolation_loop(A: sp.csr_matrix, S: sp.csr_matrix, is_c: np.ndarray) -> sp.csr_matrix:
    """Row-by-row statement of the direct interpolation (kept as the readable reference and the test twin of the
    vectorised `_direct_interpolation` below: tests/test_setup.py compares the two bit for bit)."""
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


def _direct_interpolation(A: sp.csr_matrix, S: sp.csr_matrix, is_c: np.ndarray) -> sp.csr_matrix:
    """Direct interpolation, vectorised over the entries of A.  Every per-row sum is a `np.bincount` over the
    entries in storage order, i.e. the same sequence of additions as the row loop above: identical bits."""
    n = A.shape[0]
    A = sp.csr_matrix(A)
    is_c = np.asarray(is_c, dtype=bool)
    cidx = np.cumsum(is_c) - 1
    nc = int(is_c.sum())
    Ap, Aj, Ax = A.indptr, A.indices.astype(np.int64), A.data
    row = np.repeat(np.arange(n, dtype=np.int64), np.diff(Ap))
    mask = getattr(S, "strong_mask", None)
    if mask is not None and mask.shape == Ax.shape:
        in_S = mask                # S was cut out of this very A by _classical_strength: its mask over A's entries
    else:
        Sb = sp.csr_matrix(S).astype(bool).tocoo()
        in_S = np.isin(row * n + Aj, Sb.row.astype(np.int64) * n + Sb.col.astype(np.int64))
    offd = Aj != row
    neg = Ax < 0
    strong_c = offd & is_c[Aj] & in_S
    bc = lambda mask: np.bincount(row[mask], weights=Ax[mask], minlength=n)
    diag = bc(~offd)
    sum_all_neg, sum_all_pos = bc(offd & neg), bc(offd & ~neg)
    sum_s_neg, sum_s_pos = bc(strong_c & neg), bc(strong_c & ~neg)
    with np.errstate(divide="ignore", invalid="ignore"):
        alpha = np.where(sum_s_neg != 0, sum_all_neg / sum_s_neg, 0.0)
        beta = np.where(sum_s_pos == 0, 0.0, sum_all_pos / sum_s_pos)
        diag = np.where(sum_s_pos == 0, diag + sum_all_pos, diag)
        neg_c, pos_c = -alpha / diag, -beta / diag
    keep = strong_c & ~is_c[row] & (diag[row] != 0)
    f_vals = np.where(neg[keep], neg_c[row[keep]], pos_c[row[keep]]) * Ax[keep]
    c_rows = np.nonzero(is_c)[0]
    rows = np.concatenate([c_rows, row[keep]])
    cols = np.concatenate([cidx[c_rows], cidx[Aj[keep]]])
    vals = np.concatenate([np.ones(c_rows.size), f_vals])
    return sp.csr_matrix((vals, (rows, cols)), shape=(n, nc))


def ruge_stuben_prolongations(K: sp.spmatrix, max_coarse: int = 2, max_levels: int = 10,
                              theta: float = 0.25, diag_quirk: bool = False) -> List[sp.csr_matrix]:
    """Level prolongations, finest first (the `[lvl.P for lvl in ...levels]` of the reference)."""
    A = sp.csr_matrix(K).astype(np.float64)
    Ps: List[sp.csr_matrix] = []
    while len(Ps) + 1 < max_levels and A.shape[0] > max_coarse:
        S = _classical_strength(A, theta)
        is_c = _rs_cf_splitting(S, diag_quirk)
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
