"""CPU ORACLE -- test infrastructure, NOT part of the product path.

A plain NumPy/SciPy restatement of the reference's inner Newton hot path, used only
by `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg as the
*checker* for the HIP library.  Nothing under `multigridbarrier.jl_amd/` imports it.

Pinning: the reference is pure Julia and Julia is not available in the build
container, so the oracle is pinned by the reference's own golden
vectors transcribed into `tests/golden/` (test/runtests.jl:13-32,
test/test_algebraic.jl:38-69, test/test_feasibility.jl) -- see tests/test_oracle_golden.py.

Each function cites the reference lines it follows (paths relative to /root/reference).
Inputs come from the package's setup layer (`assemble`), which is itself CPU/NumPy
as in the reference.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass
from typing import Any, Callable, List, Optional

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

EPS = np.finfo(np.float64).eps
KIND_EP, KIND_LINEAR = 1, 2


class MGBConvergenceFailure(Exception):
    """reference: src/utils.jl:178-184."""

    def __init__(self, message, code="failure"):
        super().__init__(message)
        self.message = message
        self.code = code


# ---------------------------------------------------------------------------
# per-node functors, vectorised over the n mesh nodes
# ---------------------------------------------------------------------------

def Log(x):
    """src/utils.jl:14 -- the convex programmer's log: -Inf off the domain."""
    x = np.asarray(x, dtype=np.float64)
    out = np.full(x.shape, -np.inf)
    m = x > 0
    out[m] = np.log(x[m])
    return out


def _safe_pow(s, a):
    """src/convex_linear.jl:388-390."""
    with np.errstate(invalid="ignore"):
        return np.exp(a * Log(s))


def _ep_parts(pc, y, slack=None):
    """[q; s] = A y[idx] + b  (src/convex_euclidian_power.jl:18-63)."""
    n = y.shape[0]
    nz = pc.ni
    # pc.A rows are column-major flattened nz x nz matrices: A[i][r, c] = flat[r + nz*c]
    A = pc.A.reshape(n, nz, nz).transpose(0, 2, 1)
    z = np.einsum("nrc,nc->nr", A, y[:, list(pc.idx)]) + pc.b
    q = z[:, : nz - 1]
    s = z[:, nz - 1].copy()
    if slack is not None:
        s = s + slack
    return A, q, s


def _ep_core(q, s, p0, mu, order):
    """value / core_grad / core_hess in (q, s) (src/convex_euclidian_power.jl:79-90, :387-433)."""
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        alpha = 2.0 / p0
        qsq = np.sum(q * q, axis=1)
        s_a = _safe_pow(s, alpha)
        r = s_a - qsq
        if order == 0:
            return -Log(r) - mu * Log(s)
        inv_r = 1.0 / r
        s_am1 = _safe_pow(s, alpha - 1.0)
        if order == 1:
            gq = (2.0 * inv_r)[:, None] * q
            gs = -alpha * s_am1 * inv_r - mu / s
            return np.concatenate([gq, gs[:, None]], axis=1)
        inv_r2 = inv_r * inv_r
        coef_qs = -2.0 * alpha * s_am1 * inv_r2
        s_am2 = _safe_pow(s, alpha - 2.0)
        s_2am2 = _safe_pow(s, 2.0 * alpha - 2.0)
        H_ss = -alpha * (alpha - 1.0) * s_am2 * inv_r + alpha * alpha * s_2am2 * inv_r2 + mu / (s * s)
        n, nq = q.shape
        H = np.zeros((n, nq + 1, nq + 1))
        H[:, :nq, :nq] = 4.0 * q[:, :, None] * q[:, None, :] * inv_r2[:, None, None]
        for i in range(nq):
            H[:, i, i] += 2.0 * inv_r
        H[:, :nq, nq] = coef_qs[:, None] * q
        H[:, nq, :nq] = coef_qs[:, None] * q
        H[:, nq, nq] = H_ss
        return H


def _piece_eval(pc, y, order, co):
    """One piece's barrier (co=False) or cobarrier (co=True: last entry of y is the slack).
    Returns F (n,), G (n, NY) or H (n, NY, NY) in the y (or yhat) layout."""
    n, NY = y.shape
    slack = y[:, NY - 1] if co else None
    idx = list(pc.idx)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if pc.kind == KIND_EP:
            A, q, s = _ep_parts(pc, y, slack)
            core = _ep_core(q, s, pc.p, pc.mu, order)
            if order == 0:
                return core
            nz = pc.ni
            if order == 1:
                gidx = np.einsum("nrc,nr->nc", A, core)               # A' grad_z
                G = np.zeros((n, NY))
                G[:, idx] = gidx
                if co:
                    G[:, NY - 1] = core[:, nz - 1]
                return G
            Hidx = np.einsum("nra,nrs,nsb->nab", A, core, A)          # A' H_z A
            H = np.zeros((n, NY, NY))
            H[np.ix_(range(n), idx, idx)] = Hidx
            if co:
                cross = np.einsum("nra,nr->na", A, core[:, :, nz - 1])
                H[:, idx, NY - 1] = cross
                H[:, NY - 1, idx] = cross
                H[:, NY - 1, NY - 1] = core[:, nz - 1, nz - 1]
            return H
        # linear inequalities (src/convex_linear.jl:119-203)
        nc, ni = pc.nc, pc.ni
        A = pc.A.reshape(n, ni, nc).transpose(0, 2, 1)                # A[i][r, c] = flat[r + nc*c]
        Fv = np.einsum("nrc,nc->nr", A, y[:, idx]) + pc.b
        if co:
            Fv = Fv + slack[:, None]
        if order == 0:
            return -np.sum(Log(Fv), axis=1)
        if order == 1:
            inv_F = 1.0 / Fv
            G = np.zeros((n, NY))
            G[:, idx] = -np.einsum("nrc,nr->nc", A, inv_F)
            if co:
                G[:, NY - 1] = -np.sum(inv_F, axis=1)
            return G
        inv_F2 = 1.0 / (Fv * Fv)
        H = np.zeros((n, NY, NY))
        H[np.ix_(range(n), idx, idx)] = np.einsum("nra,nr,nrb->nab", A, inv_F2, A)
        if co:
            cross = np.einsum("nra,nr->na", A, inv_F2)
            H[:, idx, NY - 1] = cross
            H[:, NY - 1, idx] = cross
            H[:, NY - 1, NY - 1] = np.sum(inv_F2, axis=1)
        return H


def _piece_slack(pc, y):
    """src/convex_euclidian_power.jl:243-253, src/convex_linear.jl:205-214."""
    n = y.shape[0]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if pc.kind == KIND_EP:
            _, q, s = _ep_parts(pc, y)
            qsq = np.sum(q * q, axis=1)
            return -np.minimum(s - _safe_pow(qsq, pc.p / 2.0), s)
        nc, ni = pc.nc, pc.ni
        A = pc.A.reshape(n, ni, nc).transpose(0, 2, 1)
        Fv = np.einsum("nrc,nc->nr", A, y[:, list(pc.idx)]) + pc.b
        return -np.min(Fv, axis=1)


def convex_eval(Q, y, order, co=False):
    """Sum over active pieces (src/convex_piecewise.jl:15-60; a single-piece Convex is
    the piece itself).  Inactive pieces contribute an exact zero, never 0*Inf."""
    out = None
    for k, pc in enumerate(Q.pieces):
        val = _piece_eval(pc, y, order, co)
        if Q.select is not None:
            act = Q.select[:, k] != 0
            val = np.where(act.reshape((-1,) + (1,) * (val.ndim - 1)), val, 0.0)
        out = val if out is None else out + val
    return out


def convex_slack(Q, y):
    """src/convex_piecewise.jl:62-75."""
    out = None
    for k, pc in enumerate(Q.pieces):
        val = _piece_slack(pc, y)
        if Q.select is not None:
            val = np.where(Q.select[:, k] != 0, val, -np.inf)
        out = val if out is None else np.maximum(out, val)
    return out


@dataclass
class FeasConvex:
    """Phase-I barrier (src/mgb.jl:217-287): cobarrier(yy[:NC]) - log(b-u) - log(b+u)
    - sum_i [log(R-v_i) + log(R+v_i)]."""

    Q: Any
    b: float
    R: float
    NC: int


def node_eval(Q, y, order):
    """Barrier value / gradient / Hessian per node for a Convex or a FeasConvex."""
    if not isinstance(Q, FeasConvex):
        return convex_eval(Q, y, order)
    n, NF = y.shape
    NC, bb, RR = Q.NC, Q.b, Q.R
    yc = y[:, :NC]
    u = yc[:, NC - 1]
    v = y[:, NC:]
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        if order == 0:
            ret = convex_eval(Q.Q, yc, 0, co=True) - Log(bb - u) - Log(bb + u)
            return ret + np.sum(-Log(RR - v) - Log(RR + v), axis=1)
        if order == 1:
            G = np.zeros((n, NF))
            G[:, :NC] = convex_eval(Q.Q, yc, 1, co=True)
            G[:, NC - 1] += 1.0 / (bb - u) - 1.0 / (bb + u)
            G[:, NC:] = 1.0 / (RR - v) - 1.0 / (RR + v)
            return G
        H = np.zeros((n, NF, NF))
        H[:, :NC, :NC] = convex_eval(Q.Q, yc, 2, co=True)
        H[:, NC - 1, NC - 1] += 1.0 / (bb - u) ** 2 + 1.0 / (bb + u) ** 2
        for i in range(NC, NF):
            H[:, i, i] = 1.0 / (RR - y[:, i]) ** 2 + 1.0 / (RR + y[:, i]) ** 2
        return H


# ---------------------------------------------------------------------------
# barrier functional f0 / f1 / f2 (src/convex.jl:147-257)
# ---------------------------------------------------------------------------

def _as_mat(op):
    if hasattr(op, "to_sparse"):
        return op.to_sparse()
    return op


class OracleAMG:
    """Host matrices of one AMG (src/multigrid.jl:278-288) in SciPy/NumPy form."""

    def __init__(self, M):
        self.w = np.asarray(M.w, dtype=np.float64)
        self.R_fine = [sp.csr_matrix(R) if sp.issparse(R) else np.asarray(R) for R in M.R_fine]
        self.D_fine = [_as_mat(D) for D in M.D_fine]
        self.dense = not sp.issparse(self.D_fine[0])
        self.n = self.w.size
        self.nD = len(self.D_fine)


def apply_D(D, z):
    """src/convex.jl:125."""
    return np.stack([np.asarray(Dk @ z).reshape(-1) for Dk in D], axis=1)


class Barrier:
    """`barrier(Q; barrier_weights)` (src/convex.jl:147-205; masked twin :213-257)."""

    def __init__(self, Q, barrier_weights=None):
        self.Q = Q
        self.bw = barrier_weights

    def _scale(self, n, arr):
        if self.bw is None:
            return arr * (1.0 / n)
        bw = self.bw.reshape((-1,) + (1,) * (arr.ndim - 1))
        with np.errstate(invalid="ignore"):
            return np.where(bw == 0, 0.0, bw * arr)

    def f0(self, s, w, c, R, D, z0):
        Dz = apply_D(D, z0 + R @ s)
        y = node_eval(self.Q, Dz, 0)
        n = w.size
        if self.bw is None:
            bar = (1.0 / n) * np.sum(y)
        else:
            bar = np.sum(self._scale(n, y))
        return float(bar + np.sum(w * np.sum(c * Dz, axis=1)))

    def f1(self, s, w, c, R, D, z0):
        Dz = apply_D(D, z0 + R @ s)
        G = node_eval(self.Q, Dz, 1)
        y = self._scale(w.size, G) + w[:, None] * c
        ret = D[0].T @ y[:, 0]
        for k in range(1, len(D)):
            ret = ret + D[k].T @ y[:, k]
        return np.asarray(R.T @ ret).reshape(-1)

    def f2(self, s, w, c, R, D, z0):
        Dz = apply_D(D, z0 + R @ s)
        H = self._scale(w.size, node_eval(self.Q, Dz, 2))
        nD = len(D)
        dense = not sp.issparse(D[0])
        diag = (lambda v: np.diag(v)) if dense else (lambda v: sp.diags(v))
        # src/convex.jl:191-200 (the reference's y[:, (j-1)*n + k] is H[:, k, j])
        ret = D[0].T @ diag(H[:, 0, 0]) @ D[0]
        for j in range(1, nD):
            ret = ret + D[j].T @ diag(H[:, j, j]) @ D[j]
            for k in range(j):
                foo = diag(H[:, k, j])
                ret = ret + (D[j].T @ foo @ D[k] + D[k].T @ foo @ D[j])
        # R' * H_blk * R  (structured assembly in the reference: src/BlockMatrices.jl:506-555)
        out = R.T @ ret @ R
        return out if dense else sp.csc_matrix(out)


def solve_symmetric(H, g):
    """`solve(symmetric(H), g)` (src/newton.jl:253, src/utils.jl:142-145): a direct solve
    of the matrix whose upper triangle is H's."""
    if sp.issparse(H):
        U = sp.triu(H, format="csc")
        Hs = U + sp.triu(H, 1, format="csc").T
        lu = spla.splu(sp.csc_matrix(Hs))
        return lu.solve(np.asarray(g, dtype=np.float64))
    Hs = np.triu(H) + np.triu(H, 1).T
    return np.linalg.solve(Hs, g)


class MfHostSolver:
    """`solve(symmetric(H), g)` through the host multifrontal Cholesky of oracle/csrc/mf_host.cpp
    (analysis cached per sparsity pattern, like CHOLMOD's symbolic reuse).  Falls back to
    SuperLU when the matrix is not numerically SPD (the reference falls back to LDLt/LU)."""

    def __init__(self):
        import ctypes as C
        import os
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_build", "libmf_host.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle`")
        self.C = C
        self.lib = C.CDLL(path)
        self.lib.mf_host_analyze.restype = C.c_void_p
        self.lib.mf_host_analyze.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]
        self.lib.mf_host_factor_solve.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        self.lib.mf_host_free.argtypes = [C.c_void_p]
        self.cache = {}

    def __call__(self, H, g):
        if not sp.issparse(H):
            return solve_symmetric(H, g)
        A = sp.csr_matrix(H)
        A.sort_indices()
        key = (A.shape[0], A.nnz, hash(A.indptr.tobytes()), hash(A.indices.tobytes()))
        ent = self.cache.get(key)
        if ent is None:
            ip = np.ascontiguousarray(A.indptr, dtype=np.int32)
            ii = np.ascontiguousarray(A.indices, dtype=np.int32)
            h = self.lib.mf_host_analyze(A.shape[0], ip.ctypes.data, ii.ctypes.data, 0)
            if not h:
                return solve_symmetric(H, g)
            ent = (h, ip, ii)
            self.cache[key] = ent
        v = np.ascontiguousarray(A.data, dtype=np.float64)
        b = np.ascontiguousarray(g, dtype=np.float64)
        x = np.zeros_like(b)
        rc = self.lib.mf_host_factor_solve(ent[0], v.ctypes.data, b.ctypes.data, x.ctypes.data)
        if rc != 0:
            return solve_symmetric(H, g)
        return x

    def close(self):
        for h, _, _ in self.cache.values():
            self.lib.mf_host_free(h)
        self.cache = {}


_SOLVER = [solve_symmetric]


def set_solver(kind: str):
    """'splu' (SciPy SuperLU, the default) or 'mf' (host multifrontal Cholesky)."""
    _SOLVER[0] = solve_symmetric if kind == "splu" else MfHostSolver()


# ---------------------------------------------------------------------------
# Newton, line searches, stopping rules (src/newton.jl)
# ---------------------------------------------------------------------------

def illinois(f, a, b, fa=None, fb=None, maxit=10000):
    """src/newton.jl:4-27."""
    fa = f(a) if fa is None else fa
    fb = f(b) if fb is None else fb
    assert math.isfinite(fa) and math.isfinite(fb)
    if fa == 0:
        return a
    if fa * fb >= 0:
        return b
    for _ in range(maxit):
        c = (a * fb - b * fa) / (fb - fa)
        fc = f(c)
        assert math.isfinite(fc)
        if c <= min(a, b) or c >= max(a, b) or fc * fa == 0 or fc * fb == 0:
            return c
        if fb * fc < 0:
            a, fa = b, fb
        else:
            fa /= 2
        b, fb = c, fc
    raise RuntimeError("Illinois solver failed to converge.")


def _linesearch_loop(attempt, x, y, g, beta):
    """src/newton.jl:35-50."""
    s = 1.0
    xn, yn, gn = x, y, g
    while s > 0.0:
        try:
            xn, yn, gn, done = attempt(s)
            if done:
                break
        except KeyboardInterrupt:
            raise
        except Exception:
            pass
        s = s * beta
    return xn, yn, gn


def linesearch_backtracking(beta=0.5, c1=0.1):
    """src/newton.jl:139-154."""

    def ls(x, y, g, n, F0, F1):
        inc = float(np.dot(g, n))

        def attempt(s):
            xn = x - s * n
            stalled = np.linalg.norm(xn - x) == 0
            yn, gn = F0(xn), F1(xn)
            if not (math.isfinite(yn) and np.all(np.isfinite(gn))):
                raise FloatingPointError("line search: non-finite step")
            return xn, yn, gn, bool(stalled or yn <= y - c1 * inc * s)

        return _linesearch_loop(attempt, x, y, g, beta)

    return ls


def linesearch_illinois(beta=0.5):
    """src/newton.jl:84-103."""

    def ls(x, y, g, n, F0, F1):
        inc = float(np.dot(g, n))

        def attempt(s):
            def phi(sigma):
                xn = x - sigma * n
                if not math.isfinite(F0(xn)):
                    raise FloatingPointError("line search: non-finite barrier value")
                return float(np.dot(F1(xn), n))

            s2 = illinois(phi, 0.0, s, fa=inc)
            xn = x - s2 * n
            yn, gn = F0(xn), F1(xn)
            if not (math.isfinite(yn) and np.all(np.isfinite(gn))):
                raise FloatingPointError("line search: non-finite step")
            return xn, yn, gn, True

        return _linesearch_loop(attempt, x, y, g, beta)

    return ls


def stopping_exact(theta):
    """src/newton.jl:187."""
    return lambda ymin, ynext, gmin, gnext, n, ndecmin, ndec: bool(ynext >= ymin and np.linalg.norm(gnext) >= theta * gmin)


def stopping_inexact(lambda_tol, theta):
    """src/newton.jl:222-225."""
    ex = stopping_exact(theta)
    return lambda ymin, ynext, gmin, gnext, n, ndecmin, ndec: bool(ndec < lambda_tol or ex(ymin, ynext, gmin, gnext, n, ndecmin, ndec))


def newton(F0, F1, F2, x, maxit=10000, stopping_criterion=None, line_search=None, solve=None,
           stats=None):
    """src/newton.jl:227-287."""
    if stopping_criterion is None:
        stopping_criterion = stopping_exact(0.1)
    if line_search is None:
        line_search = linesearch_illinois()
    if solve is None:
        solve = _SOLVER[0]
    if not np.all(np.isfinite(x)):
        raise FloatingPointError("newton: initial point has non-finite entries")
    y = F0(x)
    if not math.isfinite(y):
        raise FloatingPointError("newton: initial objective value is not finite")
    ymin = y
    ys = [y]
    converged = False
    k = 0
    g = F1(x)
    if not np.all(np.isfinite(g)):
        raise FloatingPointError("newton: initial gradient has non-finite entries")
    gmin = float(np.linalg.norm(g))
    incmin = math.inf
    while k < maxit and not converged:
        k += 1
        H = F2(x)
        t0 = time.perf_counter()
        n = solve(H, g)
        if stats is not None:
            stats["solve_s"] = stats.get("solve_s", 0.0) + time.perf_counter() - t0
            stats["newton_its"] = stats.get("newton_its", 0) + 1
            if "y_hist" in stats:                       # objective before this iteration (test instrumentation)
                stats["y_hist"].append(y)
            if "deadline" in stats and time.perf_counter() > stats["deadline"]:
                raise TimeoutError("oracle time budget exhausted")
            if "max_its" in stats and stats["newton_its"] >= stats["max_its"]:
                raise TimeoutError("oracle iteration budget exhausted")
        if not np.all(np.isfinite(n)):
            raise FloatingPointError("newton: Newton direction has non-finite entries")
        inc = float(np.dot(g, n))
        if stats is not None and "trace" in stats:          # test instrumentation: full-length logs (tests/dev/)
            stats["trace"](k, y, inc, H, g, n)
        if inc <= 0:
            converged = abs(inc) <= EPS * max(abs(y), 1.0)
            break
        xn, yn, gn = line_search(x, y, g, n, F0, F1)
        if stopping_criterion(ymin, yn, gmin, gn, n, math.sqrt(incmin), math.sqrt(inc)):
            converged = True
        x, y, g = xn, yn, gn
        gmin = min(gmin, float(np.linalg.norm(g)))
        ymin = min(ymin, y)
        incmin = min(inc, incmin)
        ys.append(y)
    return dict(x=x, y=y, k=k, converged=converged, ys=ys)


# ---------------------------------------------------------------------------
# MGB outer loops (src/mgb.jl)
# ---------------------------------------------------------------------------

class NoFinalize:
    pass


def divide_and_conquer(eta, j, J):
    """src/mgb.jl:10-15."""
    if eta(j, J):
        return True
    jmid = (j + J) // 2
    if jmid == j or jmid == J:
        return False
    return divide_and_conquer(eta, j, jmid) and divide_and_conquer(eta, jmid, J)


def mgb_step(Q, M, z, c, maxit, max_newton, line_search, stopping_criterion, finalize,
             initial_step=False, barrier_weights=None, stats=None):
    """src/mgb.jl:16-82.  Levels are 1-based (J = 1..L) as in the reference."""
    L = len(M.R_fine)
    B = Barrier(Q, barrier_weights)
    its = np.zeros(L, dtype=np.int64)
    w, D = M.w, M.D_fine
    state = {"z": z}

    def eta(j, J, sc, mi, ls):
        R = M.R_fine[J - 1]
        s0 = np.zeros(R.shape[1])
        zJ = state["z"]
        SOL = newton(lambda s: B.f0(s, w, c, R, D, zJ),
                     lambda s: B.f1(s, w, c, R, D, zJ),
                     lambda s: B.f2(s, w, c, R, D, zJ),
                     s0, maxit=mi, stopping_criterion=sc, line_search=ls, stats=stats)
        its[J - 1] += SOL["k"]
        if SOL["converged"]:
            state["z"] = zJ + R @ SOL["x"]
        return SOL["converged"]

    mn = lambda j, J: maxit if (initial_step and J - j == 1) else max_newton
    converged = divide_and_conquer(lambda j, J: eta(j, J, stopping_criterion, mn(j, J), line_search), 0, L)
    z_unfinalized = state["z"]
    if not isinstance(finalize, NoFinalize):
        foo = eta(L - 1, L, finalize, maxit, line_search)
        converged = converged and foo
    return dict(z=state["z"], z_unfinalized=z_unfinalized, its=its, converged=converged)


def _early_stop(f, z, t):
    """src/mgb.jl:89."""
    code = getattr(f, "__code__", None)
    if code is not None and code.co_argcount >= 2:
        return f(z, t)
    return f(z)


def mgb_core(Q, M, z, c, tol=math.sqrt(EPS), t=0.1, maxit=10000, kappa=10.0, early_stop=None,
             max_newton=None, finalize=None, barrier_weights=None, line_search=None,
             stopping_criterion=None, stats=None):
    """src/mgb.jl:91-183."""
    if max_newton is None:
        max_newton = int(math.ceil(math.log2(-math.log2(EPS)) + 2))
    if early_stop is None:
        early_stop = lambda z: False
    t_begin = time.time()
    tinit = t
    target = 1.0 / tol
    kappa0 = kappa
    L = len(M.R_fine)
    its, ts, kappas, times, c_dot_Dz = [], [], [], [], []
    times.append(time.time())
    kw = dict(max_newton=max_newton, maxit=maxit, barrier_weights=barrier_weights,
              line_search=line_search, stopping_criterion=stopping_criterion, stats=stats)
    initial_finalize = finalize if t >= target else NoFinalize()
    SOL = mgb_step(Q, M, z, t * c, finalize=initial_finalize, initial_step=True, **kw)
    if not SOL["converged"]:
        raise MGBConvergenceFailure(f"Initial centering failed in mgb_solve at t={t}, tol={tol}, maxit={maxit}.", "stall")
    its.append(SOL["its"].copy())
    kappas.append(kappa)
    ts.append(t)
    z = SOL["z"]
    z_unfinalized = SOL["z_unfinalized"]

    def cdot(z):
        Dz = apply_D(M.D_fine, z)
        return float(sum(np.dot(M.w * c[:, j], Dz[:, j]) for j in range(len(M.D_fine))))

    c_dot_Dz.append(cdot(z))
    k = 1
    while t < target and kappa > 1 and k < maxit and not _early_stop(early_stop, z, t):
        k += 1
        its.append(np.zeros(L, dtype=np.int64))
        times.append(time.time())
        while kappa > 1:
            t1 = kappa * t
            fin = finalize if t1 >= target else NoFinalize()
            SOL = mgb_step(Q, M, z, t1 * c, finalize=fin, **kw)
            its[-1] += SOL["its"]
            if SOL["converged"]:
                if SOL["its"].max() <= max_newton * 0.5:
                    kappa = min(kappa0, kappa ** 2)
                z = SOL["z"]
                z_unfinalized = SOL["z_unfinalized"]
                t = t1
                break
            kappa = math.sqrt(kappa)
        ts.append(t)
        kappas.append(kappa)
        c_dot_Dz.append(cdot(z))
    converged = (t >= target) or _early_stop(early_stop, z, t)
    if not converged:
        code = "stall" if kappa <= 1 else "iteration_limit"
        raise MGBConvergenceFailure(f"Convergence failure in mgb_solve at t={t}, k={k}, kappa={kappa}, tol={tol}, maxit={maxit}.", code)
    t_end = time.time()
    return dict(z=z, z_unfinalized=z_unfinalized, c=c, its=np.stack(its, axis=1), ts=np.array(ts),
                kappas=np.array(kappas), t_begin=t_begin, t_end=t_end, t_elapsed=t_end - t_begin,
                times=np.array(times), c_dot_Dz=np.array(c_dot_Dz))


def _matched_t(Q, M, z, c, t_default, barrier_weights=None):
    """src/mgb.jl:307-330."""
    B = Barrier(Q, barrier_weights)
    R = M.R_fine[-1]
    D, w = M.D_fine, M.w
    s0 = np.zeros(R.shape[1])
    c0 = 0.0 * c
    gphi = B.f1(s0, w, c0, R, D, z)
    gc = B.f1(s0, w, c, R, D, z) - gphi
    H = B.f2(s0, w, c, R, D, z)
    nphi = _SOLVER[0](H, gphi)
    nc = _SOLVER[0](H, gc)
    d = float(np.dot(gc, nc))
    b = float(np.dot(gphi, nc) + np.dot(gc, nphi))
    if not d > 0:
        return t_default
    tstar = -b / (2 * d)
    if not (math.isfinite(tstar) and tstar > 0):
        return t_default
    return min(max(tstar, math.sqrt(EPS)), t_default)


def _barrier_weights(w, barrier_nodes):
    """src/convex.jl:279-304.  `None`/'colon' = all nodes."""
    if barrier_nodes is None:
        return None
    sel = np.asarray(barrier_nodes)
    if sel.dtype == bool:
        if sel.size != w.size:
            raise ValueError("barrier_nodes mask has the wrong length")
        nz = sel.astype(np.float64)
    else:
        nz = np.zeros(w.size)
        nz[sel] = 1.0
    m = nz.sum()
    if m <= 0:
        raise ValueError("barrier_nodes selects no nodes")
    if m == nz.size:
        return None
    return nz / m


def mgb_driver(M, f, g, Q, t=0.1, t_feasibility=None, feasibility_Rmax=1.0 / math.sqrt(EPS),
               stopping_criterion=None, line_search=None, finalize="default", barrier_nodes="default",
               log=None, stats=None, **rest):
    """src/mgb.jl:332-584.  `M` is the (main, feasibility) pair of OracleAMG."""
    if t_feasibility is None:
        t_feasibility = t
    M1, M2 = M
    if stopping_criterion is None:
        stopping_criterion = stopping_inexact(0.25 / math.sqrt(M1.w.size), 0.9)
    if line_search is None:
        line_search = linesearch_backtracking()
    if isinstance(finalize, str):
        finalize = stopping_exact(0.9)
    elif finalize is False:
        finalize = NoFinalize()
    if isinstance(barrier_nodes, str):
        barrier_nodes = M1.w != 0
    printlog = (lambda *a: log.append("".join(str(x) for x in a))) if log is not None else (lambda *a: None)
    bw_main = _barrier_weights(M1.w, barrier_nodes)
    m = M1.n
    nD = M1.nD
    c0 = f
    z0 = g
    ncomp = z0.shape[1]
    z2 = z0.T.reshape(-1).copy()                       # vcat of the columns
    w = apply_D(M1.D_fine, z2)
    SOL_feasibility = None
    common = dict(stopping_criterion=stopping_criterion, line_search=line_search, finalize=finalize, stats=stats)
    with np.errstate(all="ignore"):
        foo = node_eval(Q, w, 0)
    if not np.all(np.isfinite(foo)):
        # infeasible start -> phase I (src/mgb.jl:421-572)
        sl = convex_slack(Q, w)
        z1cols = np.concatenate([z0, (2 * np.maximum(sl, 1.0))[:, None]], axis=1)
        b = 2 * max(1.0, float(z1cols[:, -1].max()))
        c1 = np.zeros((m, nD + 1 + ncomp))
        c1[:, nD] = 1.0
        z1 = z1cols.T.reshape(-1).copy()
        slack_of = lambda z: z[ncomp * m:(ncomp + 1) * m]
        feasible = lambda z: bool(slack_of(z).max() < 0)
        Rbox = max(10.0, 10.0 * float(np.abs(z2).max()))
        Rmax = max(float(feasibility_Rmax), Rbox)
        while True:
            printlog("mgb_driver: feasibility phase with bounding box R=", Rbox)
            Q_feas = FeasConvex(Q, float(b), Rbox, nD + 1)
            failure = None
            t_first = [math.inf]

            def feas_stop(z, tt):
                if not feasible(z):
                    return False
                t_first[0] = min(t_first[0], tt)
                return tt >= 2 * t_first[0]

            try:
                kw = dict(rest)
                kw.update(common)
                SOL_feasibility = mgb_core(Q_feas, M2, z1, c1, t=t_feasibility, early_stop=feas_stop,
                                           barrier_weights=None, **kw)
            except KeyboardInterrupt:
                raise
            except Exception as e2:   # broad on purpose, like the reference (src/mgb.jl:505-515)
                failure = e2
            if failure is None:
                if feasible(SOL_feasibility["z"]):
                    break
                zf = SOL_feasibility["z"]
                vmax = max(float(np.abs(zf[k * m:(k + 1) * m]).max()) for k in range(ncomp))
                smax = float(slack_of(zf).max())
                if vmax <= Rbox / 2:
                    raise MGBConvergenceFailure(
                        "The problem appears to be infeasible: the feasibility subproblem converged to a minimizer "
                        f"with positive constraint violation (max slack ~ {smax}) strictly inside the bounding box "
                        f"(max |nodal value| ~ {vmax} <= R/2 with R = {Rbox}).", "infeasible")
                printlog("mgb_driver: phase-I minimizer presses the box; growing R")
            else:
                printlog("mgb_driver: feasibility solve failed at R=", Rbox, ": ", failure)
            Rnext = 10 * Rbox
            if Rnext > Rmax:
                reason = ("the phase-I minimizer still presses against the bounding box" if failure is None
                          else f"the last attempt failed with: {failure}")
                raise MGBConvergenceFailure(
                    f"Could not find a strictly feasible point with nodal values bounded by R = {Rbox} "
                    f"(cap feasibility_Rmax ~ {Rmax}); {reason}.", "feasibility_Rmax")
            Rbox = Rnext
        z2 = SOL_feasibility["z"][: z2.size].copy()
        t = min(t, _matched_t(Q, M1, z2, c0, t, barrier_weights=bw_main))
    kw = dict(rest)
    kw.update(common)
    SOL_main = mgb_core(Q, M1, z2, c0, t=t, barrier_weights=bw_main, **kw)
    z = SOL_main["z"].reshape(ncomp, m).T.copy()
    return dict(z=z, SOL_feasibility=SOL_feasibility, SOL_main=SOL_main)


def mgb_solve(prob, **kw):
    """The reference's CPU path end to end (src/mgb.jl:798-842 with device=CPUDevice)."""
    M = (OracleAMG(prob.M[0]), OracleAMG(prob.M[1]))
    log: List[str] = []
    out = mgb_driver(M, prob.f, prob.g, prob.Q, log=log, **kw)
    out["log"] = "\n".join(log)
    return out
