"""Geometry / MultiGrid / AMG containers and the hierarchy composition logic.

Host-side (CPU, setup-time) mirror of the reference's data producers for the hot
path (reference: src/multigrid.jl:37-43 `Geometry`, :185-265 `MultiGrid`,
`_compose_R`, `_stretch_per_subspace`; :278-288 `AMG`; :372-412
`_assemble_amg_dicts`; :474-538 `amg_helper`, `_prepare_amg`).  The reference runs
all of this on the CPU as well; the device only ever sees the
resulting ``R_fine`` / ``D_fine`` / ``w`` arrays.

Indices are 0-based throughout (the reference is 1-based Julia); connectivity
``t[v, e]`` holds 0-based global node ids.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from collections.abc import Sequence as _SequenceABC
from typing import Any, Callable, Dict, List, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

from . import _setup_native
from .blockmatrices import BlockDiag, block_column


@dataclass
class Geometry:
    """reference: src/multigrid.jl:37-43."""

    discretization: Any
    t: np.ndarray            # (V, N) int, 0-based global node ids
    x: np.ndarray            # (V, N, D) node coordinates
    w: np.ndarray            # (V*N,) quadrature weights
    operators: Dict[str, Any]  # name -> BlockDiag (FEM) or dense ndarray (spectral)

    @property
    def xflat(self) -> np.ndarray:
        """(V*N, D) view in the reference's flat node order (element-major, local node fastest)."""
        V, N, D = self.x.shape
        return self.x.transpose(1, 0, 2).reshape(N * V, D)

    @property
    def labels(self) -> np.ndarray:
        """`vec(geom.t)`: flat labels in element-major order."""
        return self.t.T.reshape(-1)


@dataclass
class MultiGrid:
    """reference: src/multigrid.jl:185-188.  `R[X][l]` lifts level-l subspace-X
    coefficients directly to the fine broken basis."""

    geometry: Geometry
    R: Dict[str, List[Any]]


@dataclass
class AMG:
    """reference: src/multigrid.jl:278-288."""

    geometry: Geometry
    x: np.ndarray
    w: np.ndarray
    R_fine: List[Any]
    D_fine: List[Any]
    # bookkeeping the device upload needs (not in the reference struct: there the
    # BlockColumn wrappers carry it)
    state_names: List[str] = field(default_factory=list)
    D_spec: List[Tuple[int, str]] = field(default_factory=list)  # (state index, operator name)


# ---------------------------------------------------------------------------
# small helpers shared by the FEM families
# ---------------------------------------------------------------------------

def dedupe_labels(x: np.ndarray, tol_scale: float = 100.0) -> np.ndarray:
    """Connectivity labels of coincident rows of ``x`` (n, d), 0-based, numbered by
    first occurrence in scan order.

    The reference (`_dedupe`, src/TensorFEM.jl:74-110) numbers by a random-projection
    sort seeded with Julia's `hash`; only label *equality* propagates into the
    hierarchy, so a deterministic first-occurrence numbering is
    equivalent up to a permutation of the base-mesh unknowns.
    """
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    tol = max(np.abs(x).max(initial=0.0), 1.0) * tol_scale * np.finfo(np.float64).eps
    key = np.round(x / (4 * tol)).astype(np.int64)
    # robust bucketing: points within tol may straddle a rounding boundary, so verify
    # against the representative and fall back to a neighbour search if needed.
    labels = -np.ones(n, dtype=np.int64)
    buckets: Dict[tuple, List[int]] = {}
    reps: List[np.ndarray] = []
    d = x.shape[1]
    offsets = np.array(np.meshgrid(*[[-1, 0, 1]] * d, indexing="ij")).reshape(d, -1).T
    for i in range(n):
        k = key[i]
        found = -1
        for off in offsets:
            for j in buckets.get(tuple(k + off), ()):
                if np.linalg.norm(x[i] - reps[j]) <= tol:
                    found = j
                    break
            if found >= 0:
                break
        if found < 0:
            found = len(reps)
            reps.append(x[i])
            buckets.setdefault(tuple(k), []).append(found)
        labels[i] = found
    return labels


def mask_dirichlet_rows(B: sp.spmatrix, labels: np.ndarray, dd_set) -> sp.csr_matrix:
    """reference: src/multigrid.jl:98-102."""
    dd = np.zeros(int(labels.max()) + 1, dtype=bool)
    dd[np.fromiter(dd_set, dtype=np.int64, count=len(dd_set))] = True
    keep = (~dd[labels]).astype(np.float64)
    out = sp.diags(keep) @ sp.csr_matrix(B)
    out = sp.csr_matrix(out)
    out.eliminate_zeros()
    return out


def corner_labels_from_t(t: np.ndarray, corner_local: Sequence[int]):
    """reference: src/multigrid.jl:137-151.  Compact corner ids by first occurrence in
    (corner, element) flat order; returns (labels[(e*nc + ci)], n_v)."""
    nc = len(corner_local)
    N = t.shape[1]
    fid = t[list(corner_local), :].T.reshape(-1)  # element-major, corner fastest
    _, first_idx, inv = np.unique(fid, return_index=True, return_inverse=True)
    order = np.argsort(first_idx, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    out = rank[inv]
    return out.astype(np.int64), int(order.size)


def continuous_subspace(labels: np.ndarray, n_unique: int, dirichlet_set) -> sp.csr_matrix:
    """0/1 embedding of the zero-trace continuous space into the broken basis
    (reference: `_p2_continuous_subspace`, src/fem2d_P2.jl:331-346).  Column j is the
    j-th interior label in increasing label order."""
    is_dir = np.zeros(n_unique, dtype=bool)
    if len(dirichlet_set):
        is_dir[np.fromiter(dirichlet_set, dtype=np.int64, count=len(dirichlet_set))] = True
    pos = np.cumsum(~is_dir) - 1
    pos[is_dir] = -1
    p = pos[labels]
    rows = np.nonzero(p >= 0)[0]
    return sp.csr_matrix(
        (np.ones(rows.size), (rows, p[rows])), shape=(labels.size, int((~is_dir).sum()))
    )


# ---------------------------------------------------------------------------
# hierarchy composition
# ---------------------------------------------------------------------------

def _is_identity(M) -> bool:
    """True for an exact sparse identity (what `sp.identity` builds): multiplying by it is skipped."""
    if not sp.issparse(M) or M.shape[0] != M.shape[1] or M.nnz != M.shape[0]:
        return False
    M = M.tocsr()
    n = M.shape[0]
    return bool(np.array_equal(M.indices, np.arange(n)) and np.array_equal(M.indptr, np.arange(n + 1)) and np.all(M.data == 1.0))


def _matmat(A, B):
    """`A @ B` for float64 CSR operands through SciPy's own numeric kernel (`csr_matmat`: same entries, same order, same
    bits as `A @ B`), with the output sized by an upper bound -- the candidates per row -- instead of SciPy's exact counting
    pass, which costs half as much as the product itself on the composed prolongators (3-4 entries per row)."""
    try:
        from scipy.sparse import _sparsetools
    except ImportError:                                   # private module moved: the plain product is the same matrix
        return A @ B
    if not (sp.issparse(A) and sp.issparse(B)) or A.shape[1] != B.shape[0]:
        return A @ B
    A, B = sp.csr_matrix(A), sp.csr_matrix(B)
    if A.dtype != np.float64 or B.dtype != np.float64 or A.nnz == 0 or B.nnz == 0:
        return A @ B
    ub = int(np.diff(B.indptr).astype(np.int64)[A.indices].sum())
    if ub == 0 or max(ub, A.shape[0] + 1, B.shape[1]) >= 2**31 - 1:
        return A @ B
    it = np.int32
    M, N = A.shape[0], B.shape[1]
    indptr, indices, data = np.empty(M + 1, dtype=it), np.empty(ub, dtype=it), np.empty(ub, dtype=np.float64)
    _sparsetools.csr_matmat(M, N, np.asarray(A.indptr, dtype=it), np.asarray(A.indices, dtype=it), A.data,
                            np.asarray(B.indptr, dtype=it), np.asarray(B.indices, dtype=it), B.data, indptr, indices, data)
    nnz = int(indptr[M])
    return sp.csr_matrix((data[:nnz].copy(), indices[:nnz].copy(), indptr), shape=(M, N))


def _ladder(rX: List[Any]) -> List[Any]:
    """Cumulative products level -> fine of one refine ladder (reference: src/multigrid.jl:192-204): rfp[L-1] = rX[L-1],
    rfp[l] = rfp[l+1] * rX[l].  With libmgbsetup.so the chain below the leading identity runs in C++ (the same kernel as
    scipy's product, on the previous product in scipy's storage order: tests/test_setup.py compares the two bit for bit)
    and arrives with sorted rows."""
    L = len(rX)
    rfp = [None] * L
    rfp[L - 1] = rX[L - 1]
    l = L - 2
    while l >= 0 and _is_identity(rfp[l + 1]):           # products with the identity are skipped, as before
        rfp[l] = rX[l]
        l -= 1
    if l >= 0 and _setup_native.available() and sp.issparse(rfp[l + 1]):
        A0 = sp.csr_matrix(rfp[l + 1])
        factors = [sp.csr_matrix(rX[k]) for k in range(l, -1, -1)]
        chain = _setup_native.compose_chain(A0, factors)
        if chain is not None:
            for k, M in zip(range(l, -1, -1), chain):
                rfp[k] = M
            return rfp
    for k in range(l, -1, -1):
        rfp[k] = _matmat(rfp[k + 1], rX[k])
    return rfp


_PREFETCHED: Dict[int, Any] = {}        # id(refine list) -> (the list, Future of its ladder): see prefetch_ladder


def prefetch_ladder(rX: List[Any]) -> None:
    """Start composing the ladder of `rX` on a background thread (the C++ chain runs without the interpreter lock) while the
    caller goes on building the next hierarchy in Python; `_compose_R` picks the result up.  No-op without libmgbsetup.so."""
    if not _setup_native.available():
        return
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=1)
    _PREFETCHED[id(rX)] = (rX, pool.submit(_ladder, rX))
    pool.shutdown(wait=False)


def _compose_R(subspaces: Dict[str, List[Any]], refine: Dict[str, List[Any]]):
    """reference: src/multigrid.jl:192-204.  The cumulative products level->fine are shared between the
    symbols that ride the same refine ladder (`full`, `uniform` and the riders do), identity subspaces are
    not multiplied, and a single all-ones column is a row sum: same matrices, a fraction of the setup time.
    Distinct ladders (full / dirichlet) are independent: with the native chain they are composed on two threads."""
    out = {}
    distinct: Dict[int, List[Any]] = {}
    for X in subspaces:
        distinct.setdefault(id(refine[X]), refine[X])
    ladders: Dict[int, List[Any]] = {}
    futures: Dict[int, Any] = {}
    for k, v in list(distinct.items()):                  # ladders a caller started earlier (prefetch_ladder): still running, maybe
        pre = _PREFETCHED.pop(k, None)
        if pre is not None and pre[0] is v:
            futures[k] = pre[1]
            del distinct[k]
    if distinct and (len(distinct) + len(futures)) > 1 and _setup_native.available():
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(max_workers=min(len(distinct), 4))
        for k, v in distinct.items():
            futures[k] = pool.submit(_ladder, v)
        pool.shutdown(wait=False)
    else:
        ladders.update({k: _ladder(v) for k, v in distinct.items()})
    for k, fut in futures.items():
        ladders[k] = fut.result()
    for X in subspaces:
        rX, sX = refine[X], subspaces[X]
        L = len(rX)
        rfp = ladders[id(rX)]
        ops = []
        for l in range(L):
            S = sX[l]
            if _is_identity(S):
                ops.append(_as_op(rfp[l]))
            elif _is_identity(rfp[l]):
                ops.append(_as_op(S))
            elif sp.issparse(S) and S.shape[1] == 1 and S.nnz == S.shape[0] and np.all(S.data == 1.0) and sp.issparse(rfp[l]):
                rs = _setup_native.csr_row_sums(rfp[l]) if sp.isspmatrix_csr(rfp[l]) or getattr(rfp[l], "format", "") == "csr" else None
                if rs is None:
                    rs = np.asarray(rfp[l].sum(axis=1)).ravel()
                ops.append(_as_op(sp.csr_matrix(rs.reshape(-1, 1))))
            else:
                ops.append(_as_op(_matmat(rfp[l], S)))
        out[X] = ops
    return out


def _as_op(M):
    if sp.issparse(M):
        M = sp.csr_matrix(M)
        if not M.has_sorted_indices:
            _setup_native.csr_sort_rows(M)       # short rows: an in-place insertion sort (csrc/setup_host.cpp); same result
        M.sum_duplicates()
        M.sort_indices()
        return M
    return np.asarray(M)


def _stretch_per_subspace(refine, subspaces):
    """reference: src/multigrid.jl:226-265 (ceil-interpolation to a common depth)."""
    L_X = {X: len(refine[X]) for X in refine}
    L_max = max(L_X.values())
    if all(v == L_max for v in L_X.values()):
        return refine, subspaces
    refine_s, sub_s = {}, {}
    for X in refine:
        Lx = L_X[X]
        if Lx == L_max:
            refine_s[X], sub_s[X] = refine[X], subspaces[X]
            continue
        synth2nat = [int(np.ceil(Lx * i / L_max)) for i in range(1, L_max + 1)]  # 1-based natural levels
        rfX, ssX = [None] * L_max, [None] * L_max
        for i in range(L_max):
            ni = synth2nat[i]
            ssX[i] = subspaces[X][ni - 1]
            if i == L_max - 1:
                rfX[i] = refine[X][Lx - 1]
            elif synth2nat[i + 1] > ni:
                rfX[i] = refine[X][ni - 1]
            else:
                m = ssX[i].shape[0]
                rfX[i] = sp.identity(m, format="csr")
        refine_s[X], sub_s[X] = rfX, ssX
    return refine_s, sub_s


def make_multigrid(geometry: Geometry, subspaces, refine) -> MultiGrid:
    """reference: src/multigrid.jl:206-217, :271-276."""
    if not isinstance(refine, dict):
        refine = {k: refine for k in subspaces}
    refine_s, sub_s = _stretch_per_subspace(refine, subspaces)
    return MultiGrid(geometry, _compose_R(sub_s, refine_s))


def assemble_amg_ladder(P_amg: List[sp.spmatrix], bridge: sp.spmatrix, n_doubled: int):
    """reference: src/amg_prolongators.jl:48-66.  Returns (refine, sizes, L_total, K_amg)
    with K_amg the 1-based level index of the bridge."""
    K_amg = len(P_amg) + 1
    L_total = K_amg + 1
    refine = [None] * L_total
    for i, P in enumerate(P_amg):          # P_amg[0] finest
        refine[K_amg - 2 - i] = sp.csr_matrix(P)
    refine[K_amg - 1] = sp.csr_matrix(bridge)
    refine[L_total - 1] = sp.identity(n_doubled, format="csr")
    sizes = [0] * L_total
    sizes[K_amg - 1] = bridge.shape[1]
    for kk in range(K_amg - 2, -1, -1):
        sizes[kk] = refine[kk].shape[1]
    sizes[L_total - 1] = n_doubled
    return refine, sizes, L_total, K_amg


def assemble_amg_dicts(geom: Geometry, n_doubled: int,
                       dirichlet_nodes: Dict[str, List[Tuple[int, int]]],
                       refine_full, sizes_full, L_full: int, K_amg_full: int,
                       build_dirichlet: Callable, full_riders: Dict[str, sp.spmatrix] | None = None) -> MultiGrid:
    """reference: src/multigrid.jl:372-412."""
    sub_full = [None] * L_full
    sub_uniform = [None] * L_full
    for kk in range(K_amg_full):
        sub_full[kk] = sp.identity(sizes_full[kk], format="csr")
        sub_uniform[kk] = sp.csr_matrix(np.ones((sizes_full[kk], 1)))
    sub_full[L_full - 1] = sp.identity(n_doubled, format="csr")
    sub_uniform[L_full - 1] = sp.csr_matrix(np.ones((n_doubled, 1)))
    subspaces = {"full": sub_full, "uniform": sub_uniform}
    refine_d = {"full": refine_full, "uniform": refine_full}
    for sym, E in (full_riders or {}).items():
        sub = [sp.identity(sizes_full[kk], format="csr") for kk in range(K_amg_full)] + [None]
        sub[L_full - 1] = E
        subspaces[sym] = sub
        refine_d[sym] = refine_full
    prefetch_ladder(refine_full)             # composed in the background while the :dirichlet hierarchies are coarsened below
    try:
        for sym, nodes in dirichlet_nodes.items():
            if sym in subspaces:
                raise ValueError(f"dirichlet_nodes key :{sym} is reserved; choose another symbol")
            r, s = build_dirichlet(nodes)
            subspaces[sym] = s
            refine_d[sym] = r
        return make_multigrid(geom, subspaces, refine_d)
    finally:
        pre = _PREFETCHED.pop(id(refine_full), None)     # not consumed (stretched ladders, an error above): let it finish and drop it
        if pre is not None:
            pre[1].cancel()


# ---------------------------------------------------------------------------
# AMG pair consumed by the solver
# ---------------------------------------------------------------------------

def _blockdiag(mats):
    if all(sp.issparse(m) for m in mats):
        # direct CSR concatenation (scipy's block_diag goes through COO and a sort: 10x slower at 1e6 rows)
        mats = [sp.csr_matrix(m) for m in mats]
        for m in mats:
            if not m.has_sorted_indices:
                m.sort_indices()
        native = _setup_native.blockdiag(mats)       # the same concatenation in C++, without the interpreter lock
        if native is not None:
            return native
        nnz0 = np.cumsum([0] + [m.nnz for m in mats])
        col0 = np.cumsum([0] + [m.shape[1] for m in mats])
        rows = sum(m.shape[0] for m in mats)
        it = np.int32 if max(int(nnz0[-1]), int(col0[-1]), rows) < 2**31 - 1 else np.int64
        indptr = np.empty(rows + 1, dtype=it)
        indices = np.empty(int(nnz0[-1]), dtype=it)
        data = np.empty(int(nnz0[-1]), dtype=np.float64)
        indptr[0] = 0
        r = 0
        for k, m in enumerate(mats):
            np.add(m.indptr[1:], it(nnz0[k]), out=indptr[r + 1:r + 1 + m.shape[0]], casting="unsafe")
            np.add(m.indices, it(col0[k]), out=indices[nnz0[k]:nnz0[k + 1]], casting="unsafe")
            data[nnz0[k]:nnz0[k + 1]] = m.data
            r += m.shape[0]
        out = sp.csr_matrix((data, indices, indptr), shape=(rows, int(col0[-1])), copy=False)
        out.has_sorted_indices = True
        return out
    dense = [m.toarray() if sp.issparse(m) else np.asarray(m) for m in mats]
    rows = sum(m.shape[0] for m in dense)
    cols = sum(m.shape[1] for m in dense)
    out = np.zeros((rows, cols))
    r = c = 0
    for m in dense:
        out[r:r + m.shape[0], c:c + m.shape[1]] = m
        r += m.shape[0]
        c += m.shape[1]
    return out


class LazyLevels(_SequenceABC):
    """A read-only list whose entries are built on first access (`make(l)`) and kept."""

    def __init__(self, n: int, make: Callable[[int], Any]):
        self._make = make
        self._items: List[Any] = [None] * n
        self._built = [False] * n

    def __len__(self) -> int:
        return len(self._items)

    def __reduce__(self):
        """Pickles as the plain list of its (built) entries: the builder is a closure."""
        return (list, ([self[j] for j in range(len(self._items))],))

    def realize(self, workers: int = 8) -> None:
        """Build every entry that has not been built yet, on a thread pool (the builders of the package release the interpreter
        lock inside libmgbsetup.so; without the library this is the serial loop with extra steps)."""
        todo = [j for j, done in enumerate(self._built) if not done]
        if len(todo) > 1 and _setup_native.available():
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=min(workers, len(todo))) as pool:
                list(pool.map(self.__getitem__, todo))
        else:
            for j in todo:
                self[j]

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self._items)))]
        n = len(self._items)
        j = i + n if i < 0 else i
        if not 0 <= j < n:
            raise IndexError("level index out of range")
        if not self._built[j]:
            self._items[j] = self._make(j)
            self._built[j] = True
        return self._items[j]


def amg_helper(mg: MultiGrid, state_variables, D) -> AMG:
    """reference: src/multigrid.jl:474-512.  `state_variables` is a list of
    (name, subspace) rows, `D` a list of (state name, operator name) rows."""
    geometry = mg.geometry
    x = geometry.xflat
    w = geometry.w
    ops = geometry.operators
    state_variables = [tuple(r) for r in state_variables]
    D = [tuple(r) for r in D]
    nu = len(state_variables)
    if any(len(r) != 2 for r in state_variables):
        raise ValueError("state_variables must be (name, subspace) rows")
    L = len(mg.R[state_variables[0][1]])
    if w.shape != (x.shape[0],):
        raise ValueError(f"quadrature weights have length {w.size} but the mesh has {x.shape[0]} nodes")
    for sv in state_variables:
        mg.R[sv[1]]                          # an unknown subspace is an error here, not at first use
    # block-diagonal prolongators, one per level, built when first read: the phase-I pair member (three state variables) is
    # only read when a start is infeasible, and its eleven concatenations at L = 9 were a third of `assemble`
    R_fine = LazyLevels(L, lambda l: _blockdiag([mg.R[sv[1]][l] for sv in state_variables]))
    bar = {sv[0]: k for k, sv in enumerate(state_variables)}
    D_fine, D_spec = [], []
    for k, (var, opname) in enumerate(D):
        if var not in bar:
            raise ValueError(f"D row {k} references state variable :{var}, which is not in state_variables")
        if opname not in ops:
            raise ValueError(f"D row {k} references operator :{opname}; available: {list(ops)}")
        D_fine.append(block_column(ops[opname], bar[var], nu))
        D_spec.append((bar[var], opname))
    return AMG(geometry=geometry, x=x, w=w, R_fine=R_fine, D_fine=D_fine,
               state_names=[sv[0] for sv in state_variables], D_spec=D_spec)


def prepare_amg(mg: MultiGrid, state_variables, D, full_space="full", id_operator="id",
                feasibility_slack="feasibility_slack"):
    """reference: src/multigrid.jl:515-538 (`_prepare_amg`): the (main, feasibility) pair."""
    state_variables = [tuple(r) for r in state_variables]
    D = [tuple(r) for r in D]
    M1 = amg_helper(mg, state_variables, D)
    s1 = state_variables + [(feasibility_slack, full_space)]
    D1 = D + [(feasibility_slack, id_operator)] + [(sv[0], id_operator) for sv in state_variables]
    M2 = amg_helper(mg, s1, D1)
    return M1, M2
