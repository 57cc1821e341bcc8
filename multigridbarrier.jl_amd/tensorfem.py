"""Dimension-generic tensor-product Q_k elements: fem1d / fem2d / fem3d.

Setup-time data producer for BASELINE configs 1 and 4 (reference: src/TensorFEM.jl).
Element-local node order is tensor order with axis 1 fastest; corners are in the
reference's bit order (bit a-1 of c-1 selects the low/high end of axis a,
reference: src/TensorFEM.jl:241-249).
"""
from __future__ import annotations

from dataclasses import dataclass
from itertools import product
from typing import Dict, List, Tuple

import numpy as np
import scipy.sparse as sp

from .amg_prolongators import amg_prolongations, amg_ruge_stuben
from .blockmatrices import BlockDiag
from .multigrid import (Geometry, MultiGrid, assemble_amg_dicts, assemble_amg_ladder,
                        continuous_subspace, corner_labels_from_t, dedupe_labels,
                        mask_dirichlet_rows)

_AXIS_SYMS = ("dx", "dy", "dz")


@dataclass
class TensorFEM:
    """Discretization descriptor (reference: src/TensorFEM.jl:50-53)."""

    d: int          # intrinsic dimension
    e: int          # ambient dimension
    k: int          # polynomial order
    K: np.ndarray   # (2^d, N, e) corner tensor

    @property
    def dim(self):
        return self.d


# ---------------------------------------------------------------------------
# 1D reference primitives (reference: src/TensorFEM.jl:116-176)
# ---------------------------------------------------------------------------

def _tf_nodes(k: int) -> np.ndarray:
    # -cospi(i/k): exact at the endpoints and the midpoint
    i = np.arange(k + 1)
    x = -np.cos(np.pi * i / k) if k > 0 else np.zeros(1)
    x[0], x[-1] = (-1.0, 1.0) if k > 0 else (x[0], x[-1])
    if k % 2 == 0 and k > 0:
        x[k // 2] = 0.0
    return x


def _tf_weights(k: int) -> np.ndarray:
    """Clenshaw-Curtis weights on the k+1 nodes (sum 2)."""
    if k == 0:
        return np.array([2.0])
    N = k
    w = np.zeros(N + 1)
    for i in range(N + 1):
        val = 1.0
        for j in range(1, N // 2 + 1):
            c = 1.0 if 2 * j == N else 2.0
            val += c / (1 - 4.0 * j * j) * np.cos(np.pi * (2 * j * i) / N)
        w[i] = val / N if (i == 0 or i == N) else 2 * val / N
    return w


def _tf_dmat(nodes: np.ndarray) -> np.ndarray:
    k = len(nodes) - 1
    D = np.zeros((k + 1, k + 1))
    for i in range(k + 1):
        for j in range(k + 1):
            if i == j:
                D[i, j] = sum(1.0 / (nodes[i] - nodes[m]) for m in range(k + 1) if m != i)
            else:
                num = np.prod([nodes[i] - nodes[m] for m in range(k + 1) if m != j and m != i])
                den = np.prod([nodes[j] - nodes[m] for m in range(k + 1) if m != j])
                D[i, j] = num / den
    return D


def _tf_lagrange(nodes: np.ndarray, xv: float) -> np.ndarray:
    k = len(nodes) - 1
    vals = np.zeros(k + 1)
    for i in range(k + 1):
        num = den = 1.0
        for j in range(k + 1):
            if i != j:
                num *= xv - nodes[j]
                den *= nodes[i] - nodes[j]
        vals[i] = num / den
    return vals


def _multi_index(lin: int, s: int, d: int) -> Tuple[int, ...]:
    out = []
    for _ in range(d):
        out.append(lin % s)
        lin //= s
    return tuple(out)


def _tf_reference(d: int, k: int) -> dict:
    """reference: src/TensorFEM.jl:199-219."""
    s = k + 1
    nodes1 = _tf_nodes(k)
    w1 = _tf_weights(k)
    D1 = _tf_dmat(nodes1)
    I1 = np.eye(s)
    Daxis = []
    for axis in range(d):
        # kron over axes b = d..1 of (D1 if b == axis else I1); axis 0 fastest
        M = np.ones((1, 1))
        for b in range(d - 1, -1, -1):
            M = np.kron(M, D1 if b == axis else I1)
        Daxis.append(M)
    n = s ** d
    nodesref = np.zeros((n, d))
    wref = np.zeros(n)
    for lin in range(n):
        mi = _multi_index(lin, s, d)
        nodesref[lin] = [nodes1[c] for c in mi]
        wref[lin] = np.prod([w1[c] for c in mi])
    return dict(s=s, nodes1=nodes1, w1=w1, D1=D1, Daxis=Daxis, nodesref=nodesref, wref=wref, n=n)


def _tf_q1_lift(nodesref: np.ndarray, d: int) -> np.ndarray:
    """reference: src/TensorFEM.jl:223-238."""
    n = nodesref.shape[0]
    nc = 1 << d
    L = np.zeros((n, nc))
    for i in range(n):
        for c in range(nc):
            wv = 1.0
            for a in range(d):
                xi = nodesref[i, a]
                wv *= (1 - xi) * 0.5 if ((c >> a) & 1) == 0 else (1 + xi) * 0.5
            L[i, c] = wv
    return L


def _tf_corner_local(c: int, s: int, d: int) -> int:
    """0-based local node index of corner c (0-based) (reference: src/TensorFEM.jl:241-249)."""
    lin, stride = 0, 1
    for a in range(d):
        ia = 0 if ((c >> a) & 1) == 0 else s - 1
        lin += ia * stride
        stride *= s
    return lin


def _tf_extract_corners(x: np.ndarray, k: int, d: int) -> np.ndarray:
    s = k + 1
    idx = [_tf_corner_local(c, s, d) for c in range(1 << d)]
    return x[idx, :, :].copy()


# ---------------------------------------------------------------------------
# topological DOF numbering (reference: src/TensorFEM.jl:274-383)
# ---------------------------------------------------------------------------

def _entity_corner_ids(cor, mi, inter, s, d):
    nint = len(inter)
    out = []
    for combo in range(1 << nint):
        cbits = 0
        for a in range(d):
            if a in inter:
                bit = (combo >> inter.index(a)) & 1
            else:
                bit = 1 if mi[a] == s - 1 else 0
            cbits |= bit << a
        out.append(int(cor[cbits]))
    return out


def _face_pos(ids, pi, pj, k):
    g = lambda i, j: ids[i + 2 * j]
    i0 = j0 = 0
    best = g(0, 0)
    for j in (0, 1):
        for i in (0, 1):
            if g(i, j) < best:
                best, i0, j0 = g(i, j), i, j
    ri = pi if i0 == 0 else k - pi
    rj = pj if j0 == 0 else k - pj
    if g(1 - i0, j0) > g(i0, 1 - j0):
        ri, rj = rj, ri
    return ri + rj * (k + 1)


def tensor_dofmap(t_corner: np.ndarray, k: int, d: int) -> np.ndarray:
    """Full-node connectivity from corner connectivity alone (reference:
    `tensor_dofmap`, src/TensorFEM.jl:338-383).  Ids 0-based; corner ids carry
    through, shared edge/face nodes are keyed by their corner-id set, cell-interior
    nodes get fresh ids."""
    s = k + 1
    n = s ** d
    nc = 1 << d
    if t_corner.shape[0] != nc:
        raise ValueError(f"tensor_dofmap: t_corner must have 2^{d} = {nc} rows")
    N = t_corner.shape[1]
    t = np.empty((n, N), dtype=np.int64)
    next_id = int(t_corner.max()) + 1 if t_corner.size else 0
    reg: Dict[tuple, int] = {}
    for e in range(N):
        cor = t_corner[:, e]
        for v in range(n):
            mi = _multi_index(v, s, d)
            inter = [a for a in range(d) if mi[a] != 0 and mi[a] != s - 1]
            nint = len(inter)
            if nint == d:
                t[v, e] = next_id
                next_id += 1
                continue
            ids = _entity_corner_ids(cor, mi, inter, s, d)
            if nint == 0:
                t[v, e] = ids[0]
                continue
            if nint == 1:
                p = mi[inter[0]]                      # 1..k-1 from the low end
                pos = p if ids[0] <= ids[1] else k - p
                key = (tuple(sorted(ids[:2])), pos)
            elif nint == 2:
                pos = _face_pos(ids, mi[inter[0]], mi[inter[1]], k)
                key = (tuple(sorted(ids)), pos)
            else:
                raise ValueError("tensor_dofmap: interior grids on shared entities of dimension >= 3 are not supported")
            idv = reg.get(key)
            if idv is None:
                idv = next_id
                next_id += 1
                reg[key] = idv
            t[v, e] = idv
    return t


# ---------------------------------------------------------------------------
# geometry construction (reference: src/TensorFEM.jl:395-516)
# ---------------------------------------------------------------------------

def _tf_promote(K: np.ndarray, k: int, d: int) -> np.ndarray:
    ref = _tf_reference(d, k)
    Lq1 = _tf_q1_lift(ref["nodesref"], d)
    return np.einsum("ic,ced->ied", Lq1, K)


def _tf_resolve_mesh(K: np.ndarray, k: int, d: int) -> np.ndarray:
    n = (k + 1) ** d
    nc = 1 << d
    if not (d <= K.shape[2] <= 3):
        raise ValueError(f"fem{d}d: K ambient dim must satisfy {d} <= e <= 3")
    if K.shape[0] == n:
        return K
    if K.shape[0] == nc:
        return _tf_promote(K, k, d)
    raise ValueError(f"fem{d}d: K needs {nc} corners or (k+1)^{d}={n} nodes per element")


def _tf_build_geometry(d: int, e: int, k: int, x: np.ndarray, t: np.ndarray | None = None) -> Geometry:
    s = k + 1
    n = s ** d
    N = x.shape[1]
    if x.shape[0] != n or x.shape[2] != e:
        raise ValueError(f"fem{d}d: mesh tensor has the wrong shape {x.shape}")
    ref = _tf_reference(d, k)
    Daxis = ref["Daxis"]
    # grefs[b][i, el, dim] = d x_dim / d xi_b at node i
    grefs = np.stack([np.einsum("im,med->ied", Daxis[b], x) for b in range(d)], axis=-1)  # (n, N, e, d) = J
    J = grefs
    g = np.einsum("iNab,iNac->iNbc", J, J)                     # first fundamental form (d x d)
    detg = np.linalg.det(g)
    P = np.linalg.solve(g, np.swapaxes(J, -1, -2))              # (n, N, d, e): pseudo-inverse
    deriv = []
    DaxisS = np.stack(Daxis, axis=0)                            # (d, n, n)
    for dim in range(e):
        blk = np.einsum("iNb,bim->imN", P[:, :, :, dim], DaxisS)
        deriv.append(blk)
    w = (ref["wref"][:, None] * np.sqrt(np.maximum(detg, 0.0))).T.reshape(-1)
    if not np.all(w > 0):
        raise ValueError(f"fem{d}d: non-positive quadrature weight (degenerate element map)")
    idb = np.broadcast_to(np.eye(n)[:, :, None], (n, n, N)).copy()
    ops = {"id": BlockDiag(idb)}
    for a in range(e):
        ops[_AXIS_SYMS[a]] = BlockDiag(deriv[a])
    if t is None:
        flat = x.transpose(1, 0, 2).reshape(-1, e)
        t = dedupe_labels(flat).reshape(N, n).T
    disc = TensorFEM(d, e, k, _tf_extract_corners(x, k, d))
    return Geometry(disc, np.asarray(t, dtype=np.int64), np.ascontiguousarray(x), w, ops)


def _construct(k, K, t, d, e):
    x = _tf_resolve_mesh(np.asarray(K, dtype=np.float64), k, d)
    return _tf_build_geometry(d, e, k, x, t)


def fem1d(nodes=(-1.0, 1.0), k: int = 1, K: np.ndarray | None = None, ambient: int = 1, t=None) -> Geometry:
    """reference: `fem1d`, src/TensorFEM.jl:555-562."""
    if K is None:
        nodes = np.asarray(nodes, dtype=np.float64)
        K = np.stack([nodes[:-1], nodes[1:]], axis=0)[:, :, None]
    return _construct(k, K, t, 1, ambient)


def _default_square():
    return np.array([[-1.0, -1], [1, -1], [-1, 1], [1, 1]])[:, None, :]


def _default_cube():
    c = np.array(list(product([-1.0, 1.0], repeat=3)))[:, ::-1]   # axis 1 fastest
    return c[:, None, :]


def fem2d(k: int = 1, K: np.ndarray | None = None, ambient: int = 2, t=None) -> Geometry:
    """reference: `fem2d`, src/TensorFEM.jl:589-595."""
    return _construct(k, _default_square() if K is None else K, t, 2, ambient)


def fem3d(k: int = 3, K: np.ndarray | None = None, t=None) -> Geometry:
    """reference: `fem3d`, src/TensorFEM.jl:624-630."""
    return _construct(k, _default_cube() if K is None else K, t, 3, 3)


# ---------------------------------------------------------------------------
# boundary, AMG (reference: src/TensorFEM.jl:643-796)
# ---------------------------------------------------------------------------

def find_boundary(geom: Geometry) -> List[Tuple[int, int]]:
    disc = geom.discretization
    d, k = disc.d, disc.k
    s = k + 1
    n = s ** d
    N = geom.t.shape[1]
    labels = geom.labels.reshape(N, n)
    faces_local = []
    for a in range(d):
        for layer in (0, s - 1):
            faces_local.append([lin for lin in range(n) if _multi_index(lin, s, d)[a] == layer])
    facecount: Dict[tuple, int] = {}
    for e in range(N):
        for fl in faces_local:
            sig = tuple(sorted(labels[e, fl].tolist()))
            facecount[sig] = facecount.get(sig, 0) + 1
    bdry = set()
    for sig, c in facecount.items():
        if c == 1:
            bdry.update(sig)
    isb = np.zeros(int(labels.max()) + 1, dtype=bool)
    isb[list(bdry)] = True
    flat = np.nonzero(isb[labels.reshape(-1)])[0]
    return [(int(i % n), int(i // n)) for i in flat]


def _tf_interior_q1_lift(node_map_q1: np.ndarray, k: int, d: int, n_v: int, interior: np.ndarray) -> sp.csr_matrix:
    s = k + 1
    n = s ** d
    nc = 1 << d
    interior_idx = -np.ones(n_v, dtype=np.int64)
    interior_idx[interior] = np.arange(len(interior))
    Lq1 = _tf_q1_lift(_tf_reference(d, k)["nodesref"], d)        # n x nc
    N = node_map_q1.size // nc
    cui = interior_idx[node_map_q1.reshape(N, nc)]                # (N, nc)
    rows = (np.arange(N)[:, None, None] * n + np.arange(n)[None, :, None]) + np.zeros((1, 1, nc), dtype=np.int64)
    cols = np.broadcast_to(cui[:, None, :], (N, n, nc))
    vals = np.broadcast_to(Lq1[None, :, :], (N, n, nc))
    keep = (vals != 0) & (cols >= 0)
    return sp.csr_matrix((vals[keep], (rows[keep], cols[keep])), shape=(N * n, len(interior)))


def amg(geom: Geometry, prolongator=None, dirichlet_nodes=None, auxiliary_postprocess=None) -> MultiGrid:
    """reference: src/TensorFEM.jl:727-796."""
    if prolongator is None:
        prolongator = amg_ruge_stuben(max_coarse=2)
    if dirichlet_nodes is None:
        dirichlet_nodes = {"dirichlet": find_boundary(geom)}
    disc = geom.discretization
    d, k = disc.d, disc.k
    s = k + 1
    n = s ** d
    N = geom.t.shape[1]
    n_doubled = n * N
    nc = 1 << d
    full_labels = geom.labels
    n_full_unique = int(full_labels.max()) + 1
    cornerlocal = tuple(_tf_corner_local(c, s, d) for c in range(nc))
    node_map_q1, n_v = corner_labels_from_t(geom.t, cornerlocal)

    W = sp.diags(geom.w)
    A_doubled = sp.csr_matrix((n_doubled, n_doubled))
    for a in range(geom.x.shape[2]):
        Da = geom.operators[_AXIS_SYMS[a]].to_sparse()
        A_doubled = A_doubled + Da.T @ W @ Da

    full_to_corner = -np.ones(n_full_unique, dtype=np.int64)
    rows = (np.arange(N)[:, None] * n + np.array(cornerlocal)[None, :]).reshape(-1)
    full_to_corner[full_labels[rows]] = node_map_q1

    S_full = _tf_interior_q1_lift(node_map_q1, k, d, n_v, np.arange(n_v))
    M_full = sp.csr_matrix(S_full.T @ A_doubled @ S_full)
    if auxiliary_postprocess is not None:
        M_full = sp.csr_matrix(auxiliary_postprocess(M_full))

    def hierarchy(interior, amg_input):
        S_lift = _tf_interior_q1_lift(node_map_q1, k, d, n_v, interior)
        P_amg = amg_prolongations(amg_input, prolongator)
        return assemble_amg_ladder(P_amg, S_lift, n_doubled)

    refine_full, sizes_full, L_full, K_amg_full = hierarchy(np.arange(n_v), M_full)

    def build_dirichlet(nodes):
        lin = np.array([v + e * n for (v, e) in nodes], dtype=np.int64)
        dd_set = set(full_labels[lin].tolist())
        dc_set = {int(full_to_corner[f]) for f in dd_set if full_to_corner[f] >= 0}
        interior = np.array(sorted(set(range(n_v)) - dc_set), dtype=np.int64)
        refine_dir, sizes_dir, L_dir, K_amg_dir = hierarchy(interior, M_full[interior][:, interior])
        refine_dir[K_amg_dir - 1] = mask_dirichlet_rows(refine_dir[K_amg_dir - 1], full_labels, dd_set)
        sub = [sp.identity(sizes_dir[kk], format="csr") for kk in range(K_amg_dir)] + [None]
        sub[L_dir - 1] = continuous_subspace(full_labels, n_full_unique, dd_set)
        return refine_dir, sub

    return assemble_amg_dicts(geom, n_doubled, dirichlet_nodes, refine_full, sizes_full, L_full,
                              K_amg_full, build_dirichlet)


# ---------------------------------------------------------------------------
# geometric subdivision (geometry only) (reference: src/TensorFEM.jl:821-954)
# ---------------------------------------------------------------------------

def _tf_refine_connectivity(t: np.ndarray, k: int, d: int) -> np.ndarray:
    s = k + 1
    nc = 1 << d
    N = t.shape[1]
    cornerlocal = [_tf_corner_local(c, s, d) for c in range(nc)]
    child_corners = np.empty((nc, nc * N), dtype=np.int64)
    vertex_ids: Dict[tuple, int] = {}
    for e in range(N):
        parent = [int(t[cornerlocal[c], e]) for c in range(nc)]
        for ch in range(nc):
            for c in range(nc):
                # position in the parent's 3-point topological grid: 0 low, 1 centre, 2 high
                mi = tuple(((ch >> a) & 1) + ((c >> a) & 1) for a in range(d))
                inter = [a for a in range(d) if mi[a] == 1]
                ent = _entity_corner_ids(parent, mi, inter, 3, d)
                if not inter:
                    key = (0, ent[0])
                elif len(inter) == d:
                    key = (-1, e)
                else:
                    key = (len(inter),) + tuple(sorted(ent))
                vid = vertex_ids.get(key)
                if vid is None:
                    vid = len(vertex_ids)
                    vertex_ids[key] = vid
                child_corners[c, e * nc + ch] = vid
    return tensor_dofmap(child_corners, k, d)


def _tf_refine_local(k: int, d: int) -> np.ndarray:
    s = k + 1
    n = s ** d
    nc = 1 << d
    nodes1 = _tf_nodes(k)
    P = np.zeros((nc * n, n))
    for ch in range(nc):
        childnodes = [nodes1 * 0.5 + (-0.5 if ((ch >> a) & 1) == 0 else 0.5) for a in range(d)]
        lag = [[_tf_lagrange(nodes1, childnodes[a][i]) for i in range(s)] for a in range(d)]
        for i in range(n):
            ci = _multi_index(i, s, d)
            for j in range(n):
                cj = _multi_index(j, s, d)
                wv = 1.0
                for a in range(d):
                    wv *= lag[a][ci[a]][cj[a]]
                P[ch * n + i, j] = wv
    return P


def subdivide(geom: Geometry, L: int) -> Geometry:
    """`subdivide(geom, L)` for the tensor family: L-1 levels of 2^d-child subdivision,
    fine geometry only (reference: src/multigrid.jl:472 -> src/TensorFEM.jl:888-920)."""
    disc = geom.discretization
    if not isinstance(disc, TensorFEM):
        raise TypeError("subdivide: TensorFEM geometry expected")
    if L < 1:
        raise ValueError("L must be >= 1")
    if L == 1:
        return geom
    d, k = disc.d, disc.k
    n = (k + 1) ** d
    nc = 1 << d
    P_local = _tf_refine_local(k, d).reshape(nc, n, n)
    X, t = geom.x, geom.t
    for _ in range(L - 1):
        Xf = np.einsum("cij,jed->iecd", P_local, X)               # (n, N, nc, D)
        X = Xf.reshape(n, -1, X.shape[2])                          # element-major, child fastest
        t = _tf_refine_connectivity(t, k, d)
    return _tf_build_geometry(d, disc.e, k, X, t)


def _vblock_refine(P_local: np.ndarray, n: int, nc: int, n_elems: int) -> sp.csr_matrix:
    """The level transfer as a sparse matrix: child block (e*nc + ch) of the fine broken basis reads the
    n x n table P_local[ch] from parent block e (reference: `_vblock_sparse(n, n, nc, n_elems, ref_data)`,
    src/TensorFEM.jl:930-936 / src/fem2d_P2.jl:503-512).  Structural zeros of the table are dropped like
    the reference's sparse storage drops them."""
    blocks = P_local.reshape(nc, n, n)
    ch, i, j = np.nonzero(blocks)
    v = blocks[ch, i, j]
    e = np.arange(n_elems)
    rows = ((e[:, None] * nc + ch[None, :]) * n + i[None, :]).reshape(-1)
    cols = (e[:, None] * n + j[None, :]).reshape(-1)
    out = sp.csr_matrix((np.tile(v, n_elems), (rows, cols)), shape=(n * nc * n_elems, n * n_elems))
    out.sort_indices()
    return out


def _tf_continuous_subspace(X: np.ndarray, t: np.ndarray, k: int, d: int) -> sp.csr_matrix:
    """reference: src/TensorFEM.jl:804-815 (zero-trace continuous Q_k space of one level)."""
    disc = TensorFEM(d=d, e=X.shape[2], k=k, K=np.zeros((1 << d, 0, X.shape[2])))
    geomlike = Geometry(disc, t, X, np.zeros(0), {})
    labels = geomlike.labels
    n = (k + 1) ** d
    bset = {int(labels[v + e * n]) for (v, e) in find_boundary(geomlike)}
    return continuous_subspace(labels, int(labels.max()) + 1, bset)


def geometric_mg(geom: Geometry, L: int) -> MultiGrid:
    """`geometric_mg(geom, L)`: the L-level geometric-subdivision hierarchy of the tensor family
    (reference: src/TensorFEM.jl:888-954): per level the continuous zero-trace space, the broken
    identity and the constant; level transfers are the element-local 2^d-child interpolation tables."""
    from .multigrid import make_multigrid
    disc = geom.discretization
    if not isinstance(disc, TensorFEM):
        raise TypeError("geometric_mg: TensorFEM geometry expected")
    if L < 1:
        raise ValueError("L must be >= 1")
    d, k = disc.d, disc.k
    n = (k + 1) ** d
    nc = 1 << d
    P_local = _tf_refine_local(k, d)
    meshes, topo = [geom.x], [geom.t]
    for _ in range(L - 1):
        X = meshes[-1]
        Xf = np.einsum("cij,jed->iecd", P_local.reshape(nc, n, n), X)
        meshes.append(Xf.reshape(n, -1, X.shape[2]))
        topo.append(_tf_refine_connectivity(topo[-1], k, d))
    geomL = geom if L == 1 else _tf_build_geometry(d, disc.e, k, meshes[-1], topo[-1])
    refine = [_vblock_refine(P_local, n, nc, meshes[l].shape[1]) for l in range(L - 1)]
    refine.append(sp.identity(n * meshes[-1].shape[1], format="csr"))
    subspaces = {"dirichlet": [], "full": [], "uniform": []}
    for l in range(L):
        nl = n * meshes[l].shape[1]
        subspaces["dirichlet"].append(_tf_continuous_subspace(meshes[l], topo[l], k, d))
        subspaces["full"].append(sp.identity(nl, format="csr"))
        subspaces["uniform"].append(sp.csr_matrix(np.ones((nl, 1))))
    return make_multigrid(geomL, subspaces, refine)
