"""2-D P1 triangles in the broken (per-element doubled) basis: 3 nodes per element.

Host-side (CPU, setup-time) restatement of the reference's P1 discretization (reference: src/fem2d_P1.jl:13-326): the
single-level `fem2d_P1()` geometry, `subdivide` / `geometric_mg` by red refinement with the 12 x 3 child table, `amg` on the
continuous corner stiffness, `find_boundary`.  SURVEY.md section 2 lists it as out of scope ("not in any config; block size 3,
same kernels would cover it"); it is here because the reference's own backend-parity test runs it (the ninth case of
test/test_cuda.jl:34-56) and two of its golden vectors are P1 solves (test/test_algebraic.jl:45-49).  Nothing on the device
is specific to it: an element block is 3 x 3, the generic element kernels take any p <= 64.

Indices are 0-based (the reference is 1-based Julia); `t[v, e]` holds global corner ids numbered by first occurrence.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np
import scipy.sparse as sp

from .amg_prolongators import amg_prolongations, amg_ruge_stuben
from .blockmatrices import BlockDiag
from .fem2d_p2 import _assemble_p1_stiffness_full
from .multigrid import (Geometry, MultiGrid, assemble_amg_dicts, assemble_amg_ladder, dedupe_labels, make_multigrid)


@dataclass
class FEM2D_P1:
    """reference: src/fem2d_P1.jl:13-15 (`K`: the geometry's own corner tensor, informational)."""

    K: np.ndarray
    dim: int = 2


# child corners of the four red-refinement children as rows of the 12 x 3 interpolation table
# (reference: src/fem2d_P1.jl:219-234): child 0 = (P1, M12, M31), 1 = (M12, P2, M23), 2 = (M31, M23, P3), 3 = (M12, M23, M31)
_P, _M = np.eye(3), 0.5 * (np.eye(3)[[0, 1, 2]] + np.eye(3)[[1, 2, 0]])        # corners; midpoints M12, M23, M31
_REFINE = np.vstack([_P[0], _M[0], _M[2],
                     _M[0], _P[1], _M[1],
                     _M[2], _M[1], _P[2],
                     _M[0], _M[1], _M[2]])


def refine_table() -> np.ndarray:
    return _REFINE.copy()


def _operators(X: np.ndarray):
    """Element blocks of d/dx, d/dy and the quadrature weights (reference: src/fem2d_P1.jl:266-297): on a triangle with
    corners 1, 2, 3 and det = (x2 - x1)(y3 - y1) - (x3 - x1)(y2 - y1), every row of the 3 x 3 block is (b_j / det) resp.
    (c_j / det) with b = (y2 - y3, y3 - y1, y1 - y2), c = (x3 - x2, x1 - x3, x2 - x1); weights |det| / 6."""
    x1, y1 = X[0, :, 0], X[0, :, 1]
    x2, y2 = X[1, :, 0], X[1, :, 1]
    x3, y3 = X[2, :, 0], X[2, :, 1]
    det2 = (x2 - x1) * (y3 - y1) - (x3 - x1) * (y2 - y1)
    b = np.stack([y2 - y3, y3 - y1, y1 - y2], axis=0) / det2            # (3, N)
    c = np.stack([x3 - x2, x1 - x3, x2 - x1], axis=0) / det2
    N = X.shape[1]
    dx = np.broadcast_to(b[None, :, :], (3, 3, N)).copy()               # dx[i, j, e] = b_j(e) / det(e)
    dy = np.broadcast_to(c[None, :, :], (3, 3, N)).copy()
    w = np.repeat(np.abs(det2) / 6.0, 3)                                 # area / 3 per corner, element-major
    return dx, dy, w


def _build_geometry(X: np.ndarray, t: np.ndarray) -> Geometry:
    N = X.shape[1]
    dx, dy, w = _operators(X)
    ident = np.broadcast_to(np.eye(3)[:, :, None], (3, 3, N)).copy()
    ops = {"id": BlockDiag(ident), "dx": BlockDiag(dx), "dy": BlockDiag(dy)}
    return Geometry(discretization=FEM2D_P1(X), t=np.asarray(t, dtype=np.int64), x=X, w=w, operators=ops)


def fem2d_P1(K: np.ndarray | None = None, t: np.ndarray | None = None) -> Geometry:
    """Single-level geometry on the triangulation `K` (3 x N x 2); default: the square [-1, 1]^2 cut into two triangles
    (reference: src/fem2d_P1.jl:40-46)."""
    if K is None:
        pts = np.array([[-1.0, -1.0], [1.0, -1.0], [-1.0, 1.0], [1.0, -1.0], [1.0, 1.0], [-1.0, 1.0]])
        K = pts.reshape(2, 3, 2).transpose(1, 0, 2).copy()
    K = np.asarray(K, dtype=np.float64)
    if K.ndim != 3 or K.shape[0] != 3 or K.shape[2] != 2:
        raise ValueError("K must be a (3, N, 2) corner tensor")
    N = K.shape[1]
    if t is None:
        t = dedupe_labels(K.transpose(1, 0, 2).reshape(3 * N, 2)).reshape(N, 3).T
    t = np.asarray(t, dtype=np.int64)
    if t.shape != (3, N) or t.min() < 0:
        raise ValueError("t must be a (3, N) array of non-negative node ids")
    return _build_geometry(K, t)


def _refine_connectivity(t: np.ndarray) -> np.ndarray:
    """Red refinement of the connectivity: edge midpoints get new ids in order of first occurrence, scanning the elements and,
    inside an element, the edges (a, b), (b, c), (c, a) (reference: src/fem2d_P1.jl:236-264)."""
    N = t.shape[1]
    a, b, c = t[0], t[1], t[2]
    u = np.stack([a, b, c], axis=1).reshape(-1)                        # edge (u, v), element-major, three per element
    v = np.stack([b, c, a], axis=1).reshape(-1)
    M = int(t.max()) + 1
    key = np.minimum(u, v).astype(np.int64) * M + np.maximum(u, v)
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")                            # first-occurrence numbering
    rank = np.empty(uniq.size, dtype=np.int64)
    rank[order] = np.arange(uniq.size)
    mid = (M + rank[inv]).reshape(N, 3)
    ab, bc, ca = mid[:, 0], mid[:, 1], mid[:, 2]
    out = np.empty((3, 4 * N), dtype=np.int64)
    out[:, 0::4] = np.stack([a, ab, ca])
    out[:, 1::4] = np.stack([ab, b, bc])
    out[:, 2::4] = np.stack([ca, bc, c])
    out[:, 3::4] = np.stack([ab, bc, ca])
    return out


def _boundary_corners(t: np.ndarray) -> np.ndarray:
    """Corner ids on the boundary: end points of the edges used by exactly one triangle."""
    a = t[[0, 1, 2]].T.reshape(-1)
    b = t[[1, 2, 0]].T.reshape(-1)
    M = int(t.max()) + 1
    key = np.minimum(a, b).astype(np.int64) * M + np.maximum(a, b)
    uniq, counts = np.unique(key, return_counts=True)
    once = uniq[counts == 1]
    return np.unique(np.concatenate([once // M, once % M]))


def find_boundary(geom: Geometry) -> List[Tuple[int, int]]:
    """(v, e) pairs (0-based) of the corners on the boundary, one per triangle that owns the corner
    (reference: src/fem2d_P1.jl:57-68)."""
    isb = np.zeros(int(geom.t.max()) + 1, dtype=bool)
    isb[_boundary_corners(geom.t)] = True
    flat = np.nonzero(isb[geom.labels])[0]
    return [(int(i % 3), int(i // 3)) for i in flat]


def _continuous(t: np.ndarray) -> sp.csr_matrix:
    """Zero-trace continuous P1 space of one level: one column per interior corner, ascending id
    (reference: src/fem2d_P1.jl:299-326)."""
    labels = t.T.reshape(-1)
    n_v = int(labels.max()) + 1
    pos = -np.ones(n_v, dtype=np.int64)
    interior = np.setdiff1d(np.arange(n_v), _boundary_corners(t))
    pos[interior] = np.arange(interior.size)
    rows = np.nonzero(pos[labels] >= 0)[0]
    return sp.csr_matrix((np.ones(rows.size), (rows, pos[labels[rows]])), shape=(labels.size, interior.size))


def geometric_mg(geom: Geometry, L: int) -> MultiGrid:
    """reference: src/fem2d_P1.jl:131-212."""
    from .tensorfem import _vblock_refine
    if not isinstance(geom.discretization, FEM2D_P1):
        raise TypeError("geometric_mg: FEM2D_P1 geometry expected")
    if L < 1:
        raise ValueError("L must be >= 1")
    X, t = geom.x, geom.t
    topo, sizes = [t], [X.shape[1]]
    for _ in range(L - 1):
        t = _refine_connectivity(t)
        topo.append(t)
        sizes.append(4 * sizes[-1])
    refine = [_vblock_refine(_REFINE, 3, 4, sizes[l]) for l in range(L - 1)]
    refine.append(sp.identity(3 * sizes[-1], format="csr"))
    xf = X.transpose(1, 0, 2).reshape(-1, 2)
    for l in range(L - 1):
        xf = refine[l] @ xf
    Xfine = np.ascontiguousarray(xf.reshape(sizes[-1], 3, 2).transpose(1, 0, 2))
    geomL = _build_geometry(Xfine, topo[-1])
    subspaces = {"dirichlet": [], "full": [], "uniform": []}
    for l in range(L):
        nl = 3 * sizes[l]
        subspaces["dirichlet"].append(_continuous(topo[l]))
        subspaces["full"].append(sp.identity(nl, format="csr"))
        subspaces["uniform"].append(sp.csr_matrix(np.ones((nl, 1))))
    return make_multigrid(geomL, subspaces, refine)


def subdivide(geom: Geometry, L: int) -> Geometry:
    return geometric_mg(geom, L).geometry


def _bridge(tri_conn: np.ndarray, n_v: int, interior: np.ndarray) -> sp.csr_matrix:
    """Interior corners -> doubled corners: each of an element's three nodes takes its corner's coefficient, Dirichlet
    corners get no entry (reference: src/fem2d_P1.jl:195-217)."""
    idx = -np.ones(n_v, dtype=np.int64)
    idx[interior] = np.arange(interior.size)
    col = idx[tri_conn.reshape(-1)]                                      # element-major, corner fastest = the doubled row order
    rows = np.nonzero(col >= 0)[0]
    return sp.csr_matrix((np.ones(rows.size), (rows, col[rows])), shape=(tri_conn.size, interior.size))


def _hierarchy(tri_conn, K_full, interior, n_v, n_doubled, prolongator):
    """reference: src/fem2d_P1.jl:74-84."""
    interior = np.asarray(interior, dtype=np.int64)
    K_loc = sp.csr_matrix(K_full)[interior][:, interior]
    P_amg = amg_prolongations(K_loc, prolongator)
    return assemble_amg_ladder(P_amg, _bridge(tri_conn, n_v, interior), n_doubled)


def amg(geom: Geometry, prolongator=None, dirichlet_nodes: Dict[str, List[Tuple[int, int]]] | None = None) -> MultiGrid:
    """AMG hierarchy on the continuous P1 stiffness (reference: src/fem2d_P1.jl:86-126)."""
    if prolongator is None:
        prolongator = amg_ruge_stuben(max_coarse=2)
    if dirichlet_nodes is None:
        dirichlet_nodes = {"dirichlet": find_boundary(geom)}
    N = geom.t.shape[1]
    n_doubled = 3 * N
    labels = geom.labels
    n_v = int(labels.max()) + 1
    x_fine = geom.xflat
    corners = np.zeros((n_v, 2))
    _, first = np.unique(labels, return_index=True)                      # first occurrence wins (src/multigrid.jl:113-125)
    corners[labels[first]] = x_fine[first]
    tri_conn = labels.reshape(N, 3)
    K_full = _assemble_p1_stiffness_full(corners, tri_conn)
    refine_full, sizes_full, L_full, K_amg_full = _hierarchy(tri_conn, K_full, np.arange(n_v), n_v, n_doubled, prolongator)

    def build_dirichlet(nodes):
        lin = np.array([v + e * 3 for (v, e) in nodes], dtype=np.int64)
        dset = set(labels[lin].tolist()) if lin.size else set()
        interior = np.array(sorted(set(range(n_v)) - dset), dtype=np.int64)
        refine_dir, sizes_dir, L_dir, K_amg_dir = _hierarchy(tri_conn, K_full, interior, n_v, n_doubled, prolongator)
        sub = [sp.identity(sizes_dir[kk], format="csr") for kk in range(K_amg_dir)] + [None]
        sub[L_dir - 1] = sp.csr_matrix(refine_dir[K_amg_dir - 1])        # the fine subspace IS the bridge (P1 has corners only)
        return refine_dir, sub

    return assemble_amg_dicts(geom, n_doubled, dirichlet_nodes, refine_full, sizes_full, L_full, K_amg_full, build_dirichlet)
