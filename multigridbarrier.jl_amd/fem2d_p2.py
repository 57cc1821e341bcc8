"""2D simplicial P2(+bubble) discretization: mesh, operators, subdivision, AMG hierarchy.

Setup-time data producer for BASELINE configs 2-3 (reference: src/fem2d_P2.jl).  The
per-triangle node layout is the reference's
``corner1, edge(1,2), corner2, edge(2,3), corner3, edge(3,1)[, centroid]``
(reference: src/fem2d_P2.jl:19-20).

The reference element tables (node matrix K, weights w, derivative matrices) are
*derived* here from their definition -- the nodal basis of span{P2} (+) span{l1 l2 l3}
on the 7 (or 6) nodes, differentiated along the barycentric directions (xi = l1,
eta = l2 with corner3 at the origin), weights w_j = 2 * int phi_j -- in exact
rational arithmetic.  `tests/test_setup.py` checks them against the values the
reference tabulates (reference: src/fem2d_P2.jl:74-96, :109-128).
"""
from __future__ import annotations

from dataclasses import dataclass
from fractions import Fraction
from math import factorial
from typing import Dict, List, Tuple

import numpy as np
import scipy.sparse as sp

from .amg_prolongators import amg_prolongations, amg_ruge_stuben
from .blockmatrices import BlockDiag
from .multigrid import (Geometry, MultiGrid, assemble_amg_dicts, assemble_amg_ladder,
                        continuous_subspace, corner_labels_from_t, dedupe_labels,
                        mask_dirichlet_rows)


@dataclass
class FEM2D_P2:
    """Discretization descriptor (reference: src/fem2d_P2.jl:31-34)."""

    bubble: bool
    K: np.ndarray       # (3, N, 2) corner mesh
    Kfull: np.ndarray   # (V, N, 2) full node mesh

    dim = 2


# ---------------------------------------------------------------------------
# reference element
# ---------------------------------------------------------------------------

def _bary_nodes(bubble: bool):
    h = Fraction(1, 2)
    t = Fraction(1, 3)
    nodes = [(1, 0, 0), (h, h, 0), (0, 1, 0), (0, h, h), (0, 0, 1), (h, 0, h)]
    if bubble:
        nodes.append((t, t, t))
    return [tuple(Fraction(v) for v in nd) for nd in nodes]


def _poly_mul(a, b):
    out = {}
    for (i, j), ca in a.items():
        for (k, l), cb in b.items():
            out[(i + k, j + l)] = out.get((i + k, j + l), 0) + ca * cb
    return out


def _poly_eval(p, x, y):
    return sum(c * x ** i * y ** j for (i, j), c in p.items())


def _poly_diff(p, axis):
    out = {}
    for (i, j), c in p.items():
        if axis == 0 and i > 0:
            out[(i - 1, j)] = out.get((i - 1, j), 0) + c * i
        if axis == 1 and j > 0:
            out[(i, j - 1)] = out.get((i, j - 1), 0) + c * j
    return out


def _poly_int(p):
    """Integral over the unit reference triangle {x, y >= 0, x + y <= 1}."""
    return sum(c * Fraction(factorial(i) * factorial(j), factorial(i + j + 2)) for (i, j), c in p.items())


def _solve_exact(A, B):
    n = len(A)
    M = [list(A[i]) + list(B[i]) for i in range(n)]
    for c in range(n):
        piv = next(r for r in range(c, n) if M[r][c] != 0)
        M[c], M[piv] = M[piv], M[c]
        inv = 1 / M[c][c]
        M[c] = [v * inv for v in M[c]]
        for r in range(n):
            if r != c and M[r][c] != 0:
                f = M[r][c]
                M[r] = [a - f * b for a, b in zip(M[r], M[c])]
    return [row[n:] for row in M]


_REF_CACHE: Dict[bool, dict] = {}


def reference_triangle(bubble: bool = True) -> dict:
    """K (V x 3), w (V), dx, dy (V x V) of the reference element, as float64 arrays
    (reference: src/fem2d_P2.jl:74-96 and :109-128 tabulate the same quantities)."""
    if bubble in _REF_CACHE:
        return _REF_CACHE[bubble]
    nodes = _bary_nodes(bubble)
    V = len(nodes)
    one = Fraction(1)
    # reference coordinates: x = l1, y = l2, l3 = 1 - x - y
    l1 = {(1, 0): one}
    l2 = {(0, 1): one}
    l3 = {(0, 0): one, (1, 0): -one, (0, 1): -one}
    mons = [{(0, 0): one}, l1, l2, _poly_mul(l1, l1), _poly_mul(l1, l2), _poly_mul(l2, l2)]
    if bubble:
        mons.append(_poly_mul(_poly_mul(l1, l2), l3))
    Vand = [[_poly_eval(m, nd[0], nd[1]) for m in mons] for nd in nodes]
    eye = [[one if i == j else 0 * one for j in range(V)] for i in range(V)]
    coef = _solve_exact(Vand, eye)            # coef[m][j]: coefficient of monomial m in phi_j
    basis = []
    for j in range(V):
        pj = {}
        for m in range(V):
            for key, c in mons[m].items():
                pj[key] = pj.get(key, 0) + coef[m][j] * c
        basis.append(pj)
    dx = np.array([[float(_poly_eval(_poly_diff(basis[j], 0), nd[0], nd[1])) for j in range(V)] for nd in nodes])
    dy = np.array([[float(_poly_eval(_poly_diff(basis[j], 1), nd[0], nd[1])) for j in range(V)] for nd in nodes])
    w = np.array([float(2 * _poly_int(basis[j])) for j in range(V)])
    K = np.array([[float(v) for v in nd] for nd in nodes])
    _REF_CACHE[bubble] = dict(K=K, w=w, dx=dx, dy=dy)
    return _REF_CACHE[bubble]


# child corner slots of the four red-refinement children, as parent local slots
# (reference: src/fem2d_P2.jl:183-184: (ca,a,ab), (ab,b,bc), (bc,c,ca), (ab,bc,ca))
_CHILD_CORNERS = ((5, 0, 1), (1, 2, 3), (3, 4, 5), (1, 3, 5))


def _first_occurrence_rank(keys: np.ndarray) -> Tuple[np.ndarray, int]:
    """0-based id of every key, ids handed out in order of first occurrence; also the number of distinct keys."""
    _, first, inv = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.size)
    return rank[inv], int(order.size)


def _refine_p2_connectivity(t: np.ndarray) -> np.ndarray:
    """Red-refine full P2(+bubble) connectivity (reference: src/fem2d_P2.jl:169-207).
    Node ids are renumbered by first occurrence, child-edge nodes keyed by their
    (sorted) endpoint pair, bubbles element-local: the reference hands out the new ids child by child -- the
    three edge nodes a child meets first, then its bubble -- which is the first-occurrence rank of the event
    sequence (edge, edge, edge, bubble) per child."""
    V, N = t.shape
    ids, n0 = _first_occurrence_rank(np.ascontiguousarray(t[:6].T).reshape(-1))
    ids = ids.reshape(N, 6)
    corners = ids[:, np.array(_CHILD_CORNERS)].reshape(4 * N, 3)       # child j = 4 e + s
    nxt = corners[:, [1, 2, 0]]
    events = np.minimum(corners, nxt).astype(np.int64) * n0 + np.maximum(corners, nxt)
    if V == 7:
        events = np.concatenate([events, (np.int64(n0) * n0 + np.arange(4 * N, dtype=np.int64))[:, None]], axis=1)
    new_ids, _ = _first_occurrence_rank(events.reshape(-1))
    new_ids = n0 + new_ids.reshape(4 * N, -1)
    out = np.empty((V, 4 * N), dtype=np.int64)
    out[0], out[2], out[4] = corners[:, 0], corners[:, 1], corners[:, 2]
    out[1], out[3], out[5] = new_ids[:, 0], new_ids[:, 1], new_ids[:, 2]
    if V == 7:
        out[6] = new_ids[:, 3]
    return out


def _default_Kfull(bubble: bool) -> np.ndarray:
    """reference: src/fem2d_P2.jl:210-217 (two triangles on [-1,1]^2)."""
    R = reference_triangle(bubble)
    corners = np.array([[-1.0, -1], [1, -1], [-1, 1], [1, -1], [1, 1], [-1, 1]]).reshape(2, 3, 2)
    Kf = np.einsum("vc,ecd->ved", R["K"], corners)   # (V, N, 2)
    return Kf


def _extract_corner_mesh(Kfull: np.ndarray) -> np.ndarray:
    return Kfull[[0, 2, 4], :, :].copy()


def _build_geometry(Kfull: np.ndarray, t: np.ndarray) -> Geometry:
    """Fine-level isoparametric operators and weights (reference: src/fem2d_P2.jl:518-596)."""
    p, N, _ = Kfull.shape
    bubble = p == 7
    R = reference_triangle(bubble)
    Rdx, Rdy, Rw = R["dx"], R["dy"], R["w"]
    X = Kfull[:, :, 0]            # (p, N)
    Y = Kfull[:, :, 1]
    x_xi, x_eta = Rdx @ X, Rdy @ X
    y_xi, y_eta = Rdx @ Y, Rdy @ Y
    detJ = x_xi * y_eta - x_eta * y_xi      # (p, N) at node j of element k
    if not np.all(detJ > 0):
        bad = np.argwhere(detJ.T <= 0)
        raise ValueError(f"fem2d_P2: non-positive Jacobian at {len(bad)} node(s); "
                         "supply orientation-preserving, non-self-intersecting elements")
    invdet = 1.0 / detJ
    # dx_block[j, m, k] = ( y_eta[j,k] Rdx[j,m] - y_xi[j,k] Rdy[j,m]) / detJ[j,k], formed as [k, m, j] arrays whose transposed
    # view IS the (p, p, N) Fortran-order image BlockDiag stores (the same products and differences, no 51 MB reorder per operator)
    RdxT, RdyT = np.ascontiguousarray(Rdx.T)[None, :, :], np.ascontiguousarray(Rdy.T)[None, :, :]      # [1, m, j]
    cf = lambda a: np.ascontiguousarray(a.T)[:, None, :]                                                # [k, 1, j]
    dxb = (cf(y_eta * invdet) * RdxT - cf(y_xi * invdet) * RdyT).transpose(2, 1, 0)
    dyb = (cf(-x_eta * invdet) * RdxT + cf(x_xi * invdet) * RdyT).transpose(2, 1, 0)
    idb = np.broadcast_to(np.eye(p)[None, :, :], (N, p, p)).copy().transpose(2, 1, 0)
    w = (detJ * Rw[:, None]).T.reshape(-1)
    ops = {"id": BlockDiag(idb), "dx": BlockDiag(dxb), "dy": BlockDiag(dyb)}
    disc = FEM2D_P2(bubble, _extract_corner_mesh(Kfull), Kfull)
    return Geometry(disc, np.asarray(t, dtype=np.int64), Kfull, w, ops)


def fem2d_P2(bubble: bool | None = None, K: np.ndarray | None = None, t: np.ndarray | None = None) -> Geometry:
    """Single-level P2(+bubble) geometry (reference: `fem2d_P2`, src/fem2d_P2.jl:262-277)."""
    b = (K is None or K.shape[0] == 7) if bubble is None else bubble
    Kf = _default_Kfull(b) if K is None else np.asarray(K, dtype=np.float64)
    V = 7 if b else 6
    if Kf.shape[0] != V:
        raise ValueError(f"K must have {V} vertices per triangle for bubble={b}")
    if Kf.shape[2] != 2:
        raise ValueError("K must have spatial dim 2")
    if t is None:
        flat = Kf.transpose(1, 0, 2).reshape(-1, 2)
        t = dedupe_labels(flat).reshape(Kf.shape[1], V).T
    return _build_geometry(Kf, t)


def subdivide(geom: Geometry, L: int) -> Geometry:
    """`subdivide(geom, L)`: L-1 red refinements, fine geometry only (reference:
    src/multigrid.jl:472 -> src/fem2d_P2.jl:468-596).

    Child node coordinates are the parent element map evaluated at the child nodes.
    For the straight-sided elements the package builds this is the affine image of
    the child corners, identical (up to roundoff) to the reference's `refine * x`;
    curved parents would need the reference's bubble-distribution table and are refused.
    """
    if not isinstance(geom.discretization, FEM2D_P2):
        raise TypeError("subdivide: FEM2D_P2 geometry expected")
    if L < 1:
        raise ValueError("L must be >= 1")
    Kf, t = geom.x, geom.t
    p = Kf.shape[0]
    RK = reference_triangle(p == 7)["K"]
    straight = np.einsum("vc,ced->ved", RK, Kf[[0, 2, 4], :, :])
    if not np.allclose(straight, Kf, rtol=0, atol=1e-13 * max(1.0, np.abs(Kf).max())):
        raise NotImplementedError("subdivide: curved (isoparametric) P2 elements are not supported")
    for _ in range(L - 1):
        N = Kf.shape[1]
        six = Kf[:6]                                       # (6, N, 2)
        cc = np.array(_CHILD_CORNERS)                      # (4, 3)
        child_corners = six[cc]                            # (4, 3, N, 2)
        child_corners = child_corners.transpose(1, 2, 0, 3).reshape(3, 4 * N, 2)  # element-major, child fastest
        Kf = np.einsum("vc,ced->ved", RK, child_corners)
        t = _refine_p2_connectivity(t)
    return _build_geometry(Kf, t)


# ---------------------------------------------------------------------------
# boundary + subspaces
# ---------------------------------------------------------------------------

def _p2_boundary_dedup_set(labels: np.ndarray, N: int) -> set:
    """reference: src/fem2d_P2.jl:309-327 (half-edge use counts)."""
    V = labels.size // N
    t = labels.reshape(N, V)
    a = t[:, [0, 1, 2, 3, 4, 5]].reshape(-1)
    b = t[:, [1, 2, 3, 4, 5, 0]].reshape(-1)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    key = lo.astype(np.int64) * (int(labels.max()) + 1) + hi
    uniq, counts = np.unique(key, return_counts=True)
    once = uniq[counts == 1]
    M = int(labels.max()) + 1
    return set((once // M).tolist()) | set((once % M).tolist())


def find_boundary(geom: Geometry) -> List[Tuple[int, int]]:
    """(v, e) pairs (0-based) of P2 DOFs on the boundary (reference: src/fem2d_P2.jl:292-301)."""
    V, N = geom.t.shape
    labels = geom.labels
    bset = _p2_boundary_dedup_set(labels, N)
    isb = np.zeros(int(labels.max()) + 1, dtype=bool)
    isb[list(bset)] = True
    flat = np.nonzero(isb[labels])[0]
    return [(int(i % V), int(i // V)) for i in flat]


def _broken_p1_embedding(N: int, V: int) -> sp.csr_matrix:
    """reference: src/fem2d_P2.jl:355-380."""
    slot = np.array([[1, -1, 1], [1, 0, 0], [1, 1, -1], [0, 1, 0], [-1, 1, 1], [0, 0, 1]], dtype=float)
    if V == 7:
        slot = np.vstack([slot, np.full((1, 3), 1.0 / 3)])
    blk = sp.csr_matrix(slot)
    out = sp.kron(sp.identity(N, format="csr"), blk, format="csr")
    out.eliminate_zeros()
    return out


def _extract_corners_and_connectivity(t: np.ndarray, x_fine: np.ndarray):
    """reference: src/fem2d_P2.jl:608-624."""
    V, N = t.shape
    labels, n_v = corner_labels_from_t(t, (0, 2, 4))
    tri_conn = labels.reshape(N, 3)
    rows = (np.arange(N)[:, None] * V + np.array([0, 2, 4])[None, :]).reshape(-1)
    corners = np.zeros((n_v, x_fine.shape[1]))
    # first occurrence wins (any occurrence has the same coordinates up to roundoff)
    _, first = np.unique(labels, return_index=True)
    corners[labels[first]] = x_fine[rows[first]]
    return corners, tri_conn


def _assemble_p1_stiffness_full(corners: np.ndarray, tri_conn: np.ndarray) -> sp.csr_matrix:
    """reference: src/fem2d_P2.jl:646-669."""
    i1, i2, i3 = tri_conn[:, 0], tri_conn[:, 1], tri_conn[:, 2]
    x1, y1 = corners[i1, 0], corners[i1, 1]
    x2, y2 = corners[i2, 0], corners[i2, 1]
    x3, y3 = corners[i3, 0], corners[i3, 1]
    det2 = (x2 - x1) * (y3 - y1) - (x3 - x1) * (y2 - y1)
    bs = np.stack([y2 - y3, y3 - y1, y1 - y2], axis=1)
    cs = np.stack([x3 - x2, x1 - x3, x2 - x1], axis=1)
    s = 1.0 / (2 * np.abs(det2))
    vals = (bs[:, :, None] * bs[:, None, :] + cs[:, :, None] * cs[:, None, :]) * s[:, None, None]
    rows = np.repeat(tri_conn[:, :, None], 3, axis=2)
    cols = np.repeat(tri_conn[:, None, :], 3, axis=1)
    n_v = corners.shape[0]
    return sp.csr_matrix((vals.reshape(-1), (rows.reshape(-1), cols.reshape(-1))), shape=(n_v, n_v))


def _interior_corners_to_doubled_p2(tri_conn: np.ndarray, n_v: int, interior_corners: np.ndarray, V: int) -> sp.csr_matrix:
    """P1-corner -> broken P2(+bubble) bridge (reference: src/fem2d_P2.jl:675-708)."""
    interior_idx = -np.ones(n_v, dtype=np.int64)
    interior_idx[interior_corners] = np.arange(len(interior_corners))
    N = tri_conn.shape[0]
    ai, bi, ci = (interior_idx[tri_conn[:, k]] for k in range(3))
    base = V * np.arange(N)
    rows, cols, vals = [], [], []

    def push(slot, col, val):
        m = col >= 0
        rows.append(base[m] + slot)
        cols.append(col[m])
        vals.append(np.full(int(m.sum()), val))

    push(0, ai, 1.0); push(2, bi, 1.0); push(4, ci, 1.0)
    push(1, ai, 0.5); push(1, bi, 0.5)
    push(3, bi, 0.5); push(3, ci, 0.5)
    push(5, ci, 0.5); push(5, ai, 0.5)
    if V == 7:
        for col in (ai, bi, ci):
            push(6, col, 1.0 / 3)
    return sp.csr_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                         shape=(V * N, len(interior_corners)))


def _hierarchy(tri_conn, K_full, interior, n_v, n_doubled, prolongator, V):
    """reference: src/fem2d_P2.jl:388-398."""
    interior = np.asarray(interior, dtype=np.int64)
    K_loc = sp.csr_matrix(K_full)[interior][:, interior]
    P_amg = amg_prolongations(K_loc, prolongator)
    bridge = _interior_corners_to_doubled_p2(tri_conn, n_v, interior, V)
    return assemble_amg_ladder(P_amg, bridge, n_doubled)


def amg(geom: Geometry, prolongator=None, dirichlet_nodes: Dict[str, List[Tuple[int, int]]] | None = None) -> MultiGrid:
    """AMG hierarchy on the continuous corners (reference: src/fem2d_P2.jl:400-455)."""
    if prolongator is None:
        prolongator = amg_ruge_stuben(max_coarse=2)
    if dirichlet_nodes is None:
        dirichlet_nodes = {"dirichlet": find_boundary(geom)}
    x_fine = geom.xflat
    V, N = geom.t.shape
    n_doubled = V * N
    full_labels = geom.labels
    n_full_unique = int(full_labels.max()) + 1
    corners, tri_conn = _extract_corners_and_connectivity(geom.t, x_fine)
    n_v = corners.shape[0]
    full_to_corner = -np.ones(n_full_unique, dtype=np.int64)
    rows = (np.arange(N)[:, None] * V + np.array([0, 2, 4])[None, :]).reshape(-1)
    full_to_corner[full_labels[rows]] = tri_conn.reshape(-1)
    K_full = _assemble_p1_stiffness_full(corners, tri_conn)
    refine_full, sizes_full, L_full, K_amg_full = _hierarchy(
        tri_conn, K_full, np.arange(n_v), n_v, n_doubled, prolongator, V)

    def build_dirichlet(nodes):
        lin = np.array([v + e * V for (v, e) in nodes], dtype=np.int64)
        dd_set = set(full_labels[lin].tolist())
        dcorner = {int(full_to_corner[f]) for f in dd_set if full_to_corner[f] >= 0}
        interior = np.array(sorted(set(range(n_v)) - dcorner), dtype=np.int64)
        refine_dir, sizes_dir, L_dir, K_amg_dir = _hierarchy(
            tri_conn, K_full, interior, n_v, n_doubled, prolongator, V)
        refine_dir[K_amg_dir - 1] = mask_dirichlet_rows(refine_dir[K_amg_dir - 1], full_labels, dd_set)
        sub = [sp.identity(sizes_dir[kk], format="csr") for kk in range(K_amg_dir)] + [None]
        sub[L_dir - 1] = continuous_subspace(full_labels, n_full_unique, dd_set)
        return refine_dir, sub

    return assemble_amg_dicts(geom, n_doubled, dirichlet_nodes, refine_full, sizes_full, L_full,
                              K_amg_full, build_dirichlet,
                              full_riders={"broken_P1": _broken_p1_embedding(N, V)})


# ---------------------------------------------------------------------------
# geometric_mg (reference: src/fem2d_P2.jl:468-596)
# ---------------------------------------------------------------------------

# The reference's 28 x 7 child-interpolation table `reference_triangle(...).refine` x 648
# (src/fem2d_P2.jl:97), as data: (row, value) lists per parent basis function, 1-based rows
# = 7*(child-1) + child node.  It is the parent's P2+bubble basis evaluated at the child nodes
# EXCEPT for the six rows of child mid-edge nodes that lie inside the parent, where the reference
# carries integer-rounded numerators (61, 80, -20, -82, 549 for the exact 60.75, 81, -20.25, -81,
# 546.75; the rows still sum to one and still reproduce linears).  tests/test_geometric_mg.py checks both facts.
_REFINE_BUBBLE_648 = (
    ((2, 243), (3, 648), (4, 243), (6, 61), (7, 180), (9, -81), (13, -20), (14, -36), (18, -81), (20, -20), (21, -36), (23, -20), (25, -20), (27, 61)),
    ((4, 486), (5, 648), (6, 80), (7, 144), (8, 648), (9, 486), (13, 80), (14, 144), (20, -82), (21, -72), (22, 648), (23, 80), (25, -82), (27, 80)),
    ((4, -81), (6, -20), (7, -36), (9, 243), (10, 648), (11, 243), (13, 61), (14, 180), (16, -81), (20, -20), (21, -36), (23, 61), (25, -20), (27, -20)),
    ((6, -82), (7, -72), (11, 486), (12, 648), (13, 80), (14, 144), (15, 648), (16, 486), (20, 80), (21, 144), (23, 80), (24, 648), (25, 80), (27, -82)),
    ((2, -81), (6, -20), (7, -36), (11, -81), (13, -20), (14, -36), (16, 243), (17, 648), (18, 243), (20, 61), (21, 180), (23, -20), (25, 61), (27, -20)),
    ((1, 648), (2, 486), (6, 80), (7, 144), (13, -82), (14, -72), (18, 486), (19, 648), (20, 80), (21, 144), (23, -82), (25, 80), (26, 648), (27, 80)),
    ((6, 549), (7, 324), (13, 549), (14, 324), (20, 549), (21, 324), (23, 549), (25, 549), (27, 549), (28, 648)),
)


def refine_table(bubble: bool = True) -> np.ndarray:
    """(4V x V) child-interpolation table.  bubble: the reference's table (see above); pure P2: the
    P2 basis evaluated at the child nodes (exact; reference: src/fem2d_P2.jl:129-131)."""
    if bubble:
        T = np.zeros((28, 7))
        for j, col in enumerate(_REFINE_BUBBLE_648):
            for (r, v) in col:
                T[r - 1, j] = v / 648.0
        return T
    # P2: phi_i at the child nodes, barycentric closed forms l(2l-1) / 4 l_a l_b
    nodes = _bary_nodes(False)
    T = np.zeros((24, 6))
    for s_, cc in enumerate(_CHILD_CORNERS):
        corners = [nodes[c] for c in cc]
        for v, nd in enumerate(nodes):
            lam = [sum(nd[c] * corners[c][k] for c in range(3)) for k in range(3)]
            vals = [lam[0] * (2 * lam[0] - 1), 4 * lam[0] * lam[1], lam[1] * (2 * lam[1] - 1), 4 * lam[1] * lam[2],
                    lam[2] * (2 * lam[2] - 1), 4 * lam[2] * lam[0]]
            T[6 * s_ + v] = [float(x) for x in vals]
    return T


def continuous(t: np.ndarray) -> sp.csr_matrix:
    """reference: src/fem2d_P2.jl:159-163 (zero-trace continuous space of one level)."""
    labels = t.T.reshape(-1)
    bdry = _p2_boundary_dedup_set(labels, t.shape[1])
    return continuous_subspace(labels, int(labels.max()) + 1, bdry)


def geometric_mg(geom: Geometry, L: int) -> MultiGrid:
    """`geometric_mg(geom, L)`: L levels of red refinement with the element-local transfer table
    (reference: src/fem2d_P2.jl:468-596).  Per level: continuous zero-trace P2(+bubble), broken identity,
    constants and the broken-P1 rider."""
    from .multigrid import make_multigrid
    from .tensorfem import _vblock_refine
    if not isinstance(geom.discretization, FEM2D_P2):
        raise TypeError("geometric_mg: FEM2D_P2 geometry expected")
    if L < 1:
        raise ValueError("L must be >= 1")
    p = geom.x.shape[0]
    T = refine_table(p == 7)
    X, t = geom.x, geom.t
    topo, sizes = [t], [X.shape[1]]
    for _ in range(L - 1):
        t = _refine_p2_connectivity(t)
        topo.append(t)
        sizes.append(4 * sizes[-1])
    refine = [_vblock_refine(T, p, 4, sizes[l]) for l in range(L - 1)]
    refine.append(sp.identity(p * sizes[-1], format="csr"))
    # fine coordinates: x[l+1] = refine[l] * x[l] like the reference (src/fem2d_P2.jl:513)
    xf = X.transpose(1, 0, 2).reshape(-1, X.shape[2])
    for l in range(L - 1):
        xf = refine[l] @ xf
    Kfine = xf.reshape(sizes[-1], p, X.shape[2]).transpose(1, 0, 2)
    geomL = _build_geometry(np.ascontiguousarray(Kfine), topo[-1])
    subspaces = {"dirichlet": [], "full": [], "uniform": [], "broken_P1": []}
    for l in range(L):
        nl = p * sizes[l]
        subspaces["dirichlet"].append(continuous(topo[l]))
        subspaces["full"].append(sp.identity(nl, format="csr"))
        subspaces["uniform"].append(sp.csr_matrix(np.ones((nl, 1))))
        subspaces["broken_P1"].append(_broken_p1_embedding(sizes[l], p))
    return make_multigrid(geomL, subspaces, refine)
