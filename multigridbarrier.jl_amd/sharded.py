"""Element-sharded evaluate / assemble across ranks.

The broken basis duplicates shared nodes per element, so the operator blocks, weights, cone grids,
cost grid and the iterate `z0` are element-local: a rank that owns a contiguous element range
`[e0, e1)` (spatially compact, because `subdivide` keeps the children of a coarse element contiguous,
reference: src/fem2d_P2.jl:181-204, src/TensorFEM.jl:905-910) needs no halo on the input side.  The
only coupling is through `R`:

  * the level-J coefficients `s` are replicated (10 MB at L = 9),
  * `f0`      = sum over ranks of the local partial sums              -> one scalar all-reduce,
  * `f1`      = sum over ranks of  R_loc' * (local element gradient)   -> only the entries of DoFs whose
                support crosses a rank boundary (the interface) differ from zero on more than one rank,
  * `R' H R`  = sum over ranks of  R_loc' * H_blk,loc * R_loc          -> likewise only interface rows.

`ShardedBarrier` evaluates the three closures of the reference's `Barrier` (src/convex.jl:155-202) that
way: every rank runs the ordinary single-GPU code on its slice of the problem (a `DeviceProblem` built
from `slice_problem`, or any object with the same f0/f1/f2 methods), then interface entries are summed
with ONE `all_reduce` on a compact buffer; the batched scalar all-reduce of a Newton iteration
(f0, <g, n>, |g|^2, finite flag) rides in the same call when the caller passes `extra`.

The 1/n of the flat barrier average is global, so the local problems carry
`barrier_weights = 1/n_global` (the masked-barrier path, src/convex.jl:213-257) -- no kernel changes.

What does NOT shard is the reference's direct solve: the factorization needs the whole `H`
(`gather_hessian`).  DESIGN.md section 7 has the traffic numbers and the subtree-to-rank design
that would shard it.
"""
from __future__ import annotations

from dataclasses import dataclass, replace
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np
import scipy.sparse as sp

from .blockmatrices import BlockDiag, block_column
from .convex import Convex, Piece
from .multigrid import AMG, Geometry
from .problem import MGBProblem


def element_partition(N: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous element ranges, sizes differing by at most one."""
    base, rem = divmod(int(N), int(world))
    out, e = [], 0
    for r in range(world):
        cnt = base + (1 if r < rem else 0)
        out.append((e, e + cnt))
        e += cnt
    return out


def _rows_of(nodes: np.ndarray, n: int, nu: int) -> np.ndarray:
    return np.concatenate([a * n + nodes for a in range(nu)])


def slice_amg(M: AMG, e0: int, e1: int) -> AMG:
    """The AMG restricted to elements [e0, e1): operator blocks, weights, coordinates and the ROWS of
    every prolongation; the column spaces (level-J coefficients) stay global."""
    geom = M.geometry
    first = M.D_fine[0]
    p, N = first.active_block.p, first.active_block.N
    n, nu = p * N, first.nu
    nodes = np.arange(p * e0, p * e1)
    ops = {k: BlockDiag(v.data[:, :, e0:e1]) for k, v in geom.operators.items()}
    g2 = Geometry(discretization=geom.discretization, t=geom.t[:, e0:e1], x=geom.x[:, e0:e1], w=geom.w[nodes], operators=ops)
    rows = _rows_of(nodes, n, nu)
    R_loc = [sp.csr_matrix(sp.csr_matrix(R)[rows]) for R in M.R_fine]
    D_loc = [block_column(ops[name], state, nu) for (state, name) in M.D_spec]
    return AMG(geometry=g2, x=M.x[nodes], w=M.w[nodes], R_fine=R_loc, D_fine=D_loc, state_names=M.state_names, D_spec=M.D_spec)


def slice_convex(Q: Convex, nodes: np.ndarray) -> Convex:
    pieces = [Piece(pc.kind, pc.idx, pc.A[nodes], pc.b[nodes], None if pc.p is None else pc.p[nodes],
                    None if pc.mu is None else pc.mu[nodes], pc.colon) for pc in Q.pieces]
    return Convex(pieces, None if Q.select is None else Q.select[nodes])


def slice_problem(prob: MGBProblem, e0: int, e1: int) -> MGBProblem:
    p = prob.M[0].D_fine[0].active_block.p
    nodes = np.arange(p * e0, p * e1)
    M = tuple(slice_amg(Mk, e0, e1) for Mk in prob.M)
    return MGBProblem(M, prob.f[nodes], prob.g[nodes], slice_convex(prob.Q, nodes), M[0].geometry)


@dataclass
class _LevelPlan:
    m: int
    iface: np.ndarray                 # DoFs whose support meets more than one rank (sorted)
    rowptr: np.ndarray                # global CSR pattern of R'HR (union of the ranks' patterns)
    colidx: np.ndarray
    loc2glob: np.ndarray              # position of every local structural nonzero in the global pattern
    shared: np.ndarray                # global nnz positions touched by more than one rank (sorted)
    shared_of_local: np.ndarray       # for local nnz in `shared`: (local position, slot in the shared buffer)


class ShardedBarrier:
    """The Barrier closures on an element-sharded problem.

    `local` is this rank's evaluator on its slice: f0(level, s, c_loc, z0_loc) -> float,
    f1(...) -> (m_J,) array, f2(...) -> scipy CSR (m_J x m_J) holding R_loc' H_loc R_loc.
    `dist` is `torch.distributed` (initialised) or None for a single rank; `device` the torch device
    of the reduction buffers ("cpu" with gloo, "cuda" with RCCL)."""

    def __init__(self, prob: MGBProblem, rank: int, world: int, local, dist=None, device: str = "cpu", which: int = 0):
        self.rank, self.world, self.dist, self.device = rank, world, dist, device
        M = prob.M[which]
        first = M.D_fine[0]
        self.p, self.N, self.nu = first.active_block.p, first.active_block.N, first.nu
        self.n = self.p * self.N
        self.parts = element_partition(self.N, world)
        self.e0, self.e1 = self.parts[rank]
        self.nodes = np.arange(self.p * self.e0, self.p * self.e1)
        self.local = local
        self.M = M
        self._plans: dict = {}
        self.bytes_reduced = 0          # payload of the data-path collectives so far (diagnostics)

    # ---- slices of the per-node inputs ------------------------------------------------------------
    def c_local(self, c: np.ndarray) -> np.ndarray:
        return np.asarray(c)[self.nodes]

    def z_local(self, z0: np.ndarray) -> np.ndarray:
        return np.asarray(z0)[_rows_of(self.nodes, self.n, self.nu)]

    # ---- collectives ---------------------------------------------------------------------------------
    def _allreduce(self, buf: np.ndarray) -> np.ndarray:
        if self.dist is None or self.world == 1:
            return buf
        import torch
        t = torch.from_numpy(np.ascontiguousarray(buf, dtype=np.float64)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        self.bytes_reduced += t.numel() * 8
        return t.cpu().numpy()

    def _gather_objects(self, obj):
        if self.dist is None or self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj)
        return out

    # ---- one-time plans per level (pattern algebra only, no values) -----------------------------
    def plan(self, level: int, H_loc: Optional[sp.csr_matrix] = None) -> _LevelPlan:
        if level in self._plans and (H_loc is None or self._plans[level].rowptr is not None):
            return self._plans[level]
        R = sp.csr_matrix(self.M.R_fine[level])
        m = R.shape[1]
        # interface DoFs: columns of R with rows in more than one rank's node range
        rows, cols = R.nonzero()
        owner = np.searchsorted(np.array([self.p * e1 for (_, e1) in self.parts]), rows % self.n, side="right")
        touch = sp.csr_matrix((np.ones(rows.size), (cols, owner)), shape=(m, self.world))
        iface = np.flatnonzero(touch.getnnz(axis=1) > 1)
        pl = _LevelPlan(m=m, iface=iface, rowptr=None, colidx=None, loc2glob=None, shared=None, shared_of_local=None)
        if H_loc is not None:
            H_loc = sp.csr_matrix(H_loc)
            H_loc.sort_indices()
            key_loc = H_loc.indptr, H_loc.indices
            keys = np.repeat(np.arange(m, dtype=np.int64), np.diff(H_loc.indptr)) * m + H_loc.indices
            all_keys = self._gather_objects(keys)
            uni, counts = np.unique(np.concatenate(all_keys), return_counts=True)
            pl.rowptr = np.concatenate([[0], np.cumsum(np.bincount(uni // m, minlength=m))]).astype(np.int64)
            pl.colidx = (uni % m).astype(np.int64)
            pl.loc2glob = np.searchsorted(uni, keys)
            pl.shared = np.flatnonzero(counts > 1)
            slot = np.searchsorted(pl.shared, pl.loc2glob)
            hit = (slot < pl.shared.size) & (pl.shared[np.minimum(slot, pl.shared.size - 1)] == pl.loc2glob) if pl.shared.size else np.zeros(keys.size, bool)
            pl.shared_of_local = np.stack([np.flatnonzero(hit), slot[hit]], axis=0)
        self._plans[level] = pl
        return pl

    # ---- the closures -------------------------------------------------------------------------------
    def f0(self, level: int, s, c, z0, extra: Sequence[float] = ()) -> Any:
        """Global objective; `extra` scalars are summed over ranks in the same all-reduce."""
        y = self.local.f0(level, s, self.c_local(c), self.z_local(z0))
        out = self._allreduce(np.array([y, *extra], dtype=np.float64))
        return float(out[0]) if not extra else (float(out[0]), out[1:])

    def f1(self, level: int, s, c, z0) -> np.ndarray:
        """Global gradient: interior entries are already complete on their owner and zero elsewhere, so only
        the interface entries travel; every rank returns the full vector (s is replicated)."""
        g = np.asarray(self.local.f1(level, s, self.c_local(c), self.z_local(z0)), dtype=np.float64)
        if self.world == 1:
            return g
        # replicated result: the sum over ranks is a gather for interior entries (one rank holds a non-zero) and the
        # interface sum for the others.  The interface-only form is `f1_interface_only`; the domain-decomposed solver
        # below never replicates a gradient at all.
        return self._allreduce(g)

    def f1_interface_only(self, level: int, s, c, z0) -> Tuple[np.ndarray, np.ndarray]:
        """(local gradient with interface entries summed, interface index list): the distributed form in
        which a sharded solver would consume it -- payload = |interface| doubles."""
        g = np.asarray(self.local.f1(level, s, self.c_local(c), self.z_local(z0)), dtype=np.float64).copy()
        pl = self.plan(level)
        g[pl.iface] = self._allreduce(g[pl.iface])
        return g, pl.iface

    def f2(self, level: int, s, c, z0) -> Tuple[np.ndarray, _LevelPlan]:
        """Values of this rank's part of R'HR on the GLOBAL pattern, with the entries that several ranks
        contribute to (interface rows) already summed -- payload = |shared| doubles.  Entries no other
        rank touches stay where they were computed; `gather_hessian` replicates them when a direct solve
        needs the whole matrix."""
        H_loc = sp.csr_matrix(self.local.f2(level, s, self.c_local(c), self.z_local(z0)))
        H_loc.sort_indices()
        pl = self.plan(level, H_loc)
        vals = np.zeros(pl.colidx.size)
        vals[pl.loc2glob] = H_loc.data
        if self.world > 1 and pl.shared.size:
            buf = np.zeros(pl.shared.size)
            buf[pl.shared_of_local[1]] = H_loc.data[pl.shared_of_local[0]]
            vals[pl.shared] = self._allreduce(buf)
        return vals, pl

    def gather_hessian(self, vals: np.ndarray, pl: _LevelPlan) -> sp.csr_matrix:
        """Replicate the whole matrix (what the reference's direct solve needs): shared entries are final
        on every rank already; the others are non-zero on exactly one rank, so a sum over ranks of the
        non-shared part is a gather."""
        if self.world > 1:
            own = vals.copy()
            own[pl.shared] = 0.0
            tot = self._allreduce(own)
            tot[pl.shared] = vals[pl.shared]
            vals = tot
        return sp.csr_matrix((vals, pl.colidx, pl.rowptr), shape=(pl.m, pl.m))


class DeviceLocalEvaluator:
    """This rank's slice on its GPU: an ordinary `DeviceProblem` whose flat barrier average uses the
    GLOBAL node count (barrier_weights = 1/n_global, or the slice of the caller's weights)."""

    def __init__(self, prob: MGBProblem, rank: int, world: int, device_id: int = 0, barrier_weights=None, which: int = 0):
        from .device import DeviceProblem, HipContext
        first = prob.M[which].D_fine[0]
        N, p = first.active_block.N, first.active_block.p
        e0, e1 = element_partition(N, world)[rank]
        nodes = np.arange(p * e0, p * e1)
        sub = slice_problem(prob, e0, e1)
        bw = np.full(nodes.size, 1.0 / (p * N)) if barrier_weights is None else np.asarray(barrier_weights)[nodes]
        self.ctx = HipContext(device_id)
        self.P = DeviceProblem(self.ctx, sub.M[which], sub.Q, barrier_weights=bw)
        self.f0, self.f1, self.f2 = self.P.f0, self.P.f1, self.P.f2

    def close(self):
        self.P.close()
        self.ctx.close()


# =====================================================================================================================
# Domain-decomposed Newton solve: one process per GPU (round 3)
# =====================================================================================================================
# Every rank keeps its contiguous element range and, per level J, the unknowns its elements touch (its interior I_r)
# plus ALL interface unknowns Gamma_J (those whose support meets more than one rank), in ascending global order.  The
# level-J coefficient vectors live on that local index set: interior entries exist on one rank, interface entries are
# replicated bit for bit.  Per Newton iteration the library (csrc/driver.cpp, problem.cpp, mf_numeric.hip) exchanges
#   * one small all-reduce of scalars per evaluation (f0, <g, n>, |g|^2, flags),
#   * |Gamma_J| doubles per gradient (the interface entries of R' v),
#   * the assembled interface front of the factorization, (|Gamma_J| + 1)^2 doubles: each rank eliminates its interior
#     unknowns (its subtree of the elimination tree), the Schur complements are summed, and every rank factors the
#     interface front redundantly -- the same numbers on every rank, so the replicated entries stay identical.
# The fine iterate z is never exchanged: each rank updates the rows of its own elements.


@dataclass
class LevelShard:
    cols: np.ndarray          # global column (unknown) ids of this rank's local unknowns, ascending
    iface: np.ndarray         # local positions of the interface unknowns, ascending
    own: np.ndarray           # 1.0 where this rank counts the unknown in dot products (interior: its one holder; interface: rank 0)


def _rank_of_rows(R: sp.csr_matrix, n: int, p: int, parts) -> np.ndarray:
    ends = np.array([p * e1 for (_, e1) in parts])
    return np.searchsorted(ends, np.arange(R.shape[0]) % n, side="right")


def shard_level(R, n: int, p: int, parts, rank: int) -> LevelShard:
    R = sp.csr_matrix(R)
    world = len(parts)
    rows, cols = R.nonzero()
    owner = _rank_of_rows(R, n, p, parts)[rows]
    touch = sp.csr_matrix((np.ones(rows.size), (cols, owner)), shape=(R.shape[1], world))
    touch.sum_duplicates()
    nt = touch.getnnz(axis=1)
    iface_glob = np.flatnonzero(nt > 1)
    mine = np.zeros(R.shape[1], dtype=bool)
    mine[touch[:, rank].nonzero()[0]] = True
    local = np.flatnonzero(mine | (nt > 1))
    is_if = np.isin(local, iface_glob, assume_unique=True)
    own = np.where(is_if, 1.0 if rank == 0 else 0.0, 1.0)
    return LevelShard(cols=local, iface=np.flatnonzero(is_if).astype(np.int32), own=own)


def shard_amg(M: AMG, rank: int, world: int):
    """This rank's slice of an AMG: element range, and per level the columns of R restricted to the local unknowns."""
    first = M.D_fine[0]
    p, N = first.active_block.p, first.active_block.N
    n = p * N
    parts = element_partition(N, world)
    e0, e1 = parts[rank]
    Mrows = slice_amg(M, e0, e1)
    shards, R_loc = [], []
    for R, Rr in zip(M.R_fine, Mrows.R_fine):
        sh = shard_level(R, n, p, parts, rank)
        shards.append(sh)
        R_loc.append(sp.csr_matrix(sp.csr_matrix(Rr)[:, sh.cols]))
    return replace(Mrows, R_fine=R_loc), shards


def shard_problem(prob: MGBProblem, rank: int, world: int):
    """(local MGBProblem, (shards of M[0], shards of M[1]), node index array of the rank)."""
    first = prob.M[0].D_fine[0]
    p, N = first.active_block.p, first.active_block.N
    e0, e1 = element_partition(N, world)[rank]
    nodes = np.arange(p * e0, p * e1)
    M0, s0 = shard_amg(prob.M[0], rank, world)
    M1, s1 = shard_amg(prob.M[1], rank, world)
    sub = MGBProblem((M0, M1), prob.f[nodes], prob.g[nodes], slice_convex(prob.Q, nodes), M0.geometry)
    return sub, (s0, s1), nodes


class _Reducer:
    """The reductions mgb_driver's host-side orchestration needs on a sharded problem (phase-I bookkeeping)."""

    def __init__(self, dist, device="cpu"):
        self.dist, self.device = dist, device

    def _r(self, x, op):
        import torch
        t = torch.tensor([float(x)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, x):
        return self._r(x, self.dist.ReduceOp.MAX)

    def all(self, flag: bool) -> bool:
        return self._r(1.0 if flag else 0.0, self.dist.ReduceOp.MIN) > 0.5


_ABORT_KEY = "mgbhip_sharded_abort"


def _store(dist):
    try:
        return dist.distributed_c10d._get_default_store()
    except Exception:                                   # noqa: BLE001 -- no store: fail-fast is unavailable, nothing else changes
        return None


def make_collective(dist, torch_device: str = "cpu", device_pointers: bool = False, device_id: int = 0):
    """The all-reduce the library calls (include/mgbhip.h: mgbhip_allreduce_fn) on top of torch.distributed.  Host
    buffers are wrapped in place; with `device_pointers` (RCCL) large buffers arrive as device pointers of the context's
    device `device_id` and are wrapped through __cuda_array_interface__ -- nothing crosses PCIe.

    Fail-fast: a rank that leaves its solve with a rank-local error (a HIP error, a failed MGB_REQUIRE between two
    collectives) sets a key in the process group's store (`ShardedSolver.solve_local`); a rank waiting in an all-reduce
    polls the asynchronous work object and looks at that key every 50 ms, so peers end with an error instead of
    blocking forever in a collective the failed rank will never enter."""
    import ctypes as C
    import time
    import torch
    from .device import ALLREDUCE_FN
    errors = []
    store = _store(dist)

    class _DevView:
        def __init__(self, ptr, count):
            self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}

    def _reduce(t, rop):
        work = dist.all_reduce(t, op=rop, async_op=True)
        t0 = time.monotonic()
        next_check = 0.05
        while not work.is_completed():
            waited = time.monotonic() - t0
            if waited > next_check:
                next_check = waited + 0.05
                if store is not None and store.check([_ABORT_KEY]):
                    raise RuntimeError("a peer rank left the sharded solve with an error")
                time.sleep(0.001)
        work.wait()

    def _cb(_user, buf, count, op, on_device):
        try:
            rop = dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM
            if on_device:
                with torch.cuda.device(device_id):
                    t = torch.as_tensor(_DevView(buf, count), device=f"cuda:{device_id}")
                    _reduce(t, rop)
                    torch.cuda.current_stream().synchronize()
            else:
                a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(int(count),))
                t = torch.from_numpy(a)
                if torch_device == "cpu":
                    _reduce(t, rop)
                else:                                  # RCCL reduces device tensors only
                    td = t.to(torch_device)
                    _reduce(td, rop)
                    t.copy_(td)
            return 0
        except BaseException as e:                     # noqa: BLE001 -- reported by the library as a failed collective
            errors.append(e)
            return 1
    thunk = ALLREDUCE_FN(_cb)
    thunk.errors = errors
    return thunk


class ShardedSolver:
    """One problem resident across all ranks of `dist` (torch.distributed, initialised): every rank constructs this with
    the SAME assembled problem and keeps its element range on its GPU; `solve()` is `mgb_solve` (reference semantics:
    src/mgb.jl:798-842; the partition: SURVEY.md section 8e) and returns the full solution on every rank."""

    def __init__(self, prob: MGBProblem, dist, device_id: int = 0, torch_device: str = "cpu", device_pointers: bool = False):
        from .device import DeviceMGBProblem
        self.prob, self.dist = prob, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        sub, shards, self.nodes = shard_problem(prob, self.rank, self.world)
        self.shards = shards
        self.coll = make_collective(dist, torch_device, device_pointers, device_id)
        self.D = DeviceMGBProblem(sub, device_id=device_id, shards=shards, collective=self.coll, accepts_device_ptr=device_pointers)
        n_glob = prob.M[0].w.size
        nz = prob.M[0].w != 0
        # the flat barrier averages are global (src/convex.jl:279-304) ...
        # ... and so is the node count in the default stopping rule lambda < 0.25 / sqrt(n) (src/mgb.jl:360): round 3 passed
        # the slice's count, a sqrt(world) looser tolerance -- 432 instead of 548 Newton iterations at L = 9 on two ranks
        self._shard = dict(reduce=_Reducer(dist, torch_device), bw_main=(nz.astype(np.float64) / nz.sum())[self.nodes],
                           bw_feas=np.full(self.nodes.size, 1.0 / n_glob), n_global=int(n_glob))

    # ---- user callables: every rank must take the same branch ------------------------------------------------------------
    def _collective_flags(self, res: bool, err: bool):
        import torch
        t = torch.tensor([1.0 if res else 0.0, 1.0 if err else 0.0], dtype=torch.float64, device=self._shard["reduce"].device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return bool(t[0].item() > 0.5), bool(t[1].item() > 0.5)

    def _wrap_stopping(self, fn):
        """`stopping_criterion(ymin, ynext, gmin, gnext, n, ndecmin, ndec)`: its arguments are global (already reduced)
        scalars, so every rank computes the same answer -- unless the callable raises or is not a pure function.  The
        answers are MAX-reduced together with an error flag: one raising rank ends the solve on all of them."""
        def stop(*a):
            res, exc = False, None
            try:
                res = bool(fn(*a))
            except BaseException as e:                 # noqa: BLE001 -- made collective below
                exc = e
            res, err = self._collective_flags(res, exc is not None)
            if err:
                raise exc if exc is not None else RuntimeError("stopping_criterion raised on a peer rank")
            return res
        return stop

    def _wrap_early_stop(self, fn):
        """`early_stop(z)` / `early_stop(z, t)` sees the WHOLE stacked iterate (gathered over ranks), like on one device."""
        import inspect
        two = len(inspect.signature(fn).parameters) >= 2
        nu = self.prob.g.shape[1]

        def early(z, t=None):
            parts = [None] * self.world
            self.dist.all_gather_object(parts, np.asarray(z).reshape(nu, -1))
            zfull = np.concatenate(parts, axis=1).reshape(-1)
            res, exc = False, None
            try:
                res = bool(fn(zfull, t) if two else fn(zfull))
            except BaseException as e:                 # noqa: BLE001
                exc = e
            res, err = self._collective_flags(res, exc is not None)
            if err:
                raise exc if exc is not None else RuntimeError("early_stop raised on a peer rank")
            return res
        return early if two else (lambda z: early(z))

    def solve_local(self, **kw):
        """The solve without the final gather: this rank's rows of z (timed loops use this)."""
        from .solve import MGBConvergenceFailure, mgb_driver
        if "barrier_nodes" in kw:
            raise ValueError("barrier_nodes is not supported on a domain-decomposed problem (the barrier weights of the "
                             "slices are fixed by the partition); solve on one device or drop the argument")
        if callable(kw.get("line_search")):
            raise ValueError("a custom line_search closure is not supported on a domain-decomposed problem: it would run "
                             "rank-local vector algebra on interface-replicated vectors; use ('backtracking', ...) or ('illinois', ...)")
        if callable(kw.get("stopping_criterion")):
            kw["stopping_criterion"] = self._wrap_stopping(kw["stopping_criterion"])
        if kw.get("early_stop") is not None:
            kw["early_stop"] = self._wrap_early_stop(kw["early_stop"])
        lines = []
        try:
            SOL = mgb_driver(self.D, printlog=lambda *a: lines.append("".join(str(x) for x in a)), _shard=self._shard, **kw)
        except MGBConvergenceFailure:
            raise                                      # decided on reduced scalars: every rank raises it
        except BaseException:
            st = _store(self.dist)                     # rank-local failure: tell the peers waiting in a collective
            if st is not None:
                try:
                    st.set(_ABORT_KEY, "1")
                except Exception:                      # noqa: BLE001
                    pass
            if self.coll.errors:
                raise self.coll.errors[0]
            raise
        SOL["log"] = "\n".join(lines)
        return SOL

    def solve(self, **kw):
        from .solve import MGBSOL
        SOL = self.solve_local(**kw)
        parts = [None] * self.world                  # ranks hold consecutive node ranges
        self.dist.all_gather_object(parts, SOL["z"])
        return MGBSOL(np.concatenate(parts, axis=0), SOL["SOL_feasibility"], SOL["SOL_main"], SOL["log"], self.prob.geometry)

    def close(self):
        self.D.close()


def sharded_mgb_solve(prob: MGBProblem, dist, device_id: int = 0, torch_device: str = "cpu",
                      device_pointers: bool = False, **kw):
    """`mgb_solve` of one problem across all ranks of `dist`: every rank calls this with the SAME assembled problem and
    gets the full solution."""
    S = ShardedSolver(prob, dist, device_id, torch_device, device_pointers)
    try:
        return S.solve(**kw)
    finally:
        S.close()
