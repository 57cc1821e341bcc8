"""Row (e): the domain-decomposed Newton solve, world_size 2 over gloo.

CPU part (-m "not gpu"): the decomposition itself -- local unknown sets, interface lists, ownership masks, local R --
and the algebra the device runs (each rank eliminates its interior, the interface Schur complements are summed, every
rank solves the interface system and back-substitutes) with the ORACLE as local evaluator: the assembled Newton direction
must be the single-rank one.
GPU part (-m gpu): `sharded_mgb_solve` on two processes sharing GPU 0 against the single-rank device solve -- same z
(1e-8, the reference's cross-backend bar) and the same Newton iteration counts."""
import os
import socket

import numpy as np
import pytest
import scipy.sparse as sp
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _problem(kind):
    import mgb_amd as m
    if kind == "fem2d_P2_L3":
        return m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 3)), p=1.5)
    if kind == "fem2d_P2_L5":
        return m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 5)), p=1.0)
    if kind == "fem2d_P2_L7":
        return m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 7)), p=1.0)
    if kind == "fem3d_L6_config4":
        return m.assemble(m.amg(m.subdivide(m.fem3d(k=1), 6), prolongator=m.amg_ruge_stuben(max_coarse=500)), p=4.0)
    if kind == "fem3d_L3":
        return m.assemble(m.amg(m.subdivide(m.fem3d(k=1), 3)), p=1.5)
    if kind == "phase1":                      # infeasible start: phase I, box, _matched_t hand-off across ranks
        prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 3)), p=1.5)
        prob.g[:, 1] = 0.5
        return prob
    raise ValueError(kind)


def _init(rank, world, port):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)


# ---------------------------------------------------------------------------------------------------------------------
# CPU: structure + Schur-complement Newton direction with the oracle
# ---------------------------------------------------------------------------------------------------------------------

def _cpu_worker(rank, world, port, kind, out):
    _init(rank, world, port)
    import torch
    from helpers import stacked
    from mgb_amd.sharded import shard_problem
    from oracle import mgb_oracle as O
    prob = _problem(kind)
    sub, shards, nodes = shard_problem(prob, rank, world)
    n_glob = prob.M[0].w.size
    Mo = O.OracleAMG(sub.M[0])
    B = O.Barrier(sub.Q, np.full(nodes.size, 1.0 / n_glob))
    z0, c = stacked(sub.g), 0.1 * sub.f
    res = {}
    rng = np.random.default_rng(3)
    for level in range(len(shards[0])):
        sh = shards[0][level]
        m_glob = prob.M[0].R_fine[level].shape[1]
        s_glob = 1e-3 * rng.standard_normal(m_glob)            # same stream on every rank
        s = s_glob[sh.cols]
        R = Mo.R_fine[level]
        g = B.f1(s, Mo.w, c, R, Mo.D_fine, z0)                 # this rank's partial sums on its local unknowns
        H = np.asarray(sp.csr_matrix(B.f2(s, Mo.w, c, R, Mo.D_fine, z0)).todense())
        y = B.f0(s, Mo.w, c, R, Mo.D_fine, z0)
        I = np.setdiff1d(np.arange(sh.cols.size), sh.iface)
        G = sh.iface
        # eliminate the interior, sum the interface Schur complements, solve, back-substitute
        if I.size:
            HII_inv_HIG = np.linalg.solve(H[np.ix_(I, I)], H[np.ix_(I, G)])
            HII_inv_g = np.linalg.solve(H[np.ix_(I, I)], g[I])
            S = H[np.ix_(G, G)] - H[np.ix_(G, I)] @ HII_inv_HIG
            r = g[G] - H[np.ix_(G, I)] @ HII_inv_g
        else:
            S, r = H[np.ix_(G, G)], g[G]
        buf = torch.from_numpy(np.concatenate([S.ravel(), r, [y]]))
        dist.all_reduce(buf)
        tot = buf.numpy()
        ng = G.size
        S, r, y = tot[:ng * ng].reshape(ng, ng), tot[ng * ng:ng * ng + ng], tot[-1]
        x = np.zeros(sh.cols.size)
        x[G] = np.linalg.solve(S, r)
        if I.size:
            x[I] = HII_inv_g - HII_inv_HIG @ x[G]
        res[level] = dict(cols=sh.cols, iface=sh.iface, own=sh.own, x=x, y=y, s=s_glob)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,world", [("fem2d_P2_L3", 2), ("fem3d_L3", 2), ("fem2d_P2_L3", 3), ("fem2d_P2_L3", 4)])
def test_domain_decomposition_reproduces_single_rank_newton_direction(kind, world):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import stacked
    from oracle import mgb_oracle as O
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_cpu_worker, args=(world, _free_port(), kind, out), nprocs=world, join=True)
        res = dict(out)
    prob = _problem(kind)
    Mo, B = O.OracleAMG(prob.M[0]), O.Barrier(prob.Q)
    z0, c = stacked(prob.g), 0.1 * prob.f
    for level, R in enumerate(Mo.R_fine):
        rk = [res[r][level] for r in range(world)]
        a = rk[0]
        m_J = R.shape[1]
        # structure: the local sets cover every unknown, interiors are disjoint, interfaces identical, one owner each
        cover = np.zeros(m_J, dtype=int)
        owners = np.zeros(m_J)
        gam = a["cols"][a["iface"]]
        for q in rk:
            assert np.array_equal(q["cols"][q["iface"]], gam)                  # every rank carries the WHOLE interface
            interior = np.setdiff1d(q["cols"], gam)
            cover[interior] += 1
            owners[q["cols"]] += q["own"]
        cover[gam] += 1
        assert np.array_equal(cover, np.ones(m_J, dtype=int)) and np.array_equal(owners, np.ones(m_J))
        # algebra: the assembled direction is the single-rank one
        s = a["s"]
        g = B.f1(s, Mo.w, c, R, Mo.D_fine, z0)
        H = sp.csc_matrix(B.f2(s, Mo.w, c, R, Mo.D_fine, z0))
        x_ref = O.solve_symmetric(H, g)
        x = np.zeros(m_J)
        for q in rk:
            x[q["cols"]] = q["x"]
            assert np.array_equal(q["x"][q["iface"]], a["x"][a["iface"]])     # replicated interface: bit for bit
        assert np.linalg.norm(x - x_ref) <= 1e-9 * np.linalg.norm(x_ref)
        assert abs(a["y"] - B.f0(s, Mo.w, c, R, Mo.D_fine, z0)) <= 1e-12 * abs(a["y"])
    if kind == "fem2d_P2_L3" and world == 2:
        assert res[0][len(Mo.R_fine) - 1]["iface"].size < 0.1 * Mo.R_fine[-1].shape[1]    # a mesh line


# ---------------------------------------------------------------------------------------------------------------------
# GPU: the complete sharded mgb_solve on two ranks sharing the device
# ---------------------------------------------------------------------------------------------------------------------

def _gpu_worker(rank, world, port, kind, out):
    _init(rank, world, port)
    from mgb_amd.sharded import sharded_mgb_solve
    prob = _problem(kind)
    sol = sharded_mgb_solve(prob, dist, device_id=0, torch_device="cpu")
    out[rank] = dict(z=sol.z, its=sol.SOL_main["its"], feas=None if sol.SOL_feasibility is None else sol.SOL_feasibility["its"],
                     t=sol.SOL_main["ts"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["fem2d_P2_L3", "fem2d_P2_L5", "fem3d_L3", "phase1"])
def test_sharded_mgb_solve_matches_single_rank_world2(kind):
    import mgb_amd as m
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_gpu_worker, args=(world, _free_port(), kind, out), nprocs=world, join=True)
        res = dict(out)
    ref = m.mgb_solve(_problem(kind))
    for rank in range(world):
        r = res[rank]
        assert np.abs(r["z"] - ref.z).max() < 1e-8                          # the reference's criterion (test/test_cuda.jl:51)
        a, b = np.asarray(r["its"]), np.asarray(ref.SOL_main["its"])
        # Newton counts: the sums run in another order (interface entries, Schur complements), and the stopping
        # rules compare quantities at rounding level -- a level solve may stop one iteration earlier or later
        # (seen: 8/12 instead of 9/13 iterations in the initial centring of the two coarsest levels; in the phase-I case
        # one t-step whose fine solve needs 8 iterations, the cap of max_newton, converges on one side and bisects the
        # levels on the other -- same t ramp, same z)
        assert a.shape == b.shape
        assert abs(int(a.sum()) - int(b.sum())) <= max(3, (0.10 if kind == "phase1" else 0.02) * b.sum())
        if kind != "phase1":
            assert np.abs(a - b).max() <= 3
        assert np.allclose(r["t"], ref.SOL_main["ts"])
        assert (r["feas"] is None) == (ref.SOL_feasibility is None)
    assert np.array_equal(res[0]["z"], res[1]["z"])                         # every rank returns the same solution


@pytest.mark.gpu
def test_sharded_mgb_solve_L7_same_z_and_iteration_counts_as_single_rank():
    """BASELINE configs[1]'s mesh (fem2d_P2 L = 7, 57 344 nodes), p = 1.0, two ranks sharing the device: the sharded solve
    must follow the single-rank trajectory -- same z (1e-8, the reference's cross-backend bar, test/test_cuda.jl:51) and
    the same Newton iteration count in EVERY level solve of every t-step (+-1: sums run in another order).  Round 3's
    rehearsal at L = 9 took 432 iterations where the single-rank solve takes 548: the default stopping rule
    lambda < 0.25 / sqrt(n) (src/mgb.jl:360) was fed the slice's node count instead of the mesh's."""
    import mgb_amd as m
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_gpu_worker, args=(world, _free_port(), "fem2d_P2_L7", out), nprocs=world, join=True)
        res = dict(out)
    ref = m.mgb_solve(_problem("fem2d_P2_L7"))
    b = np.asarray(ref.SOL_main["its"])
    for rank in range(world):
        r = res[rank]
        a = np.asarray(r["its"])
        assert a.shape == b.shape and np.abs(a - b).max() <= 1, (a.sum(axis=1), b.sum(axis=1))
        assert abs(int(a.sum()) - int(b.sum())) <= 3
        assert np.abs(r["z"] - ref.z).max() < 1e-8
        assert np.allclose(r["t"], ref.SOL_main["ts"])
    assert np.array_equal(res[0]["z"], res[1]["z"])


# ---------------------------------------------------------------------------------------------------------------------
# CPU: host-side pieces of the sharded solve that need no GPU
# ---------------------------------------------------------------------------------------------------------------------

def test_sharded_default_stopping_rule_uses_the_global_node_count():
    """src/mgb.jl:360: stopping_inexact(0.25 / sqrt(length(w)), 0.9) with w of the WHOLE mesh."""
    import math
    import sys
    sys.path.insert(0, ROOT)
    from mgb_amd import device as dev
    from mgb_amd import solve

    class SliceImage:                       # stands in for one rank's DeviceProblem (a slice of 1 000 nodes)
        lib = dev.load_library()
        n, nu = 1000, 2
        default_options = dev.DeviceProblem.default_options
    kw = dict(tol=None, t=0.1, kappa=None, maxit=None, max_newton=None, line_search=None, stopping_criterion=None,
              finalize=None, early_stop=0)
    assert solve._options(SliceImage(), **kw).stop_lambda_tol == 0.25 / math.sqrt(1000)
    assert solve._options(SliceImage(), n_nodes=4000, **kw).stop_lambda_tol == 0.25 / math.sqrt(4000)


def _collective_worker(rank, world, port, out):
    _init(rank, world, port)
    import ctypes as C
    import time
    from mgb_amd.sharded import _ABORT_KEY, _store, make_collective
    coll = make_collective(dist, "cpu")
    buf = np.array([1.0 + rank, 10.0 * (rank + 1)])
    rc = coll(None, buf.ctypes.data_as(C.c_void_p), 2, 0, 0)                 # SUM, host buffer, in place
    mx = np.array([float(rank)])
    rc2 = coll(None, mx.ctypes.data_as(C.c_void_p), 1, 1, 0)                 # MAX
    res = dict(rc=(rc, rc2), sum=buf.tolist(), max=mx.tolist())
    dist.barrier()
    # fail-fast: rank 1 "fails" (sets the abort key like ShardedSolver.solve_local does) and never enters the collective;
    # rank 0 must come back from its all-reduce with an error instead of blocking
    if rank == 1:
        _store(dist).set(_ABORT_KEY, "1")
        res["waited"] = 0.0
    else:
        t0 = time.monotonic()
        rc3 = coll(None, buf.ctypes.data_as(C.c_void_p), 2, 0, 0)
        res["waited"] = time.monotonic() - t0
        res["rc_abort"] = rc3
        res["err"] = str(coll.errors[-1]) if coll.errors else ""
    out[rank] = res
    # no barrier / destroy here: the group is poisoned by design after an abort


def test_collective_sums_in_place_and_fails_fast_when_a_peer_aborts():
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_collective_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = dict(out)
    for rank in range(world):
        assert res[rank]["rc"] == (0, 0)
        assert res[rank]["sum"] == [3.0, 30.0] and res[rank]["max"] == [1.0]
    assert res[0]["rc_abort"] == 1 and "peer rank" in res[0]["err"]
    assert res[0]["waited"] < 20.0


@pytest.mark.gpu
def test_sharded_config4_fem3d_L6_phase1_matches_single_rank():
    """BASELINE configs[3] (`fem3d() Q1 p = 4, L = 6, domain-decomposed`) at full size from the default, infeasible start
    -- phase I with box escalation, `_matched_t`, main ramp -- as ONE solve over two ranks sharing the device (the
    rehearsal form of the 8-GPU configuration: this pipeline's build loop has one GPU).  Same z as the single-rank device
    solve (1e-8, test/test_cuda.jl:51), same phase-I / main t-steps, Newton counts within 2 %."""
    import mgb_amd as m
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_gpu_worker, args=(world, _free_port(), "fem3d_L6_config4", out), nprocs=world, join=True)
        res = dict(out)
    ref = m.mgb_solve(_problem("fem3d_L6_config4"))
    assert ref.SOL_feasibility is not None
    b = int(np.asarray(ref.SOL_main["its"]).sum())
    bf = int(np.asarray(ref.SOL_feasibility["its"]).sum())
    for rank in range(world):
        r = res[rank]
        assert r["feas"] is not None
        assert np.abs(r["z"] - ref.z).max() < 1e-8
        assert np.allclose(r["t"], ref.SOL_main["ts"])
        assert abs(int(np.asarray(r["its"]).sum()) - b) <= max(3, 0.02 * b)
        assert abs(int(np.asarray(r["feas"]).sum()) - bf) <= max(3, 0.02 * bf)
    assert np.array_equal(res[0]["z"], res[1]["z"])


def _callback_worker(rank, world, port, out):
    """The collective wrappers of user callables (ShardedSolver._wrap_stopping / _wrap_early_stop) without a GPU: a bare
    ShardedSolver object with just the attributes the wrappers use."""
    _init(rank, world, port)
    import types
    from mgb_amd.sharded import ShardedSolver, _Reducer
    S = object.__new__(ShardedSolver)
    S.dist, S.rank, S.world = dist, rank, world
    S._shard = dict(reduce=_Reducer(dist, "cpu"))
    S.prob = types.SimpleNamespace(g=np.zeros((6 * world, 2)))          # nu = 2 state components
    res = {}
    # stopping rule: rank-dependent answers are MAX-reduced (every rank takes the same branch) ...
    stop = S._wrap_stopping(lambda *a: rank == 1)
    res["stop"] = stop(0.0, 0.0, 0.0, np.array([1.0]), None, 0.0, 0.0)
    # ... and an exception on ONE rank is raised on ALL ranks
    def bad(*a):
        if rank == 1:
            raise ValueError("boom on rank 1")
        return False
    try:
        S._wrap_stopping(bad)(0.0, 0.0, 0.0, np.array([1.0]), None, 0.0, 0.0)
        res["raised"] = None
    except Exception as e:                                               # noqa: BLE001
        res["raised"] = type(e).__name__
    # early_stop sees the WHOLE stacked iterate [u; s] in global node order, on every rank
    n_loc = 6
    u = np.arange(rank * n_loc, (rank + 1) * n_loc, dtype=float)
    z_loc = np.concatenate([u, 100.0 + u])
    seen = {}
    def early(z):
        seen["z"] = z.copy()
        return bool(z[: n_loc * world].max() > 10.0) and rank == 0       # only rank 0 says yes: MAX-reduced
    res["early"] = S._wrap_early_stop(early)(z_loc)
    res["z"] = seen["z"]
    res["two"] = S._wrap_early_stop(lambda z, t: t > 1.0)(z_loc, 2.0)
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def test_user_callables_are_collective_on_a_sharded_problem():
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_callback_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = dict(out)
    full = np.concatenate([np.arange(12.0), 100.0 + np.arange(12.0)])
    for rank in range(world):
        r = res[rank]
        assert r["stop"] is True                                         # one rank said "stop": all stop
        assert r["raised"] in ("ValueError", "RuntimeError")             # the raising rank its own error, the peer a RuntimeError
        assert r["early"] is True and r["two"] is True
        assert np.array_equal(r["z"], full)
    assert res[1]["raised"] == "ValueError" and res[0]["raised"] == "RuntimeError"


@pytest.mark.gpu
def test_sharded_mgb_solve_four_ranks_sharing_the_device():
    """G = 4 (the largest world the one-GPU box allows: at most six processes on the card): fem2d_P2 L = 5 over four ranks --
    the interface is the union of three cuts, every rank factors it redundantly -- against the
    single-rank solve: z to 1e-8, every level solve within +-1 iteration."""
    import mgb_amd as m
    world = 4
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_gpu_worker, args=(world, _free_port(), "fem2d_P2_L5", out), nprocs=world, join=True)
        res = dict(out)
    ref = m.mgb_solve(_problem("fem2d_P2_L5"))
    b = np.asarray(ref.SOL_main["its"])
    for rank in range(world):
        r = res[rank]
        a = np.asarray(r["its"])
        assert a.shape == b.shape and np.abs(a - b).max() <= 1, (a.sum(axis=1), b.sum(axis=1))
        assert np.abs(r["z"] - ref.z).max() < 1e-8
    assert all(np.array_equal(res[0]["z"], res[k]["z"]) for k in range(1, world))
