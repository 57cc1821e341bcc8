"""Row (e): element-sharded evaluate/assemble, world_size 2 over gloo.

CPU part (-m "not gpu"): the sharding, slicing and interface-reduction logic of
multigridbarrier.jl_amd/sharded.py with the ORACLE as each rank's local evaluator -- sharded f0 / f1 /
R'HR must equal the single-rank oracle to 1e-13, and only interface entries may travel.
GPU part (-m gpu): the same with each rank's slice on the device (two processes sharing GPU 0)."""
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


class OracleLocalEvaluator:
    def __init__(self, prob, rank, world):
        from mgb_amd.sharded import element_partition, slice_problem
        from oracle import mgb_oracle as O
        first = prob.M[0].D_fine[0]
        N, p = first.active_block.N, first.active_block.p
        e0, e1 = element_partition(N, world)[rank]
        sub = slice_problem(prob, e0, e1)
        self.M = O.OracleAMG(sub.M[0])
        self.B = O.Barrier(sub.Q, np.full(p * (e1 - e0), 1.0 / (p * N)))

    def f0(self, level, s, c, z0):
        return self.B.f0(s, self.M.w, c, self.M.R_fine[level], self.M.D_fine, z0)

    def f1(self, level, s, c, z0):
        return self.B.f1(s, self.M.w, c, self.M.R_fine[level], self.M.D_fine, z0)

    def f2(self, level, s, c, z0):
        return sp.csr_matrix(self.B.f2(s, self.M.w, c, self.M.R_fine[level], self.M.D_fine, z0))


def _problem(kind):
    import mgb_amd as m
    if kind == "fem2d_P2":
        return m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 3)), p=1.5)
    return m.assemble(m.amg(m.subdivide(m.fem3d(k=1), 3)), p=4.0, g_grid=None)


def _worker(rank, world, port, kind, use_device, out):
    import sys
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from mgb_amd.sharded import DeviceLocalEvaluator, ShardedBarrier
    from helpers import stacked
    prob = _problem(kind)
    if kind == "fem3d":
        prob.g[:, 1] = 1.0e4                     # a feasible start for p = 4 (the default one needs phase I)
    local = DeviceLocalEvaluator(prob, rank, world, device_id=0) if use_device else OracleLocalEvaluator(prob, rank, world)
    SB = ShardedBarrier(prob, rank, world, local, dist=dist, device="cpu")
    rng = np.random.default_rng(5)               # same stream on every rank: s is replicated
    z0, c = stacked(prob.g), 0.1 * prob.f
    res = {}
    for level in (len(prob.M[0].R_fine) - 1, 1):
        m_J = prob.M[0].R_fine[level].shape[1]
        s = 1e-3 * rng.standard_normal(m_J)
        y, extra = SB.f0(level, s, c, z0, extra=(float(rank + 1), 2.0))
        g = SB.f1(level, s, c, z0)
        gi, iface = SB.f1_interface_only(level, s, c, z0)
        vals, pl = SB.f2(level, s, c, z0)
        payload_h = pl.shared.size
        H = SB.gather_hessian(vals, pl)
        res[level] = dict(y=y, extra=extra.tolist(), g=g, gi=gi, iface=iface, H=H, m=m_J, shared=payload_h, nnz=pl.colidx.size)
    if use_device:
        local.close()
    out[rank] = res
    dist.barrier()
    dist.destroy_process_group()


def _reference(kind, use_device):
    """Single-rank values of the same closures (oracle on CPU; the device itself on the GPU box)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import stacked
    from oracle import mgb_oracle as O
    prob = _problem(kind)
    if kind == "fem3d":
        prob.g[:, 1] = 1.0e4
    rng = np.random.default_rng(5)
    z0, c = stacked(prob.g), 0.1 * prob.f
    ref = {}
    D = None
    if use_device:
        from mgb_amd.device import DeviceMGBProblem
        D = DeviceMGBProblem(prob)
    Mo, B = O.OracleAMG(prob.M[0]), O.Barrier(prob.Q)
    for level in (len(prob.M[0].R_fine) - 1, 1):
        R = Mo.R_fine[level]
        s = 1e-3 * rng.standard_normal(R.shape[1])
        if use_device:
            ref[level] = (D.main.f0(level, s, c, z0), D.main.f1(level, s, c, z0), D.main.f2(level, s, c, z0))
        else:
            ref[level] = (B.f0(s, Mo.w, c, R, Mo.D_fine, z0), B.f1(s, Mo.w, c, R, Mo.D_fine, z0),
                          sp.csr_matrix(B.f2(s, Mo.w, c, R, Mo.D_fine, z0)))
    if D is not None:
        D.close()
    return ref


def _run(kind, use_device):
    world = 2
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_worker, args=(world, _free_port(), kind, use_device, out), nprocs=world, join=True)
        res = dict(out)
    ref = _reference(kind, use_device)
    for level, (y, g, H) in ref.items():
        for rank in range(world):
            r = res[rank][level]
            assert abs(r["y"] - y) <= 1e-13 * abs(y)
            assert r["extra"] == [3.0, 4.0]                               # batched scalars ride in the same all-reduce
            assert np.linalg.norm(r["g"] - g) <= 1e-13 * np.linalg.norm(g)
            assert abs(r["H"] - H).max() <= 1e-13 * abs(H).max()
            # interface-only form: the summed interface entries are the global ones
            assert np.linalg.norm(r["gi"][r["iface"]] - g[r["iface"]]) <= 1e-13 * np.linalg.norm(g)
        a, b = res[0][level], res[1][level]
        assert np.array_equal(a["iface"], b["iface"])
        interior = np.setdiff1d(np.arange(a["m"]), a["iface"])
        assert np.all((a["gi"][interior] == 0) | (b["gi"][interior] == 0))  # an interior DoF lives on one rank only
        assert np.allclose(a["gi"][interior] + b["gi"][interior], g[interior], rtol=1e-13, atol=1e-300)
        if level == max(ref):                                              # fine level: the interface is a thin layer
            assert a["iface"].size < 0.1 * a["m"] and a["shared"] < 0.1 * a["nnz"]


@pytest.mark.parametrize("kind", ["fem2d_P2", "fem3d"])
def test_sharded_closures_match_single_rank_oracle_world2(kind):
    _run(kind, use_device=False)


def test_element_partition_and_slices():
    import mgb_amd as m
    from mgb_amd.sharded import element_partition, slice_problem
    assert element_partition(10, 4) == [(0, 3), (3, 6), (6, 8), (8, 10)]
    prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 2)), p=1.0)
    n = prob.M[0].w.size
    a, b = slice_problem(prob, 0, 3), slice_problem(prob, 3, 8)
    assert a.M[0].w.size + b.M[0].w.size == n and a.f.shape[0] == 21 and b.g.shape[0] == 35
    for l, R in enumerate(prob.M[0].R_fine):                               # row slices: same column space, rows add up
        assert a.M[0].R_fine[l].shape[1] == R.shape[1] == b.M[0].R_fine[l].shape[1]
        assert a.M[0].R_fine[l].nnz + b.M[0].R_fine[l].nnz == sp.csr_matrix(R).nnz


@pytest.mark.gpu
def test_sharded_closures_match_single_rank_device_world2():
    _run("fem2d_P2", use_device=True)
