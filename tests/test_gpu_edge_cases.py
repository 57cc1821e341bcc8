"""Boundary behaviour of the C ABI on the GPU: malformed descriptors are refused with
MGBHIP_ERR_INVALID and a message (never a crash, never a CPU fallback); numerical infeasibility
is a value, not an error; the smallest meshes and a singular system."""
import numpy as np
import pytest

import mgb_amd as m
from mgb_amd import device as dev
from helpers import assert_z_close, stacked
from oracle import mgb_oracle as O

pytestmark = pytest.mark.gpu

INVALID, NOT_SPD = 1, 3


class _MutatingLib:
    """Forwards to libmgbhip but lets a test edit the descriptor just before problem_create."""

    def __init__(self, lib, mutate):
        self._lib, self._mutate = lib, mutate

    def __getattr__(self, name):
        return getattr(self._lib, name)

    def mgbhip_problem_create(self, ctx, dref, share, out):
        self._mutate(dref._obj)
        return self._lib.mgbhip_problem_create(ctx, dref, share, out)


def _small_problem():
    return m.assemble(m.amg(m.fem1d(nodes=np.linspace(-1, 1, 5))), p=1.5)


def _set(path, value):
    def mutate(d):
        obj = d
        for key in path[:-1]:
            obj = obj[key] if isinstance(key, int) else getattr(obj, key)
        if isinstance(path[-1], int):
            obj[path[-1]] = value
        else:
            setattr(obj, path[-1], value)
    return mutate


def _shrink_rows(d):
    d.R[0].rows -= 1


@pytest.mark.parametrize("mutate", [
    _set(("p",), 0), _set(("N",), 0), _set(("nu",), 0), _set(("nu",), 9), _set(("nD",), 0), _set(("nD",), 64),
    _set(("n_ops",), 0), _set(("L",), 0), _set(("D_state", 1), 7), _set(("D_op", 1), 6),
    _set(("cone", "npieces"), 0), _set(("cone", "npieces"), 9), _set(("cone", "pieces", 0, "kind"), 99),
    _set(("cone", "pieces", 0, "ni"), 0), _set(("cone", "pieces", 0, "idx", 0), 17), _shrink_rows,
])
def test_malformed_descriptor_is_refused_with_a_message(mutate):
    prob = _small_problem()
    ctx = dev.HipContext(0)
    real = ctx.lib
    ctx.lib = _MutatingLib(real, mutate)
    try:
        with pytest.raises(dev.MGBHipError) as ei:
            dev.DeviceProblem(ctx, dev.dense_as_block(prob.M[0]), prob.Q)
        assert ei.value.status == INVALID
        assert len(str(ei.value)) > 30          # mgbhip_last_error() carries the reason
    finally:
        ctx.lib = real
        ctx.close()


def test_out_of_range_level_and_solve_before_assembly():
    prob = _small_problem()
    D = dev.DeviceMGBProblem(prob)
    try:
        P = D.main
        L = len(P.level_sizes)
        z0, c = stacked(prob.g), 0.1 * prob.f
        for bad in (-1, L, L + 5):
            with pytest.raises(dev.MGBHipError) as ei:
                P.f0(bad, np.zeros(1), c, z0)
            assert ei.value.status == INVALID
        with pytest.raises(dev.MGBHipError) as ei:
            P.solve(L - 1, np.ones(P.level_sizes[L - 1]))          # no Hessian assembled yet
        assert ei.value.status == INVALID
    finally:
        D.close()


def test_nonfinite_inputs_propagate_as_values():
    prob = _small_problem()
    D = dev.DeviceMGBProblem(prob)
    try:
        P = D.main
        J = len(P.level_sizes) - 1
        z0, c = stacked(prob.g), 0.1 * prob.f
        s = np.zeros(P.level_sizes[J])
        s[0] = np.nan
        assert np.isnan(P.f0(J, s, c, z0))                         # a value, like the reference's Log protocol
        assert not np.all(np.isfinite(P.f1(J, s, c, z0)))
        s[0] = 1e300                                               # far outside the cone: +Inf / NaN, no error
        assert not np.isfinite(P.f0(J, s, c, z0))
    finally:
        D.close()


def test_singular_hessian_reports_not_spd():
    prob = _small_problem()
    D = dev.DeviceMGBProblem(prob)
    try:
        P = D.main
        J = len(P.level_sizes) - 1
        P.set_barrier_weights(np.zeros(prob.M[0].w.size))          # barrier switched off everywhere: H = 0
        z0 = stacked(prob.g)
        P.f2(J, np.zeros(P.level_sizes[J]), 0.1 * prob.f, z0)
        with pytest.raises(dev.MGBHipError) as ei:
            P.solve(J, np.ones(P.level_sizes[J]))
        assert ei.value.status == NOT_SPD
    finally:
        D.close()


@pytest.mark.parametrize("geom", ["fem1d_1element", "fem2d_P2_2elements", "fem3d_1element", "spectral1d_n2"])
def test_smallest_meshes_match_oracle(geom):
    g = {"fem1d_1element": lambda: m.fem1d(nodes=np.array([-1.0, 1.0])),
         "fem2d_P2_2elements": lambda: m.fem2d_P2(),
         "fem3d_1element": lambda: m.fem3d(k=1),
         "spectral1d_n2": lambda: m.spectral1d(n=2)}[geom]()
    prob = m.assemble(m.amg(g), p=1.5)
    sol = m.mgb_solve(prob)
    so = O.mgb_solve(prob)
    assert_z_close(sol.z, so["z"], f"smallest mesh {geom}")


def test_handles_are_independent_and_reusable():
    """Two images on one context and repeated solves on one image (the reference flushes its
    caches per solve, src/mgb.jl:840; a handle here keeps plans and symbolic factorizations)."""
    pa = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), 2)), p=1.0)
    pb = m.assemble(m.amg(m.fem1d(nodes=np.linspace(-1, 1, 9))), p=2.0)
    sa = m.mgb_solve(pa, keep_device=True)
    sb = m.mgb_solve(pb)
    from mgb_amd.solve import mgb_driver
    again = mgb_driver(sa.device)
    sa.device.close()
    assert np.array_equal(again["z"], sa.z)
    assert_z_close(sb.z, O.mgb_solve(pb)["z"], "second image on one context (fem1d 9 nodes p=2)")


def test_intersect_of_linear_cones_with_phase1_single_state():
    # test/test_cuda.jl:58-79: minimise int u subject to 1 <= u <= 5 from the infeasible start u = 0
    # (one state variable, intersect of two linear cones, phase-I slack path)
    mg = m.amg(m.fem2d_P2())
    Qa = m.convex_linear(mg, idx=(1,), A=lambda x: np.array([[1.0]]), b=lambda x: np.array([-1.0]))
    Qb = m.convex_linear(mg, idx=(1,), A=lambda x: np.array([[-1.0]]), b=lambda x: np.array([5.0]))
    prob = m.assemble(mg, state_variables=[("u", "full")], D=[("u", "id")], f=lambda x: np.array([1.0]),
                      g=lambda x: np.array([0.0]), Q=m.intersect(mg, Qa, Qb))
    sol = m.mgb_solve(prob)
    so = O.mgb_solve(prob)
    assert sol.SOL_feasibility is not None
    assert_z_close(sol.z, so["z"], "intersect of linear cones, phase I, single state")
    assert np.abs(sol.z - 1.0).max() < 1e-5


def test_explicit_stream_worker_thread_and_recovery_after_a_failed_solve():
    # test/test_cuda.jl:118-139: the handle is bound to the caller's stream, works off the main
    # thread, and a throwing solve leaves nothing stale behind
    import ctypes
    import threading
    prob = m.assemble(m.amg(m.fem2d_P2()), p=1.5)
    ref = O.mgb_solve(prob)["z"]
    hip = ctypes.CDLL("libamdhip64.so")                # the runtime libmgbhip itself is linked against
    stream = ctypes.c_void_p()
    assert hip.hipStreamCreate(ctypes.byref(stream)) == 0
    try:
        sol_stream = m.mgb_solve(prob, stream=stream.value)
        assert_z_close(sol_stream.z, ref, "explicit stream")
    finally:
        hip.hipStreamDestroy(stream)
    box = {}
    t = threading.Thread(target=lambda: box.setdefault("sol", m.mgb_solve(prob)))
    t.start()
    t.join()
    assert_z_close(box["sol"].z, ref, "worker thread")
    mgd = m.amg(m.fem1d(nodes=np.linspace(-1.0, 1.0, 9)))
    Qd = m.convex_linear(mgd, idx=(1, 3), A=lambda x: np.array([[1.0, 0.0]]), b=lambda x: np.array([2.0]))
    degenerate = m.assemble(mgd, Q=Qd)                 # slack unconstrained: must fail
    with pytest.raises(Exception):
        m.mgb_solve(degenerate)
    assert_z_close(m.mgb_solve(prob).z, ref, "solve after a failed solve")


def test_compiled_maxima_3d_parabolic_phase1():
    """The largest descriptor the build accepts: 3-D parabolic step = 3 states + phase-I slack
    (nu = 4 = MAX_NU), (dim + 3) + 1 + 3 = 10 operator rows (MAX_ND), index lists of 4 (MAX_IDX)."""
    mg = m.amg(m.subdivide(m.fem3d(k=1), 2))
    kw = dict(h=0.5, t1=0.5, p=1.5)
    sol = m.parabolic_solve(mg, **kw)
    so = m.parabolic_solve(mg, solver=O.mgb_solve, **kw)
    assert all(s.SOL_feasibility is not None for s in sol.steps)
    assert_z_close(np.stack(sol.u), np.stack(so.u), "3-D parabolic step, phase I, compiled maxima")


def test_four_piece_intersection_max_pieces():
    """MAX_PIECES = 4: power cone intersected with three linear cones (box on u, sign of the slack)."""
    mg = m.amg(m.subdivide(m.fem2d_P2(), 2))
    n = mg.geometry.w.size
    one = lambda a, b: m.convex_linear(mg, idx=(1,), A=lambda x: np.array([[a]]), b=lambda x: np.array([b]))
    Q = m.intersect(mg, m.convex_Euclidian_power(mg, idx=(2, 3, 4), p_grid=np.full(n, 1.5)),
                    one(1.0, 0.5), one(-1.0, 3.0),
                    m.convex_linear(mg, idx=(4,), A=lambda x: np.array([[1.0]]), b=lambda x: np.array([0.0])))
    prob = m.assemble(mg, Q=Q)
    sol = m.mgb_solve(prob)
    so = O.mgb_solve(prob)
    assert_z_close(sol.z, so["z"], "four-piece intersection")
    with pytest.raises(ValueError):                       # a fifth piece exceeds this build
        Q5 = m.intersect(mg, Q, one(1.0, 9.0))
        dev.DeviceMGBProblem(m.assemble(mg, Q=Q5))


def test_exception_in_user_callable_propagates_and_vectors_outlive_nothing():
    """A user `stopping_criterion` / `early_stop` that raises must surface from mgb_solve as in the reference
    (its exception propagates out of newton, src/newton.jl:273), not be swallowed by the ctypes thunk; and a
    DeviceVector collected after its context was closed must not touch the freed context."""
    prob = _small_problem()
    calls = {"n": 0}

    def bad_stop(*a):
        calls["n"] += 1
        if calls["n"] == 3:
            raise ValueError("user stopping rule failed")
        return False
    with pytest.raises(ValueError, match="user stopping rule failed"):
        m.mgb_solve(prob, stopping_criterion=bad_stop)
    assert calls["n"] == 3                                     # nothing called it again after the failure

    def bad_early(z):
        raise KeyError("user early_stop failed")
    with pytest.raises(KeyError):
        m.mgb_solve(prob, early_stop=bad_early)
    sol = m.mgb_solve(prob)                                    # the library is usable afterwards
    assert np.isfinite(sol.z).all()
    D = dev.DeviceMGBProblem(prob)
    v = D.main.vec(np.ones(D.main.level_sizes[0]))
    w = v + v                                                  # temporaries are DeviceVectors too
    D.close()
    assert v.handle is None and w.handle is None               # closed with their context
    del v, w                                                   # __del__ after the context is gone: no native call
