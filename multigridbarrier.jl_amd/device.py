"""Device markers and the native<->device transfer for the HIP backend.

Mirror of the reference's `Device` dispatch (reference: src/device.jl:18-92) and of the
CUDA extension's converters (ext/MultiGridBarrierCUDAExt/conversion.jl:152-264): a
CPU-assembled `MGBProblem` is uploaded once through the C ABI of `libmgbhip.so`
(`include/mgbhip.h`); the solution comes back as NumPy arrays.

There is deliberately no CPU compute path here: `CPUDevice` exists as a marker for API
parity, but this package ships only the MI355X backend and raises if the HIP library (or
a GPU) is missing.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import List, Optional

import numpy as np
import scipy.sparse as sp

from .blockmatrices import BlockColumn, BlockDiag
from .convex import KIND_EP, KIND_LINEAR, Convex
from .multigrid import AMG
from .problem import MGBProblem

MAX_PIECES, MAX_IDX, MAX_ND, MAX_NU, MAX_OPS = 4, 4, 10, 4, 8
OK, ERR_INVALID, ERR_HIP, ERR_NOT_SPD, ERR_NONFINITE, ERR_CONVERGENCE = range(6)


class Device:
    """reference: src/device.jl:18"""


class CPUDevice(Device):
    """Marker only: the native CPU path is the reference itself (src/device.jl:25)."""


class HIPDevice(Device):
    """The MI355X backend (the reference's `CUDADevice` slot, src/device.jl:32)."""


_DEFAULT = [HIPDevice]


def default_device():
    return _DEFAULT[0]


def default_device_set(D):
    _DEFAULT[0] = D
    return D


# ---------------------------------------------------------------------------
# ctypes mirror of include/mgbhip.h
# ---------------------------------------------------------------------------

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class _Piece(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ni", C.c_int32), ("nc", C.c_int32), ("idx", C.c_int32 * MAX_IDX),
                ("A", _dp), ("b", _dp), ("p", _dp), ("mu", _dp), ("p_const", C.c_double),
                ("mu_const", C.c_double), ("select", _dp)]


class _Cone(C.Structure):
    _fields_ = [("npieces", C.c_int32), ("pieces", _Piece * MAX_PIECES), ("feasibility", C.c_int32),
                ("NC", C.c_int32)]


class _CSR(C.Structure):
    _fields_ = [("rows", C.c_int64), ("cols", C.c_int64), ("rowptr", _ip), ("colidx", _ip), ("values", _dp)]


class _Desc(C.Structure):
    _fields_ = [("p", C.c_int32), ("N", C.c_int64), ("nu", C.c_int32), ("nD", C.c_int32), ("n_ops", C.c_int32),
                ("ops", _dp * MAX_OPS), ("D_state", C.c_int32 * MAX_ND), ("D_op", C.c_int32 * MAX_ND),
                ("w", _dp), ("L", C.c_int32), ("R", C.POINTER(_CSR)), ("cone", _Cone),
                ("barrier_weights", _dp), ("x", _dp), ("dim", C.c_int32)]


class Options(C.Structure):
    _fields_ = [("tol", C.c_double), ("t", C.c_double), ("kappa", C.c_double), ("maxit", C.c_int32),
                ("max_newton", C.c_int32), ("ls_beta", C.c_double), ("ls_c1", C.c_double),
                ("line_search", C.c_int32), ("stop_lambda_tol", C.c_double), ("stop_theta", C.c_double),
                ("finalize", C.c_int32), ("finalize_theta", C.c_double), ("early_stop", C.c_int32),
                ("stopping_criterion", C.c_void_p), ("early_stop_fn", C.c_void_p), ("user", C.c_void_p)]


STOP_FN = C.CFUNCTYPE(C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_void_p)
EARLY_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_double, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32)


class _CoreResult(C.Structure):
    _fields_ = [("k", C.c_int32), ("L", C.c_int32), ("failure_code", C.c_int32), ("t_final", C.c_double),
                ("t_elapsed", C.c_double), ("solve_seconds", C.c_double), ("newton_iterations", C.c_int64),
                ("f0_evals", C.c_int64), ("f1_evals", C.c_int64), ("f2_evals", C.c_int64),
                ("factorizations", C.c_int64), ("cap_steps", C.c_int32), ("its", C.POINTER(C.c_int64)),
                ("ts", _dp), ("kappas", _dp), ("times", _dp), ("c_dot_Dz", _dp)]


EXPORTS = [
    "mgbhip_create", "mgbhip_destroy", "mgbhip_last_error", "mgbhip_version", "mgbhip_problem_create",
    "mgbhip_problem_destroy", "mgbhip_problem_set_box", "mgbhip_problem_set_barrier_weights",
    "mgbhip_level_size", "mgbhip_f0", "mgbhip_f1", "mgbhip_f2", "mgbhip_hessian_pattern", "mgbhip_solve",
    "mgbhip_set_hessian", "mgbhip_solve_newton", "mgbhip_newton_direction", "mgbhip_problem_set_sharding",
    "mgbhip_problem_set_collective",
    "mgbhip_node_barrier", "mgbhip_node_slack", "mgbhip_mgb_core", "mgbhip_matched_t",
    "mgbhip_default_options", "mgbhip_stage_ms", "mgbhip_reset_stage_timers", "mgbhip_solver_stats", "mgbhip_solver_chain",
    "mgbhip_vec_alloc", "mgbhip_vec_free", "mgbhip_vec_len", "mgbhip_vec_upload", "mgbhip_vec_download",
    "mgbhip_vec_fill", "mgbhip_vec_copy", "mgbhip_vec_axpy", "mgbhip_vec_scale", "mgbhip_vec_dot",
    "mgbhip_vec_norm", "mgbhip_vec_isfinite", "mgbhip_f0_d", "mgbhip_f1_d", "mgbhip_f2_d", "mgbhip_solve_d",
    "mgbhip_prolong_add",
]


def library_path() -> str:
    """The in-tree build; MGBHIP_LIB overrides it (A/B builds of the same library, like the Julia glue's variable)."""
    here = os.path.dirname(os.path.abspath(__file__))
    return os.environ.get("MGBHIP_LIB") or os.path.join(here, "lib", "libmgbhip.so")


_LIB = None


def load_library():
    """Load libmgbhip.so (in-tree build).  No fallback: a missing library is an error."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = library_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build it with `make -C multigridbarrier.jl_amd/csrc` "
                           "(or __graft_entry__.build()); this package has no CPU fallback")
    lib = C.CDLL(path)
    lib.mgbhip_last_error.restype = C.c_char_p
    lib.mgbhip_version.restype = C.c_char_p
    lib.mgbhip_level_size.restype = C.c_int64
    lib.mgbhip_level_size.argtypes = [C.c_void_p, C.c_int32]
    lib.mgbhip_create.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_void_p]
    lib.mgbhip_destroy.argtypes = [C.c_void_p]
    lib.mgbhip_problem_create.argtypes = [C.c_void_p, C.POINTER(_Desc), C.c_void_p, C.POINTER(C.c_void_p)]
    lib.mgbhip_problem_destroy.argtypes = [C.c_void_p]
    lib.mgbhip_problem_set_box.argtypes = [C.c_void_p, C.c_double, C.c_double]
    lib.mgbhip_problem_set_barrier_weights.argtypes = [C.c_void_p, _dp]
    for name in ("mgbhip_f0", "mgbhip_f1", "mgbhip_f2"):
        getattr(lib, name).argtypes = [C.c_void_p, C.c_int32, _dp, _dp, _dp, _dp]
    lib.mgbhip_hessian_pattern.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(_ip), C.POINTER(_ip)]
    lib.mgbhip_solve.argtypes = [C.c_void_p, C.c_int32, _dp, _dp]
    lib.mgbhip_set_hessian.argtypes = [C.c_void_p, C.c_int32, _dp]
    lib.mgbhip_solve_newton.argtypes = [C.c_void_p, C.c_int32, _dp, _dp, _dp]
    lib.mgbhip_newton_direction.argtypes = [C.c_void_p, C.c_int32, _dp, _dp, _dp, _dp, _dp, _ip]
    lib.mgbhip_problem_set_sharding.argtypes = [C.c_void_p, C.c_int32, C.c_int64, _ip, _dp]
    lib.mgbhip_problem_set_collective.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]
    lib.mgbhip_node_barrier.argtypes = [C.c_void_p, _dp, _dp, _dp]
    lib.mgbhip_node_slack.argtypes = [C.c_void_p, _dp, _dp]
    lib.mgbhip_mgb_core.argtypes = [C.c_void_p, _dp, _dp, C.POINTER(Options), C.POINTER(_CoreResult)]
    lib.mgbhip_matched_t.argtypes = [C.c_void_p, _dp, _dp, C.c_double, _dp]
    lib.mgbhip_default_options.argtypes = [C.POINTER(Options), C.c_int64]
    lib.mgbhip_default_options.restype = None
    lib.mgbhip_stage_ms.argtypes = [C.c_void_p, C.c_char_p, _dp, C.POINTER(C.c_int64)]
    lib.mgbhip_reset_stage_timers.argtypes = [C.c_void_p, C.c_int]
    lib.mgbhip_solver_stats.argtypes = [C.c_void_p, C.c_int32, _dp]
    lib.mgbhip_solver_chain.argtypes = [C.c_void_p, C.c_int32, _dp]
    vp = C.c_void_p
    lib.mgbhip_vec_alloc.argtypes = [vp, C.c_int64, C.POINTER(vp)]
    lib.mgbhip_vec_free.argtypes = [vp]
    lib.mgbhip_vec_len.argtypes = [vp]
    lib.mgbhip_vec_len.restype = C.c_int64
    lib.mgbhip_vec_upload.argtypes = [vp, _dp, C.c_int64]
    lib.mgbhip_vec_download.argtypes = [vp, _dp, C.c_int64]
    lib.mgbhip_vec_fill.argtypes = [vp, C.c_double]
    lib.mgbhip_vec_copy.argtypes = [vp, vp]
    lib.mgbhip_vec_axpy.argtypes = [C.c_double, vp, vp]
    lib.mgbhip_vec_scale.argtypes = [C.c_double, vp]
    lib.mgbhip_vec_dot.argtypes = [vp, vp, _dp]
    lib.mgbhip_vec_norm.argtypes = [vp, _dp]
    lib.mgbhip_vec_isfinite.argtypes = [vp, C.POINTER(C.c_int32)]
    lib.mgbhip_f0_d.argtypes = [vp, C.c_int32, vp, vp, vp, _dp]
    lib.mgbhip_f1_d.argtypes = [vp, C.c_int32, vp, vp, vp, vp]
    lib.mgbhip_f2_d.argtypes = [vp, C.c_int32, vp, vp, vp]
    lib.mgbhip_solve_d.argtypes = [vp, C.c_int32, vp, vp]
    lib.mgbhip_prolong_add.argtypes = [vp, C.c_int32, vp, vp]
    _LIB = lib
    return lib


class MGBHipError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libmgbhip status {status}: {message}")
        self.status = status


def _check(lib, status):
    if status != OK:
        raise MGBHipError(status, lib.mgbhip_last_error().decode())


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(_dp)


def _f64(a, order="C") -> np.ndarray:
    return np.require(a, dtype=np.float64, requirements=["C" if order == "C" else "F", "ALIGNED"])


class HipContext:
    """`mgbhip_ctx`: one device + one stream (explicit, never the NULL stream)."""

    def __init__(self, device_id: int = 0, stream: Optional[int] = None):
        self.lib = load_library()
        h = C.c_void_p()
        _check(self.lib, self.lib.mgbhip_create(C.byref(h), int(device_id), C.c_void_p(stream) if stream else None))
        self.handle = h
        self._vectors = weakref.WeakSet()      # live DeviceVectors: freed before the context they point into

    def close(self):
        if self.handle:
            for v in list(self._vectors):
                v.close()
            self.lib.mgbhip_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceVector:
    """`mgbhip_vec`: a device-resident vector with the algebra the reference's generic `newton`
    needs from its vector type (`+`, `-`, scalar `*`, `dot`, `norm`, `all(isfinite)`; the CUDA
    extension gets these from CuArray, ext/MultiGridBarrierCUDAExt/mgb_interface.jl:14-41)."""

    def __init__(self, ctx: "HipContext", data=None, length: Optional[int] = None):
        self.ctx, self.lib = ctx, ctx.lib
        if data is not None:
            data = _f64(np.asarray(data, dtype=np.float64).reshape(-1))
            length = data.size
        h = C.c_void_p()
        _check(self.lib, self.lib.mgbhip_vec_alloc(ctx.handle, int(length), C.byref(h)))
        self.handle, self.n = h, int(length)
        ctx._vectors.add(self)
        if data is not None and length:
            _check(self.lib, self.lib.mgbhip_vec_upload(h, _ptr(data), length))

    def to_host(self) -> np.ndarray:
        out = np.empty(self.n)
        if self.n:
            _check(self.lib, self.lib.mgbhip_vec_download(self.handle, _ptr(out), self.n))
        return out

    def copy(self) -> "DeviceVector":
        out = DeviceVector(self.ctx, length=self.n)
        _check(self.lib, self.lib.mgbhip_vec_copy(out.handle, self.handle))
        return out

    def axpy(self, alpha: float, x: "DeviceVector") -> "DeviceVector":
        _check(self.lib, self.lib.mgbhip_vec_axpy(float(alpha), x.handle, self.handle))
        return self

    def __add__(self, o):
        return self.copy().axpy(1.0, o)

    def __sub__(self, o):
        return self.copy().axpy(-1.0, o)

    def __mul__(self, a):
        out = self.copy()
        _check(self.lib, self.lib.mgbhip_vec_scale(float(a), out.handle))
        return out

    __rmul__ = __mul__

    def fill(self, value: float):
        _check(self.lib, self.lib.mgbhip_vec_fill(self.handle, float(value)))
        return self

    def dot(self, o) -> float:
        out = C.c_double()
        _check(self.lib, self.lib.mgbhip_vec_dot(self.handle, o.handle, C.cast(C.byref(out), _dp)))
        return out.value

    def norm(self) -> float:
        out = C.c_double()
        _check(self.lib, self.lib.mgbhip_vec_norm(self.handle, C.cast(C.byref(out), _dp)))
        return out.value

    def all_isfinite(self) -> bool:
        out = C.c_int32()
        _check(self.lib, self.lib.mgbhip_vec_isfinite(self.handle, C.byref(out)))
        return bool(out.value)

    def close(self):
        if self.handle:
            if self.ctx.handle:                # mgbhip_vec_free dereferences the context: never after mgbhip_destroy
                self.lib.mgbhip_vec_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DeviceProblem:
    """One (AMG, Convex) pair resident on the device: the `native_to_device` image the
    Barrier closures run on (reference: src/convex.jl:147-205 consume (w, R, D); the CUDA
    twin converts the same fields, conversion.jl:122-159)."""

    def __init__(self, ctx: HipContext, M: AMG, Q: Convex, feasibility: bool = False, NC: int = 0,
                 barrier_weights: Optional[np.ndarray] = None, share: "DeviceProblem | None" = None):
        self.ctx = ctx
        self.lib = ctx.lib
        geom = M.geometry
        first = M.D_fine[0]
        if not isinstance(first, BlockColumn):
            raise NotImplementedError("dense (spectral) operators are uploaded through dense_as_block()")
        p, N = first.active_block.p, first.active_block.N
        self.p, self.N, self.n = p, N, p * N
        self.nu = first.nu
        self.nD = len(M.D_fine)
        if self.nD > MAX_ND or self.nu > MAX_NU:
            raise ValueError("problem exceeds the compiled MAX_ND / MAX_NU")
        keep: List[np.ndarray] = []
        d = _Desc()
        d.p, d.N, d.nu, d.nD = p, N, self.nu, self.nD
        op_names: List[str] = []
        for k, (state, name) in enumerate(M.D_spec):
            if name not in op_names:
                op_names.append(name)
            d.D_state[k] = state
            d.D_op[k] = op_names.index(name)
        if len(op_names) > MAX_OPS:
            raise ValueError("too many distinct operators")
        d.n_ops = len(op_names)
        for o, name in enumerate(op_names):
            op: BlockDiag = geom.operators[name]
            if op.is_identity():
                d.ops[o] = None
            else:
                arr = np.asfortranarray(op.data, dtype=np.float64)     # Julia Array{T,3} memory image
                keep.append(arr)
                d.ops[o] = arr.ctypes.data_as(_dp)
        w = _f64(M.w)
        keep.append(w)
        d.w = _ptr(w)
        L = len(M.R_fine)
        d.L = L
        csr = (_CSR * L)()
        self.level_sizes = []
        if hasattr(M.R_fine, "realize"):
            M.R_fine.realize()           # block-diagonal prolongators of all levels at once (multigrid.LazyLevels)
        for l, R in enumerate(M.R_fine):
            Rs = sp.csr_matrix(R)
            Rs.sum_duplicates()
            Rs.sort_indices()
            ip = np.ascontiguousarray(Rs.indptr, dtype=np.int32)
            ii = np.ascontiguousarray(Rs.indices, dtype=np.int32)
            vv = np.ascontiguousarray(Rs.data, dtype=np.float64)
            keep += [ip, ii, vv]
            csr[l].rows, csr[l].cols = Rs.shape
            csr[l].rowptr = ip.ctypes.data_as(_ip)
            csr[l].colidx = ii.ctypes.data_as(_ip)
            csr[l].values = vv.ctypes.data_as(_dp)
            self.level_sizes.append(Rs.shape[1])
        d.R = csr
        # cone
        if len(Q.pieces) > MAX_PIECES:
            raise ValueError("too many convex pieces for this build")
        d.cone.npieces = len(Q.pieces)
        d.cone.feasibility = 1 if feasibility else 0
        d.cone.NC = NC
        n = self.n
        for k, pc in enumerate(Q.pieces):
            P = d.cone.pieces[k]
            P.kind = pc.kind
            P.ni = pc.ni
            P.nc = pc.nc
            if pc.ni > MAX_IDX or pc.nc > MAX_IDX:
                raise ValueError("functor family size exceeds this build (MAX_IDX)")
            for c, i in enumerate(pc.idx):
                P.idx[c] = int(i)

            def grid(a, cols, default=None):
                if a is None:
                    return None
                a = np.asarray(a, dtype=np.float64).reshape(n, cols)
                if default is not None and np.array_equal(a, np.broadcast_to(default, a.shape)):
                    return None
                g = np.asfortranarray(a)        # n x K column-major, like the reference's Q.args
                keep.append(g)
                return g.ctypes.data_as(_dp)

            nc = pc.ni if pc.kind == KIND_EP else pc.nc
            ident = np.eye(nc, pc.ni).reshape(-1, order="F")[None, :] if nc == pc.ni else None
            P.A = grid(pc.A, nc * pc.ni, ident)
            P.b = grid(pc.b, nc, np.zeros((1, nc)))
            if pc.kind == KIND_EP:
                pu, mu = np.unique(pc.p), np.unique(pc.mu)
                if pu.size == 1 and mu.size == 1:
                    P.p, P.mu = None, None
                    P.p_const, P.mu_const = float(pu[0]), float(mu[0])
                else:
                    P.p = grid(pc.p, 1)
                    P.mu = grid(pc.mu, 1)
            if Q.select is not None:
                P.select = grid(Q.select[:, k], 1)
        xc = np.asfortranarray(np.asarray(M.x, dtype=np.float64).reshape(self.n, -1))     # AMG.x, ordering hint
        keep.append(xc)
        d.x, d.dim = _ptr(xc), xc.shape[1]
        bw = None
        if barrier_weights is not None:
            bw = _f64(barrier_weights)
            keep.append(bw)
            d.barrier_weights = _ptr(bw)
        h = C.c_void_p()
        _check(self.lib, self.lib.mgbhip_problem_create(ctx.handle, C.byref(d), share.handle if share else None, C.byref(h)))
        self.handle = h
        del keep

    # -- primitives ------------------------------------------------------------------------
    def f0(self, level: int, s, c, z0) -> float:
        s, c, z0 = _f64(s), np.asfortranarray(c, dtype=np.float64), _f64(z0)
        out = C.c_double()
        _check(self.lib, self.lib.mgbhip_f0(self.handle, level, _ptr(s), _ptr(c), _ptr(z0), C.cast(C.byref(out), _dp)))
        return out.value

    def f1(self, level: int, s, c, z0) -> np.ndarray:
        s, c, z0 = _f64(s), np.asfortranarray(c, dtype=np.float64), _f64(z0)
        g = np.empty(self.level_sizes[level])
        _check(self.lib, self.lib.mgbhip_f1(self.handle, level, _ptr(s), _ptr(c), _ptr(z0), _ptr(g)))
        return g

    # -- the same closures on device-resident vectors (nothing but scalars crosses PCIe) --------
    def vec(self, data=None, length=None) -> DeviceVector:
        return DeviceVector(self.ctx, data, length)

    def f0_d(self, level: int, s: DeviceVector, c: DeviceVector, z0: DeviceVector) -> float:
        out = C.c_double()
        _check(self.lib, self.lib.mgbhip_f0_d(self.handle, level, s.handle, c.handle, z0.handle, C.cast(C.byref(out), _dp)))
        return out.value

    def f1_d(self, level: int, s: DeviceVector, c: DeviceVector, z0: DeviceVector) -> DeviceVector:
        g = DeviceVector(self.ctx, length=self.level_sizes[level])
        _check(self.lib, self.lib.mgbhip_f1_d(self.handle, level, s.handle, c.handle, z0.handle, g.handle))
        return g

    def f2_d(self, level: int, s: DeviceVector, c: DeviceVector, z0: DeviceVector) -> None:
        _check(self.lib, self.lib.mgbhip_f2_d(self.handle, level, s.handle, c.handle, z0.handle))

    def solve_d(self, level: int, g: DeviceVector) -> DeviceVector:
        x = DeviceVector(self.ctx, length=g.n)
        _check(self.lib, self.lib.mgbhip_solve_d(self.handle, level, g.handle, x.handle))
        return x

    def prolong_add(self, level: int, s: DeviceVector, z: DeviceVector) -> DeviceVector:
        _check(self.lib, self.lib.mgbhip_prolong_add(self.handle, level, s.handle, z.handle))
        return z

    def hessian_pattern(self, level: int):
        nnz = C.c_int64()
        rp, ci = _ip(), _ip()
        _check(self.lib, self.lib.mgbhip_hessian_pattern(self.handle, level, C.byref(nnz), C.byref(rp), C.byref(ci)))
        m = self.level_sizes[level]
        indptr = np.ctypeslib.as_array(rp, shape=(m + 1,)).copy()
        indices = np.ctypeslib.as_array(ci, shape=(max(nnz.value, 1),))[: nnz.value].copy()
        return indptr, indices

    def f2(self, level: int, s, c, z0, want_matrix: bool = True):
        s, c, z0 = _f64(s), np.asfortranarray(c, dtype=np.float64), _f64(z0)
        if not want_matrix:
            _check(self.lib, self.lib.mgbhip_f2(self.handle, level, _ptr(s), _ptr(c), _ptr(z0), None))
            return None
        indptr, indices = self.hessian_pattern(level)
        vals = np.empty(indices.size)
        _check(self.lib, self.lib.mgbhip_f2(self.handle, level, _ptr(s), _ptr(c), _ptr(z0), _ptr(vals)))
        m = self.level_sizes[level]
        return sp.csr_matrix((vals, indices, indptr), shape=(m, m))

    def solve(self, level: int, g) -> np.ndarray:
        g = _f64(g)
        x = np.empty_like(g)
        _check(self.lib, self.lib.mgbhip_solve(self.handle, level, _ptr(g), _ptr(x)))
        return x

    def set_hessian(self, level: int, values) -> None:
        """Replace the values of the level's H (CSR order of `hessian_pattern`); the next solve factors them."""
        v = _f64(np.asarray(values, dtype=np.float64).reshape(-1))
        _check(self.lib, self.lib.mgbhip_set_hessian(self.handle, level, _ptr(v)))

    def solve_newton(self, level: int, g, check: bool = True):
        """x = H^{-1} g through the bordered factorization of the resident Newton loop; returns (x, lambda^2, status)."""
        g = _f64(g)
        x = np.empty_like(g)
        lam = C.c_double()
        status = self.lib.mgbhip_solve_newton(self.handle, level, _ptr(g), _ptr(x), C.cast(C.byref(lam), _dp))
        if check:
            _check(self.lib, status)
        return x, lam.value, status

    def newton_direction(self, level: int, s, c, z0):
        """(x, lambda^2, condensed): one Newton direction formed exactly as in the resident loop."""
        s, c, z0 = _f64(s), np.asfortranarray(c, dtype=np.float64), _f64(z0)
        x = np.empty(self.level_sizes[level])
        lam, cond = C.c_double(), C.c_int32()
        _check(self.lib, self.lib.mgbhip_newton_direction(self.handle, level, _ptr(s), _ptr(c), _ptr(z0), _ptr(x),
                                                          C.cast(C.byref(lam), _dp), C.byref(cond)))
        return x, lam.value, bool(cond.value)

    def node_barrier(self, z, want_Dz: bool = False):
        z = _f64(z)
        F = np.empty(self.n)
        Dz = np.empty((self.n, self.nD), order="F") if want_Dz else None
        _check(self.lib, self.lib.mgbhip_node_barrier(self.handle, _ptr(z), _ptr(F), _ptr(Dz)))
        return (F, Dz) if want_Dz else F

    def node_slack(self, z) -> np.ndarray:
        z = _f64(z)
        out = np.empty(self.n)
        _check(self.lib, self.lib.mgbhip_node_slack(self.handle, _ptr(z), _ptr(out)))
        return out

    def set_sharding(self, shards, collective, accepts_device_ptr: bool = False):
        """One process per GPU (include/mgbhip.h): per level the interface columns + ownership mask of this rank's slice,
        and the all-reduce the library calls (`collective`: an ALLREDUCE_FN thunk, kept alive by this object)."""
        for level, sh in enumerate(shards):
            ic = np.ascontiguousarray(sh.iface, dtype=np.int32)
            own = _f64(sh.own)
            _check(self.lib, self.lib.mgbhip_problem_set_sharding(self.handle, level, ic.size, ic.ctypes.data_as(_ip), _ptr(own)))
        self._collective = collective
        _check(self.lib, self.lib.mgbhip_problem_set_collective(self.handle, C.cast(collective, C.c_void_p), None,
                                                               1 if accepts_device_ptr else 0))

    def set_box(self, b: float, R: float):
        _check(self.lib, self.lib.mgbhip_problem_set_box(self.handle, float(b), float(R)))

    def set_barrier_weights(self, bw):
        a = None if bw is None else _f64(bw)
        _check(self.lib, self.lib.mgbhip_problem_set_barrier_weights(self.handle, _ptr(a)))

    def default_options(self, n_nodes: Optional[int] = None) -> Options:
        """Reference defaults (src/mgb.jl:95-101, :360); the stopping tolerance 0.25 / sqrt(n) uses the node count of
        the WHOLE mesh (`n_nodes`: a domain-decomposed problem passes the global count, this image holds a slice)."""
        o = Options()
        self.lib.mgbhip_default_options(C.byref(o), int(self.n if n_nodes is None else n_nodes))
        return o

    def mgb_core(self, z, c, opt: Options, cap_steps: int = 256):
        """Run the t-ramp on the device.  Returns (status, z, diagnostics dict)."""
        z = _f64(z).copy()
        c = np.asfortranarray(c, dtype=np.float64)
        L = len(self.level_sizes)
        its = np.zeros((L, cap_steps), dtype=np.int64, order="F")
        ts, kappas, times, cdz = (np.zeros(cap_steps) for _ in range(4))
        r = _CoreResult()
        r.cap_steps = cap_steps
        r.its = its.ctypes.data_as(C.POINTER(C.c_int64))
        r.ts, r.kappas, r.times, r.c_dot_Dz = _ptr(ts), _ptr(kappas), _ptr(times), _ptr(cdz)
        status = self.lib.mgbhip_mgb_core(self.handle, _ptr(z), _ptr(c), C.byref(opt), C.byref(r))
        if status not in (OK, ERR_CONVERGENCE):
            _check(self.lib, status)
        k = min(r.k, cap_steps)
        diag = dict(its=its[:, :k].copy(), ts=ts[:k].copy(), kappas=kappas[:k].copy(), times=times[:k].copy(),
                    c_dot_Dz=cdz[:k].copy(), t_elapsed=r.t_elapsed, t_final=r.t_final,
                    solve_seconds=r.solve_seconds, newton_iterations=int(r.newton_iterations),
                    f0_evals=int(r.f0_evals), f1_evals=int(r.f1_evals), f2_evals=int(r.f2_evals),
                    factorizations=int(r.factorizations), failure_code=int(r.failure_code), k=int(r.k))
        return status, z, diag

    def matched_t(self, z, c, t_default: float) -> float:
        z = _f64(z)
        c = np.asfortranarray(c, dtype=np.float64)
        out = C.c_double()
        _check(self.lib, self.lib.mgbhip_matched_t(self.handle, _ptr(z), _ptr(c), float(t_default), C.cast(C.byref(out), _dp)))
        return out.value

    def reset_stage_timers(self, enable: bool = True):
        _check(self.lib, self.lib.mgbhip_reset_stage_timers(self.handle, 1 if enable else 0))

    def stage_ms(self, stage: str):
        ms, cnt = C.c_double(), C.c_int64()
        _check(self.lib, self.lib.mgbhip_stage_ms(self.handle, stage.encode(), C.cast(C.byref(ms), _dp), C.byref(cnt)))
        return ms.value, cnt.value

    def solver_stats(self, level: int) -> dict:
        out = np.zeros(8)
        _check(self.lib, self.lib.mgbhip_solver_stats(self.handle, level, _ptr(out)))
        keys = ("fronts", "max_front", "arena_doubles", "factor_flops", "peeled", "tree_levels", "nnz", "unknowns")
        return dict(zip(keys, out.tolist()))

    def solver_chain(self, level: int) -> dict:
        out = np.zeros(8)
        _check(self.lib, self.lib.mgbhip_solver_chain(self.handle, level, _ptr(out)))
        keys = ("pivot_blocks_on_critical_path", "large_front_tree_levels", "launches_per_factorization",
                "launches_per_backward_sweep", "arena_doubles", "factor_flops", "extra_trailing_doubles", "reserved")
        return dict(zip(keys, out.tolist()))

    def close(self):
        if self.handle:
            self.lib.mgbhip_problem_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dense_as_block(M: AMG) -> AMG:
    """Spectral geometries carry dense operators (one notional element, reference:
    src/spectral1d.jl:100-108).  A dense operator *is* a single element block (N = 1,
    p = n_nodes): up to 64 nodes it runs through the element kernels, above that the library
    switches to its dense path (GEMV + node kernel + fp64 MFMA GEMM, csrc/dense.hip)."""
    if isinstance(M.D_fine[0], BlockColumn):
        return M
    n = M.w.size
    geom = M.geometry
    ops = {k: BlockDiag(np.asarray(v, dtype=np.float64).reshape(n, n, 1)) for k, v in geom.operators.items()}
    from dataclasses import replace
    g2 = replace(geom, operators=ops)
    nu = len(M.state_names)
    D_fine = [BlockColumn(ops[name], state, nu) for (state, name) in M.D_spec]
    return AMG(geometry=g2, x=M.x, w=M.w, R_fine=M.R_fine, D_fine=D_fine, state_names=M.state_names, D_spec=M.D_spec)


class DeviceMGBProblem:
    """`native_to_device(HIPDevice, prob)`: the (main, feasibility) pair on one context.
    The feasibility image is created lazily (most starts are feasible) and shares the
    operator arrays of the main image."""

    def __init__(self, prob: MGBProblem, device_id: int = 0, stream: Optional[int] = None, shards=None, collective=None,
                 accepts_device_ptr: bool = False):
        self.prob = prob
        self.ctx = HipContext(device_id, stream)
        self.M1 = dense_as_block(prob.M[0])
        self.main = DeviceProblem(self.ctx, self.M1, prob.Q)
        self._feas = None
        self._shards, self._collective, self._coll_dev = shards, collective, accepts_device_ptr
        if shards is not None:           # this process holds one rank's slice of a domain-decomposed problem (sharded.py)
            self.main.set_sharding(shards[0], collective, accepts_device_ptr)

    @property
    def feasibility(self) -> DeviceProblem:
        if self._feas is None:
            M2 = dense_as_block(self.prob.M[1])
            self._feas = DeviceProblem(self.ctx, M2, self.prob.Q, feasibility=True, NC=self.main.nD + 1,
                                       share=self.main)
            if self._shards is not None:
                self._feas.set_sharding(self._shards[1], self._collective, self._coll_dev)
        return self._feas

    def close(self):
        if self._feas is not None:
            self._feas.close()
        self.main.close()
        self.ctx.close()


def native_to_device(D, prob: MGBProblem, **kw) -> DeviceMGBProblem:
    """reference: src/device.jl:40, conversion.jl:263."""
    if D is HIPDevice:
        return DeviceMGBProblem(prob, **kw)
    raise RuntimeError(f"native_to_device: device {getattr(D, '__name__', D)} is unavailable in this package "
                       "(only HIPDevice is implemented; the CPU path is the reference itself)")


def device_to_native(D, sol):
    """Solutions are returned as NumPy arrays already (reference: src/device.jl:50)."""
    return sol


def mgb_cleanup(dev: Optional[DeviceMGBProblem] = None):
    """Plans and factorizations live in the handle; destroying it is the cache flush the
    reference performs in `mgb_cleanup` (src/BlockMatrices.jl:737-751)."""
    if dev is not None:
        dev.close()
