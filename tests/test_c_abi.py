"""A compiled C caller of include/mgbhip.h (tests/csrc/abi_smoke.c): no ctypes mirror between the
header and the library.  The CPU part checks that the header is valid C99 and that the program
links against libmgbhip.so; the GPU part solves the reference's fem2d_P2() p=1 golden problem
(test/runtests.jl:20-22) from C and checks the device-vector entry points against the host ones."""
import os
import struct
import subprocess

import numpy as np
import pytest
import scipy.sparse as sp

import mgb_amd as m
from helpers import gold_z, stacked

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "csrc", "abi_smoke.c")
EXE = os.path.join(ROOT, "tests", "csrc", "abi_smoke")


def build_c_caller():
    libdir = os.path.join(ROOT, "multigridbarrier.jl_amd", "lib")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"), SRC,
                           "-o", EXE, "-L", libdir, "-lmgbhip", "-Wl,-rpath,$ORIGIN/../../multigridbarrier.jl_amd/lib", "-lm"])
    return EXE


def write_problem(path, prob):
    """The byte image abi_smoke.c reads: exactly the arrays a binding passes in mgbhip_problem_desc."""
    M = prob.M[0]
    geom = M.geometry
    first = M.D_fine[0]
    p, N = first.active_block.p, first.active_block.N
    n = p * N
    op_names = []
    rows = []
    for (state, name) in M.D_spec:
        if name not in op_names:
            op_names.append(name)
        rows.append((state, op_names.index(name)))
    with open(path, "wb") as f:
        f.write(struct.pack("<iqiiii", p, N, first.nu, len(M.D_fine), len(op_names), len(M.R_fine)))
        for (state, o) in rows:
            f.write(struct.pack("<ii", state, o))
        for name in op_names:
            op = geom.operators[name]
            f.write(struct.pack("<i", 1 if op.is_identity() else 0))
            if not op.is_identity():
                f.write(np.asfortranarray(op.data, dtype=np.float64).tobytes(order="F"))
        f.write(np.asarray(M.w, dtype=np.float64).tobytes())
        x = np.asarray(M.x, dtype=np.float64).reshape(n, -1)
        f.write(struct.pack("<i", x.shape[1]))
        f.write(np.asfortranarray(x).tobytes(order="F"))
        for R in M.R_fine:
            R = sp.csr_matrix(R)
            R.sum_duplicates()
            R.sort_indices()
            f.write(struct.pack("<qqq", R.shape[0], R.shape[1], R.nnz))
            f.write(R.indptr.astype(np.int32).tobytes())
            f.write(R.indices.astype(np.int32).tobytes())
            f.write(R.data.astype(np.float64).tobytes())
        pc = prob.Q.pieces[0]
        f.write(struct.pack("<i", pc.ni))
        f.write(np.asarray(pc.idx, dtype=np.int32).tobytes())
        f.write(struct.pack("<dd", float(pc.p[0]), float(pc.mu[0])))
        f.write(np.asfortranarray(prob.f, dtype=np.float64).tobytes(order="F"))
        f.write(stacked(prob.g).astype(np.float64).tobytes())


def test_header_is_c99_and_the_c_caller_links():
    exe = build_c_caller()
    assert os.access(exe, os.X_OK)
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 2 and "usage" in out.stderr          # argument check only: no GPU call without a file


@pytest.mark.gpu
def test_c_caller_reproduces_the_reference_golden(tmp_path, golden):
    exe = build_c_caller()
    case = golden["fem2d_P2_L1_p1"]
    prob = m.assemble(m.amg(m.fem2d_P2()), p=1.0)
    path = str(tmp_path / "problem.bin")
    write_problem(path, prob)
    out = subprocess.run([exe, path], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    lines = {ln.split()[0]: ln.split()[1:] for ln in out.stdout.strip().splitlines()}
    z = np.array([float(v) for v in lines["z"]]).reshape(2, -1).T
    assert np.linalg.norm(z - gold_z(case)) < case["tol"]
    sol = m.mgb_solve(prob)
    assert np.array_equal(z, sol.z)                                  # same library, same bits as the Python host path
    ns = lines["newton_step"]
    y0, inc, y0d, incd = float(ns[1]), float(ns[2]), float(ns[4]), float(ns[5])
    gn, gnd, fin = float(ns[7]), float(ns[8]), int(ns[10])
    y1, y1b, armijo = float(ns[12]), float(ns[13]), float(ns[14])
    assert y0d == y0 and abs(incd - inc) <= 1e-13 * abs(inc) and abs(gnd - gn) <= 1e-13 * gn and fin == 1
    assert inc > 0 and y1 <= armijo                                  # sufficient decrease of the damped Newton step
    assert abs(y1b - y1) <= 1e-12 * max(1.0, abs(y1))                # z += R s on the device, then s = 0: same point
