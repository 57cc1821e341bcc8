"""C-ABI surface (CPU): the library loads, exports every symbol the header declares, fails
loudly without a GPU, and the product package never touches the oracle."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _lib():
    from mgb_amd import device
    if not os.path.exists(device.library_path()):
        pytest.skip("libmgbhip.so not built (run __graft_entry__.build())")
    return device.load_library(), device


def test_exports_every_declared_symbol():
    lib, device = _lib()
    hdr = open(os.path.join(ROOT, "include", "mgbhip.h")).read()
    declared = sorted(set(re.findall(r"\b(mgbhip_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed from include/mgbhip.h"
    for name in declared:
        assert hasattr(lib, name), f"libmgbhip.so does not export {name}"
    assert sorted(device.EXPORTS) == declared


def test_struct_layouts_match_the_compiled_header():
    """The ctypes mirrors in device.py against the C compiler's own sizeof/offsetof of include/mgbhip.h
    (printed by the compiled caller tests/csrc/abi_smoke.c --layout): no hand-computed sizes."""
    import subprocess
    from test_c_abi import build_c_caller
    lib, device = _lib()
    out = subprocess.run([build_c_caller(), "--layout"], capture_output=True, text=True, check=True).stdout
    mirror = {"mgbhip_piece": device._Piece, "mgbhip_cone": device._Cone, "mgbhip_csr": device._CSR,
              "mgbhip_problem_desc": device._Desc, "mgbhip_options": device.Options,
              "mgbhip_core_result": device._CoreResult}
    seen = 0
    for ln in out.strip().splitlines():
        kind, what, val = ln.split()
        if kind == "sizeof":
            assert C.sizeof(mirror[what]) == int(val), ln
        else:
            st, field = what.split(".")
            assert getattr(mirror[st], field).offset == int(val), ln
        seen += 1
    assert seen >= 35


def test_no_gpu_is_an_error_not_a_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib, device = _lib()
    h = C.c_void_p()
    rc = lib.mgbhip_create(C.byref(h), 0, None)
    assert rc == device.ERR_HIP and b"hipGetDeviceCount" in lib.mgbhip_last_error()
    import numpy as np
    import mgb_amd as m
    prob = m.assemble(m.amg(m.fem1d(nodes=np.linspace(-1, 1, 3))), p=1.0)
    with pytest.raises(device.MGBHipError):
        m.mgb_solve(prob)
    with pytest.raises(RuntimeError):
        m.mgb_solve(prob, device=m.CPUDevice)       # there is no CPU path in this package


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "multigridbarrier.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", txt, re.M), f
                assert not re.search(r"#include\s+\".*oracle", txt), f
