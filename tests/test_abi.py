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


def test_struct_layouts_match_header_sizes():
    lib, device = _lib()
    # sizes computed from the header's field lists (natural alignment, LP64)
    assert C.sizeof(device._Piece) == 3 * 4 + 4 * 4 + 4 + 4 * 8 + 2 * 8 + 8       # 28 -> pad to 32, pointers, consts, select
    assert C.sizeof(device._CSR) == 40
    assert C.sizeof(device.Options) == 8 * 3 + 4 * 2 + 8 * 2 + 8 + 8 * 2 + 8 + 8 + 8


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
