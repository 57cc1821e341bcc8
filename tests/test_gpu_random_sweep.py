"""Device vs oracle `mgb_solve` over a spread of small problems (-m gpu): 27 complete solves -- four mesh families,
AMG and geometric ladders, p from 1 to 3.5, one of them a creeping solve of 5 355 Newton iterations in a 37-unknown
space -- at north_star's end-to-end bar (1e-10 relative; observed <= 2.0e-11, tests/dev/logs/gpu_random_sweep_r03_final.txt)
with Newton iteration counts within +-3 of the oracle's (stopping rules compare rounded quantities; sums run in another
order on the device).  Promoted from tests/dev/gpu_random_sweep.py.

Reference: `mgb_solve` end to end, src/mgb.jl:798-842; the reference's own cross-backend criterion is 1e-8 absolute
(test/test_cuda.jl:51)."""
import numpy as np
import pytest

import mgb_amd as m
from helpers import assert_z_close, record_observation
from oracle import mgb_oracle as O

pytestmark = pytest.mark.gpu

CASES = {}
for L in (2, 3, 4, 5):
    for p in (1.0, 1.3, 2.0, 3.5):
        CASES[f"fem2d_P2 L={L} p={p}"] = (lambda L=L, p=p: m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p))
for L in (3, 5, 7):
    for p in (1.0, 2.5):
        CASES[f"fem1d L={L} p={p}"] = (lambda L=L, p=p: m.assemble(m.amg(m.subdivide(m.fem1d(), L)), p=p))
for L in (1, 2, 3):
    CASES[f"fem3d L={L} p=2"] = (lambda L=L: m.assemble(m.amg(m.subdivide(m.fem3d(k=1), L)), p=2.0))
for L in (2, 3):
    CASES[f"fem2d_P2 geometric_mg L={L} p=1.5"] = (lambda L=L: m.assemble(m.geometric_mg(m.fem2d_P2(), L), p=1.5))


@pytest.mark.parametrize("name", list(CASES))
def test_sweep_device_vs_oracle(name):
    prob = CASES[name]()
    sol = m.mgb_solve(prob)
    ref = O.mgb_solve(prob)
    its_d, its_o = np.asarray(sol.SOL_main["its"]), np.asarray(ref["SOL_main"]["its"])
    record_observation(f"sweep {name}: Newton iterations device/oracle {int(its_d.sum())}/{int(its_o.sum())}")
    assert_z_close(sol.z, ref["z"], f"sweep {name}")
    assert its_d.shape == its_o.shape                         # same number of t-steps and levels
    assert abs(int(its_d.sum()) - int(its_o.sum())) <= 3
