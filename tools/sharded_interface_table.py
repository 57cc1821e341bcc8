"""Interface sizes and collective payloads of the domain-decomposed Newton solve, computed from the
actual hierarchies on the CPU:  python tools/sharded_interface_table.py fem2d 9 1.0   |   fem3d 6 4.0 '{"max_coarse":500}'
Per level J and world size G: |Gamma_J| (unknowns whose support meets more than one rank), the doubles one rank sends per
Newton iteration -- the interface gradient + the packed lower triangle of the interface front (|Gamma| + 1)(|Gamma| + 2) / 2
(the border row rides along) -- and the flops of the redundant interface factorization (|Gamma|^3 / 3)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mgb_amd as m
from mgb_amd.sharded import element_partition, shard_level

fam, L, p = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
kw = json.loads(sys.argv[4]) if len(sys.argv) > 4 else {}
geo = m.fem2d_P2() if fam == "fem2d" else m.fem3d(k=1)
prob = m.assemble(m.amg(m.subdivide(geo, L), prolongator=m.amg_ruge_stuben(**kw)) if kw else m.amg(m.subdivide(geo, L)), p=p)
M = prob.M[0]
first = M.D_fine[0]
pn, N = first.active_block.p, first.active_block.N
n = pn * N
print(f"{fam} L={L}: {n} nodes, {N} elements, levels {[R.shape[1] for R in M.R_fine]}")
print("| G | level: unknowns | interface per level | doubles per rank per fine Newton iteration (gradient + packed front) | MB | redundant GFLOP (fine) |")
print("|---|---|---|---|---|---|")
for G in (2, 4, 8):
    parts = element_partition(N, G)
    gam = [int(shard_level(R, n, pn, parts, 0).iface.size) for R in M.R_fine]
    fine = gam[-1]
    dbl = fine + (fine + 1) * (fine + 2) // 2
    print(f"| {G} | {[R.shape[1] for R in M.R_fine]} | {gam} | {fine} + {(fine + 1) * (fine + 2) // 2} = {dbl} | {8e-6 * dbl:.2f} | {fine ** 3 / 3 * 1e-9:.2f} |", flush=True)
