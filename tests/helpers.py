import numpy as np

import mgb_amd as m


def build_case(c):
    g = c["geom"]
    if g == "fem1d":
        geom = m.fem1d(nodes=np.linspace(-1, 1, c["nodes"]))
    elif g == "fem2d_P2":
        geom = m.subdivide(m.fem2d_P2(), c["L"])
    elif g == "fem3d":
        geom = m.subdivide(m.fem3d(k=c["k"]), c["L"])
    elif g == "spectral1d":
        geom = m.spectral1d(n=c["n"])
    elif g == "spectral2d":
        geom = m.spectral2d(n=c["n"])
    else:
        return None
    return m.assemble(m.amg(geom), p=c["p"])


def build_geom(c):
    g = c["geom"]
    if g == "fem1d":
        return m.fem1d(nodes=np.linspace(-1, 1, c["nodes"]))
    if g == "fem2d_P2":
        return m.subdivide(m.fem2d_P2(), c["L"])
    if g == "spectral1d":
        return m.spectral1d(n=c["n"])
    if g == "spectral2d":
        return m.spectral2d(n=c["n"])
    raise ValueError(g)


def parabolic_goldens():
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_parabolic.json")) as fh:
        return {c["name"]: c for c in json.load(fh)["cases"]}


def gold_z(c):
    return np.array(c["z_colmajor"]).reshape(c["ncols"], -1).T


def stacked(z):
    return np.ascontiguousarray(z.T).reshape(-1).copy()


def lower_bound_problem(lower, nodes=5):
    """The phase-I test problem of the reference (test/test_feasibility.jl:13-24): minimise
    int u subject to u >= lower, from the start u = 0."""
    mg = m.amg(m.fem1d(nodes=np.linspace(-1.0, 1.0, nodes)))
    Q = m.convex_linear(mg, idx=(1,), A=lambda x: np.array([[1.0]]), b=lambda x: np.array([-lower]))
    return m.assemble(mg, state_variables=[("u", "full")], D=[("u", "id")], f=lambda x: np.array([1.0]),
                      g=lambda x: np.array([0.0]), Q=Q)
