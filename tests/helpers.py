import numpy as np

import mgb_amd as m


def build_case(c):
    g = c["geom"]
    if g == "fem1d":
        geom = m.fem1d(nodes=np.linspace(-1, 1, c["nodes"]))
    elif g == "fem2d_P2":
        geom = m.subdivide(m.fem2d_P2(), c["L"])
    elif g == "fem2d_P1":
        geom = m.subdivide(m.fem2d_P1(), c["L"])
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


# ---- end-to-end tolerance (north_star: "result within 1e-10 relative of reference") ---------------------------------
E2E_RTOL = 1e-10


def record_observation(line):
    """Append one line to gpurun_out/parity_observed.txt (merged back from the GPU box): the observed errors behind the
    asserted tolerances, so that a tolerance can be checked against what the hardware actually delivers."""
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = os.path.join(root, "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        with open(os.path.join(d, "parity_observed.txt"), "a") as fh:
            fh.write(line + "\n")
    except OSError:
        pass


def assert_z_close(z_dev, z_ref, label, rtol=E2E_RTOL):
    """max|z_dev - z_ref| <= rtol * max(1, max|z_ref|): north_star's end-to-end bar (the reference's own cross-backend
    criterion is the much looser absolute 1e-8, test/test_cuda.jl:51)."""
    z_dev, z_ref = np.asarray(z_dev), np.asarray(z_ref)
    assert z_dev.shape == z_ref.shape, (z_dev.shape, z_ref.shape)
    err = float(np.abs(z_dev - z_ref).max())
    scale = max(1.0, float(np.abs(z_ref).max()))
    record_observation(f"e2e {label}: max|dz| {err:.2e} relative {err / scale:.2e} (asserted {rtol:.0e})")
    assert err <= rtol * scale, (label, err, scale)


def literal_fem1d_problem(nodes, p):
    """An MGBProblem written out by hand -- fem1d, k = 1, the three nodes `nodes`, default f / g / D / state variables,
    power cone with exponent p -- from the reference's definitions (src/TensorFEM.jl:199-219, :428-490; src/mgb.jl:587-613;
    src/multigrid.jl:474-538).  The containers are plain dataclasses; no setup function of the package is called."""
    import scipy.sparse as sp
    from mgb_amd.blockmatrices import BlockColumn, BlockDiag
    from mgb_amd.convex import KIND_EP, Convex, Piece
    from mgb_amd.multigrid import AMG, Geometry
    from mgb_amd.problem import MGBProblem
    a, b, c = nodes
    h1, h2 = b - a, c - b
    x = np.array([a, b, b, c])                                     # broken nodes, element-major
    w = np.array([h1 / 2, h1 / 2, h2 / 2, h2 / 2])
    ident = np.zeros((2, 2, 2))
    ident[:, :, 0] = [[1.0, 0.0], [0.0, 1.0]]
    ident[:, :, 1] = [[1.0, 0.0], [0.0, 1.0]]
    dx = np.zeros((2, 2, 2))
    dx[:, :, 0] = [[-1.0 / h1, 1.0 / h1], [-1.0 / h1, 1.0 / h1]]
    dx[:, :, 1] = [[-1.0 / h2, 1.0 / h2], [-1.0 / h2, 1.0 / h2]]
    ops = {"id": BlockDiag(ident), "dx": BlockDiag(dx)}
    geom = Geometry(discretization=None, t=np.array([[0, 1], [1, 2]]), x=x.reshape(2, 2, 1).transpose(1, 0, 2).copy(),
                    w=w, operators=ops)
    # level -> fine broken basis, rows = [u at the 4 broken nodes; s at the 4 broken nodes]
    R0 = sp.csr_matrix(np.array([[0, 0], [1, 0], [1, 0], [0, 0],          # u: interior vertex (Dirichlet ends masked)
                                 [0, 1], [0, 1], [0, 1], [0, 1.0]]))      # s: constants
    R1 = sp.csr_matrix(np.array([[0, 0, 0, 0], [1, 0, 0, 0], [1, 0, 0, 0], [0, 0, 0, 0],
                                 [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 1, 0], [0, 0, 0, 1.0]]))   # s: continuous P1
    R2 = sp.csr_matrix(np.array([[0, 0, 0, 0, 0], [1, 0, 0, 0, 0], [1, 0, 0, 0, 0], [0, 0, 0, 0, 0],
                                 [0, 1, 0, 0, 0], [0, 0, 1, 0, 0], [0, 0, 0, 1, 0], [0, 0, 0, 0, 1.0]]))   # s: broken
    D_spec = [(0, "id"), (0, "dx"), (1, "id")]
    D_fine = [BlockColumn(ops[name], state, 2) for (state, name) in D_spec]
    main = AMG(geometry=geom, x=x.reshape(4, 1), w=w, R_fine=[R0, R1, R2], D_fine=D_fine, state_names=["u", "s"], D_spec=D_spec)
    # phase-I image (src/multigrid.jl:515-538): states (u, s, feasibility_slack), the slack in the :full space like s;
    # D rows = the user's three, the slack id row, then one id row per user state
    S0 = np.array([[1.0], [1.0], [1.0], [1.0]])
    S1 = np.array([[1, 0, 0], [0, 1, 0], [0, 1, 0], [0, 0, 1.0]])
    S2 = np.eye(4)
    U = np.array([[0.0], [1.0], [1.0], [0.0]])
    Rf = [sp.csr_matrix(sp.block_diag([U, S, S])) for S in (S0, S1, S2)]
    D_spec2 = [(0, "id"), (0, "dx"), (1, "id"), (2, "id"), (0, "id"), (1, "id")]
    D_fine2 = [BlockColumn(ops[name], state, 3) for (state, name) in D_spec2]
    feas = AMG(geometry=geom, x=x.reshape(4, 1), w=w, R_fine=Rf, D_fine=D_fine2, state_names=["u", "s", "feasibility_slack"],
               D_spec=D_spec2)
    n = 4
    Q = Convex([Piece(KIND_EP, (1, 2), np.tile([1.0, 0.0, 0.0, 1.0], (n, 1)), np.zeros((n, 2)), np.full(n, float(p)),
                      np.full(n, 0.0 if p in (1.0, 2.0) else (1.0 if p < 2 else 2.0)))])
    f = np.tile([0.5, 0.0, 1.0], (n, 1))
    g = np.stack([x, np.full(n, 2.0)], axis=1)
    return MGBProblem((main, feas), f, g, Q, geom)
