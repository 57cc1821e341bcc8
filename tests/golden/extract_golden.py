"""Transcribe the reference's golden solution vectors into JSON fixtures.

Development-time helper (reads the reference's *test* sources as text; nothing is
executed and no source text is stored): every `z = reshape([...], (:, 2))` literal in
test/runtests.jl:13-32 and test/test_algebraic.jl:38-69 is parsed to numbers and
written, with the problem description that produces it, to tests/golden/golden.json.
Run from the repo root:  python tests/golden/extract_golden.py
"""
import json
import os
import re

REF = "/root/reference/test"


def _literals(path):
    txt = open(path).read()
    out = []
    for m in re.finditer(r"z\s*=\s*reshape\((?:Float64)?\[([^\]]*)\]\s*,\s*\(:\s*,\s*2\)\)", txt):
        vals = [float(v) for v in m.group(1).replace("\n", " ").split(",") if v.strip()]
        line = txt[: m.start()].count("\n") + 1
        out.append((line, vals))
    return out


def _parabolic_literals(path):
    """`z = [a b c; d e f;;; ...]` 3-D literals followed by a `parabolic_solve(...)` call
    (test/runtests.jl:35-52): rows = nodes, columns = (u, s1, s2), slices = time stamps."""
    txt = open(path).read()
    out = []
    for m in re.finditer(r"z\s*=\s*\[([^\]]*;;;[^\]]*)\]\s*\n\s*sol\s*=\s*parabolic_solve\(([^\n]*)\)\n", txt):
        slices = []
        for sl in m.group(1).split(";;;"):
            rows = [[float(v) for v in r.split()] for r in sl.split(";") if r.strip()]
            slices.append(rows)
        line = txt[: m.start()].count("\n") + 1
        out.append((line, slices, m.group(2)))
    return out


def main():
    par = _parabolic_literals(os.path.join(REF, "runtests.jl"))
    names_par = [("fem1d_3nodes", dict(geom="fem1d", nodes=3)), ("fem2d_P2_L1", dict(geom="fem2d_P2", L=1)),
                 ("spectral1d_n4", dict(geom="spectral1d", n=4)), ("spectral2d_n4", dict(geom="spectral2d", n=4))]
    assert len(par) == len(names_par), len(par)
    pcases = []
    for (name, desc), (line, slices, call) in zip(names_par, par):
        assert "h=0.5" in call and "p=1.0" in call, call
        pcases.append(dict(name=name, source=f"test/runtests.jl:{line}", tol=1e-6, h=0.5, p=1.0, u=slices, **desc))
    here0 = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here0, "golden_parabolic.json"), "w") as fh:
        json.dump(dict(note="Golden parabolic_solve trajectories transcribed from the reference's test sources "
                            "(u[time][node][component]; criterion norm(cat(u) - gold) < tol).", cases=pcases), fh, indent=1)
    print("wrote", len(pcases), "parabolic cases")
    rt = _literals(os.path.join(REF, "runtests.jl"))
    ta = _literals(os.path.join(REF, "test_algebraic.jl"))
    cases = []
    names_rt = [
        ("fem1d_3nodes_p1", dict(geom="fem1d", nodes=3, L=1, p=1.0)),
        ("fem2d_P2_L1_p1", dict(geom="fem2d_P2", L=1, p=1.0)),
        ("spectral1d_n5_p1", dict(geom="spectral1d", n=5, p=1.0)),
        ("spectral2d_n5_p1", dict(geom="spectral2d", n=5, p=1.0)),
    ]
    for (name, desc), (line, vals) in zip(names_rt, rt):
        cases.append(dict(name=name, source=f"test/runtests.jl:{line}", tol=1e-6, z_colmajor=vals, ncols=2, **desc))
    names_ta = [
        ("fem1d_5nodes_p1", dict(geom="fem1d", nodes=5, L=1, p=1.0)),
        ("fem1d_5nodes_p1.5", dict(geom="fem1d", nodes=5, L=1, p=1.5)),
        ("fem2d_P1_L2_p1", dict(geom="fem2d_P1", L=2, p=1.0)),
        ("fem2d_P1_L2_p1.5", dict(geom="fem2d_P1", L=2, p=1.5)),
        ("fem2d_P2_L2_p1", dict(geom="fem2d_P2", L=2, p=1.0)),
        ("fem2d_P2_L2_p1.5", dict(geom="fem2d_P2", L=2, p=1.5)),
        ("fem3d_k1_L2_p1", dict(geom="fem3d", k=1, L=2, p=1.0)),
        ("fem3d_k1_L2_p1.5", dict(geom="fem3d", k=1, L=2, p=1.5)),
    ]
    assert len(ta) == len(names_ta), len(ta)
    for (name, desc), (line, vals) in zip(names_ta, ta):
        cases.append(dict(name=name, source=f"test/test_algebraic.jl:{line}", tol=1e-6, z_colmajor=vals, ncols=2, **desc))
    here = os.path.dirname(os.path.abspath(__file__))
    with open(os.path.join(here, "golden.json"), "w") as fh:
        json.dump(dict(note="Golden end-to-end solutions transcribed from the reference's test sources "
                            "(default f, g, tolerances; criterion norm(z - gold) < tol).", cases=cases), fh, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
