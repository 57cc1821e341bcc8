"""The threaded parts of the symbolic analysis (csrc/mf_analysis.cpp: clique tests and neighbourhood hashes of the
peeling rounds, relative indices and A scatter lists) must reproduce the serial plan array for array.  The thread
count is read once per process (MGBHIP_ANALYZE_THREADS), hence the subprocesses; the hash covers every array of
MfPlan (oracle/csrc/mf_host.cpp: mf_host_plan_hash -- the checker's build of the product's analysis)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hash(threads, nx):
    env = dict(os.environ, MGBHIP_ANALYZE_THREADS=str(threads))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "plan_hash_worker.py"), str(nx)], env=env,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    n, h, _ = out.stdout.split()
    return int(n), h


def test_threaded_analysis_reproduces_the_serial_plan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    n1, h1 = _hash(1, 300)          # 180 000 unknowns: above the threading threshold of the analysis
    n4, h4 = _hash(4, 300)
    n7, h7 = _hash(7, 300)
    assert n1 == n4 == n7 == 180000
    assert h1 != "0x0" and h1 == h4 == h7


def test_plan_of_the_fem2d_P2_fine_level_is_pinned_and_thread_independent():
    """The real thing rather than a synthetic stencil: the fine-level Hessian pattern of fem2d_P2 at L = 7 (81 665 unknowns:
    8 192 slack groups and as many bubbles peeled in rounds 1 and 2, whose structures the analysis forms on host threads since
    round 4).  The plan hash is pinned to the value the serial symbolic loop of round 3 produced, for one and for eight threads."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
    for threads in (1, 8):
        env = dict(os.environ, MGBHIP_ANALYZE_THREADS=str(threads))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dev", "plan_hash_fem_worker.py"), "7"], env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr
        rows = [ln.split() for ln in out.stdout.strip().splitlines()]
        assert [r[5] for r in rows] == ["0x3bea6e2ad675678e", "0xd891e7d4b075eea5"], rows
