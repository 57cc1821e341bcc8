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
