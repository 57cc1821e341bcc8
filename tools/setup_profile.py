"""Where the host setup time goes (row f4): cProfile of subdivide / amg / assemble at fem2d_P2 level L, then the upload and
the first solve with the library's own plan / analysis timings (MGBHIP_DEBUG=2 on stderr).
Usage: python tools/setup_profile.py [L] [p] [top]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def profiled(label, fn, top):
    pr = cProfile.Profile()
    t = time.perf_counter()
    pr.enable()
    out = fn()
    pr.disable()
    print(f"== {label}: {time.perf_counter() - t:.3f} s", flush=True)
    pstats.Stats(pr).sort_stats("tottime").print_stats(top)
    return out


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    p = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 18
    import mgb_amd as m
    g = profiled("subdivide", lambda: m.subdivide(m.fem2d_P2(), L), top)
    mg = profiled("amg", lambda: m.amg(g), top)
    prob = profiled("assemble", lambda: m.assemble(mg, p=p), top)
    import torch
    if not torch.cuda.is_available():
        return
    from mgb_amd.device import DeviceMGBProblem
    from mgb_amd.solve import mgb_driver
    os.environ["MGBHIP_DEBUG"] = "2"
    import torch.cuda
    torch.cuda.init()                      # context creation is not part of the upload
    t = time.perf_counter(); D = DeviceMGBProblem(prob, device_id=0); print(f"== upload: {time.perf_counter() - t:.3f} s", flush=True)
    t = time.perf_counter(); mgb_driver(D); print(f"== first solve: {time.perf_counter() - t:.3f} s", flush=True)
    os.environ.pop("MGBHIP_DEBUG")
    t = time.perf_counter(); mgb_driver(D); print(f"== second solve: {time.perf_counter() - t:.3f} s", flush=True)
    D.close()


if __name__ == "__main__":
    main()
