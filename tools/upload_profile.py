"""cProfile of the device upload (DeviceMGBProblem) and the phases of the fine-level symbolic analysis (MGBHIP_DEBUG=3).
Usage: python tools/upload_profile.py [L] [p]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    L = int(sys.argv[1]) if len(sys.argv) > 1 else 9
    p = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    import mgb_amd as m
    import torch
    torch.cuda.init()
    from mgb_amd.device import DeviceMGBProblem
    from mgb_amd.solve import mgb_driver
    prob = m.assemble(m.amg(m.subdivide(m.fem2d_P2(), L)), p=p)
    pr = cProfile.Profile()
    t = time.perf_counter()
    pr.enable()
    D = DeviceMGBProblem(prob, device_id=0)
    pr.disable()
    print(f"== upload: {time.perf_counter() - t:.3f} s", flush=True)
    pstats.Stats(pr).sort_stats("tottime").print_stats(16)
    os.environ["MGBHIP_DEBUG"] = "3"
    t = time.perf_counter(); mgb_driver(D); print(f"== first solve: {time.perf_counter() - t:.3f} s", flush=True)
    D.close()


if __name__ == "__main__":
    main()
