"""Idle time of the GPU after the kernels the host waits for (finish_kernel: the read-backs of the Newton loop) from a
`rocprofv3 --kernel-trace --output-format csv` directory: gap between the end of that kernel and the start of the next one.
Usage: python tools/trace_gaps.py DIR"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
if len(sys.argv) > 2:            # keep the last N kernels (e.g. the second, warm solve of TWICE=1 tools/gpu_try.py)
    rows = rows[-int(sys.argv[2]):]
gaps = {}
allgap = 0
for a, b in zip(rows, rows[1:]):
    g = int(b["Start_Timestamp"]) - int(a["End_Timestamp"])
    name = a["Kernel_Name"].replace("(anonymous namespace)::", "").replace("mgbhip::", "").replace("void ", "").split("(")[0].split("<")[0][:40]
    gaps.setdefault(name, []).append(g)
    allgap += max(g, 0)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
span = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"kernels {len(rows)} busy {busy/1e6:.1f} ms span {span/1e6:.1f} ms idle between kernels {allgap/1e6:.1f} ms")
for name, g in sorted(gaps.items(), key=lambda kv: -sum(kv[1]))[:12]:
    g2 = sorted(g)
    print(f"{name:42s} n={len(g):6d} gap after: median {g2[len(g2)//2]/1e3:7.1f} us  mean {sum(g)/len(g)/1e3:7.1f} us  total {sum(g)/1e6:7.1f} ms")
