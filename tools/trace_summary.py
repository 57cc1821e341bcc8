"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid): python tools/trace_summary.py DIR [min_calls]"""
import csv, glob, sys, collections, re
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"]
    mm = re.search(r"::(\w+)(<[^>]*>)?\(", name)
    key = (mm.group(1) + (mm.group(2) or "")) if mm else name[:50]
    g = (r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Grid_Size_Y", ""), r.get("Workgroup_Size_X", ""))
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    a = acc[(key, g)]
    a[0] += dur; a[1] += 1
tot = sum(v[0] for v in acc.values())
print(f"total kernel time {tot:.1f} us")
for (k, g), (s, c) in sorted(acc.items(), key=lambda kv: -kv[1][0])[:60]:
    print(f"{k[:44]:44s} grid={g[0]:>8s}x{g[1]:>4s} wg={g[2]:>4s} calls={c:5d} avg={s/c:9.1f} us total={s:10.1f} ({100*s/tot:4.1f}%)")
