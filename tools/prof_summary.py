"""Summarise a rocprofv3 kernel trace: per kernel and per launch configuration."""
import csv, collections, re, glob, sys
d = sys.argv[1]
f = glob.glob(d + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
agg = collections.defaultdict(lambda: [0, 0])
byk = collections.defaultdict(lambda: [0, 0])
for r in rows:
    m = re.search(r"::(\w+)[<(]", r["Kernel_Name"]); k = m.group(1) if m else r["Kernel_Name"][:25]
    g = int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1); gy = int(r["Grid_Size_Y"])
    dt = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg[(k, g, gy)][0] += dt; agg[(k, g, gy)][1] += 1
    byk[k][0] += dt; byk[k][1] += 1
tot = sum(v[0] for v in agg.values())
print("total GPU ms %.1f launches %d" % (tot / 1e6, len(rows)))
for k, v in sorted(byk.items(), key=lambda kv: -kv[1][0])[:22]:
    print(f"  {k:28s} calls={v[1]:7d} total={v[0]/1e6:8.1f}ms avg={v[0]/v[1]/1e3:8.1f}us {100*v[0]/tot:5.1f}%")
print("-- by launch configuration")
for key, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"  {key[0]:24s} gx={key[1]:7d} gy={key[2]:4d} calls={v[1]:6d} total={v[0]/1e6:8.1f}ms avg={v[0]/v[1]/1e3:8.1f}us {100*v[0]/tot:5.1f}%")
