"""Condense a rocprofv3 *kernel_stats.csv into a short text table: python tools/stats_to_md.py CSV [rows]"""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over {sum(int(r['Calls']) for r in rows)} launches")
for r in rows[:top]:
    m = re.search(r"(\w+)(<[^>]*>)?\(", r["Name"])
    n = (m.group(1) + (m.group(2) or ""))[:52] if m else r["Name"][:52]
    print(f"{n:52s} calls={r['Calls']:>7s} avg_us={float(r['AverageNs'])/1e3:9.1f} total_ms={float(r['TotalDurationNs'])/1e6:9.1f} pct={float(r['Percentage']):5.1f}")
