"""Average a rocprofv3 --pmc counter per kernel: python tools/pmc_summary.py DIR COUNTER [name filter...]"""
import csv, glob, sys, collections, re, json
d, counter = sys.argv[1], sys.argv[2]
f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(f)):
    if r.get("Counter_Name") != counter:
        continue
    name = r["Kernel_Name"]
    m = re.search(r"::(\w+)[<(]", name)
    key = m.group(1) if m else name[:40]
    if "elem_kernel" in key:
        mm = re.search(r"elem_kernel<(\d+), *\(?[\w:]*\)?(\d+)>", name)
        if mm: key = f"elem_kernel<{mm.group(1)}, {mm.group(2)}>"
    acc[(key, r.get("Grid_Size", ""))][0] += float(r["Counter_Value"]); acc[(key, r.get("Grid_Size", ""))][1] += 1
out = {}
for (k, g), (s, c) in sorted(acc.items(), key=lambda kv: -kv[1][0]):
    out[f"{k} grid={g}"] = dict(avg=s / c, launches=c)
print(json.dumps(out, indent=1))
