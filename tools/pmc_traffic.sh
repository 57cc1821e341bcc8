# HBM traffic per kernel of one warm L=9 solve: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB; FETCH is doubled
# when converted to bytes on gfx950, MI355X_MICROARCH.md).  Output: gpurun_out/pmc_traffic_{FETCH,WRITE}.json
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmct_$C
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmct_$C -- python3 $R/tools/gpu_try.py 9 1.0 '{}' > $R/gpurun_out/pmc_traffic_$C.out 2>&1
  python3 $R/tools/pmc_summary.py /tmp/pmct_$C $C > $R/gpurun_out/pmc_traffic_$C.json
done
python3 - <<PY
import json
F = json.load(open("$R/gpurun_out/pmc_traffic_FETCH_SIZE.json")); W = json.load(open("$R/gpurun_out/pmc_traffic_WRITE_SIZE.json"))
rows = []
for k in F:
    if k in W:
        rows.append((k, 1024 * (2 * F[k]["avg"] + W[k]["avg"]), F[k]["launches"]))
rows.sort(key=lambda r: -r[1] * r[2])
for k, b, n in rows[:40]:
    print(f"{k:60s} launches {n:6d}  bytes/launch {b/1e6:9.2f} MB")
PY
