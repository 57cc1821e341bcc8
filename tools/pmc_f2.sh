# SQ counters of the fine-level element kernels (one pass: 8 SQ slots): where do elem_f2_fast's wave cycles go?
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace --output-format csv -d /tmp/pmcsq -- python3 $R/tools/gpu_kernels.py 9 1.0 3 '{}' > $R/gpurun_out/pmc_f2.out 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/pmcsq/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name']
    for key in ('elem_f2_fast', 'elem_f01_fast', 'gather_assemble_kernel', 'elem_kernel'):
        if key in k:
            acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
for key, d in acc.items():
    print(key, {c: round(sum(v) / len(v)) for c, v in d.items()})
PY
