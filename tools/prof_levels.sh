# kernel trace of ten fine-level factorizations + solves, summarised per kernel and launch configuration
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/pl -- python3 $R/tools/gpu_solver_levels.py 9 1.0 10 "{}" > $O/prof_levels.out 2>&1
python3 $R/tools/prof_summary.py /tmp/pl 60 > $O/prof_levels.txt 2>&1
tail -70 $O/prof_levels.txt
