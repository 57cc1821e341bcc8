# quick check after a solver-kernel change: solver tests, per-level timings, headline solve, spectral config
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_solver.py tests/test_plan_device.py -x -q -m gpu > $O/tests.txt 2>&1 && \
MGBHIP_DEBUG=2 timeout -k 10 300 python tools/gpu_solver_levels.py 9 1.0 10 "{}" > $O/solver_levels.txt 2>&1 && \
timeout -k 10 300 python tools/gpu_bench_quick.py 9 1.0 > $O/quick_L9.txt 2>&1 && \
timeout -k 10 300 python tools/gpu_spectral.py 32 1.5 > $O/spectral.txt 2>&1
tail -3 $O/tests.txt; grep -A5 "^factor" $O/solver_levels.txt; tail -4 $O/quick_L9.txt; tail -3 $O/spectral.txt
