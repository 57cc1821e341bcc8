# Round-4 measurement set (one gpurun call; outputs under gpurun_out/final_r04, copied to profiles/r04_* afterwards)
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final_r04
mkdir -p $O
cd $R
python bench.py > $O/bench_L9.json 2> $O/bench_L9.err
echo bench done >> $O/progress
python bench.py --L 7 --p 1.5 --no-traffic --no-p15 --cpu-budget 10 > $O/bench_L7.json 2> $O/bench_L7.err
echo bench7 done >> $O/progress
MGBHIP_LEVEL_TIMING=1 python tools/gpu_bench_quick.py 9 1.0 > $O/newton_levels.txt 2>&1
python tools/setup_timing.py 9 1.0 > $O/setup_timing.json 2>$O/setup_timing.err
REPS=2 python tools/gpu_fem3d.py 6 4.0 '{"max_coarse":500}' > $O/config4.txt 2>&1
MGB_BENCH_REHEARSAL=1 timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --sharded --steps 2 --warmup 1 2>/dev/null | tail -1 > $O/bench_sharded_rehearsal_2ranks_1gpu.json
echo levels done >> $O/progress
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $R/bench.py --no-cpu-baseline --no-traffic --no-p15 --steps 2 > $O/prof_bench.out 2>&1
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/bench_L9_kernel_stats.csv
echo prof1 done >> $O/progress
bash $R/tools/pmc_traffic.sh > $O/pmc_traffic.txt 2>&1
echo pmc done >> $O/progress
REPS=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p4 -- python3 $R/tools/gpu_fem3d.py 6 4.0 '{"max_coarse":500}' > $O/prof_c4.out 2>&1
cp $(find /tmp/p4 -name "*kernel_stats.csv" | head -1) $O/config4_kernel_stats.csv
echo prof4 done >> $O/progress
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p5 -- python3 $R/tools/gpu_spectral.py 32 1.5 > $O/prof_c5.out 2>&1
cp $(find /tmp/p5 -name "*kernel_stats.csv" | head -1) $O/config5_kernel_stats.csv
echo prof5 done >> $O/progress
tail -2 $O/prof_c4.out $O/prof_c5.out
cut -c1-300 $O/bench_L9.json
