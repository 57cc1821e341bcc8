# copy the outputs of tools/final_measure_r03.sh (merged into gpurun_out/final_r03 by gpurun) to profiles/r03_*
set -e
cd "$(dirname "$0")/.."
O=gpurun_out/final_r03
cp $O/bench_L9.json profiles/r03_bench_L9_p1.0.json
cp $O/bench_L7.json profiles/r03_bench_L7_p1.5.json
cp $O/bench_L9_kernel_stats.csv profiles/r03_bench_L9_kernel_stats.csv
cp $O/pmc_traffic.txt profiles/r03_pmc_traffic_L9.txt
cp $O/newton_levels.txt profiles/r03_newton_path_levels_L9.txt
cp $O/solver_levels.txt profiles/r03_solver_levels_L9.txt
cp $O/setup_timing.json profiles/r03_setup_timing_L9.json
cp $O/default_ladder_p15.txt profiles/r03_default_ladder_p15.txt
cp $O/config4.txt profiles/r03_config4_solve.txt
cp $O/config4_kernel_stats.csv profiles/r03_config4_fem3d_L6_kernel_stats.csv
cp $O/config5_kernel_stats.csv profiles/r03_config5_spectral32_kernel_stats.csv
cp $O/bench_sharded_rehearsal_2ranks_1gpu.json profiles/r03_bench_sharded_rehearsal_2ranks_1gpu.json
for n in bench_L9 config4_fem3d_L6 config5_spectral32; do python tools/stats_to_md.py profiles/r03_${n}_kernel_stats.csv > profiles/r03_${n}_kernel_stats_top.txt; done
