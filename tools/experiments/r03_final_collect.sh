#!/bin/bash
# copy what tools/experiments/r03_final.sh measured (merged back into gpurun_out/) into the tracked profiles/ tree
cd "$(dirname "$0")/../.."
cp gpurun_out/r03_pmc_final/pt_kernel_model.json profiles/pt_kernel_model.json
cp gpurun_out/r03_pmc_final/pmc_summary_headline.csv profiles/r03/pmc_summary_headline.csv
for c in scene_10 scene_500_c2 cornell cornell_mix c4 c5r; do
  cp gpurun_out/r03_cfgpmc_$c/model_$c.json profiles/r03/model_$c.json
  cp gpurun_out/r03_cfgpmc_$c/pmc_summary_$c.csv profiles/r03/pmc_summary_$c.csv
done
cp gpurun_out/r03_final/config_bench_1gpu.json profiles/r03/config_bench_1gpu.json
cp gpurun_out/r03_prof/bench_default.json gpurun_out/r03_prof/bench_under_rocprof.json gpurun_out/r03_prof/kernel_stats_bench_default.csv gpurun_out/r03_prof/bench_forced_rccl_world1.json profiles/r03/
cp gpurun_out/r03_share/share_scaling.txt gpurun_out/r03_share/tail_stats.txt gpurun_out/r03_share/step_wall.txt profiles/r03/
cp gpurun_out/r03_final/schedule_soak.log profiles/r03/schedule_soak.log
cp gpurun_out/r03_final/c4_share.txt profiles/r03/c4_share.txt
