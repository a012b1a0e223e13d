#!/bin/bash
# End-of-round records, all measured on ONE build: PMC model of the headline kernel, PMC models of the other configurations, the
# configuration table with its roofline objects, the bench lines and the rocprof kernel statistics of the bench command.
# Outputs under gpurun_out/r03_final/ (copy into profiles/ afterwards: tools/experiments/r03_final_collect.sh).
set -o pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r03_final; rm -rf $OUT; mkdir -p $OUT
echo "== headline PMC (the bench workload itself: 1000 spp)"; SPP=1000 tools/experiments/r03_headline_pmc.sh final > $OUT/headline_pmc.log 2>&1 || { tail -5 $OUT/headline_pmc.log; exit 1; }
cp gpurun_out/r03_pmc_final/pt_kernel_model.json profiles/pt_kernel_model.json
echo "== config PMC"; tools/experiments/r03_config_pmc.sh > $OUT/config_pmc.log 2>&1 || { tail -5 $OUT/config_pmc.log; exit 1; }
for c in scene_10 scene_500_c2 cornell cornell_mix c4 c5r; do cp gpurun_out/r03_cfgpmc_$c/model_$c.json profiles/r03/model_$c.json; done
echo "== config bench"; timeout -k 10 900 python3 tools/config_bench.py > $OUT/config_bench.log 2>&1 || { tail -5 $OUT/config_bench.log; exit 1; }
cp gpurun_out/config_bench.json $OUT/config_bench_1gpu.json
echo "== bench records"; tools/experiments/r03_profile.sh > $OUT/profile.log 2>&1 || { tail -5 $OUT/profile.log; exit 1; }
echo "== share scaling"; tools/experiments/r03_share.sh > $OUT/share.log 2>&1 || { tail -5 $OUT/share.log; exit 1; }
echo "== schedule soak"; timeout -k 10 600 python3 tools/schedule_soak.py 60 > $OUT/schedule_soak.log 2>$OUT/schedule_soak.err || { tail -5 $OUT/schedule_soak.log $OUT/schedule_soak.err; exit 1; }
tail -2 $OUT/schedule_soak.log
tail -c 300 gpurun_out/r03_prof/bench_default.json
