#!/bin/bash
# End-of-round records, all measured on ONE build, in one GPU call: PMC model of the headline kernel and of every BASELINE configuration,
# the configuration table with its roofline objects, the bench lines (default, under rocprofv3 --kernel-trace --stats, forced RCCL process
# group at world 1, 2-rank rehearsal, single-process rt_render_multi), the share-scaling rehearsal, the phase statistics.
#   tools/final_records.sh [pmc|rest] on the GPU box (gpurun): writes gpurun_out/final/ (pmc: the counter passes only; rest: everything
#                                     after them, with the models the pmc stage left in profiles/ -- two calls fit gpurun's time limit)
#   tools/final_records.sh collect R  here, afterwards: copies what came back into profiles/rR/ and profiles/pt_kernel_model.json
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
if [ "$1" = collect ]; then
  R=profiles/r$2; mkdir -p $R
  cp gpurun_out/pmc_headline/model_headline.json profiles/pt_kernel_model.json
  for c in headline scene_10 scene_500_c2 cornell cornell_mix c4 c5r c5; do
    [ -f gpurun_out/pmc_$c/model_$c.json ] && cp gpurun_out/pmc_$c/model_$c.json gpurun_out/pmc_$c/pmc_summary_$c.csv $R/
  done
  cp gpurun_out/final/*.json gpurun_out/final/*.csv gpurun_out/final/*.txt $R/ 2>/dev/null
  [ -d gpurun_out/r05_phase ] && for f in gpurun_out/r05_phase/*.txt; do cp $f $R/phase_$(basename $f); done
  ls $R | wc -l
  exit 0
fi
OUT=gpurun_out/final; STAGE=${1:-all}
if [ "$STAGE" = all ] || [ "$STAGE" = pmc ]; then
rm -rf $OUT; mkdir -p $OUT
echo "== headline PMC (the bench workload itself, 1000 spp)"; tools/pmc_passes.sh headline 2>&1 | tail -1 || exit 1
mkdir -p profiles; cp gpurun_out/pmc_headline/model_headline.json profiles/pt_kernel_model.json   # (so that the bench lines below carry this build's model)
echo "== config PMC"; for c in scene_10 scene_500_c2 cornell cornell_mix c4 c5r c5; do tools/pmc_passes.sh $c 2>&1 | tail -1 || exit 1; mkdir -p profiles/r05; cp gpurun_out/pmc_$c/model_$c.json profiles/r05/model_$c.json; done
fi
[ "$STAGE" = pmc ] && exit 0
mkdir -p $OUT
echo "== config bench"; timeout -k 10 900 python3 tools/config_bench.py > $OUT/config_bench.log 2>&1 || { tail -5 $OUT/config_bench.log; exit 1; }
cp gpurun_out/config_bench.json $OUT/config_bench_1gpu.json
echo "== bench records"
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2>>$OUT/err.log || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 bench.py --cpu-spp 0 > $OUT/bench_under_rocprof.json 2>>$OUT/err.log || exit 1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_default.csv; rm -rf $OUT/trace
timeout -k 10 400 python3 bench.py --cpu-spp 0 --force-pg > $OUT/bench_forced_rccl_world1.json 2>>$OUT/err.log || exit 1
RTAMD_BENCH_REHEARSE=1 timeout -k 10 400 python3 bench.py --gpus 2 --cpu-spp 0 --steps 3 > $OUT/bench_rehearsal_2ranks_on_1gpu.json 2>>$OUT/err.log || exit 1
timeout -k 10 400 python3 bench.py --single-process --gpus 1 --cpu-spp 0 > $OUT/bench_single_process_1gpu.json 2>>$OUT/err.log || exit 1
timeout -k 10 400 python3 bench.py --single-process --devices 0,0 --cpu-spp 0 --steps 3 > $OUT/bench_single_process_2ranks_on_1gpu.json 2>>$OUT/err.log || exit 1
python3 - <<PY
import json
for n in ("bench_default", "bench_under_rocprof", "bench_forced_rccl_world1", "bench_rehearsal_2ranks_on_1gpu", "bench_single_process_1gpu", "bench_single_process_2ranks_on_1gpu"):
    d = json.load(open("$OUT/%s.json" % n))
    print("%-40s %8.1f Msamples/s  %7.2f ms/step  frac %s%s" % (n, d["value"], d["ms_per_step"], d["roofline"].get("frac"), "  (rehearsal)" if "rehearsal" in d else ""))
PY
echo "== share scaling (rank 0's share of the headline frame, kernel ms, best of 3)"
for W in 1 2 4 8 16; do timeout 200 python3 tools/share_repeat.py $W 1000 3 2>/dev/null | tail -1; done | tee $OUT/share_scaling.txt
python3 tools/share_ranks.py 2>/dev/null | tee $OUT/share_ranks.txt | tail -3
echo "== phase statistics"; tools/phase_stats.sh > $OUT/phase_stats.log 2>&1; tail -2 $OUT/phase_stats.log
