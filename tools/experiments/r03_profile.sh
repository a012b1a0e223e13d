#!/bin/bash
# Round-3 records of the bench command: the default bench line, the same command under rocprofv3 --kernel-trace --stats, the
# forced-RCCL line; copied into profiles/r03/ by hand afterwards (gpurun_out/ is scratch).
set -o pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r03_prof; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2>>$OUT/err.log || exit 1
tail -c 400 $OUT/bench_default.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 bench.py --cpu-spp 0 > $OUT/bench_under_rocprof.json 2>>$OUT/err.log || exit 1
find $OUT/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_bench_default.csv
head -4 $OUT/kernel_stats_bench_default.csv | cut -c1-200
timeout -k 10 400 python3 bench.py --cpu-spp 0 --force-pg > $OUT/bench_forced_rccl_world1.json 2>>$OUT/err.log || exit 1
tail -c 300 $OUT/bench_forced_rccl_world1.json
