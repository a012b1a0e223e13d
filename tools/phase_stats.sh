#!/bin/bash
# Phase statistics of pt_kernel (kernels 1 / 2) on the BASELINE configurations: wave time and lane utilisation per phase -- node loop,
# leaf tests, materialize, the common part of shade, each material branch -- from a tools-only build (-DRTAMD_PHASE_STATS).
# usage: tools/phase_stats.sh [config ...]   -> gpurun_out/r05_phase/<config>.txt   (is material stream sorting worth building?)
set -o pipefail
cd "$(dirname "$0")/.."
V=rust-raytracer_amd/variants/librtamd_phase.so   # (built here or beforehand: the variants directory travels with gpurun)
if [ ! -f $V ] || [ rust-raytracer_amd/csrc/device/kernels.hip -nt $V ]; then tools/build_variant.sh phase -DRTAMD_PHASE_STATS > /dev/null 2>&1 || { echo "variant build failed"; exit 1; }; fi
OUT=gpurun_out/r05_phase; mkdir -p $OUT
declare -A SPP=( [scene_10]=100 [scene_500]=64 [cornell]=128 [cornell_mix]=128 [c5r]=32 )
for CFG in ${@:-scene_500 cornell cornell_mix c5r}; do
  RTAMD_LIB=$PWD/rust-raytracer_amd/variants/librtamd_phase.so timeout -k 10 300 python3 tools/config_run.py $CFG ${SPP[$CFG]} 2> $OUT/$CFG.err > $OUT/$CFG.json || { tail -5 $OUT/$CFG.err; exit 1; }
  # the warm-up render prints a block too: keep the last one
  grep "^\[phase\]" $OUT/$CFG.err | tail -18 > $OUT/$CFG.txt
  echo "== $CFG"; cat $OUT/$CFG.txt
done
