#!/bin/bash
# Strong-scaling rehearsal records on one GPU: kernel time (best of 3) of rank 0's share of the headline frame at world 1..32 for the
# default build and the variant builds present (tools/build_variant.sh: t0 = -DTAPER_R=0 -DFOLD_PERIOD=0 -DRING_UNITS=6 -DSINGLE_UNITS_BELOW_WAVES=0, the schedule at the start of round 3; tail = -DRT_TAIL_STATS), the end-of-launch statistics of the tail build, and the per-step host overhead.
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_share; rm -rf $OUT; mkdir -p $OUT
WORLDS="1 2 4 8 16 32" tools/share_ab.sh | tee $OUT/share_scaling.txt
if [ -f rust-raytracer_amd/variants/librtamd_tail.so ]; then
  for W in 1 8 16; do RTAMD_LIB=$PWD/rust-raytracer_amd/variants/librtamd_tail.so python tools/share_one.py $W 1000 2>&1 | grep -A4 "measured render" | grep "tail stats. kernel" | sed "s/^/world $W: /"; done | tee $OUT/tail_stats.txt
fi
if [ -f rust-raytracer_amd/variants/librtamd_tail0.so ]; then
  for W in 1 8 16; do RTAMD_LIB=$PWD/rust-raytracer_amd/variants/librtamd_tail0.so python tools/share_one.py $W 1000 2>&1 | grep -A4 "measured render" | grep "tail stats. kernel" | sed "s/^/world $W (uniform schedule, no early fold): /"; done | tee -a $OUT/tail_stats.txt
fi
for W in 1 8; do python tools/step_wall.py $W 4 2>/dev/null | tail -1; done | tee $OUT/step_wall.txt
