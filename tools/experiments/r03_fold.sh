#!/bin/bash
# What the ordered fold costs a rank's share (tools-only builds: foldstats = -DRT_FOLD_STATS, foldstats0 = the same on the schedule of the
# start of round 3, nofold = -DRT_NOFOLD_TEST: no fold at all, wrong image, the floor of the timing).
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r03_fold; rm -rf $OUT; mkdir -p $OUT
for V in foldstats foldstats0; do for W in 1 8 16; do
  RTAMD_LIB=$PWD/rust-raytracer_amd/variants/librtamd_$V.so python tools/share_one.py $W 1000 2>&1 | grep -A4 "measured render" | grep "fold stats" | sed "s/^/$V world $W: /"
done; done | tee $OUT/fold_stats.txt
WORLDS="1 8 16" tools/share_ab.sh | grep -v foldstats | tee -a $OUT/fold_stats.txt
