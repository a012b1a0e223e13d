#!/bin/bash
# schedule statistics of the cooperative kernel on C4 (stats build of the library; see tools/build_variant.sh)
cd "$(dirname "$0")/../.."
for T in ${TORI:-160x320 20x40}; do
  echo "== torus $T"
  RTAMD_LIB=$PWD/rust-raytracer_amd/variants/librtamd_stats.so C4_TORUS=$T C4_KERNEL=5 timeout -k 10 200 python tools/c4_bench.py 64 2>&1 | grep -v amdgpu.ids | cut -c1-110 | tail -16
done
