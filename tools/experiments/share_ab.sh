#!/bin/bash
# kernel time (best of 3) of one rank's share of the headline frame (WORLDS, default 1 8 16; 1000 spp) for the default build and every variant build
cd "$(dirname "$0")/../.."
for V in default $(ls rust-raytracer_amd/variants/*.so 2>/dev/null); do
  L=$PWD/$V; [ "$V" = default ] && L=$PWD/rust-raytracer_amd/librtamd.so
  echo -n "$(basename $V): "
  for W in ${WORLDS:-1 8 16}; do RTAMD_LIB=$L timeout 200 python tools/share_repeat.py $W 1000 3 2>/dev/null | tail -1 | awk '{printf "w%s %s ms   ", $2, $3}'; done; echo
done
