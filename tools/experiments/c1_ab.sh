#!/bin/bash
# C1 (scene_10 400x225x100) kernel time, best of 5, for the default build and every variant build
cd "$(dirname "$0")/../.."
for V in default $(ls rust-raytracer_amd/variants/*.so 2>/dev/null); do
  L=$PWD/$V; [ "$V" = default ] && L=$PWD/rust-raytracer_amd/librtamd.so
  echo -n "$(basename $V): "; RTAMD_LIB=$L python tools/c1_sub.py 2>/dev/null | grep "sub_spp 0" | tr '\n' ' '; echo
done
