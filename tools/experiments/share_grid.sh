#!/bin/bash
# kernel time of rank 0's share for (world, spp) pairs: separates "few tiles per rank" from "short launch".  RTAMD_LIB picks the build.
cd "$(dirname "$0")/../.."
for P in "$@"; do W=${P%%:*}; S=${P##*:}; timeout 300 python tools/share_one.py $W $S 2>/dev/null | tail -1; done
