#!/bin/bash
# usage: tools/build_variant.sh NAME [-DFLAG ...]  -> rust-raytracer_amd/variants/librtamd_NAME.so (A/B builds of the kernels)
set -e
cd "$(dirname "$0")/../rust-raytracer_amd"
name=$1; shift
mkdir -p variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../include -Icsrc --offload-arch=gfx950 "$@" \
  -Rpass-analysis=kernel-resource-usage -c csrc/device/kernels.hip -o variants/kernels_$name.o 2>&1 \
  | grep -E "Function Name|VGPRs:|ScratchSize" | grep -A2 "pt_kernelILb1ELb0ELi2ELi0" | grep -E "VGPRs|Scratch" | sed 's/.*remark: //' | cut -c1-50 | tr '\n' ' '
echo " <- $name"
# the schedule (host/schedule.cpp) takes the same -D switches (-DTAPER_R=...)
g++ -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -I../include -Icsrc $(for f in "$@"; do case $f in -D*) echo $f;; esac; done) -c csrc/host/schedule.cpp -o variants/schedule_$name.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o variants/librtamd_$name.so csrc/abi.o $(ls csrc/host/*.o | grep -v schedule.o) variants/schedule_$name.o variants/kernels_$name.o csrc/device/exchange.o -lrccl
