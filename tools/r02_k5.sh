#!/bin/bash
# kernel 5 (cooperative instance service) against kernel 2 on C4 and on the Cornell box (12-triangle instance)
set -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02_k5; mkdir -p $OUT
for K in ${KERNELS:-2 5}; do
  echo "== C4 kernel $K" | tee -a $OUT/k5.log
  C4_KERNEL=$K timeout -k 10 300 python tools/c4_bench.py ${C4_SPP:-64} 2>>$OUT/err.log | cut -c1-120 | tee -a $OUT/k5.log || exit 1
done
for K in 2 5; do
  echo "== cornell kernel $K" | tee -a $OUT/k5.log
  timeout -k 10 300 python -c "
import sys; sys.path.insert(0,'rust-raytracer_amd')
import rtamd
w,c = rtamd.select_scene('tests/golden/scenes/cube.obj'); w.render(c,width=800,height=800,spp=8,kernel=$K)
_,st = w.render(c,width=800,height=800,spp=500,kernel=$K); print(round(st['samples']/(st['kernel_ms']*1e-3)/1e6,1), 'lds', st['scene_in_lds'], 'kernel', st['kernel_used'])" 2>>$OUT/err.log | tee -a $OUT/k5.log || exit 1
done
