#!/bin/bash
# kernel 4 (8-wide quantised BVH) against kernel 2 (BVH2): C4, the headline scene and the Cornell box
set -o pipefail
cd "$(dirname "$0")/.."
OUT=gpurun_out/r02_k4; mkdir -p $OUT
for K in 2 4; do
  echo "== C4 kernel $K" | tee -a $OUT/k4.log
  C4_KERNEL=$K timeout -k 10 300 python tools/c4_bench.py ${C4_SPP:-64} 2>>$OUT/err.log | tee -a $OUT/k4.log || exit 1
  echo "== headline kernel $K" | tee -a $OUT/k4.log
  timeout -k 10 300 python bench.py --steps 2 --warmup 1 --cpu-spp 0 --kernel $K 2>>$OUT/err.log | tee -a $OUT/k4.log || exit 1
  echo "== cornell kernel $K" | tee -a $OUT/k4.log
  timeout -k 10 300 python -c "
import sys; sys.path.insert(0,'rust-raytracer_amd')
import rtamd
w,c = rtamd.select_scene('tests/golden/scenes/cube.obj'); w.render(c,width=800,height=800,spp=8,kernel=$K)
_,st = w.render(c,width=800,height=800,spp=500,kernel=$K); print(round(st['samples']/(st['kernel_ms']*1e-3)/1e6,1), 'lds', st['scene_in_lds'], 'kernel', st['kernel_used'])" 2>>$OUT/err.log | tee -a $OUT/k4.log || exit 1
done
