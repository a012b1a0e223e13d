#!/bin/bash
# A/B of kernel build variants (rust-raytracer_amd/variants/librtamd_<name>.so) on C4, interleaved; C4_KERNEL picks the kernel
cd "$(dirname "$0")/../.."
OUT=gpurun_out/ab_c4; mkdir -p $OUT
for R in 1 2; do for V in default $(ls rust-raytracer_amd/variants/*.so); do
  L=$PWD/$V; [ "$V" = default ] && L=$PWD/rust-raytracer_amd/librtamd.so
  echo -n "$(basename $V) k${C4_KERNEL:-5} " | tee -a $OUT/ab.log
  RTAMD_LIB=$L C4_KERNEL=${C4_KERNEL:-5} timeout -k 10 200 python tools/c4_bench.py ${C4_SPP:-128} 2>>$OUT/err.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["msamples_per_s"],1), round(d["kernel_ms"],2))' | tee -a $OUT/ab.log || exit 1
done; done
echo -n "default k2 " | tee -a $OUT/ab.log
C4_KERNEL=2 timeout -k 10 200 python tools/c4_bench.py ${C4_SPP:-128} 2>>$OUT/err.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["msamples_per_s"],1), round(d["kernel_ms"],2))' | tee -a $OUT/ab.log
grep "coop stats" $OUT/err.log | tail -12
