#!/bin/bash
# kernel 5 (cooperative instance service) against kernel 2 on C4: parity first, then throughput
set -o pipefail
cd "$(dirname "$0")/../.."
OUT=gpurun_out/r02_k5; mkdir -p $OUT
if [ -z "$SKIP_TESTS" ]; then
  timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "c4 or torus" 2>&1 | tail -15 | tee $OUT/tests.log || exit 1
fi
for K in ${KERNELS:-2 5}; do
  echo "== C4 kernel $K" | tee -a $OUT/k5.log
  C4_KERNEL=$K timeout -k 10 300 python tools/c4_bench.py ${C4_SPP:-64} 2>>$OUT/err.log | cut -c1-120 | tee -a $OUT/k5.log || exit 1
done
tail -20 $OUT/err.log
