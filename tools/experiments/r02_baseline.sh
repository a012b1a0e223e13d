#!/bin/bash
# Round-2 "before" measurements on the round-1 kernels: block-size A/B on the final schedule (headline + C4) and the
# C4 PMC counters (lane utilisation, wait share).  Run on the GPU box: bash tools/r02_baseline.sh
set -o pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r02_baseline
mkdir -p $OUT
for B in 1024 768 512; do
  L=$PWD/rust-raytracer_amd/variants/librtamd_b$B.so
  echo "== headline, PT_BLOCK=$B" | tee -a $OUT/ab_block.log
  RTAMD_LIB=$L timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-spp 0 2>>$OUT/err.log | tee -a $OUT/ab_block.log || exit 1
  echo "== C4, PT_BLOCK=$B" | tee -a $OUT/ab_block.log
  RTAMD_LIB=$L timeout -k 10 300 python tools/c4_bench.py 64 2>>$OUT/err.log | tee -a $OUT/ab_block.log || exit 1
done
# PMC passes for C4 (default build), SQ counters in two sets
timeout -k 10 400 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU \
  -d $OUT/pmc_c4_a --output-format csv -- python3 tools/c4_bench.py 16 > $OUT/pmc_c4_a.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM \
  -d $OUT/pmc_c4_b --output-format csv -- python3 tools/c4_bench.py 16 > $OUT/pmc_c4_b.log 2>&1 || exit 1
timeout -k 10 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum \
  -d $OUT/pmc_c4_c --output-format csv -- python3 tools/c4_bench.py 16 > $OUT/pmc_c4_c.log 2>&1 || exit 1
python tools/pmc_summary.py $OUT/pmc_c4_a $OUT/pmc_c4_b $OUT/pmc_c4_c > $OUT/pmc_c4_summary.csv
cat $OUT/pmc_c4_summary.csv | head -40
