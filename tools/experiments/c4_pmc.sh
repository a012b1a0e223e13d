#!/bin/bash
# PMC passes of C4 with the kernel C4_KERNEL picks (separate runs per counter set); summary + instruction-mix model
set -o pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
K=${C4_KERNEL:-5}
OUT=gpurun_out/c4pmc_k$K; rm -rf $OUT; mkdir -p $OUT
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $SET | cut -d' ' -f1)
  echo "== pmc $N"
  C4_KERNEL=$K timeout -k 10 400 rocprofv3 --pmc $SET -d $OUT/pmc_$N --output-format csv -- python3 tools/c4_bench.py 16 > $OUT/pmc_$N.log 2>>$OUT/err.log || exit 1
done
python3 tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_summary_c4_k$K.csv
python3 tools/make_pt_model.py --samples $((1200*1200*18)) --source "C4 (2 + 16 spp), kernel $K" --out $OUT/c4_model_k$K.json $OUT/pmc_*
