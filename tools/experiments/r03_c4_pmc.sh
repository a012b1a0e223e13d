#!/bin/bash
# PMC passes of C4 (Cornell + 102,400-triangle torus, 1200x1200) with the kernel C4_KERNEL picks, SPP spp (+ the 2-spp warm-up of
# c4_bench.py); separate runs per counter set; one model file per device kernel.   usage: tools/experiments/r03_c4_pmc.sh [tag]
set -o pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
K=${C4_KERNEL:-6}
SPP=${SPP:-128}
TAG=${1:-k$K}
OUT=gpurun_out/r03_c4pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $SET | cut -d' ' -f1)
  echo "== pmc $N"
  C4_KERNEL=$K timeout -k 10 400 rocprofv3 --pmc $SET -d $OUT/pmc_$N --output-format csv -- python3 tools/c4_bench.py $SPP > $OUT/pmc_$N.log 2>>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
done
python3 tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_summary_c4_$TAG.csv
S=$((1200*1200*(SPP+2)))
if [ "$K" = "6" ]; then
  python3 tools/make_pt_model.py --kernel pt_kernel_wf --samples $S --source "C4 (2 + $SPP spp), kernel 6: pt_kernel_wf" --out $OUT/c4_model_pt_kernel_wf.json $OUT/pmc_* > $OUT/model_pt.log
  python3 tools/make_pt_model.py --kernel wf_walk_kernel --samples $S --source "C4 (2 + $SPP spp), kernel 6: wf_walk_kernel" --out $OUT/c4_model_wf_walk_kernel.json $OUT/pmc_* > $OUT/model_walk.log
  grep -E "valu_insts_per_sample|lane_util|valu_busy|share_|kernel_ms|hbm_bytes_per" $OUT/model_pt.log $OUT/model_walk.log
else
  python3 tools/make_pt_model.py --kernel pt_kernel --samples $S --source "C4 (2 + $SPP spp), kernel $K" --out $OUT/c4_model_k$K.json $OUT/pmc_* | grep -E "valu_insts_per_sample|lane_util|valu_busy|share_|kernel_ms|hbm_bytes_per"
fi
