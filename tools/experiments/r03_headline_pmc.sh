#!/bin/bash
# Round-3 PMC passes of the headline workload (scene_500 1200x1200, SPP spp): separate runs per counter set, summary + model.
# usage: tools/experiments/r03_headline_pmc.sh [tag]   -> gpurun_out/r03_pmc_<tag>/
set -o pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
TAG=${1:-headline}
OUT=gpurun_out/r03_pmc_$TAG; rm -rf $OUT; mkdir -p $OUT
SPP=${SPP:-96}
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $SET | cut -d' ' -f1)
  echo "== pmc $N"
  timeout -k 10 400 rocprofv3 --pmc $SET -d $OUT/pmc_$N --output-format csv -- python3 bench.py --steps 1 --warmup 0 --spp $SPP --cpu-spp 0 > $OUT/pmc_$N.json 2>>$OUT/err.log || exit 1
done
python3 tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_summary_headline.csv
python3 tools/make_pt_model.py --samples $((1200*1200*SPP)) --source "profiles/r03/pmc_summary_headline.csv (rocprofv3 --pmc, separate passes, bench.py --steps 1 --warmup 0 --spp $SPP --cpu-spp 0)" --out $OUT/pt_kernel_model.json $OUT/pmc_*
