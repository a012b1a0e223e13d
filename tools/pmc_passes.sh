#!/bin/bash
# rocprofv3 --pmc passes of ONE workload (separate runs per counter set, as MI355X_MICROARCH.md prescribes), their per-kernel summary and
# the PMC model the `roofline` objects carry over (tools/make_pt_model.py).
#   tools/pmc_passes.sh headline [spp]        the bench workload itself (bench.py --steps 1 --warmup 0 --spp S --cpu-spp 0; default 1000)
#   tools/pmc_passes.sh <config> [spp]        a BASELINE configuration of tools/configs.py through tools/config_run.py (2-spp warm-up + S spp)
# -> gpurun_out/pmc_<name>/{pmc_summary_<name>.csv, model_<name>.json}
set -o pipefail
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
NAME=$1
declare -A SPP=( [headline]=1000 [scene_10]=100 [scene_500_c2]=500 [cornell]=2000 [cornell_mix]=2000 [c4]=1000 [c5r]=4000 [c5]=4000 )
declare -A PIX=( [headline]=$((1200*1200)) [scene_10]=$((400*225)) [scene_500_c2]=$((1200*800)) [cornell]=$((800*800)) [cornell_mix]=$((800*800)) [c4]=$((1200*1200)) [c5r]=$((1600*1600)) [c5]=$((1600*1600)) )
S=${2:-${SPP[$NAME]}}
OUT=gpurun_out/pmc_$NAME; rm -rf $OUT; mkdir -p $OUT
if [ "$NAME" = headline ]; then CMD="python3 bench.py --steps 1 --warmup 0 --spp $S --cpu-spp 0"; N=$(( ${PIX[$NAME]} * S )); else CMD="python3 tools/config_run.py $NAME $S"; N=$(( ${PIX[$NAME]} * (S + 2) )); fi
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  C=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $SET -d $OUT/pmc_$C --output-format csv -- $CMD > $OUT/pmc_$C.json 2>>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
done
python3 tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_summary_$NAME.csv
python3 tools/make_pt_model.py --samples $N --source "profiles/r05/pmc_summary_$NAME.csv (rocprofv3 --pmc, separate passes: $CMD)" --out $OUT/model_$NAME.json $OUT/pmc_* \
  | grep -E "valu_insts_per_sample|lane_util|valu_busy_measured|kernel_ms|hbm_bytes_per|share_sq_wait_any" | tr '\n' ' '
echo " <- $NAME"
