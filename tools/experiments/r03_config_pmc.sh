#!/bin/bash
# PMC model per BASELINE configuration (C1, C2, C3, C4 with the library's automatic kernel): separate rocprofv3 --pmc passes of
# tools/config_run.py, one model file each (tools/make_pt_model.py).   usage: tools/experiments/r03_config_pmc.sh [config ...]
set -o pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
declare -A SPP=( [scene_10]=100 [scene_500_c2]=500 [cornell]=2000 [cornell_mix]=2000 [c4]=1000 [c5r]=500 )  # the spp tools/config_bench.py times them with
declare -A PIX=( [scene_10]=$((400*225)) [scene_500_c2]=$((1200*800)) [cornell]=$((800*800)) [cornell_mix]=$((800*800)) [c4]=$((1200*1200)) [c5r]=$((1600*1600)) )
for CFG in ${@:-scene_10 scene_500_c2 cornell cornell_mix c4 c5r}; do
  OUT=gpurun_out/r03_cfgpmc_$CFG; rm -rf $OUT; mkdir -p $OUT
  S=${SPP[$CFG]}
  for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
             "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
             "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "FETCH_SIZE" "WRITE_SIZE"; do
    N=$(echo $SET | cut -d' ' -f1)
    timeout -k 10 400 rocprofv3 --pmc $SET -d $OUT/pmc_$N --output-format csv -- python3 tools/config_run.py $CFG $S > $OUT/pmc_$N.json 2>>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
  done
  python3 tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_summary_$CFG.csv
  # samples of the pt_kernel dispatches of one pass: the 2-spp warm-up and the S-spp render
  python3 tools/make_pt_model.py --samples $(( ${PIX[$CFG]} * (S + 2) )) --source "profiles/r03/pmc_summary_$CFG.csv (rocprofv3 --pmc, separate passes, tools/config_run.py $CFG $S: 2-spp warm-up + $S spp)" \
      --out $OUT/model_$CFG.json $OUT/pmc_* | grep -E "valu_insts_per_sample|lane_util|valu_busy|kernel_ms|hbm_bytes_per|\"kernel\"" -A0 | tr '\n' ' '
  echo " <- $CFG"
done
