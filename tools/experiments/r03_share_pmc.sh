#!/bin/bash
# PMC pass of ONE rank's share of the headline frame (tools/share_one.py WORLD 1000): what a share executes per sample compared with the
# whole frame (profiles/pt_kernel_model.json).  usage: tools/experiments/r03_share_pmc.sh [world ...]
set -o pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for W in ${@:-8}; do
  OUT=gpurun_out/r03_sharepmc_w$W; rm -rf $OUT; mkdir -p $OUT
  for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
             "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
    N=$(echo $SET | cut -d' ' -f1)
    timeout -k 10 300 rocprofv3 --pmc $SET -d $OUT/pmc_$N --output-format csv -- python3 tools/share_one.py $W 1000 > $OUT/pmc_$N.json 2>>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
  done
  python3 tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_summary_share_w$W.csv
  echo "world $W:"; cat $OUT/pmc_summary_share_w$W.csv | head -40
done
