#!/bin/bash
# Round-2 profiles of the bench workload (headline: scene_500 1200x1200) and of C4, for profiles/r02/ and profiles/pt_kernel_model.json.
# PMC passes are separate runs (SQ set, GRBM, FETCH_SIZE, WRITE_SIZE); --kernel-trace/--stats in their own run.
set -o pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=gpurun_out/r02_prof; rm -rf $OUT; mkdir -p $OUT
SPP=${SPP:-96}
echo "== bench default" | tee $OUT/log.txt
timeout -k 10 400 python3 bench.py > $OUT/bench_default.json 2>>$OUT/err.log || exit 1
tail -c 600 $OUT/bench_default.json | tee -a $OUT/log.txt
echo "== kernel trace of the same command" | tee -a $OUT/log.txt
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 bench.py --cpu-spp 0 > $OUT/bench_under_rocprof.json 2>>$OUT/err.log || exit 1
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_IFETCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $SET | cut -d' ' -f1)
  echo "== pmc $N" | tee -a $OUT/log.txt
  timeout -k 10 400 rocprofv3 --pmc $SET -d $OUT/pmc_$N --output-format csv -- python3 bench.py --steps 1 --warmup 0 --spp $SPP --cpu-spp 0 > $OUT/pmc_$N.json 2>>$OUT/err.log || exit 1
done
python3 tools/pmc_summary.py $OUT/pmc_* > $OUT/pmc_summary_headline.csv
python3 tools/make_pt_model.py --samples $((1200*1200*SPP)) --source "profiles/r02/pmc_summary_headline.csv (rocprofv3 --pmc, separate passes, bench.py --steps 1 --warmup 0 --spp $SPP --cpu-spp 0)" --out $OUT/pt_kernel_model.json $OUT/pmc_* | tee -a $OUT/log.txt
echo "== C4 (Cornell box + 102,400-triangle instance, 1200x1200): kernel 2 and the automatic choice (kernel 5)" | tee -a $OUT/log.txt
for K in 2 0; do
  C4_KERNEL=$K timeout -k 10 300 python3 tools/c4_bench.py 128 2>>$OUT/err.log | cut -c1-130 | tee -a $OUT/log.txt
done
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/c4trace --output-format csv -- python3 tools/c4_bench.py 128 > $OUT/c4_under_rocprof.log 2>>$OUT/err.log || exit 1
for SET in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" \
           "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64" \
           "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "FETCH_SIZE" "WRITE_SIZE"; do
  N=$(echo $SET | cut -d' ' -f1)
  timeout -k 10 400 rocprofv3 --pmc $SET -d $OUT/c4pmc_$N --output-format csv -- python3 tools/c4_bench.py 16 > $OUT/c4pmc_$N.log 2>>$OUT/err.log || exit 1
done
python3 tools/pmc_summary.py $OUT/c4pmc_* > $OUT/pmc_summary_c4.csv
# the model counts the pt_kernel_coop dispatches only (16 spp); the 2-spp warm-up of c4_bench.py runs the same kernel
python3 tools/make_pt_model.py --kernel pt_kernel_coop --samples $((1200*1200*18)) --source "C4 (2 + 16 spp, kernel 5), profiles/r02/pmc_summary_c4.csv" --out $OUT/c4_model.json $OUT/c4pmc_* | tee -a $OUT/log.txt
echo "== configs" | tee -a $OUT/log.txt
timeout -k 10 600 python3 tools/config_bench.py > $OUT/config_bench.log 2>>$OUT/err.log || exit 1
cp gpurun_out/config_bench.json $OUT/ 2>/dev/null
tail -8 $OUT/config_bench.log | cut -c1-200
