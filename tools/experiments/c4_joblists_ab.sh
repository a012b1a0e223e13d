#!/bin/bash
# C4 locality experiment: jobs dealt from ONE counter against per-XCD / per-CU job lists (chunks of consecutive tiles
# dealt round-robin to the lists; tools-only builds -DXCD_JOBS=2 -DXCD_LISTS=.. -DXCD_CHUNK=..).  Kernel rate, L2 hit / miss and fabric bytes per sample for both builds.
# usage: tools/experiments/c4_joblists_ab.sh [spp]  -> gpurun_out/r04_xcd/
set -o pipefail
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
SPP=${1:-256}
OUT=gpurun_out/r04_xcd; rm -rf $OUT; mkdir -p $OUT
for V in xcd2:-DXCD_JOBS=2 l32:"-DXCD_JOBS=2 -DXCD_LISTS=32u -DXCD_CHUNK=64u" l256:"-DXCD_JOBS=2 -DXCD_LISTS=256u -DXCD_CHUNK=16u" l256b:"-DXCD_JOBS=2 -DXCD_LISTS=256u -DXCD_CHUNK=64u"; do
  F=rust-raytracer_amd/variants/librtamd_${V%%:*}.so
  if [ ! -f $F ] || [ rust-raytracer_amd/csrc/device/kernels.hip -nt $F ]; then tools/build_variant.sh ${V%%:*} ${V#*:} > /dev/null 2>&1 || { echo "variant build failed"; exit 1; }; fi
done
for B in product xcd2 l32 l256 l256b; do
  L=$PWD/rust-raytracer_amd/librtamd.so; [ $B != product ] && L=$PWD/rust-raytracer_amd/variants/librtamd_$B.so
  for i in 1 2 3 4; do RTAMD_LIB=$L python3 tools/config_run.py c4 $SPP 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$B run $i: %.1f Msamples/s' % d['msamples_per_s_kernel'])"; done | tee -a $OUT/rates.txt
  for SET in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU"; do
    N=$(echo $SET | cut -d' ' -f1)
    RTAMD_LIB=$L timeout -k 10 300 rocprofv3 --pmc $SET -d $OUT/${B}_$N --output-format csv -- python3 tools/config_run.py c4 $SPP > $OUT/${B}_$N.json 2>>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
  done
  python3 tools/pmc_summary.py $OUT/${B}_* | grep "pt_kernel" > $OUT/pmc_$B.csv
done
python3 - <<PY
import csv
S = 1200 * 1200 * ($SPP + 2)
for b in ("product", "xcd2", "l32", "l256", "l256b"):
    d = {}
    for r in csv.reader(open("$OUT/pmc_%s.csv" % b)):
        d[r[1]] = float(r[4])
    hit, miss = d.get("TCC_HIT_sum", 0), d.get("TCC_MISS_sum", 0)
    print("%-8s TCC hit rate %.3f (hit %.3g miss %.3g), TCC requests / sample %.1f, fabric bytes / sample %.0f, wait share %.3f, VALU / sample %.1f" % (
        b, hit / max(1, hit + miss), hit, miss, d.get("TCC_REQ_sum", 0) / S, 2 * 1024 * d.get("FETCH_SIZE", 0) / S,
        d.get("SQ_WAIT_ANY", 0) / max(1, d.get("SQ_WAVE_CYCLES", 1)), d.get("SQ_INSTS_VALU", 0) / S))
PY
