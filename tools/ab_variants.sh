# A/B of kernel build variants (rust-raytracer_amd/variants/librtamd_<name>.so), interleaved, 2 rounds
for R in 1 2; do for V in $(ls rust-raytracer_amd/variants/*.so); do
  echo -n "$(basename $V) "
  RTAMD_LIB=$PWD/$V timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-spp 0 ${BENCH_ARGS:-} 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["ms_per_launch"],3))'
done; done
