# A/B of kernel build variants (rust-raytracer_amd/variants/librtamd_<name>.so) against the default build, interleaved, 2 rounds
for R in 1 2; do for V in default $(ls rust-raytracer_amd/variants/*.so); do
  echo -n "$(basename $V) "
  L=$PWD/$V; [ "$V" = default ] && L=$PWD/rust-raytracer_amd/librtamd.so
  RTAMD_LIB=$L timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-spp 0 ${BENCH_ARGS:-} 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["ms_per_launch"],3))'
done; done
