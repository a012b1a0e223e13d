# A/B: path-trace block size (waves per SIMD) for kernels 2 and 3
for V in "" variants/librtamd_768.so variants/librtamd_1024.so; do
  for K in 2 3; do
    echo -n "lib=${V:-default512} kernel=$K "
    RTAMD_SM_RESTART=56 RTAMD_LIB=${V:+$PWD/rust-raytracer_amd/$V} timeout -k 10 200 python bench.py --steps 2 --warmup 1 --spp 100 --cpu-spp 0 --kernel $K 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"], d["config"]["block_threads"], d["config"]["grid_blocks"])'
  done
done
