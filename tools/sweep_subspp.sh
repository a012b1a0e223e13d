for T in 1 2 4 8 16 23 46; do
  echo -n "sub_spp=$T "
  RTAMD_SUB_SPP=$T timeout -k 10 200 python bench.py --steps 2 --warmup 1 --cpu-spp 0 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1), round(d["roofline"]["ms_per_launch"],3))'
done
