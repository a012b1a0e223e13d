for T in 64 56 48 40 32 16 8; do
  echo -n "TH=$T "
  RTAMD_SM_RESTART=$T timeout -k 10 200 python bench.py --steps 2 --warmup 1 --spp 100 --cpu-spp 0 --kernel 3 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["value"], d["roofline"]["ms_per_launch"])'
done
