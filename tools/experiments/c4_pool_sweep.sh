#!/bin/bash
# C4 against the number of parked-path slots per workgroup (rt_tuning.coop_pool): the pool's records share L2 with the mesh
cd "$(dirname "$0")/../.."
for R in 1 2; do for P in 1024 768 512 384 256 192 128; do
  echo -n "coop_pool=$P "
  C4_TUNING=coop_pool=$P C4_KERNEL=5 timeout -k 10 200 python tools/c4_bench.py ${C4_SPP:-128} 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["msamples_per_s"],1), round(d["kernel_ms"],2))'
done; done
