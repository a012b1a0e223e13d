#!/bin/bash
# C4 against the BVH builder's knobs (items per leaf, SAH box cost), kernels 2 and 5
cd "$(dirname "$0")/../.."
for T in "" max_leaf=1 max_leaf=2 max_leaf=3 max_leaf=4 max_leaf=2,sah_box_cost=0.5 max_leaf=2,sah_box_cost=2.0 max_leaf=4,sah_box_cost=0.5 max_leaf=4,sah_box_cost=2.0; do for K in 5 2; do
  echo -n "[$T] k$K "
  C4_TUNING=$T C4_KERNEL=$K timeout -k 10 200 python tools/c4_bench.py 64 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["msamples_per_s"],1), round(d["kernel_ms"],2), d["info"]["accel_nodes"], d["info"]["accel_stack"])'
done; done
