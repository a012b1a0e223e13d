cd /root/repo
for T in 20x40 40x80 80x160 160x320 320x640; do for K in 2 5; do echo -n "$T k$K "; C4_TORUS=$T C4_KERNEL=$K timeout -k 10 200 python tools/c4_bench.py 64 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["msamples_per_s"],1), round(d["kernel_ms"],2), d["info"]["n_tris"], d["info"].get("accel_stack"))'; done; done
