echo -n "kernel 2: "; python tools/c4_bench.py 32 2>/dev/null | cut -c1-45
for T in 64 48 32 16; do echo -n "kernel 3 TH=$T: "; C4_KERNEL=3 RTAMD_SM_RESTART=$T python tools/c4_bench.py 32 2>/dev/null | cut -c1-45; done
