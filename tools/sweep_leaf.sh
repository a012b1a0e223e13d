for M in 1 2 4; do
  echo -n "max_leaf=$M  C4: "; RTAMD_MAX_LEAF=$M python tools/c4_bench.py 32 2>/dev/null | cut -c1-45
  echo -n "             scene_500: "; RTAMD_MAX_LEAF=$M python bench.py --steps 2 --warmup 1 --cpu-spp 0 2>/dev/null | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["value"],1))'
  echo -n "             cornell: "; RTAMD_MAX_LEAF=$M python -c "
import sys; sys.path.insert(0,'rust-raytracer_amd')
import rtamd
w,c = rtamd.select_scene('tests/golden/scenes/cube.obj'); w.render(c,width=800,height=800,spp=8)
_,st = w.render(c,width=800,height=800,spp=500); print(round(st['samples']/(st['kernel_ms']*1e-3)/1e6,1))" 2>/dev/null
done
