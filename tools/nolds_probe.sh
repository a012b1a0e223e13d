probe() { python -c "
import sys; sys.path.insert(0,'rust-raytracer_amd')
import rtamd
w,c = rtamd.select_scene('tests/golden/scenes/cube.obj'); w.render(c,width=800,height=800,spp=8)
_,st = w.render(c,width=800,height=800,spp=300); print(round(st['samples']/(st['kernel_ms']*1e-3)/1e6,1), 'lds', st['scene_in_lds'])" 2>/dev/null; }
echo -n "LDS scene: "; probe
echo -n "NO_LDS, top cache: "; RTAMD_NO_LDS=1 probe
echo -n "NO_LDS, no top cache: "; RTAMD_NO_LDS=1 RTAMD_N2_TOP=0 probe
