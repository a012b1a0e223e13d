"""Wall time per bench step of ONE rank's share on one GPU (device-resident path: rt_render_tiles_device + rt_assemble_frame_device, no
gather): what the host adds to the kernel time per step at world W.  argv: world [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "rust-raytracer_amd"))
import torch
import rtamd
from rtamd.distributed import TileLayout
W = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
world, cam = rtamd.load_scene_file(os.path.join(ROOT, "tests", "golden", "scenes", "scene_500.json"))
lay = TileLayout(1200, 1200, W)
p = rtamd.default_params(width=1200, height=1200, spp=1000, seed=1, rank=0, world=W)
p0 = rtamd.default_params(width=1200, height=1200, rank=0, world=W)
dev = torch.device("cuda:0")
d_tiles = torch.zeros(lay.stride * 64 * 3, dtype=torch.float64, device=dev)
gathered = torch.zeros(W * lay.stride * 64 * 3, dtype=torch.float64, device=dev)
frame = torch.zeros(1200 * 1200 * 3, dtype=torch.float64, device=dev)
stream = torch.cuda.current_stream().cuda_stream
def step():
    st = world.render_tiles_device(cam, p, d_tiles.data_ptr(), stream)
    gathered[: d_tiles.numel()].copy_(d_tiles)  # stands in for the gather's local copy
    rtamd.assemble_frame_device(p0, gathered.data_ptr(), lay.stride, frame.data_ptr(), stream)
    return st["kernel_ms"]
step(); torch.cuda.synchronize()
t0 = time.perf_counter(); k = 0.0
for _ in range(steps):
    k += step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) * 1e3 / steps
print("world %d: wall %.2f ms per step, pt kernel %.2f ms, host + finalize + assemble %.2f ms" % (W, dt, k / steps, dt - k / steps))
