"""Structure statistics of k_frame on a frame (counting build, RT_DEBUG_STATS=1 prints the raw dbg_* words):
python3 tools/lab/r4/frame_stats.py W H spp depth"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from raytrace_amd import abi, render, world
W, H, spp, depth = (int(x) for x in sys.argv[1:5])
noise = np.fromfile(os.path.join(ROOT, "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
mats, mine = world.generate_region(world.DEFAULT_SEED)
u = render.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, 0.0, 0.0, seed=1)
cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_FRAME, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS)
with render.Context(cfg) as ctx:
    ctx.upload_world(mats, mine); ctx.upload_noise(noise)
    ctx.draw_frame(u); ctx.sync()
    c = ctx.counters().as_dict()
print({k: c[k] for k in ("rays", "rays_primary", "rays_shadow", "rays_diffuse", "iterations", "pixels")})
