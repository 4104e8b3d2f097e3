"""Draw N frames of one small configuration on one kernel (for rocprofv3 passes): python3 tools/lab/r4/frame_loop.py W H spp depth kernel_id nframes"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from raytrace_amd import abi, render, world
W, H, spp, depth, kernel, n = (int(x) for x in sys.argv[1:7])
noise = np.fromfile(os.path.join(ROOT, "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
mats, mine = world.generate_region(world.DEFAULT_SEED)
u = render.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, 0.0, 0.0, seed=1)
cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=abi.RT_FLAG_CACHE_PRIMARY)
with render.Context(cfg) as ctx:
    ctx.upload_world(mats, mine); ctx.upload_noise(noise)
    for _ in range(n):
        ctx.draw_frame(u)
        ctx.sync()
    print("kernel in use", ctx.kernel_in_use())
