"""Draws one frame and runs the post passes N times; meant to run under `rocprofv3 --kernel-trace --stats`
(tools/post_kstats.sh) so that k_denoise_pass / k_finalize get per-launch durations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from raytrace_amd import abi, render, world

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
N = int(sys.argv[3]) if len(sys.argv) > 3 else 10
noise = np.fromfile(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
mats, mine = world.generate_region()
u = render.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, 0.0, 0.0, 1)
cfg = render.make_config(W, H, spp=1, depth=2, flags=abi.RT_FLAG_CACHE_PRIMARY)
with render.Context(cfg) as ctx:
    ctx.upload_world(mats, mine)
    ctx.upload_noise(noise)
    ctx.draw_frame(u)
    for _ in range(N):
        ctx.denoise(True)
        ctx.finalize()
    ctx.sync()
print("done", W, H, N)
