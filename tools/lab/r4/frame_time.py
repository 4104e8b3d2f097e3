"""Time small frames on one kernel: python3 tools/lab/r4/frame_time.py kernel_id [W H spp depth ...]; prints ms per frame back to back
(one frame slot), latency, and the kernel's own event time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from raytrace_amd import abi, render, world
kernel = int(sys.argv[1])
cases = [tuple(int(x) for x in sys.argv[i:i + 4]) for i in range(2, len(sys.argv), 4)] or [(1024, 1024, 1, 2), (256, 256, 1, 2), (1920, 1080, 1, 2)]
noise = np.fromfile(os.path.join(ROOT, "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
mats, mine = world.generate_region(world.DEFAULT_SEED)
u = render.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, 0.0, 0.0, seed=1)
tag = " ".join("%s=%s" % (k, v) for k, v in sorted(os.environ.items()) if k.startswith("RT_"))
for (W, H, spp, depth) in cases:
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_TIMING)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine); ctx.upload_noise(noise)
        for _ in range(20):
            ctx.draw_frame(u)
        ctx.sync(); ctx.timing()
        n = 200
        t0 = time.perf_counter()
        for _ in range(n):
            ctx.draw_frame(u)
        ctx.sync()
        ms = (time.perf_counter() - t0) * 1e3 / n
        tm = ctx.timing()
        lat = []
        for _ in range(50):
            t0 = time.perf_counter(); ctx.draw_frame(u); ctx.sync(); lat.append((time.perf_counter() - t0) * 1e3)
        print("%s | %dx%d spp %d depth %d kernel %d: %.4f ms/frame, latency %.4f, kernel events %.4f ms" % (tag, W, H, spp, depth, ctx.kernel_in_use(), ms, float(np.median(lat)), tm.trace_ms / max(tm.trace_launches, 1)), flush=True)
