"""Drives rt_upload_slice a few dozen times at R = 256 and R = 512 so that rocprofv3 --kernel-trace --stats shows the slab's
two launches (k_flatten_slab over 16 R^2 voxels, k_build_coarse over the nibble-map words it touches)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace_amd import abi, render, world

noise = np.fromfile(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
for region in (256, 512):
    mats, mine = world.generate_region(world.DEFAULT_SEED, region=region)
    cfg = render.make_config(64, 64, spp=1, depth=2, region=region, flags=abi.RT_FLAG_TRUSTED_WORLD | abi.RT_FLAG_TIMING_ALL)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(noise)
        n = 0
        for rep in range(4):
            for axis in range(3):
                for off in (0, 64, region - 16):
                    sl = [slice(None)] * 3
                    sl[2 - axis] = slice(off, off + 16)
                    ctx.upload_slice(axis, off, np.ascontiguousarray(mats[tuple(sl)]), np.ascontiguousarray(mine[tuple(sl)]))
                    n += 1
        ctx.draw_frame(render.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, 0.0, 0.0, seed=1))
        ctx.sync()
        tm = ctx.timing()
        print("region %d: %d slabs, %d launches, %.3f ms of kernels in total" % (region, n, tm.other_launches, tm.shade_ms))
