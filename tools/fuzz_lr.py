# extended random cases for k_paths' scrolled-region build (not part of the suite): many lr, poses near the window's edges
import sys, numpy as np
sys.path.insert(0, ".")
from raytrace_amd import abi, render, world
from oracle import pyoracle as po
from tests import scenes
noise = np.fromfile("tests/golden/blue_noise_512.rgba", dtype=np.uint8)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
regions = {}
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    lr = tuple(int(v) * 16 for v in rng.integers(-6, 7, size=3))
    kind = rng.integers(0, 3)
    if kind == 0:
        key = ("tor", lr)
        if key not in regions: regions[key] = world.toroidal_region(lr)
    elif kind == 1:
        key = ("blocks",)
        if key not in regions: regions[key] = world.region_from_ids(scenes.random_blocks_ids())
    else:
        key = ("stairs",)
        if key not in regions: regions[key] = world.region_from_ids(scenes.staircase_ids())
    mats, mine = regions[key]
    # origins anywhere in the window, some right at its faces
    o = np.array(lr, dtype=np.float64) + rng.uniform(-127.9, 127.9, size=3)
    if rng.random() < 0.3: o[rng.integers(0, 3)] = lr[rng.integers(0, 3)] + rng.choice([-127.99, 127.99, -128.0, 127.5])
    W = int(rng.integers(4, 16)) * 8; H = int(rng.integers(3, 10)) * 8
    spp = int(rng.integers(1, 4)); depth = int(rng.integers(1, 7))
    u = po.camera_uniforms(tuple(float(x) for x in o), float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.5, 1.5)), int(rng.integers(0, abi.NOISE_BYTES)), lr)
    cpu, ccn = po.render(mats, mine, noise, u, W, H, spp, depth)
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_PATHS, flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine); ctx.upload_noise(noise); ctx.draw_frame(u); ctx.sync()
        assert ctx.kernel_in_use() == abi.RT_KERNEL_PATHS
        gpu = ctx.readback_all()
    for name in cpu:
        if not np.array_equal(gpu[name], cpu[name], equal_nan=True):
            bad += 1; print("MISMATCH", i, name, lr, tuple(o), W, H, spp, depth, int(np.count_nonzero(gpu[name] != cpu[name]))); break
    if len(regions) > 6: regions.pop(next(iter(regions)))
print("cases done, mismatching:", bad)
