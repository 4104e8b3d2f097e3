# extended random cases for k_paths (FUZZ_KERNEL=frame: k_frame) (not part of the suite): tools/fuzz_lr.py SEED N — scrolled windows (many lr, some far from the
# origin where rays stall and meet the loop limit), lr = 0, poses near and beyond the window's faces; planes against the oracle, and
# for every fifth case the exact cached-primary counters of the counting build
import os, sys, numpy as np
sys.path.insert(0, ".")
from raytrace_amd import abi, render, world
from oracle import pyoracle as po
from tests import scenes
KERNEL = abi.RT_KERNEL_FRAME if os.environ.get('FUZZ_KERNEL') == 'frame' else abi.RT_KERNEL_PATHS
noise = np.fromfile("tests/golden/blue_noise_512.rgba", dtype=np.uint8)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
bad = 0
regions = {}
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    lr = tuple(int(v) * 16 for v in rng.integers(-6, 7, size=3))
    r = rng.random()
    if r < 0.35: lr = (0, 0, 0)                                                     # the lr = 0 build
    elif r < 0.45: lr = tuple(int(v) * 16 for v in rng.integers(-140, 141, size=3))   # up to 2240 voxels out: stalled rays, loop limit
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
    if rng.random() < 0.1: o[rng.integers(0, 3)] += rng.choice([-200.0, 180.0])   # a camera outside the window
    W = int(rng.integers(4, 16)) * 8; H = int(rng.integers(3, 10)) * 8
    spp = int(rng.integers(1, 4)); depth = int(rng.integers(1, 7))
    u = po.camera_uniforms(tuple(float(x) for x in o), float(rng.uniform(-3.2, 3.2)), float(rng.uniform(-1.5, 1.5)), float(rng.uniform(-1.5, 1.5)), int(rng.integers(0, abi.NOISE_BYTES)), lr)
    cpu, ccn = po.render(mats, mine, noise, u, W, H, spp, depth)
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=KERNEL, flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine); ctx.upload_noise(noise); ctx.draw_frame(u); ctx.sync()
        assert ctx.kernel_in_use() == KERNEL
        gpu = ctx.readback_all()
    for name in cpu:
        if not np.array_equal(gpu[name], cpu[name], equal_nan=True):
            bad += 1; print("MISMATCH", i, name, lr, tuple(o), W, H, spp, depth, int(np.count_nonzero(gpu[name] != cpu[name]))); break
    if i % 5 == 0:   # the counting build: exact counters (the oracle's, minus the primaries the cache saves)
        cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=KERNEL, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine); ctx.upload_noise(noise); ctx.draw_frame(u); ctx.sync()
            gcn = ctx.counters().as_dict()
        _, c0 = po.render(mats, mine, noise, u, W, H, 1, 0)
        want, p0 = ccn.as_dict(), c0.as_dict()
        for k in ("rays", "rays_primary", "iterations", "minefield_fetches", "material_fetches", "hits", "sky_exits", "limit_exits", "border_fetches"):
            want[k] -= (spp - 1) * p0[k]
        if gcn != want:
            bad += 1; print("COUNTERS", i, lr, tuple(o), W, H, spp, depth, {k: (gcn[k], want[k]) for k in want if gcn[k] != want[k]})
    if len(regions) > 6: regions.pop(next(iter(regions)))
print("cases done, mismatching:", bad)
