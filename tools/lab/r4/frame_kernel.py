"""k_frame (RT_KERNEL_FRAME) on the GPU box: first parity against the oracle on a few small frames (planes and exact counters,
cached primaries), then wall-clock timing of small frames next to the persistent path (prepass + k_persist).

    python3 tools/lab/r4/frame_kernel.py [check] [time]
Environment read by the library: RT_FRAME_THRESHOLD (parked lanes per pass)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
from raytrace_amd import abi, render, world  # noqa: E402


def uniforms(seed=1, origin=(-30.0, -128.0, 100.0), heading=np.pi / 2, pitch=0.0, sun=0.0, lr=(0, 0, 0)):
    return render.camera_uniforms(origin, heading, pitch, sun, seed=seed, lr=lr)


def check():
    from oracle import pyoracle as po
    noise = np.fromfile(os.path.join(ROOT, "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
    mats, mine = world.generate_region(world.DEFAULT_SEED)
    bad = 0
    for (W, H, spp, depth, lr, origin) in ((64, 64, 1, 2, (0, 0, 0), (-30.0, -128.0, 100.0)), (100, 60, 3, 4, (0, 0, 0), (-30.0, -128.0, 100.0)),
                                           (96, 96, 2, 8, (0, 0, 0), (100.0, 100.0, 60.0)), (96, 72, 2, 3, (16, 32, 0), (-14.0, -100.0, 100.0)),
                                           (128, 128, 1, 0, (0, 0, 0), (-30.0, -128.0, 100.0)), (1024, 520, 2, 3, (0, 0, 0), (-30.0, -128.0, 100.0))):
        u = po.camera_uniforms(origin, np.pi / 2, -0.1, 0.3, 5, lr)
        cpu, ccn = po.render(mats, mine, noise, u, W, H, spp, depth)
        _, c0 = po.render(mats, mine, noise, u, W, H, 1, 0)
        want = ccn.as_dict()
        p = c0.as_dict()
        for k in ("rays", "rays_primary", "iterations", "minefield_fetches", "material_fetches", "hits", "sky_exits", "limit_exits", "border_fetches"):
            want[k] -= (spp - 1) * p[k]
        for flags in (abi.RT_FLAG_CACHE_PRIMARY, abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS):
            cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_FRAME, flags=flags)
            with render.Context(cfg) as ctx:
                ctx.upload_world(mats, mine)
                ctx.upload_noise(noise)
                ctx.draw_frame(u)
                ctx.sync()
                gpu = ctx.readback_all()
                got = ctx.counters().as_dict()
                kiu = ctx.kernel_in_use()
            diffs = [n for n in cpu if not np.array_equal(gpu[n], cpu[n], equal_nan=True)]
            cdiff = {k: (got[k], want[k]) for k in want if got[k] != want[k]} if flags & abi.RT_FLAG_COUNTERS else {}
            ok = not diffs and not cdiff and kiu == abi.RT_KERNEL_FRAME
            bad += 0 if ok else 1
            print("check %dx%d spp %d depth %d lr %s flags %#x kernel %d: %s %s %s" % (W, H, spp, depth, lr, flags, kiu, "OK" if ok else "DIFF", diffs, cdiff), flush=True)
    return bad


def timing():
    noise = np.fromfile(os.path.join(ROOT, "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
    mats, mine = world.generate_region(world.DEFAULT_SEED)
    u = uniforms()
    for (W, H, spp, depth) in ((1024, 1024, 1, 2), (256, 256, 1, 2), (1920, 1080, 1, 2), (1920, 1080, 2, 4), (1920, 1080, 4, 4), (512, 512, 16, 4)):
        for name, kernel in (("frame", abi.RT_KERNEL_FRAME), ("persistent", abi.RT_KERNEL_PERSISTENT), ("paths", abi.RT_KERNEL_PATHS)):
            for fif in (1, 2):
                flags = abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_TIMING | (abi.RT_FLAG_FRAMES_IN_FLIGHT_2 if fif == 2 else 0)
                cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=flags)
                with render.Context(cfg) as ctx:
                    ctx.upload_world(mats, mine)
                    ctx.upload_noise(noise)
                    for _ in range(20):
                        ctx.draw_frame(u)
                    ctx.sync()
                    ctx.timing()
                    n = 200
                    t0 = time.perf_counter()
                    for _ in range(n):
                        ctx.draw_frame(u)
                    ctx.sync()
                    ms = (time.perf_counter() - t0) * 1e3 / n
                    tm = ctx.timing()
                    lat = []
                    for _ in range(50):
                        t0 = time.perf_counter()
                        ctx.draw_frame(u)
                        ctx.sync()
                        lat.append((time.perf_counter() - t0) * 1e3)
                    print("time %dx%d spp %d depth %d %-10s fif %d: %.4f ms/frame back to back, latency %.4f ms (median), kernel events %.4f ms x %d, in use %d"
                          % (W, H, spp, depth, name, fif, ms, float(np.median(lat)), tm.trace_ms / max(tm.trace_launches, 1), tm.trace_launches // n, ctx.kernel_in_use()),
                          flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["check", "time"]
    rc = 0
    if "check" in what:
        rc = check()
    if "time" in what and rc == 0:
        timing()
    sys.exit(1 if rc else 0)
