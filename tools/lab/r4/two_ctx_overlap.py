"""Round-4 experiment, before any library change: what do two frames in flight buy?  Two CONTEXTS on one GPU are two complete sets
of per-frame resources on two streams — the upper bound of what double-buffering inside one context can reach.  Frames are
enqueued alternately on the two contexts (the second one's first frame a little later, so that the two path kernels relay
instead of sharing the CUs from the start) and the wall clock per frame is compared with one context drawing back to back.

    python tools/lab/r4/two_ctx_overlap.py [headline|share8|c4|small] ...
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))
from raytrace_amd import abi, render, world

WORK = {"headline": (1920, 1080, 64, 4, 0, 1, 20), "share8": (1920, 1080, 64, 4, 0, 8, 40), "c4": (3840, 2160, 256, 8, 0, 1, 2),
        "c4share8": (3840, 2160, 256, 8, 0, 8, 4), "small": (1024, 1024, 1, 2, 0, 1, 100), "spp4": (1920, 1080, 4, 4, 0, 1, 40)}
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..")
noise = np.fromfile(os.path.join(ROOT, "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
mats, mine = world.generate_region(world.DEFAULT_SEED)
p = render.DEFAULT_POSE
u = render.camera_uniforms(p["origin"], p["heading"], p["pitch"], p["sun_angle"], seed=1)


def make(W, H, spp, depth, r, N):
    cfg = render.make_config(W, H, spp=spp, depth=depth, tile_rank=r, tile_world=N, flags=abi.RT_FLAG_CACHE_PRIMARY)
    ctx = render.Context(cfg)
    ctx.upload_world(mats, mine)
    ctx.upload_noise(noise)
    ctx.draw_frame(u); ctx.sync()
    return ctx


for name in (sys.argv[1:] or ["headline", "share8"]):
    W, H, spp, depth, r, N, frames = WORK[name]
    a, b = make(W, H, spp, depth, r, N), make(W, H, spp, depth, r, N)
    out = {"workload": name, "frames": frames}
    for rep in range(2):
        t0 = time.perf_counter()
        for _ in range(frames):
            a.draw_frame(u)
        a.sync()
        out["single_ms_%d" % rep] = round((time.perf_counter() - t0) * 1e3 / frames, 4)
        t0 = time.perf_counter()
        for _ in range(frames):
            a.draw_frame(u); a.sync()
        out["latency_ms_%d" % rep] = round((time.perf_counter() - t0) * 1e3 / frames, 4)
        for stagger in (0.0, 0.3, 0.5):
            one = out["single_ms_%d" % rep] * 1e-3
            t0 = time.perf_counter()
            a.draw_frame(u)
            if stagger:
                time.sleep(one * stagger)
            b.draw_frame(u)
            for _ in range(frames // 2 - 1):
                a.draw_frame(u); b.draw_frame(u)
            a.sync(); b.sync()
            dt = time.perf_counter() - t0 - (0.0 if not stagger else 0.0)
            out["dual_stagger%.1f_ms_%d" % (stagger, rep)] = round(dt * 1e3 / (frames // 2 * 2), 4)
    a.destroy(); b.destroy()
    print(json.dumps(out), flush=True)
