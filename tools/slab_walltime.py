"""rt_upload_slice as the host sees it: wall time of the CALL (time.perf_counter around the ctypes call, arrays
prepared beforehand), R = 256 and 512, validated / RT_FLAG_TRUSTED_WORLD, caller-owned buffers / the library's pinned staging
(rt_slice_staging), with the stream idle and with a frame in flight (1920x1080 spp 16: the call must not wait for it).
Round 4: also THREE slabs back to back behind a frame in flight — with two staging sets the first two calls return at
once and the third waits for the first slab's re-tile, i.e. for the frame in front of it (one slab per frame never waits).
Prints one JSON line per case -> profiles/r4_slab_walltime.jsonl."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace_amd import abi, render, world

noise = np.fromfile(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
for region in (256, 512):
    mats, mine = world.generate_region(world.DEFAULT_SEED, region=region)
    scale = region // 256
    u = render.camera_uniforms((-30.0 * scale, -128.0 * scale, 100.0 * scale), np.pi / 2, 0.0, 0.0, seed=1)
    slabs = []
    for axis in range(3):
        for off in (0, 64, region - 16):
            sl = [slice(None)] * 3
            sl[2 - axis] = slice(off, off + 16)
            slabs.append((axis, off, np.ascontiguousarray(mats[tuple(sl)]).reshape(-1), np.ascontiguousarray(mine[tuple(sl)]).reshape(-1)))
    for trusted in (False, True):
        flags = abi.RT_FLAG_CACHE_PRIMARY | (abi.RT_FLAG_TRUSTED_WORLD if trusted else 0)
        cfg = render.make_config(1920, 1080, spp=16, depth=4, region=region, flags=flags)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(noise)
            ctx.draw_frame(u); ctx.sync()
            t0 = time.perf_counter(); ctx.draw_frame(u); ctx.sync(); frame_ms = (time.perf_counter() - t0) * 1e3
            for staging in (False, True):
                for busy in (False, True):
                    times, fill = [], []
                    for rep in range(4):
                        for axis, off, sm, sf in slabs:
                            ctx.sync()
                            if staging:
                                t0 = time.perf_counter()
                                pm, pf = ctx.slice_staging()
                                pm[:] = sm; pf[:] = sf          # the host assembles its slab in place (here: a plain copy)
                                fill.append((time.perf_counter() - t0) * 1e3)
                                sm_, sf_ = pm, pf
                            else:
                                sm_, sf_ = sm, sf
                            if busy:
                                ctx.draw_frame(u)
                            t0 = time.perf_counter()
                            ctx.upload_slice(axis, off, sm_, sf_)
                            times.append((time.perf_counter() - t0) * 1e3)
                    ctx.sync()
                    times.sort()
                    print(json.dumps({"region": region, "slab_MiB": round(5 * 16 * region * region / 2 ** 20, 1), "validated": not trusted,
                                      "buffers": "rt_slice_staging (pinned, filled in place)" if staging else "caller-owned (pageable), copied into the staging",
                                      "frame_in_flight": busy, "frame_ms": round(frame_ms, 3) if busy else None, "calls": len(times),
                                      "call_ms_median": round(times[len(times) // 2], 4), "call_ms_min": round(times[0], 4), "call_ms_max": round(times[-1], 4),
                                      "fill_ms_median": round(sorted(fill)[len(fill) // 2], 4) if fill else None}), flush=True)
            # three slabs back to back behind a frame in flight: per position in the burst
            burst = [[], [], []]
            for rep in range(6):
                ctx.sync()
                ctx.draw_frame(u)
                for k in range(3):
                    axis, off, sm, sf = slabs[(3 * rep + k) % len(slabs)]
                    t0 = time.perf_counter()
                    ctx.upload_slice(axis, off, sm, sf)
                    burst[k].append((time.perf_counter() - t0) * 1e3)
            ctx.sync()
            print(json.dumps({"region": region, "validated": not trusted, "case": "three slabs back to back behind a frame in flight (caller-owned buffers)",
                              "frame_ms": round(frame_ms, 3), "call_ms_median_by_position": [round(sorted(b)[len(b) // 2], 4) for b in burst]}), flush=True)
