"""What each rank of an N-GPU run would render, measured one rank after the other on ONE GPU (no 8-GPU node has been available
to this build: SCALE_r01/r02 were skipped).  For N = 1, 2, 4, 8 and every rank r the frame's tiles t % N == r are rendered by a
context with tile_rank = r, tile_world = N — exactly what bench.py --gpus N creates on rank r — and the frame period is the wall
clock over a few frames enqueued back to back (the path kernel's share from its launch events, rt_get_timing).  Reported per N:
slowest and fastest rank, and N=1 time / slowest rank = the speed-up the RENDERING allows (load balance + fixed per-launch costs); the RCCL gather (23 B per pixel to rank 0 over xGMI) is not in it.
An emulation, not a scaling measurement: each rank has the whole GPU's caches and memory system to itself, as on its own GPU.

    python tools/scale_emulation.py [headline|c4]     -> one JSON line per workload (profiles/r4_scale_emulation.jsonl; SCALE_FIF=1|2 frames in flight)
"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytrace_amd import abi, render, world

WORK = {"headline": (1920, 1080, 64, 4, 20), "c4": (3840, 2160, 256, 8, 2)}
noise = np.fromfile(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
mats, mine = world.generate_region(world.DEFAULT_SEED)
u = render.camera_uniforms(render.DEFAULT_POSE["origin"], render.DEFAULT_POSE["heading"], render.DEFAULT_POSE["pitch"],
                           render.DEFAULT_POSE["sun_angle"], seed=1)
NS = [int(x) for x in os.environ.get("SCALE_N", "1,2,4,8").split(",")]
FIF = int(os.environ.get("SCALE_FIF", "2"))      # frames in flight, as bench.py's default (RT_FLAG_FRAMES_IN_FLIGHT_2); 1 = round 3's discipline
for name in (sys.argv[1:] or ["headline", "c4"]):
    W, H, spp, depth, frames = WORK[name]
    out = {"workload": "%dx%d spp=%d depth=%d" % (W, H, spp, depth), "method": "ranks rendered one after the other on one MI355X (tools/scale_emulation.py); gather not included",
           "frames_per_rank": frames, "frames_in_flight": FIF, "n": {}}
    base = None
    for N in NS:
        per_rank, per_rank_pk = [], []
        for r in range(N):
            cfg = render.make_config(W, H, spp=spp, depth=depth, tile_rank=r, tile_world=N, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_TIMING | (abi.RT_FLAG_FRAMES_IN_FLIGHT_2 if FIF == 2 else 0))
            with render.Context(cfg) as ctx:
                ctx.upload_world(mats, mine)
                ctx.upload_noise(noise)
                ctx.draw_frame(u); ctx.sync()          # warm-up
                ctx.timing()
                t0 = time.perf_counter()
                for _ in range(frames):
                    ctx.draw_frame(u)
                ctx.sync()
                wall = (time.perf_counter() - t0) * 1e3 / frames
                t = ctx.timing()
                per_rank.append(round(wall, 4)); per_rank_pk.append(round(t.trace_ms / frames, 4))
        if N == 1:
            base = per_rank[0]
        out["n"][str(N)] = {"rank_frame_ms": per_rank, "rank_path_kernel_ms": per_rank_pk, "slowest_ms": max(per_rank), "fastest_ms": min(per_rank),
                            "render_speedup_vs_1": round(base / max(per_rank), 3), "render_efficiency": round(base / max(per_rank) / N, 3)}
    print(json.dumps(out), flush=True)
