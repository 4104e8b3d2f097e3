"""Where the step loop's minefield fetches are answered (VERDICT r3 #2a): by the nibble map in LDS (one entry per (R/64)^3 cube),
by the global per-brick nibble map behind it (regions above 256), or by a byte of the swizzled array (one 64-byte line each).
One counting frame of k_paths per case; the counters ride in the debug fields of the counting build (RT_DEBUG_STATS).

    python tools/fetch_levels.py            # headline (R = 256), R = 512, C5 pose and the terrain pose at R = 1024
    python tools/fetch_levels.py REGION W H SPP DEPTH X Y Z HEADING PITCH
"""
import json, os, re, subprocess, sys
if os.environ.get("_FETCH_CHILD"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from raytrace_amd import abi, render, world
    a = sys.argv[1:]
    R, W, H, spp, depth = (int(x) for x in a[:5])
    pose = [float(x) for x in a[5:10]]
    noise = np.fromfile(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
    mats, mine = world.generate_region(world.DEFAULT_SEED, region=R)
    u = render.camera_uniforms(tuple(pose[:3]), pose[3], pose[4], 0.0, seed=1)
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_PATHS, region=R, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine); ctx.upload_noise(noise)
        ctx.draw_frame(u); ctx.sync()
        cn = ctx.counters()
    print("COUNTERS %d %d %d" % (cn.rays, cn.minefield_fetches, cn.pixels))
    sys.exit(0)

CASES = [("headline R=256", 256, 1920, 1080, 8, 4, -30, -128, 100, 1.5707964, 0.0),
         ("R=512 default pose", 512, 1920, 1080, 8, 4, -60, -256, 200, 1.5707964, 0.0),
         ("C5 pose R=1024 (mostly sky)", 1024, 3840, 2160, 4, 8, -120, -512, 400, 1.5707964, 0.0),
         ("C5 terrain pose R=1024", 1024, 3840, 2160, 4, 8, -120, -512, 160, 1.5707964, -0.3)]
if len(sys.argv) > 10:
    CASES = [("custom",) + tuple(sys.argv[1:11])]
for case in CASES:
    env = dict(os.environ, _FETCH_CHILD="1", RT_DEBUG_STATS="1")
    out = subprocess.run([sys.executable, __file__] + [str(x) for x in case[1:]], env=env, capture_output=True, text=True)
    m = re.search(r"raw: loop_iters (\d+) s_lanes (\d+) f_lanes (\d+) passes (\d+) pass_lanes (\d+) s_execs (\d+) f_execs (\d+)", out.stderr)
    m2 = re.search(r"raw2: sky_lanes (\d+)", out.stderr)
    m3 = re.search(r"COUNTERS (\d+) (\d+) (\d+)", out.stdout)
    if not (m and m2 and m3):
        sys.exit("no counters: " + out.stderr[-1500:])
    fetch, beyond_lds, beyond_brick = int(m.group(6)), int(m.group(2)), int(m2.group(1))
    print(json.dumps({"case": case[0], "region": int(case[1]), "frame": "%sx%s spp=%s depth=%s" % tuple(case[2:6]), "pose": [float(x) for x in case[6:11]],
                      "loop_fetches_of_rays_in_flight": fetch, "answered_by_lds_map": round(1 - beyond_lds / fetch, 4),
                      "answered_by_brick_map": round((beyond_lds - beyond_brick) / fetch, 4), "bytes_from_the_array": round(beyond_brick / fetch, 4),
                      "rays": int(m3.group(1)), "sky_pixels_note": None}), flush=True)
