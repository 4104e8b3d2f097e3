"""How long the path kernel takes to drain (diagnostic build -DRT_DIAG_WAVE_TIMES, tools/variant.sh): one counting frame of a
workload, wave lifetimes from the 100 MHz clock.   RT_AMD_LIB=.../librt_amd_wt.so python tools/drain_times.py [r/N] [W H spp depth]"""
import os, re, subprocess, sys
if os.environ.get("_DRAIN_CHILD"):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    from raytrace_amd import abi, render, world
    a = sys.argv[1:]
    r, N = (int(x) for x in a[0].split("/")) if a and "/" in a[0] else (0, 1)
    if a and "/" in a[0]: a = a[1:]
    W, H, spp, depth = (int(x) for x in a) if a else (1920, 1080, 64, 4)
    noise = np.fromfile(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
    mats, mine = world.generate_region(world.DEFAULT_SEED)
    p = render.DEFAULT_POSE
    u = render.camera_uniforms(p["origin"], p["heading"], p["pitch"], p["sun_angle"], seed=1)
    cfg = render.make_config(W, H, spp=spp, depth=depth, tile_rank=r, tile_world=N, kernel=abi.RT_KERNEL_PATHS,
                             flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine); ctx.upload_noise(noise)
        ctx.draw_frame(u); ctx.sync(); ctx.reset_counters()
        ctx.draw_frame(u); ctx.sync(); ctx.counters()
    sys.exit(0)
env = dict(os.environ, _DRAIN_CHILD="1", RT_DEBUG_STATS="1")
out = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
m = re.search(r"raw: loop_iters (\d+) s_lanes (\d+) f_lanes (\d+) passes (\d+) pass_lanes (\d+) s_execs (\d+) f_execs (\d+)", out.stderr)
m2 = re.search(r"raw2: sky_lanes (\d+)", out.stderr)
if not m or not m2:
    sys.exit("no counters: " + out.stderr[-2000:])
first_exh_inv, exh_sum, exh_t_sum, waves, end_sum, life_sum, last_end = (int(x) for x in m.groups())
M = (1 << 64) - 1
t0 = M - int(m2.group(1)); first_exh = M - first_exh_inv
us = lambda ticks: ticks / 100.0
t0m = t0 & 0xFFFFFFFFFF
print("waves %d | kernel span %.1f us | first wave out of paths at %.1f us | mean wave: out of paths at %.1f us, ends at %.1f us "
      "(%.1f us in the drain) | last wave ends %.1f us after the mean" % (
      waves, us(last_end - t0), us(first_exh - t0), us(exh_t_sum / waves - t0m), us(end_sum / waves - t0m), us(exh_sum / waves),
      us(last_end - t0) - us(end_sum / waves - t0m)))
