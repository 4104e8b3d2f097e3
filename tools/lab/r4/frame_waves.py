"""Per-wave timeline of k_frame (RT_DEBUG_WAVE_DUMP; a -DRT_DIAG_FRAME_TIMES variant for shipped-speed times, or FRAME_WAVES_COUNT=1 for
the counting build's step counts): python3 tools/lab/r4/frame_waves.py W H spp depth"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
os.environ["RT_DEBUG_WAVE_DUMP"] = "/tmp/frame_waves.bin"
from raytrace_amd import abi, render, world
W, H, spp, depth = (int(x) for x in sys.argv[1:5])
noise = np.fromfile(os.path.join(ROOT, "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
mats, mine = world.generate_region(world.DEFAULT_SEED)
u = render.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, 0.0, 0.0, seed=1)
cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_FRAME,
                         flags=abi.RT_FLAG_CACHE_PRIMARY | (abi.RT_FLAG_COUNTERS if os.environ.get("FRAME_WAVES_COUNT") else 0))
with render.Context(cfg) as ctx:
    ctx.upload_world(mats, mine); ctx.upload_noise(noise)
    for _ in range(3):
        ctx.draw_frame(u); ctx.sync()
    ctx.counters()
r = np.fromfile("/tmp/frame_waves.bin", dtype=np.uint64).reshape(-1, 4)
r = r[r[:, 0] != 0]
t0 = r[:, 0].min()
r = r[r[:, 0] - t0 < 10**7]          # (stale rows of an earlier, larger dump)
st, mid, en = (r[:, 0] - t0) / 100.0, (r[:, 1] - t0) / 100.0, (r[:, 2] - t0) / 100.0     # us
a, b, ps = (r[:, 3] & 0xFFFF).astype(np.int64), (r[:, 3] >> 16 & 0xFFFF).astype(np.int64), (r[:, 3] >> 32).astype(np.int64)
print("waves %d, kernel span %.1f us" % (len(r), en.max()))
print("wave start us: p50 %.1f p99 %.1f max %.1f | phase A ends: p50 %.1f p90 %.1f max %.1f | wave ends: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f"
      % (*np.percentile(st, [50, 99]), st.max(), *np.percentile(mid, [50, 90]), mid.max(), *np.percentile(en, [10, 50, 90, 99]), en.max()))
print("phase A duration: mean %.1f p90 %.1f max %.1f | phase B (+ barrier wait): mean %.1f p90 %.1f max %.1f"
      % ((mid - st).mean(), np.percentile(mid - st, 90), (mid - st).max(), (en - mid).mean(), np.percentile(en - mid, 90), (en - mid).max()))
if a.max() > 0:
    print("steps A per wave: mean %.1f p90 %d max %d | steps B: mean %.1f p90 %d max %d | passes mean %.1f max %d"
          % (a.mean(), np.percentile(a, 90), a.max(), b.mean(), np.percentile(b, 90), b.max(), ps.mean(), ps.max()))
