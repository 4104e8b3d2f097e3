import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, time
from raytrace_amd import abi, render, world
noise = np.fromfile('tests/golden/blue_noise_512.rgba', dtype=np.uint8)
mats, mine = world.generate_region()
u = render.camera_uniforms((-30.0,-128.0,100.0), np.pi/2, 0.0, 0.0, 1)
for (W,H,spp,D) in ((1024,1024,1,2),(1920,1080,64,4)):
    cfg = render.make_config(W,H,spp=spp,depth=D,flags=abi.RT_FLAG_CACHE_PRIMARY|abi.RT_FLAG_TIMING_ALL)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine); ctx.upload_noise(noise)
        for it in range(3):
            ctx.draw_frame(u); ctx.sync(); t=ctx.timing()
        fr=t.frame_ms
    # post passes on a context without per-launch timing (so that the denoise chain replays its HIP graph)
    cfg2 = render.make_config(W,H,spp=spp,depth=D,flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg2) as ctx:
        ctx.upload_world(mats, mine); ctx.upload_noise(noise)
        ctx.draw_frame(u); ctx.sync()
        ts=[]
        for it in range(8):
            t0=time.perf_counter(); ctx.denoise(True); ctx.sync(); t1=time.perf_counter(); ctx.finalize(); ctx.sync(); t2=time.perf_counter()
            ts.append(((t1-t0)*1e3,(t2-t1)*1e3))
        print(W,H,spp,D,"raytrace frame ms %.3f (trace %.3f other %.3f)"%(fr,t.trace_ms,t.shade_ms),"denoise x6 ms %.3f finalize ms %.3f"%(min(a for a,b in ts),min(b for a,b in ts)))
