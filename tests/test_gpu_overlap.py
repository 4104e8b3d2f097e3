"""Round 4: two launches in flight.  The library alternates its path launches between two streams (the sample batches of a
frame always; the frames themselves with RT_FLAG_FRAMES_IN_FLIGHT_2, which then render into two frame slots).  None of it may
change a bit of any frame: every test here compares against the oracle, through the C ABI, frames enqueued WITHOUT a wait in
between.  Also here: the one-sample "direct" path of the path kernels and the reference's whole frame — ray trace,
denoise x6, finalize — behind the mirror's draw_frame."""
import numpy as np
import pytest

from raytrace_amd import abi, render, tiles, world
from oracle import pyoracle as po

pytestmark = pytest.mark.gpu

PLANES = ("depth_r16", "normal_r8", "albedo_rgba8", "emission_rgba8", "fog_rgba8", "lighting_rgba16", "lighting_f32", "fog_f32", "depth_f32")


def _same(gpu, cpu, what=""):
    for name in cpu:
        assert np.array_equal(gpu[name], cpu[name], equal_nan=True), "%s: plane %s differs at %d values" % (
            what, name, int(np.count_nonzero(gpu[name] != cpu[name])))


def _peek(ctx, ptrs, W, H):
    """Planes behind device pointers captured earlier (the slot of a frame that is no longer the context's current one)."""
    import torch
    import bench
    out = {}
    for b, ptr in ptrs.items():
        dt, ch = abi.BUFFER_FORMATS[b]
        n = W * H * ch * np.dtype(dt).itemsize
        raw = torch.as_tensor(bench._DevArray(ptr, n), device=torch.device("cuda", 0)).cpu().numpy()
        out[abi.BUFFER_NAMES[b]] = raw.view(dt).reshape((H, W, ch) if ch > 1 else (H, W))
    return out


def _ptrs(ctx):
    return {b: ctx.device_ptr(b) for b in range(abi.RT_BUF_FINAL_BGRA8)}


@pytest.mark.parametrize("kernel", [abi.RT_KERNEL_PATHS, abi.RT_KERNEL_PERSISTENT, abi.RT_KERNEL_DEFAULT, abi.RT_KERNEL_FRAME])
@pytest.mark.parametrize("batch", [None, "2"])
def test_two_frames_in_flight_are_the_frames_of_their_own_uniforms(procedural_region, blue_noise, kernel, batch, monkeypatch):
    """Six frames with six different cameras, seeds and TWO sun angles (the per-frame tables are rebuilt between frames that are
    both in flight) enqueued back to back with RT_FLAG_FRAMES_IN_FLIGHT_2: after one wait the context's planes are the last
    frame and the other slot still holds the frame before it — both equal to the oracle's.  With RT_PERSIST_BATCH=2 every frame
    is three launches (2 + 2 + 1 samples) that alternate between the lanes as well."""
    if batch:
        monkeypatch.setenv("RT_PERSIST_BATCH", batch)
    mats, mine = procedural_region
    W, H, spp, depth = 104, 56, 5, 4
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_FRAMES_IN_FLIGHT_2)
    us = [po.camera_uniforms((-30.0 + 3 * i, -128.0, 100.0 - 2 * i), np.pi / 2 + 0.05 * i, -0.02 * i, 0.0 if i % 3 else 0.7, 11 + 5 * i) for i in range(6)]
    with render.Context(cfg) as ctx:
        inf = ctx.info()
        assert inf.frames_in_flight == 2 and inf.launches_in_flight == 2
        if batch:
            assert inf.samples_per_launch == 2
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        seen = []
        for u in us:
            ctx.draw_frame(u)
            seen.append(_ptrs(ctx))
        ctx.sync()
        assert seen[0] == seen[2] == seen[4] and seen[1] == seen[3] == seen[5] and seen[0] != seen[1]      # two slots, used in turn
        last = ctx.readback_all()
        before = _peek(ctx, seen[4], W, H)
    _same(last, po.render(mats, mine, blue_noise, us[5], W, H, spp, depth)[0], "last frame")
    _same(before, po.render(mats, mine, blue_noise, us[4], W, H, spp, depth)[0], "the frame before it")


def test_frames_in_flight_with_post_passes_and_readback_between_them(procedural_region, blue_noise):
    """rt_denoise / rt_finalize / rt_readback act on the frame drawn last and are ordered after it, whichever stream it ended on:
    draw, denoise, finalize four times without a wait, then the swapchain image of the last frame against the oracle's chain —
    and again with a readback (a wait for that frame only) after every frame."""
    mats, mine = procedural_region
    W, H, spp, depth = 96, 64, 2, 2
    cfg = render.make_config(W, H, spp=spp, depth=depth, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_FRAMES_IN_FLIGHT_2)
    us = [po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.1, 0.3, 3 + i) for i in range(4)]

    def chain(u):
        g, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
        den = po.denoise(g["lighting_rgba16"], g["depth_r16"], g["normal_r8"], faithful=True)
        return den, po.finalize(g["albedo_rgba8"], g["emission_rgba8"], g["fog_rgba8"], den, g["depth_r16"], blue_noise)

    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        for u in us:
            ctx.draw_frame(u)
            ctx.denoise(True)
            ctx.finalize()
        fin = ctx.readback(abi.RT_BUF_FINAL_BGRA8)
        den = ctx.readback(abi.RT_BUF_LIGHTING_RGBA16)
        exp_den, exp_fin = chain(us[3])
        assert np.array_equal(den, exp_den) and np.array_equal(fin, exp_fin)
        for u in us[:3]:
            ctx.draw_frame(u)
            ctx.denoise(True)
            ctx.finalize()
            exp_den, exp_fin = chain(u)
            assert np.array_equal(ctx.readback(abi.RT_BUF_FINAL_BGRA8), exp_fin)


@pytest.mark.parametrize("lanes", ["1", "2"])
def test_sample_batches_alternate_between_the_lanes(procedural_region, blue_noise, lanes, monkeypatch):
    """Seven launches of one sample each per frame, two frames back to back on ONE slot (no flag): launch b + 1 runs on the other
    stream while launch b drains, its accumulate waits for launch b's (a pixel's samples are added in sample order).  Same bits as
    the oracle with two lanes and with RT_LANES=1 (everything on one stream, round 3's order); depth 6 puts part of the albedo
    stack into the lane's global stack."""
    monkeypatch.setenv("RT_PERSIST_BATCH", "1")
    monkeypatch.setenv("RT_LANES", lanes)
    mats, mine = procedural_region
    W, H, spp = 88, 48, 7
    for kernel, depth in ((abi.RT_KERNEL_PATHS, 3), (abi.RT_KERNEL_PATHS, 6), (abi.RT_KERNEL_PERSISTENT, 9)):
        cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=abi.RT_FLAG_CACHE_PRIMARY)
        with render.Context(cfg) as ctx:
            assert ctx.info().launches_in_flight == int(lanes) and ctx.info().frames_in_flight == 1
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            u1, u2 = po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.1, 0.2, 21), po.camera_uniforms((-10.0, -120.0, 90.0), 1.2, -0.2, 0.2, 40)
            ctx.draw_frame(u1)
            ctx.draw_frame(u2)
            got = ctx.readback_all()
        _same(got, po.render(mats, mine, blue_noise, u2, W, H, spp, depth)[0], "kernel %d depth %d" % (kernel, depth))


def test_a_slab_between_two_frames_in_flight(procedural_region, blue_noise):
    """rt_upload_slice with frames in flight on both lanes: frame 0 (enqueued before) must see the old region, frame 1 the new
    one — the re-tile waits for every launch submitted so far and every later launch waits for it."""
    mats, mine = procedural_region
    W, H, spp, depth = 96, 56, 3, 3
    R = 256
    m3, f3 = mats.reshape(R, R, R).copy(), mine.reshape(R, R, R).copy()
    # a slab of foreign terrain (another part of the region, shifted up) where the camera looks
    sl = slice(96, 112)
    new_m, new_f = m3.copy(), f3.copy()
    new_m[:, sl, :] = np.roll(m3[:, 16:32, :], 24, axis=0)
    new_f[:, sl, :] = np.roll(f3[:, 16:32, :], 24, axis=0)
    u0 = po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.1, 0.3, 5)
    u1 = po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.1, 0.3, 9)
    cfg = render.make_config(W, H, spp=spp, depth=depth, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_FRAMES_IN_FLIGHT_2)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u0)
        p0 = _ptrs(ctx)
        ctx.upload_slice(1, 96, new_m[:, sl, :], new_f[:, sl, :])
        ctx.draw_frame(u1)
        ctx.sync()
        f1 = ctx.readback_all()
        f0 = _peek(ctx, p0, W, H)
    old, _ = po.render(mats, mine, blue_noise, u0, W, H, spp, depth)
    new, _ = po.render(new_m.reshape(-1), new_f.reshape(-1), blue_noise, u1, W, H, spp, depth)
    assert not np.array_equal(old["depth_r16"], po.render(new_m.reshape(-1), new_f.reshape(-1), blue_noise, u0, W, H, 1, 0)[0]["depth_r16"])
    _same(f0, old, "frame before the slab")
    _same(f1, new, "frame after the slab")


@pytest.mark.parametrize("kernel", [abi.RT_KERNEL_PATHS, abi.RT_KERNEL_PERSISTENT])
@pytest.mark.parametrize("W,H,depth,world_", [(128, 96, 2, 1), (100, 60, 4, 1), (100, 60, 2, 2), (96, 64, 5, 1)])
def test_one_sample_frames_store_their_lighting_in_the_path_kernel(procedural_region, blue_noise, kernel, W, H, depth, world_):
    """with one sample per pixel k_paths (depth <= 4, region 256) and k_persist store the lighting planes themselves —
    no light record, no accumulate launch — and the host decides that by the very test the kernels make (STK = 0 / `direct`).
    Named cases: depth 2 and 4 direct, a partial-tile frame, a two-way tile split, depth 5 on k_paths (not direct: records +
    accumulate); two consecutive frames each (the lanes' cursors alternate)."""
    mats, mine = procedural_region
    us = [po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.1, 0.3, s) for s in (4, 90)]
    for rank in range(world_):
        cfg = render.make_config(W, H, spp=1, depth=depth, kernel=kernel, tile_rank=rank, tile_world=world_, flags=abi.RT_FLAG_CACHE_PRIMARY)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            for u in us:
                ctx.draw_frame(u)
                ctx.sync()
                got = ctx.readback_all()
                assert ctx.kernel_in_use() == kernel
                cpu, _ = po.render(mats, mine, blue_noise, u, W, H, 1, depth)
                if world_ == 1:
                    _same(got, cpu)
                else:   # this rank's tiles of the oracle's frame, in the rank's tile-major layout
                    for name in cpu:
                        exp = tiles.tile_major_from_frame(cpu[name], rank, world_)
                        n = tiles.tile_count(W, H, rank, world_) * 64
                        px = got[name].reshape((-1,) + exp.shape[1:])[:n]
                        # pixels of a partial tile that lie outside the frame are never written: compare the inside ones
                        inside = tiles.tile_major_from_frame(np.ones((H, W), dtype=np.uint8), rank, world_)[:n].astype(bool)
                        assert np.array_equal(px[inside], exp[:n][inside], equal_nan=True), (name, rank)


def test_default_kernel_switches_to_k_paths_for_one_sample_frames_of_three_million_pixels(procedural_region, blue_noise):
    """RT_KERNEL_DEFAULT runs launches of >= 3 M pixel-samples on k_paths: a 2304 x 1408 one-sample frame (3.24 M pixels) is on
    k_paths' direct path, a 2304 x 1152 one on k_persist's (below 2.5 M pixels k_frame takes over: tests/test_gpu_frame_kernel.py); two
    8-row bands of the big frame against the oracle."""
    mats, mine = procedural_region
    W, H, depth = 2304, 1408, 2
    u = po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.1, 0.3, 17)
    cfg = render.make_config(W, H, spp=1, depth=depth, flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        for _ in range(2):
            ctx.draw_frame(u)
        ctx.sync()
        assert ctx.kernel_in_use() == abi.RT_KERNEL_PATHS
        got = ctx.readback_all()
    for y0 in (400, 1000):
        band, _ = po.render(mats, mine, blue_noise, u, W, H, 1, depth, rows=(y0, y0 + 8))
        for name in band:
            assert np.array_equal(got[name][y0:y0 + 8], band[name][y0:y0 + 8], equal_nan=True), (name, y0)
    with render.Context(render.make_config(2304, 1152, spp=1, depth=depth, flags=abi.RT_FLAG_CACHE_PRIMARY)) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        assert ctx.kernel_in_use() == abi.RT_KERNEL_PERSISTENT


@pytest.mark.parametrize("faithful", [True, False])
def test_the_mirror_draws_the_whole_reference_frame(blue_noise, faithful):
    """Pipeline::draw_frame submits ONE command buffer holding ray trace -> denoise x6 -> finalize
    (pipeline.rs:86-123, :229-235).  With enable_post_passes the mirror's draw_frame enqueues the same three stages, and
    RT_BUF_FINAL_BGRA8 is rt_oracle_finalize(rt_oracle_denoise(oracle frame)) — three frames, the seed advancing as in :201."""
    g = render.Game(args=(-30, -128, 100, 1.5707964, -0.15, 0.3))
    g.generate_world(world.DEFAULT_SEED)
    W, H = 112, 80
    cfg = render.make_config(W, H, spp=1, depth=2, flags=abi.RT_FLAG_CACHE_PRIMARY)
    p = render.create_instance(cfg, g, blue_noise)
    p.enable_post_passes(faithful=faithful)
    mats, mine = world.generate_region(world.DEFAULT_SEED)
    for frame in range(3):
        p.draw_frame(g)
        p.wait()
        u = p.uniforms()
        assert u.seed == frame + 1
        cpu, _ = po.render(mats, mine, blue_noise, u, W, H, 1, 2)
        den = po.denoise(cpu["lighting_rgba16"], cpu["depth_r16"], cpu["normal_r8"], faithful=faithful)
        fin = po.finalize(cpu["albedo_rgba8"], cpu["emission_rgba8"], cpu["fog_rgba8"], den, cpu["depth_r16"], blue_noise)
        assert np.array_equal(p.context.readback(abi.RT_BUF_LIGHTING_RGBA16), den)
        assert np.array_equal(p.context.readback(abi.RT_BUF_FINAL_BGRA8), fin), "frame %d" % frame
        for name in ("depth_r16", "normal_r8", "albedo_rgba8", "emission_rgba8", "fog_rgba8"):
            assert np.array_equal(p.context.readback(getattr(abi, "RT_BUF_" + {"depth_r16": "DEPTH_R16UI", "normal_r8": "NORMAL_R8UI", "albedo_rgba8": "ALBEDO_RGBA8",
                                                                               "emission_rgba8": "EMISSION_RGBA8", "fog_rgba8": "FOG_RGBA8"}[name])), cpu[name])
    p.close()
    # a tile-split pipeline has no whole frame to filter
    p2 = render.create_instance(render.make_config(W, H, spp=1, depth=2, tile_rank=0, tile_world=2, flags=abi.RT_FLAG_CACHE_PRIMARY), g, blue_noise)
    with pytest.raises(render.RtError):
        p2.enable_post_passes()
    p2.close()
    g.close()


def test_rt_bench_post_draws_the_whole_frame(native_built):
    import json
    import os
    import subprocess
    from tests.conftest import ROOT
    exe = os.path.join(ROOT, "raytrace_amd", "rt_bench")
    base = [exe, "--width", "256", "--height", "128", "--frames", "6"]
    r0 = subprocess.run(base, cwd=ROOT, capture_output=True, text=True, timeout=300)
    r1 = subprocess.run(base + ["--post"], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r0.returncode == 0 and r1.returncode == 0, r0.stderr[-1000:] + r1.stderr[-1000:]
    j0 = json.loads([l for l in r0.stdout.splitlines() if l.startswith("{")][-1])
    j1 = json.loads([l for l in r1.stdout.splitlines() if l.startswith("{")][-1])
    assert j0["config"]["post_passes"] is False and j0["final_image_checksum"] == 0
    assert j1["config"]["post_passes"] is True and j1["final_image_checksum"] > 256 * 128 * 255      # alpha alone is 255 per pixel
