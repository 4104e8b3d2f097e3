"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar: bit-exact for every integer plane and — because both sides compile the same arithmetic contract
(include/rt_math.h) with contraction off — also for the fp32 planes; the stated tolerance of the path
(BASELINE.json: per-pixel RMS <= 1e-4) is asserted as well so a future relaxation of exactness still has a gate.
"""
import os

import numpy as np
import pytest

from raytrace_amd import abi, render, world
from oracle import pyoracle as po
from tests import scenes

pytestmark = pytest.mark.gpu

KERNELS = [abi.RT_KERNEL_MEGA, abi.RT_KERNEL_WAVEFRONT, abi.RT_KERNEL_PERSISTENT]
RMS_TOL = 1e-4  # BASELINE.json north_star: per-pixel RMS error <= 1e-4 vs the CPU reference


def _uniforms(origin=(-30.0, -128.0, 100.0), heading=np.pi / 2, pitch=0.0, sun=0.0, seed=1, lr=(0, 0, 0)):
    return po.camera_uniforms(origin, heading, pitch, sun, seed, lr)


def _render_gpu(mats, mine, noise, u, W, H, spp, depth, kernel, flags=abi.RT_FLAG_COUNTERS):
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=flags)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(noise)
        ctx.draw_frame(u)
        ctx.sync()
        planes = ctx.readback_all()
        cn = ctx.counters()
    return planes, cn


def _compare(gpu, cpu, gcn=None, ccn=None):
    for name in ("depth_r16", "normal_r8", "albedo_rgba8", "emission_rgba8", "fog_rgba8", "lighting_rgba16"):
        assert np.array_equal(gpu[name], cpu[name]), "plane %s differs at %d pixels" % (
            name, int(np.count_nonzero(gpu[name] != cpu[name])))
    for name in ("lighting_f32", "fog_f32", "depth_f32"):
        g, c = gpu[name].astype(np.float64), cpu[name].astype(np.float64)
        # a camera inside solid ground makes mod(x, 0) = NaN in the shader (quirk Q12); NaNs must coincide
        assert np.array_equal(np.isnan(g), np.isnan(c)), "%s: NaN pattern differs" % name
        ok = ~np.isnan(c)
        rms = float(np.sqrt(np.mean((g[ok] - c[ok]) ** 2))) if ok.any() else 0.0
        assert rms <= RMS_TOL, "%s RMS %g" % (name, rms)
        assert np.array_equal(gpu[name], cpu[name], equal_nan=True), "%s not bit-exact (rms %g)" % (name, rms)
    if gcn is not None:
        gd, cd = gcn.as_dict(), ccn.as_dict()
        assert gd == cd, "counters differ: %s" % {k: (gd[k], cd[k]) for k in gd if gd[k] != cd[k]}


def _cached_counters(mats, mine, noise, u, W, H, spp, depth, ccn, **kw):
    """Counters the oracle implies for a frame rendered with RT_FLAG_CACHE_PRIMARY: the primary ray (seed-independent) is
    traced once per pixel instead of once per sample, everything else is unchanged."""
    _, c0 = po.render(mats, mine, noise, u, W, H, 1, 0, **kw)      # the primary rays of one sample
    d, p = ccn.as_dict(), c0.as_dict()
    for k in ("rays", "rays_primary", "iterations", "minefield_fetches", "material_fetches", "hits", "sky_exits", "limit_exits",
              "border_fetches"):
        d[k] -= (spp - 1) * p[k]
    return d


PATH_KERNELS = [abi.RT_KERNEL_PERSISTENT, abi.RT_KERNEL_PATHS]

CASES = [
    # W, H, spp, depth
    (64, 64, 1, 2),      # the reference frame (1 spp, 2 levels)
    (128, 128, 1, 0),    # primary only
    (96, 72, 2, 1),
    (128, 128, 3, 4),
    (100, 60, 2, 3),     # not a multiple of 8: partial tiles
]


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("W,H,spp,depth", CASES)
def test_procedural_matches_oracle(procedural_region, blue_noise, kernel, W, H, spp, depth):
    mats, mine = procedural_region
    u = _uniforms(seed=1)
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, kernel)
    _compare(gpu, cpu, gcn, ccn)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("pose", [
    dict(origin=(100.0, 100.0, 60.0), heading=-2.0, pitch=-0.1, sun=0.7),     # capture_training_data.py pose grid
    dict(origin=(-30.0, -200.0, 100.0), heading=np.pi / 2, pitch=-0.2, sun=-0.7),  # outside the region (raytrace.comp:311-315)
    dict(origin=(10.0, 10.0, 5.0), heading=1.0, pitch=0.3, sun=1.2),           # camera inside solid ground (degenerate start)
    dict(origin=(0.0, 0.0, 120.0), heading=0.3, pitch=-1.2, sun=0.0),          # looking steeply down
])
def test_poses_match_oracle(procedural_region, blue_noise, kernel, pose):
    mats, mine = procedural_region
    u = _uniforms(pose["origin"], pose["heading"], pose["pitch"], pose["sun"], seed=77)
    cpu, ccn = po.render(mats, mine, blue_noise, u, 96, 96, 2, 3)
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, 96, 96, 2, 3, kernel)
    _compare(gpu, cpu, gcn, ccn)


@pytest.mark.parametrize("kernel", KERNELS)
@pytest.mark.parametrize("scene", ["empty", "floor", "voxel", "stairs", "blocks"])
def test_analytic_scenes_match_oracle(native_built, blue_noise, kernel, scene):
    ids = {"empty": scenes.empty_ids, "floor": scenes.floor_ids, "voxel": scenes.single_voxel_ids,
           "stairs": scenes.staircase_ids, "blocks": scenes.random_blocks_ids}[scene]()
    mats, mine = world.region_from_ids(ids)
    u = _uniforms(origin=(-40.0, -100.0, 90.0), heading=1.1, pitch=-0.5, sun=0.3, seed=5)
    cpu, ccn = po.render(mats, mine, blue_noise, u, 80, 80, 2, 3)
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, 80, 80, 2, 3, kernel)
    _compare(gpu, cpu, gcn, ccn)


@pytest.mark.parametrize("kernel", KERNELS)
def test_scrolled_region_lr_nonzero(procedural_region, blue_noise, kernel):
    """lr != 0 (TerrainUploadManager render offset, pipeline.rs:203-207): the generic wrap/border path."""
    mats, mine = procedural_region
    u = _uniforms(origin=(-14.0, -100.0, 100.0), seed=9, lr=(16, 32, 0))
    cpu, ccn = po.render(mats, mine, blue_noise, u, 96, 96, 1, 2)
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, 96, 96, 1, 2, kernel)
    _compare(gpu, cpu, gcn, ccn)


def test_seed_clamp_and_wrap(procedural_region, blue_noise):
    """Q4: seed/512 beyond the noise height clamps to row 511; seeds wrap at RT_NOISE_BYTES (pipeline.rs:201)."""
    mats, mine = procedural_region
    u = _uniforms(seed=abi.NOISE_BYTES - 1)
    cpu, ccn = po.render(mats, mine, blue_noise, u, 64, 64, 3, 2)
    for kernel in KERNELS:
        gpu, gcn = _render_gpu(mats, mine, blue_noise, u, 64, 64, 3, 2, kernel)
        _compare(gpu, cpu, gcn, ccn)


@pytest.mark.parametrize("prepass", ["1", "2"])
@pytest.mark.parametrize("W,H,spp,depth", [(64, 64, 1, 2), (96, 72, 4, 0), (100, 60, 5, 3), (128, 128, 8, 4)])
def test_primary_cache_same_pixels(procedural_region, blue_noise, W, H, spp, depth, prepass, monkeypatch):
    """RT_FLAG_CACHE_PRIMARY traces the seed-independent primary ray once per pixel: identical planes, fewer rays.
    Both prepass kernels: k_primary2 (default: nibble map in LDS) and k_primary (RT_PRIMARY_V=1: one thread per pixel), with the
    exact cached-primary counters."""
    monkeypatch.setenv("RT_PRIMARY_V", prepass)
    mats, mine = procedural_region
    u = _uniforms(seed=3)
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, abi.RT_KERNEL_PERSISTENT,
                           flags=abi.RT_FLAG_COUNTERS | abi.RT_FLAG_CACHE_PRIMARY)
    _compare(gpu, cpu)
    assert gcn.rays_primary == W * H
    assert gcn.rays_shadow == ccn.rays_shadow and gcn.rays_diffuse == ccn.rays_diffuse
    assert gcn.rays == ccn.rays - (spp - 1) * W * H
    assert gcn.as_dict() == _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn)


def test_prepass_on_poses_regions_and_windows(blue_noise, procedural_region, region512):
    """The prepass alone (depth 0: every plane it writes, and its exact counters) for camera poses inside terrain, outside the
    region (the primary ray's first step takes the generic q), looking at the sky, through a scrolled window (lr != 0) and at
    region 512, against the oracle."""
    cases = [(256, (0, 0, 0), (-30.0, -128.0, 100.0), np.pi / 2, 0.0), (256, (0, 0, 0), (-30.0, -200.0, 60.0), np.pi / 2, -0.1),
             (256, (0, 0, 0), (10.0, 10.0, -100.0), 0.3, 0.2), (256, (0, 0, 0), (300.0, 40.0, 140.0), np.pi, -0.4),
             (256, (48, 0, 32), (18.0, -128.0, 132.0), np.pi / 2, 0.0), (256, (-32, 64, -16), (-62.0, 20.0, 70.0), -0.7, -0.3),
             (512, (0, 0, 0), (-60.0, -256.0, 110.0), np.pi / 2, -0.05)]
    for region, lr, origin, heading, pitch in cases:
        if region == 512:
            mats, mine = region512
        elif lr == (0, 0, 0):
            mats, mine = procedural_region
        else:
            mats, mine = world.toroidal_region(lr)
        u = _uniforms(origin=origin, heading=heading, pitch=pitch, sun=0.2, seed=5, lr=lr)
        W, H = 104, 72
        cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, 1, 0, region=region)
        cfg = render.make_config(W, H, spp=1, depth=0, flags=abi.RT_FLAG_COUNTERS | abi.RT_FLAG_CACHE_PRIMARY, region=region)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            gpu, gcn = ctx.readback_all(), ctx.counters()
        _compare(gpu, cpu, gcn, ccn)


@pytest.mark.parametrize("kernel", PATH_KERNELS)
@pytest.mark.parametrize("W,H", [(96, 64), (100, 60)])
def test_tile_split_contexts_reassemble_to_the_full_frame(procedural_region, blue_noise, W, H, kernel):
    """Multi-GPU layout on one GPU: two contexts render the even / odd 8x8 tiles (tile_world = 2), their tile-major
    planes are concatenated rank-major on the device (what the RCCL gather produces) and rt_untile scatters them;
    the result must equal the single-context frame (and hence the oracle) on every plane."""
    import torch
    mats, mine = procedural_region
    u = _uniforms(seed=21)
    spp, depth = 2, 3
    cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    world_n = 2
    ctxs = []
    for r in range(world_n):
        cfg = render.make_config(W, H, spp=spp, depth=depth, tile_rank=r, tile_world=world_n, kernel=kernel,
                                 flags=abi.RT_FLAG_CACHE_PRIMARY)
        c = render.Context(cfg)
        c.upload_world(mats, mine)
        c.upload_noise(blue_noise)
        c.draw_frame(u)
        c.sync()
        ctxs.append(c)
    assert sum(c.tile_count() for c in ctxs) == ((W + 7) // 8) * ((H + 7) // 8)
    assert ctxs[0].tile_capacity() == ctxs[1].tile_capacity()
    dev = torch.device("cuda", 0)
    frames = {}
    for b in range(abi.RT_BUF_FINAL_BGRA8):
        nbytes = ctxs[0].buffer_bytes(b)
        parts = [torch.from_numpy(c.readback(b).reshape(-1).view(np.uint8).copy()).to(dev) for c in ctxs]
        gathered = torch.cat(parts).contiguous()
        assert gathered.numel() == world_n * nbytes
        dt, ch = abi.BUFFER_FORMATS[b]
        bpp = np.dtype(dt).itemsize * ch
        frame = torch.zeros(W * H * bpp, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()      # the context runs on its own non-blocking stream
        ctxs[0].untile(b, gathered.data_ptr(), world_n, frame.data_ptr())
        ctxs[0].sync()
        torch.cuda.synchronize()
        got = frame.cpu().numpy().view(dt).reshape((H, W, ch) if ch > 1 else (H, W))
        assert np.array_equal(got, cpu[abi.BUFFER_NAMES[b]], equal_nan=True), abi.BUFFER_NAMES[b]
        frames[b] = frame
    # the post passes on the assembled frame (what rank 0 does after the gather): rt_denoise_planes / rt_finalize_planes
    out = torch.zeros(W * H * 4, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()
    F = frames
    ctxs[0].denoise_planes(F[abi.RT_BUF_LIGHTING_RGBA16].data_ptr(), F[abi.RT_BUF_DEPTH_R16UI].data_ptr(),
                           F[abi.RT_BUF_NORMAL_R8UI].data_ptr(), faithful=True)
    ctxs[0].finalize_planes(F[abi.RT_BUF_ALBEDO_RGBA8].data_ptr(), F[abi.RT_BUF_EMISSION_RGBA8].data_ptr(),
                            F[abi.RT_BUF_FOG_RGBA8].data_ptr(), F[abi.RT_BUF_LIGHTING_RGBA16].data_ptr(),
                            F[abi.RT_BUF_DEPTH_R16UI].data_ptr(), out.data_ptr())
    ctxs[0].sync()
    exp_den = po.denoise(cpu["lighting_rgba16"], cpu["depth_r16"], cpu["normal_r8"], faithful=True)
    exp_fin = po.finalize(cpu["albedo_rgba8"], cpu["emission_rgba8"], cpu["fog_rgba8"], exp_den, cpu["depth_r16"], blue_noise)
    den = F[abi.RT_BUF_LIGHTING_RGBA16].cpu().numpy().view(np.uint16).reshape(H, W, 4)
    assert np.array_equal(den, exp_den)
    assert np.array_equal(out.cpu().numpy().reshape(H, W, 4), exp_fin)
    with pytest.raises(render.RtError):
        ctxs[0].denoise()          # the context's own planes hold tiles, not a frame
    for c in ctxs:
        c.destroy()


def test_device_pointer_view_for_collectives(procedural_region, blue_noise):
    """bench.py hands rt_device_ptr planes to torch.distributed through the CUDA array interface (zero copy)."""
    import torch
    import bench
    mats, mine = procedural_region
    u = _uniforms(seed=2)
    cfg = render.make_config(64, 64, spp=1, depth=2)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        b = abi.RT_BUF_FOG_RGBA8
        view = torch.as_tensor(bench._DevArray(ctx.device_ptr(b), ctx.buffer_bytes(b)), device=torch.device("cuda", 0))
        assert view.data_ptr() == ctx.device_ptr(b)
        assert np.array_equal(view.cpu().numpy(), ctx.readback(b).reshape(-1).view(np.uint8))


def test_pipeline_mirror_draw_frame(blue_noise):
    """The C++ mirror of render::create_instance / Pipeline::draw_frame (pipeline.rs:134-255): seed advances per frame
    ((seed + 1) % 2^20 before the first frame, pipeline.rs:201) and the frame equals the oracle fed the same uniforms."""
    g = render.Game()
    g.generate_world(world.DEFAULT_SEED)
    cfg = render.make_config(64, 48, spp=1, depth=2)
    p = render.create_instance(cfg, g, blue_noise)
    mats, mine = world.generate_region(world.DEFAULT_SEED)
    for frame in range(2):
        p.draw_frame(g)
        p.wait()
        u = p.uniforms()
        assert u.seed == frame + 1
        assert tuple(u.lr) == (0, 0, 0)
        cpu, _ = po.render(mats, mine, blue_noise, u, 64, 48, 1, 2)
        gpu = p.context.readback_all()
        for name in cpu:
            assert np.array_equal(gpu[name], cpu[name], equal_nan=True), name
    p.close()
    g.close()


@pytest.mark.parametrize("region,flags", [(256, 0), (256, abi.RT_FLAG_TRUSTED_WORLD), (512, 0)])
def test_upload_slice_matches_full_upload(procedural_region, blue_noise, region, flags):
    """rt_upload_slice (terrain_upload.rs:84-275): patching 16-thick slabs on each axis — terrain of another seed, so mixed
    bricks, materials and every minefield value move — equals uploading the edited region; the slab path re-tiles only the
    slab (any region size); slabs assembled in the library's pinned staging (rt_slice_staging) take the same path without the
    staging copy.  A slab with a minefield value above 30 is rejected BEFORE it is applied: the region stays as it was and stays
    drawable — also after two rejections in a row (the second used to hide the first)."""
    if region == 256:
        mats, mine = procedural_region
        origin = (-20.0, -120.0, 60.0)
    else:
        mats, mine = world.generate_region(world.DEFAULT_SEED, region=region)
        origin = (-40.0, -240.0, 120.0)
    other_m, other_f = world.generate_region(world.DEFAULT_SEED + 1, region=region)
    mats2, mine2 = mats.copy(), mine.copy()
    R = region
    edits = [(0, 96), (1, 32), (2, R // 2 + 16), (0, R - 16), (2, 0), (1, R // 2)]
    for axis, off in edits:
        sl = [slice(None)] * 3
        sl[2 - axis] = slice(off, off + 16)     # arrays are [z, y, x]
        mats2[tuple(sl)] = other_m[tuple(sl)]
        mine2[tuple(sl)] = other_f[tuple(sl)]
    u = _uniforms(origin=origin, pitch=-0.3, seed=4)
    cfg = render.make_config(64, 64, spp=1, depth=2, region=region, flags=flags | abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        for k, (axis, off) in enumerate(edits):
            sl = [slice(None)] * 3
            sl[2 - axis] = slice(off, off + 16)
            if k % 2 == 0:
                ctx.upload_slice(axis, off, np.ascontiguousarray(mats2[tuple(sl)]), np.ascontiguousarray(mine2[tuple(sl)]))
            else:       # assembled in place in the pinned staging buffers
                sm, sf = ctx.slice_staging()
                sm[:] = mats2[tuple(sl)].reshape(-1)
                sf[:] = mine2[tuple(sl)].reshape(-1)
                ctx.upload_slice(axis, off, sm, sf)
        ctx.draw_frame(u)
        ctx.sync()
        a = ctx.readback_all()
        with pytest.raises(render.RtError):
            ctx.upload_slice(3, 0, np.zeros(16 * R * R, np.uint32), np.zeros(16 * R * R, np.uint8))
        with pytest.raises(render.RtError):
            ctx.upload_slice(0, 8, np.zeros(16 * R * R, np.uint32), np.zeros(16 * R * R, np.uint8))
        if not flags & abi.RT_FLAG_TRUSTED_WORLD:
            bad = np.full(16 * R * R, 6, np.uint8)
            bad[12345] = 31
            def same_frame():
                ctx.draw_frame(u)
                ctx.sync()
                b = ctx.readback_all()
                for name in a:
                    assert np.array_equal(a[name], b[name], equal_nan=True), name

            with pytest.raises(render.RtError) as e:
                ctx.upload_slice(1, 64, np.zeros(16 * R * R, np.uint32), bad)      # bad A
            assert e.value.code == abi.RT_ERR_INVALID_ARG and "above 30" in str(e.value)
            same_frame()                 # rejected before anything was written: the region is intact and drawable
            with pytest.raises(render.RtError):
                ctx.upload_slice(2, 32, np.zeros(16 * R * R, np.uint32), bad)      # bad B
            same_frame()
            ctx.upload_slice(2, 32, np.ascontiguousarray(mats2[32:48, :, :]), np.ascontiguousarray(mine2[32:48, :, :]))   # good B (unchanged content)
            same_frame()                 # ... and A's values never reached the region
    cpu, _ = po.render(mats2, mine2, blue_noise, u, 64, 64, 1, 2, region=region)
    for name in cpu:
        assert np.array_equal(a[name], cpu[name], equal_nan=True), name


def test_errors_are_reported_not_fatal(procedural_region, blue_noise):
    mats, mine = procedural_region
    cfg = render.make_config(32, 32)
    with render.Context(cfg) as ctx:
        with pytest.raises(render.RtError) as e:
            ctx.draw_frame(_uniforms())
        assert e.value.code == abi.RT_ERR_NOT_READY
        bad = mine.copy()
        bad[10, 10, 10] = 77
        with pytest.raises(render.RtError) as e:
            ctx.upload_world(mats, bad)
        assert e.value.code == abi.RT_ERR_INVALID_ARG and "above 30" in str(e.value)
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(_uniforms())
        ctx.sync()
        assert ctx.timing().frame_ms == 0          # events are recorded for contexts that ask for timings only
    for flags, frame in ((abi.RT_FLAG_TIMING, False), (abi.RT_FLAG_TIMING_ALL, True)):
        with render.Context(render.make_config(32, 32, flags=flags)) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(_uniforms())
            ctx.sync()
            t = ctx.timing()
            assert t.trace_ms > 0 and t.trace_launches >= 1 and (t.frame_ms > 0) == frame and (t.other_launches > 0) == frame


@pytest.mark.parametrize("W,H,spp,depth", [(1920, 1080, 64, 4)])
def test_full_size_properties(procedural_region, blue_noise, W, H, spp, depth):
    """BASELINE.json's headline size, checked through size-independent properties instead of the (slow) oracle:
    - determinism: two frames with the same uniforms are bit-identical;
    - the primary cache changes no pixel;
    - planes that depend only on the primary ray equal an spp=1, depth=0 frame;
    - counters obey the identities fetches = rays + iterations, rays_shadow == rays_diffuse, hits + sky + limit == rays;
    - a band of rows equals the oracle."""
    mats, mine = procedural_region
    u = _uniforms(seed=1)

    def run(spp_, depth_, flags):
        cfg = render.make_config(W, H, spp=spp_, depth=depth_, flags=flags)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            first = ctx.readback_all()
            ctx.draw_frame(u)
            ctx.sync()
            second = ctx.readback_all()
            return first, second, ctx.counters()

    a1, a2, _ = run(spp, depth, abi.RT_FLAG_CACHE_PRIMARY)
    for name in a1:
        assert np.array_equal(a1[name], a2[name], equal_nan=True), name
    b1, _, cn = run(spp, depth, abi.RT_FLAG_COUNTERS)
    for name in a1:
        assert np.array_equal(a1[name], b1[name], equal_nan=True), name
    p1, _, _ = run(1, 0, 0)
    for name in ("depth_r16", "normal_r8", "albedo_rgba8", "emission_rgba8", "fog_rgba8", "depth_f32", "fog_f32"):
        assert np.array_equal(a1[name], p1[name], equal_nan=True), name
    assert cn.frames == 2 and cn.pixels == 2 * W * H
    assert cn.minefield_fetches == cn.rays + cn.iterations
    assert cn.rays_shadow == cn.rays_diffuse and cn.rays_primary == 2 * spp * W * H
    assert cn.hits + cn.sky_exits + cn.limit_exits == cn.rays and cn.material_fetches == cn.hits
    rows = (536, 544)
    cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth, rows=rows)
    for name in cpu:
        assert np.array_equal(a1[name][rows[0]:rows[1]], cpu[name][rows[0]:rows[1]], equal_nan=True), name


def test_full_size_scrolled_region(blue_noise):
    """The headline's frame size through a scrolled window (lr != 0 — every frame once the camera has travelled): RT_KERNEL_DEFAULT
    runs k_paths' scrolled-region build there; the frame is deterministic, equals k_persist's (the generic wrap_texel walk) on
    every plane, and a band of rows equals the oracle."""
    lr = (48, 0, 32)
    mats, mine = world.toroidal_region(lr)
    u = _uniforms(origin=(18.0, -128.0, 132.0), seed=1, lr=lr)
    W, H, spp, depth = 1920, 1080, 64, 4
    frames = {}
    for kernel in (abi.RT_KERNEL_DEFAULT, abi.RT_KERNEL_PERSISTENT):
        cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=abi.RT_FLAG_CACHE_PRIMARY)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            frames[kernel] = ctx.readback_all()
            assert ctx.kernel_in_use() == (abi.RT_KERNEL_PATHS if kernel == abi.RT_KERNEL_DEFAULT else abi.RT_KERNEL_PERSISTENT)
            if kernel == abi.RT_KERNEL_DEFAULT:
                ctx.draw_frame(u)
                ctx.sync()
                again = ctx.readback_all()
                for name in again:
                    assert np.array_equal(again[name], frames[kernel][name], equal_nan=True), name
    for name in frames[abi.RT_KERNEL_DEFAULT]:
        assert np.array_equal(frames[abi.RT_KERNEL_DEFAULT][name], frames[abi.RT_KERNEL_PERSISTENT][name], equal_nan=True), name
    rows = (500, 508)
    cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth, rows=rows)
    for name in cpu:
        assert np.array_equal(frames[abi.RT_KERNEL_DEFAULT][name][rows[0]:rows[1]], cpu[name][rows[0]:rows[1]], equal_nan=True), name


def test_bench_two_rank_rehearsal_assembles_the_same_frame():
    """bench.py's N>1 path on a one-GPU box: 2 ranks share GPU 0 and gather over gloo (RCCL needs one GPU per rank; the
    driver runs that).  The frame assembled on rank 0 must hash to the same value as the single-rank frame."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    common = ["--width", "200", "--height", "120", "--spp", "4", "--depth", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_SINGLE_DEVICE="1")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29533", "bench.py", "--gpus", "2"] + common[:-1], cwd=ROOT, capture_output=True, text=True,
                         timeout=600, env=env)      # common[:-1]: the CPU baseline runs at N > 1 too (rank 0, after the timed region)
    assert two.returncode == 0, two.stderr[-3000:]
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1
    # the N > 1 line is as complete as the N = 1 line: CPU baseline, per-rank spread, and the C4 frame's own roofline
    assert j2["cpu_baseline"] is not None and j2["cpu_baseline"]["value"] > 0 and j2["cpu_baseline"]["kind"] == "port"
    spread = j2["roofline"]["ranks_path_kernel_ms_per_frame"]
    assert 0 < spread["min"] <= spread["max"]
    rays = j2["roofline"]["ranks_rays_per_frame"]
    assert rays["min"] <= rays["max"] and rays["min"] + rays["max"] == j2["config"]["rays_per_frame"]    # two ranks
    c4 = j2["c4"]
    assert c4["workload"].startswith("3840x2160 spp=256 depth=8") and c4["value"] > 0
    assert c4["roofline"]["frac"] > 0 and c4["roofline"]["kernel"] == "k_paths" and c4["roofline"]["avg_launch_ms"] > 0
    assert j2["config"]["samples_per_launch"] >= 1 and j2["config"]["light_record_bytes"] <= j2["config"]["light_record_budget_bytes"]
    assert j1["config"]["frame_sha256_16"] == j2["config"]["frame_sha256_16"]
    assert j1["config"]["rays_per_frame"] == j2["config"]["rays_per_frame"]
    for j in (j1, j2):
        for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                    "data", "config", "roofline", "cpu_baseline"):
            assert key in j
        assert set(("bound", "achieved", "peak", "unit", "frac", "traffic")) <= set(j["roofline"])


def test_bench_rccl_path_single_rank():
    """bench.py's N>1 code path over RCCL ("nccl" backend: process group bound to the device, gather of the context's
    G-buffer block, MAX/SUM reductions, barrier) with a single rank — what a one-GPU box can run of it.  Same frame hash
    as the plain run."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    common = ["--width", "200", "--height", "120", "--spp", "4", "--depth", "3", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert one.returncode == 0, one.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    for overlap, port in (("1", "29537"), ("0", "29538")):   # pipelined gather and the serial one (default)
        env = dict(os.environ, RT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                   RT_BENCH_OVERLAP=overlap)
        env.pop("RT_BENCH_BACKEND", None)
        forced = subprocess.run([sys.executable, "bench.py", "--gpus", "1"] + common, cwd=ROOT, capture_output=True, text=True, timeout=300, env=env)
        assert forced.returncode == 0, forced.stderr[-3000:]
        j2 = json.loads([l for l in forced.stdout.splitlines() if l.startswith("{")][-1])
        assert j1["config"]["frame_sha256_16"] == j2["config"]["frame_sha256_16"]
        assert j1["config"]["rays_per_frame"] == j2["config"]["rays_per_frame"]
        assert ("overlapped" in j2["config"]["gather"]) == (overlap == "1") and "rt_gather_gbuffer" in j2["config"]["gather"]


@pytest.fixture(scope="module")
def region512(native_built):
    return world.generate_region(world.DEFAULT_SEED, region=512)


@pytest.mark.parametrize("kernel,flags", [(abi.RT_KERNEL_PERSISTENT, abi.RT_FLAG_COUNTERS),
                                          (abi.RT_KERNEL_PERSISTENT, abi.RT_FLAG_CACHE_PRIMARY),
                                          (abi.RT_KERNEL_PATHS, abi.RT_FLAG_CACHE_PRIMARY),
                                          (abi.RT_KERNEL_MEGA, abi.RT_FLAG_COUNTERS)])
@pytest.mark.parametrize("pose", [
    dict(origin=(-60.0, -256.0, 110.0), heading=np.pi / 2, pitch=-0.05, sun=0.0, lr=(0, 0, 0)),
    dict(origin=(200.0, 180.0, 90.0), heading=-2.3, pitch=-0.2, sun=0.8, lr=(0, 0, 0)),
    dict(origin=(-30.0, -200.0, 100.0), heading=1.4, pitch=0.0, sun=0.2, lr=(32, -16, 0)),
])
def test_region_512_matches_oracle(region512, blue_noise, kernel, flags, pose):
    """Region-size extension (config C5 family): ROOT_BLOCK_WIDTH = 512, nibble map over 8^3 cubes."""
    mats, mine = region512
    u = po.camera_uniforms(pose["origin"], pose["heading"], pose["pitch"], pose["sun"], 11, pose["lr"])
    W, H, spp, depth = 104, 72, 2, 3
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth, region=512)
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=flags, region=512)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        gpu = ctx.readback_all()
        gcn = ctx.counters()
    if flags & abi.RT_FLAG_CACHE_PRIMARY:
        _compare(gpu, cpu)
    else:
        _compare(gpu, cpu, gcn, ccn)


def test_region_size_validation(native_built):
    with pytest.raises(render.RtError):
        render.Context(render.make_config(64, 64, region=300))
    with pytest.raises(render.RtError):
        render.Context(render.make_config(64, 64, region=512, kernel=abi.RT_KERNEL_WAVEFRONT))


@pytest.fixture(scope="module")
def region1024(native_built):
    """Config C5's scene: 1024^3 (1 GiB minefield + 4 GiB materials — larger than the 256 MiB Infinity Cache)."""
    return world.generate_region(world.DEFAULT_SEED, region=1024)


def test_region_1024_matches_oracle(blue_noise, region1024):
    """Config C5's scene size at a small frame, on k_persist and k_paths, and through a scrolled window."""
    mats, mine = region1024
    u = po.camera_uniforms((-120.0, -512.0, 400.0), np.pi / 2, -0.3, 0.0, 1)
    W, H, spp, depth = 96, 64, 2, 3
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth, region=1024)
    for kernel, flags in ((abi.RT_KERNEL_PERSISTENT, abi.RT_FLAG_COUNTERS), (abi.RT_KERNEL_PERSISTENT, abi.RT_FLAG_CACHE_PRIMARY),
                          (abi.RT_KERNEL_PATHS, abi.RT_FLAG_CACHE_PRIMARY),
                          (abi.RT_KERNEL_PATHS, abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS)):
        cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=flags, region=1024)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            gpu = ctx.readback_all()
            gcn = ctx.counters()
        if flags & abi.RT_FLAG_CACHE_PRIMARY:
            _compare(gpu, cpu)
            if flags & abi.RT_FLAG_COUNTERS:
                assert gcn.as_dict() == _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn, region=1024)
        else:
            _compare(gpu, cpu, gcn, ccn)
    # the same scene seen through a scrolled window (lr != 0): k_paths' scrolled-region build at region 1024, where the voxel
    # index takes 30 bits next to the border flag
    u = _uniforms(origin=(-104.0, -520.0, 410.0), seed=7, lr=(16, -32, 48))
    W, H, spp, depth = 48, 32, 2, 3
    cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth, region=1024)
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_PATHS, flags=abi.RT_FLAG_CACHE_PRIMARY, region=1024)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        assert ctx.kernel_in_use() == abi.RT_KERNEL_PATHS
        gpu = ctx.readback_all()
    _compare(gpu, cpu)


@pytest.mark.parametrize("region,depth", [(512, 8), (512, 9), (1024, 8), (1024, 9)])
def test_deep_paths_on_the_larger_regions(blue_noise, region512, region1024, region, depth):
    """Depth 8 and 9 at regions 512 and 1024 (VERDICT r2 #2b): k_paths<., STK = 1, LOGR = 9 / 10> — the albedo stack in global
    memory next to the large swizzle tables, the instantiation the benchmarked C5 frame runs — and k_persist, planes and the
    exact counters against the oracle."""
    mats, mine = region512 if region == 512 else region1024
    s = region // 256
    u = po.camera_uniforms((-30.0 * s, -128.0 * s, 100.0 * s), np.pi / 2, -0.25, 0.4, 5)
    W, H, spp = 40, 24, 2
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth, region=region)
    cached = _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn, region=region)
    for kernel, flags in ((abi.RT_KERNEL_PATHS, abi.RT_FLAG_CACHE_PRIMARY), (abi.RT_KERNEL_PATHS, abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS),
                          (abi.RT_KERNEL_PERSISTENT, abi.RT_FLAG_COUNTERS)):
        cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=flags, region=region)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            assert ctx.kernel_in_use() == kernel
            gpu, gcn = ctx.readback_all(), ctx.counters()
        if flags == abi.RT_FLAG_COUNTERS:
            _compare(gpu, cpu, gcn, ccn)
        else:
            _compare(gpu, cpu)
            if flags & abi.RT_FLAG_COUNTERS:
                assert gcn.as_dict() == cached


def test_far_travelled_window_reaches_the_loop_limit(blue_noise):
    """raytrace.comp:109 — far from the origin the 1e-4 nudge of :119 drops below the spacing of the positions (2^-12 at
    |x| = 2048), rays stall at cell boundaries and run into the 2048-iteration limit (a quarter of this frame's rays; the
    reference's own arithmetic, restated by the oracle).  k_paths counts iterations without testing the counter in its main
    loop and switches to tested steps when a ray comes within reach of the limit (rt_paths.hip): planes, the exact counters
    and the number of limit exits must be the oracle's, on k_persist as well."""
    lr = (2048, 0, 0)
    mats, mine = world.toroidal_region(lr)
    u = _uniforms(origin=(lr[0] - 30.0, lr[1] - 128.0, lr[2] + 100.0), pitch=-0.2, sun=0.3, seed=3, lr=lr)
    W, H, spp, depth = 64, 40, 2, 3
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    assert ccn.limit_exits > 1000
    cached = _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn)
    for kernel, flags in ((abi.RT_KERNEL_PATHS, abi.RT_FLAG_CACHE_PRIMARY), (abi.RT_KERNEL_PATHS, abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS),
                          (abi.RT_KERNEL_PERSISTENT, abi.RT_FLAG_COUNTERS)):
        cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=flags)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            assert ctx.kernel_in_use() == kernel
            gpu, gcn = ctx.readback_all(), ctx.counters()
        if flags == abi.RT_FLAG_COUNTERS:
            _compare(gpu, cpu, gcn, ccn)
        else:
            _compare(gpu, cpu)
            if flags & abi.RT_FLAG_COUNTERS:
                assert gcn.as_dict() == cached and gcn.limit_exits > 1000


def test_loop_limit_in_a_region_of_unit_cells(blue_noise, native_built):
    """The loop limit with lr = 0: a 1024^3 region whose empty space carries minefield value 1 throughout (unit steps — a valid
    minefield, just not a packed one) above a solid floor.  Diffuse rays leaving the floor diagonally need ~3000 iterations to
    cross the region and stop at 2048 (Q8: a non-air hit with material 0): k_paths<., 1, 10, true> against the oracle."""
    R = 1024
    mine = np.ones((R, R, R), dtype=np.uint8)
    mats = np.zeros((R, R, R), dtype=np.uint32)
    mine[:16] = 0
    mats[:16] = world.material_pack(2)
    u = po.camera_uniforms((-490.0, -490.0, -480.0), np.pi / 4, -0.5, 0.2, 9)
    W, H, spp, depth = 32, 16, 4, 2
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth, region=R)
    assert ccn.limit_exits > 50
    cached = _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn, region=R)
    for flags in (abi.RT_FLAG_CACHE_PRIMARY, abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS):
        cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_PATHS, flags=flags, region=R)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            assert ctx.kernel_in_use() == abi.RT_KERNEL_PATHS
            gpu, gcn = ctx.readback_all(), ctx.counters()
        _compare(gpu, cpu)
        if flags & abi.RT_FLAG_COUNTERS:
            assert gcn.as_dict() == cached and gcn.limit_exits == ccn.limit_exits


C5 = (3840, 2160, 1024, 8)     # BASELINE.json config 5 on the 1024^3 region


def test_c5_frame_properties_and_oracle_bands(blue_noise, region1024):
    """The benchmarked C5 frame at its own size (VERDICT r2 #2a): 1024^3 region, 3840x2160 spp 1024 depth 8 through
    RT_KERNEL_DEFAULT — six 172-sample launches of k_paths<false, 1, 10, true>, path indices beyond 2^30.  Determinism (drawn
    twice), primary-plane identity with an spp-1 depth-0 frame, the counting build's identities (and equal planes), and two
    8-row bands against the oracle."""
    mats, mine = region1024
    W, H, spp, depth = C5
    u = po.camera_uniforms((-120.0, -512.0, 400.0), np.pi / 2, 0.0, 0.0, 1)     # bench.py --region 1024: the default pose scaled
    cfg = render.make_config(W, H, spp=spp, depth=depth, flags=abi.RT_FLAG_CACHE_PRIMARY, region=1024)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        assert ctx.kernel_in_use() == abi.RT_KERNEL_PATHS
        inf = ctx.info()
        assert 1 <= inf.samples_per_launch <= spp and inf.light_record_bytes <= inf.light_record_budget_bytes
        a1 = ctx.readback_all()
        ctx.draw_frame(u)
        ctx.sync()
        a2 = ctx.readback_all()
    for name in a1:
        assert np.array_equal(a1[name], a2[name], equal_nan=True), name
    del a2
    cfg = render.make_config(W, H, spp=1, depth=0, region=1024)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        p1 = ctx.readback_all()
    for name in ("depth_r16", "normal_r8", "albedo_rgba8", "emission_rgba8", "fog_rgba8", "depth_f32", "fog_f32"):
        assert np.array_equal(a1[name], p1[name], equal_nan=True), name
    del p1
    cfg = render.make_config(W, H, spp=spp, depth=depth, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS, region=1024)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        b1, cn = ctx.readback_all(), ctx.counters()
    for name in a1:
        assert np.array_equal(a1[name], b1[name], equal_nan=True), name
    del b1
    assert cn.frames == 1 and cn.pixels == W * H and cn.rays_primary == W * H
    assert cn.minefield_fetches == cn.rays + cn.iterations
    assert cn.rays_shadow == cn.rays_diffuse and cn.noise_fetches == cn.rays_shadow + spp
    assert cn.hits + cn.sky_exits + cn.limit_exits == cn.rays and cn.material_fetches == cn.hits
    assert cn.rays_shadow > (1 << 30)       # more first-level paths than 2^30: the 32-bit path indices of a 172-sample launch are in use
    for rows in ((536, 544), (1336, 1344)):
        cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth, rows=rows, region=1024)
        for name in cpu:
            assert np.array_equal(a1[name][rows[0]:rows[1]], cpu[name][rows[0]:rows[1]], equal_nan=True), (name, rows)


@pytest.mark.parametrize("kernel", PATH_KERNELS)
@pytest.mark.parametrize("W,H,spp,depth", [(1, 1, 3, 2), (5, 3, 2, 4), (9, 17, 1, abi.MAX_DEPTH), (24, 16, 37, 3), (40, 24, 2, 5), (40, 24, 2, 8), (40, 24, 2, 9)])
def test_edge_shapes_match_oracle(procedural_region, blue_noise, kernel, W, H, spp, depth):
    """Frames smaller than one tile / one wave, the maximum depth, and more samples than lanes."""
    mats, mine = procedural_region
    u = _uniforms(seed=77)
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    for flags in (abi.RT_FLAG_COUNTERS, abi.RT_FLAG_CACHE_PRIMARY, abi.RT_FLAG_COUNTERS | abi.RT_FLAG_CACHE_PRIMARY):
        gpu, gcn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, kernel, flags=flags)
        if flags == abi.RT_FLAG_COUNTERS:
            _compare(gpu, cpu, gcn, ccn)
        else:
            _compare(gpu, cpu)
            if flags & abi.RT_FLAG_COUNTERS:
                assert gcn.as_dict() == _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn)


@pytest.mark.parametrize("lr,origin,heading,pitch", [((48, 0, 32), (18.0, -128.0, 132.0), np.pi / 2, 0.0),
                                                    ((-32, 64, -16), (-62.0, 20.0, 70.0), -0.7, -0.3),
                                                    ((16, -48, 0), (40.0, -160.0, 30.0), 2.4, 0.4)])
def test_scrolled_regions_run_on_k_paths(blue_noise, lr, origin, heading, pitch):
    """Once the camera has travelled every frame has lr != 0 (terrain_upload.rs:84-275): k_paths' scrolled-region build (generic
    q, lr in the sky test, the shader's own mod for the texel and its border case) against the oracle on the toroidal window
    the streaming would have uploaded — planes, the cached-primary counters, and that it is k_paths that ran."""
    mats, mine = world.toroidal_region(lr)
    u = _uniforms(origin=origin, heading=heading, pitch=pitch, sun=0.3, seed=11, lr=lr)
    W, H, spp, depth = 160, 96, 3, 4
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    for flags in (abi.RT_FLAG_CACHE_PRIMARY, abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS):
        cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_PATHS, flags=flags)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            assert ctx.kernel_in_use() == abi.RT_KERNEL_PATHS
            gpu, gcn = ctx.readback_all(), ctx.counters()
        _compare(gpu, cpu)
        if flags & abi.RT_FLAG_COUNTERS:
            assert gcn.as_dict() == _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn)


def test_scrolled_region_512_runs_on_k_paths(blue_noise):
    """The same at region 512 (2 R-entry swizzle tables, nibble-map entry from the brick coordinates)."""
    lr = (32, -16, 48)
    mats, mine = world.toroidal_region(lr, region=512)
    u = _uniforms(origin=(-28.0, -272.0, 250.0), heading=np.pi / 2, pitch=-0.1, sun=0.2, seed=4, lr=lr)
    W, H, spp, depth = 96, 64, 2, 3
    cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth, region=512)
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_PATHS, flags=abi.RT_FLAG_CACHE_PRIMARY, region=512)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        assert ctx.kernel_in_use() == abi.RT_KERNEL_PATHS
        gpu = ctx.readback_all()
    _compare(gpu, cpu)


@pytest.mark.parametrize("kernel", PATH_KERNELS)
def test_sample_batches_accumulate_in_order(procedural_region, blue_noise, kernel, monkeypatch):
    """spp larger than one launch holds: RT_PERSIST_BATCH forces 4 launches of 3 + 3 + 3 + 1 samples; the per-pixel sum must
    still run in sample order (same bits as one launch and as the oracle)."""
    mats, mine = procedural_region
    u = _uniforms(seed=5)
    W, H, spp, depth = 72, 40, 10, 3
    cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    monkeypatch.setenv("RT_PERSIST_BATCH", "3")
    for flags in (0, abi.RT_FLAG_CACHE_PRIMARY):
        gpu, _ = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, kernel, flags=flags)
        _compare(gpu, cpu)


def test_per_frame_tables_follow_the_sun(procedural_region, blue_noise):
    """The shadow-direction and sky tables are rebuilt only when the sun vector changes: frames with sun 0.0, 0.9, 0.9, 0.0
    on ONE context must each match the oracle."""
    mats, mine = procedural_region
    W, H, spp, depth = 80, 48, 2, 3
    cfg = render.make_config(W, H, spp=spp, depth=depth, flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        for sun in (0.0, 0.9, 0.9, 0.0):
            u = po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.1, sun, 7)
            cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
            ctx.draw_frame(u)
            ctx.sync()
            _compare(ctx.readback_all(), cpu)


@pytest.mark.parametrize("threshold,rmin", [(1, 1), (64, 128), (17, 3)])
@pytest.mark.parametrize("kernel", PATH_KERNELS)
def test_scheduling_parameters_do_not_change_results(procedural_region, blue_noise, kernel, threshold, rmin, monkeypatch):
    """The parked-lane threshold of the transition pass and the re-arm trigger only regroup the work: planes and counters
    stay those of the oracle at the extremes too (pass per finished lane / only when the whole wave is parked)."""
    monkeypatch.setenv("RT_PERSIST_THRESHOLD", str(threshold))
    monkeypatch.setenv("RT_PERSIST_RMIN", str(rmin))
    mats, mine = procedural_region
    u = _uniforms(seed=13)
    W, H, spp, depth = 88, 56, 3, 4
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, kernel)
    _compare(gpu, cpu, gcn, ccn)
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, kernel, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS)
    _compare(gpu, cpu)
    assert gcn.as_dict() == _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn)


@pytest.mark.parametrize("mode", ["0", "1", "2", "3"])
def test_light_records_streamed_or_cached_give_the_same_frame(procedural_region, blue_noise, mode, monkeypatch):
    """RT_PL_STREAM: k_paths' light records as plain or streaming (`nt`) stores, k_accumulate_paths reading them with plain or
    streaming loads (the library chooses by the launch's size; a frame of test size never gets there by itself) — a cache policy,
    not a value: the frame is the oracle's either way, multi-launch accumulation included."""
    monkeypatch.setenv("RT_PL_STREAM", mode)
    monkeypatch.setenv("RT_PERSIST_BATCH", "3")
    mats, mine = procedural_region
    u = _uniforms(seed=29)
    W, H, spp, depth = 104, 72, 7, 5
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, abi.RT_KERNEL_PATHS, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS)
    _compare(gpu, cpu)
    assert gcn.as_dict() == _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn)


def test_headline_frame_equals_the_oracle(procedural_region, blue_noise):
    """bench.py's workload in full — 1920x1080, spp 64, depth 4, default pose — against the oracle on every pixel of every
    plane, for both path kernels; counters (349 M rays) for the kernel RT_KERNEL_DEFAULT picks at this size."""
    mats, mine = procedural_region
    W, H, spp, depth = 1920, 1080, 64, 4
    u = render.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, 0.0, 0.0, seed=1)
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    for kernel in PATH_KERNELS:
        gpu, _ = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, kernel, flags=abi.RT_FLAG_CACHE_PRIMARY)
        _compare(gpu, cpu)
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, abi.RT_KERNEL_DEFAULT)
    _compare(gpu, cpu, gcn, ccn)
    # the counting build of the kernel the benchmark times (cached primaries): exact ray / iteration / fetch counts
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, abi.RT_KERNEL_DEFAULT,
                           flags=abi.RT_FLAG_COUNTERS | abi.RT_FLAG_CACHE_PRIMARY)
    _compare(gpu, cpu)
    assert gcn.as_dict() == _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn)


# ---- BASELINE.json config C4: 3840x2160, spp 256, depth 8 (three launches of up to 86 samples on one GPU, one per tile-split
# ---- context), as one context and as the eight-way tile split the 8-GPU run uses -------------------------------------------
C4 = (3840, 2160, 256, 8)


def _frame_hash(planes):
    import hashlib
    h = hashlib.sha256()
    for name in sorted(planes):
        h.update(np.ascontiguousarray(planes[name]).tobytes())
    return h.hexdigest()


@pytest.fixture(scope="module")
def c4_frame(procedural_region, blue_noise):
    """The C4 frame from one whole-frame context on the default kernel (cached primaries), drawn twice."""
    mats, mine = procedural_region
    W, H, spp, depth = C4
    u = _uniforms(seed=1)
    cfg = render.make_config(W, H, spp=spp, depth=depth, flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        first = ctx.readback_all()
        ctx.draw_frame(u)
        ctx.sync()
        second = ctx.readback_all()
    return u, first, second


def test_c4_frame_properties_and_oracle_bands(procedural_region, blue_noise, c4_frame):
    """3840x2160 spp 256 depth 8 through RT_KERNEL_DEFAULT: determinism, primary-plane identity with an spp-1 depth-0 frame,
    counter identities of the counting build (also un-cached: same pixels), and two 8-row bands against the oracle."""
    mats, mine = procedural_region
    W, H, spp, depth = C4
    u, a1, a2 = c4_frame
    for name in a1:
        assert np.array_equal(a1[name], a2[name], equal_nan=True), name
    p1, _ = _render_gpu(mats, mine, blue_noise, u, W, H, 1, 0, abi.RT_KERNEL_DEFAULT, flags=0)
    for name in ("depth_r16", "normal_r8", "albedo_rgba8", "emission_rgba8", "fog_rgba8", "depth_f32", "fog_f32"):
        assert np.array_equal(a1[name], p1[name], equal_nan=True), name
    b1, cn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, abi.RT_KERNEL_DEFAULT,
                         flags=abi.RT_FLAG_COUNTERS | abi.RT_FLAG_CACHE_PRIMARY)
    for name in a1:
        assert np.array_equal(a1[name], b1[name], equal_nan=True), name
    assert cn.frames == 1 and cn.pixels == W * H and cn.rays_primary == W * H
    assert cn.minefield_fetches == cn.rays + cn.iterations
    assert cn.rays_shadow == cn.rays_diffuse and cn.noise_fetches == cn.rays_shadow + spp
    assert cn.hits + cn.sky_exits + cn.limit_exits == cn.rays and cn.material_fetches == cn.hits
    for rows in ((536, 544), (1336, 1344)):     # terrain near the horizon, and sky + distant terrain
        cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth, rows=rows)
        for name in cpu:
            assert np.array_equal(a1[name][rows[0]:rows[1]], cpu[name][rows[0]:rows[1]], equal_nan=True), (name, rows)


def test_c4_eight_way_tile_split_reassembles(procedural_region, blue_noise, c4_frame):
    """The layout of the 8-GPU run on one GPU: eight contexts render tiles t % 8 == rank of the C4 frame, their G-buffer blocks
    are laid out rank-major (what the gather produces) and rt_untile_gbuffer scatters them: same frame, bit for bit."""
    import torch
    mats, mine = procedural_region
    W, H, spp, depth = C4
    u, whole, _ = c4_frame
    world_n = 8
    dev = torch.device("cuda", 0)
    gathered = None
    keep = None
    for r in range(world_n):
        cfg = render.make_config(W, H, spp=spp, depth=depth, tile_rank=r, tile_world=world_n, flags=abi.RT_FLAG_CACHE_PRIMARY)
        c = render.Context(cfg)
        c.upload_world(mats, mine)
        c.upload_noise(blue_noise)
        c.draw_frame(u)
        c.sync()
        gbytes = c.gbuffer_bytes()
        if gathered is None:
            gathered = torch.zeros(world_n * gbytes, dtype=torch.uint8, device=dev)
        view = torch.as_tensor(_DevBytes(c.gbuffer_ptr(), gbytes), device=dev)
        gathered[r * gbytes:(r + 1) * gbytes].copy_(view)
        torch.cuda.synchronize()
        if r == 0:
            keep = c
        else:
            c.destroy()
    ids = list(range(abi.RT_BUF_FOG_RGBA8 + 1))
    frames = []
    for b in ids:
        dt, ch = abi.BUFFER_FORMATS[b]
        frames.append(torch.zeros(W * H * np.dtype(dt).itemsize * ch, dtype=torch.uint8, device=dev))
    torch.cuda.synchronize()
    keep.untile_gbuffer(gathered.data_ptr(), world_n, [fr.data_ptr() for fr in frames])
    keep.sync()
    torch.cuda.synchronize()
    for b, fr in zip(ids, frames):
        dt, ch = abi.BUFFER_FORMATS[b]
        got = fr.cpu().numpy().view(dt).reshape((H, W, ch) if ch > 1 else (H, W))
        assert np.array_equal(got, whole[abi.BUFFER_NAMES[b]]), abi.BUFFER_NAMES[b]
    keep.destroy()


class _DevBytes:
    """Zero-copy view of device memory for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def test_bench_gather_delivers_the_frame_of_its_own_step():
    """Every step renders a different frame (--vary-seed, no warm-up): the frame rank 0 assembled after the last step must be
    the frame of THAT step's seed — a gather that ran ahead of its frame's kernels (or an un-tile ahead of its gather) would
    deliver an older one.  RCCL path with one rank (serial and overlapped), and the two-rank gloo rehearsal."""
    import json
    import subprocess
    import sys
    from tests.conftest import ROOT
    size = ["--width", "200", "--height", "120", "--spp", "4", "--depth", "3", "--no-cpu-baseline"]

    def line(cmd, env=None):
        r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])

    want = line([sys.executable, "bench.py", "--gpus", "1", "--seed", "7", "--steps", "1", "--warmup", "0"] + size)
    prev = line([sys.executable, "bench.py", "--gpus", "1", "--seed", "6", "--steps", "1", "--warmup", "0"] + size)
    assert want["config"]["frame_sha256_16"] != prev["config"]["frame_sha256_16"]
    vary = ["--seed", "4", "--vary-seed", "--steps", "4", "--warmup", "0"]      # frames of seeds 4, 5, 6, 7
    for overlap, port in (("0", "29541"), ("1", "29542")):
        env = dict(os.environ, RT_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                   RT_BENCH_OVERLAP=overlap)
        env.pop("RT_BENCH_BACKEND", None)
        got = line([sys.executable, "bench.py", "--gpus", "1"] + vary + size, env)
        assert got["config"]["seed"] == 7 and got["config"]["frame_sha256_16"] == want["config"]["frame_sha256_16"], overlap
    env = dict(os.environ, RT_BENCH_BACKEND="gloo", RT_BENCH_SINGLE_DEVICE="1")
    got = line([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", "29543", "bench.py", "--gpus", "2", "--no-c4"] + vary + size, env)
    assert got["n_gpus"] == 2 and got["config"]["frame_sha256_16"] == want["config"]["frame_sha256_16"]


@pytest.mark.parametrize("overlapped", [False, True])
def test_gather_gbuffer_through_the_c_abi(procedural_region, blue_noise, overlapped):
    """rt_comm_unique_id / rt_comm_init_rank / rt_gather_gbuffer / rt_comm_destroy with the one-rank communicator a one-GPU
    box allows: the block travels through RCCL (send to and receive from itself on the context's stream) into the staging
    area and from there into six caller-owned planes; three frames in a row, each checked against the context's own planes."""
    import torch
    mats, mine = procedural_region
    W, H, spp, depth = 136, 72, 2, 3
    dev = torch.device("cuda", 0)
    cfg = render.make_config(W, H, spp=spp, depth=depth, flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        comm = ctx.comm_init_rank(render.comm_unique_id())
        ids = list(range(abi.RT_BUF_FOG_RGBA8 + 1))
        frames = []
        for b in ids:
            dt, ch = abi.BUFFER_FORMATS[b]
            frames.append(torch.zeros(W * H * np.dtype(dt).itemsize * ch, dtype=torch.uint8, device=dev))
        torch.cuda.synchronize()
        for seed in (3, 4, 5):
            ctx.draw_frame(_uniforms(seed=seed))
            ctx.gather_gbuffer(comm, 0, [fr.data_ptr() for fr in frames], overlapped=overlapped)
            ctx.sync()                 # both streams
            for b, fr in zip(ids, frames):
                assert np.array_equal(fr.cpu().numpy(), ctx.readback(b).reshape(-1).view(np.uint8)), (seed, abi.BUFFER_NAMES[b])
        with pytest.raises(render.RtError):
            ctx.gather_gbuffer(comm, 1, [fr.data_ptr() for fr in frames])      # root outside the communicator
        render.comm_destroy(comm)


def test_rt_bench_binary_runs_and_reports_metrics(native_built):
    """The headless counterpart of src/bin/main.rs: six positional floats (x y z heading pitch sun_angle, game/mod.rs:45-52),
    a short run, the reference-style `avg / max` line (main.rs:45-46) and one JSON line with config, rays, ms and Mrays/s —
    plain, and with the frame-end gather on a one-rank communicator (what --gpus N does per device), serial and overlapped."""
    import json
    import subprocess
    from tests.conftest import ROOT
    exe = os.path.join(ROOT, "raytrace_amd", "rt_bench")
    base = [exe, "-30", "-128", "100", "1.5707964", "-0.2", "0.3", "--width", "256", "--height", "128", "--spp", "2", "--depth", "3", "--frames", "8"]
    lines = {}
    for tag, extra in (("plain", []), ("gather", ["--gather"]), ("overlap", ["--gather", "--overlap"])):
        r = subprocess.run(base + extra, cwd=ROOT, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        out = r.stdout.splitlines()
        assert any(l.endswith("ms") and " / " in l for l in out), out
        j = json.loads([l for l in out if l.startswith("{")][-1])
        assert j["binary"] == "rt_bench" and j["frames"] == 8 and j["config"]["gpus"] == 1
        assert j["config"]["pose"][:3] == [-30, -128, 100] and abs(j["config"]["sun_angle"] - 0.3) < 1e-6
        assert j["rays_per_frame"] > 256 * 128 and j["ms_per_frame"] > 0 and j["mrays_per_s"] > 0
        lines[tag] = j
    assert lines["plain"]["depth_plane_checksum"] == 0
    assert lines["gather"]["depth_plane_checksum"] == lines["overlap"]["depth_plane_checksum"] > 0
    assert lines["gather"]["rays_per_frame"] == lines["plain"]["rays_per_frame"]


@pytest.mark.parametrize("kernel", [abi.RT_KERNEL_PATHS, abi.RT_KERNEL_PERSISTENT])
@pytest.mark.parametrize("origin,special", [((-200.0, 10.0, 30.0), True), ((-200.0, 10.0, 124.0), False), ((60.0, 300.0, 20.0), True)])
def test_cameras_outside_the_region(procedural_region, blue_noise, kernel, origin, special):
    """The region texture wraps (mod(p + 128, 256), raytrace.comp:137), so a camera outside it fetches the texel on the far side:
    under the terrain there, every primary ray is a fresh ray on a 0 — "special" (Q12: a non-air hit at a NaN position, whose
    level rays are dead) —, above it the ray takes one step and leaves.  k_paths reads "reached the sky" off a ray's position,
    and a ray that never moved from out there must not pass for one that left (the prepass analogue of this was found by the
    fuzz cases): planes and exact counters against the oracle."""
    mats, mine = procedural_region
    u = _uniforms(origin=origin, heading=0.0, pitch=-0.1, sun=0.3, seed=13)
    W, H, spp, depth = 96, 64, 3, 3
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    assert bool(np.isnan(cpu["depth_f32"]).all()) == special
    cached = _cached_counters(mats, mine, blue_noise, u, W, H, spp, depth, ccn)
    for flags in (abi.RT_FLAG_CACHE_PRIMARY, abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_COUNTERS):
        gpu, gcn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, kernel, flags=flags)
        _compare(gpu, cpu)
        if flags & abi.RT_FLAG_COUNTERS:
            assert gcn.as_dict() == cached
