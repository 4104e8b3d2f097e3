"""GPU parity of k_frame (RT_KERNEL_FRAME, raytrace_amd/csrc/rt_frame.hip): the whole frame in one launch — what RT_KERNEL_DEFAULT
runs for the reference's own frames (1024 x 1024, 1 sample, depth 2: src/render/constants.rs:9-10, pipeline.rs:44-45,86-90).

Same bar as tests/test_gpu_parity.py: every plane bit for bit against the oracle, and — the kernel traces a pixel's primary ray
once (RT_FLAG_CACHE_PRIMARY semantics) — the exact counters the oracle implies for cached primaries."""
import numpy as np
import pytest

from raytrace_amd import abi, render, tiles, world
from oracle import pyoracle as po
from tests import scenes
from tests.test_gpu_parity import _cached_counters, _compare, _uniforms

pytestmark = pytest.mark.gpu

CACHE = abi.RT_FLAG_CACHE_PRIMARY


def _render_frame_kernel(mats, mine, noise, u, W, H, spp, depth, flags=CACHE | abi.RT_FLAG_COUNTERS, region=256, frames=1, kernel=abi.RT_KERNEL_FRAME, **cfgkw):
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=flags, region=region, **cfgkw)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(noise)
        for _ in range(frames):
            ctx.reset_counters()
            ctx.draw_frame(u)
        ctx.sync()
        assert ctx.kernel_in_use() == abi.RT_KERNEL_FRAME
        return ctx.readback_all(), ctx.counters()


def _check(mats, mine, noise, u, W, H, spp, depth, region=256, **kw):
    okw = {"region": region} if region != 256 else {}
    cpu, ccn = po.render(mats, mine, noise, u, W, H, spp, depth, **okw)
    want = _cached_counters(mats, mine, noise, u, W, H, spp, depth, ccn, **okw)
    gpu, gcn = _render_frame_kernel(mats, mine, noise, u, W, H, spp, depth, region=region, **kw)
    _compare(gpu, cpu)
    got = gcn.as_dict()
    assert got == want, "counters differ: %s" % {k: (got[k], want[k]) for k in got if got[k] != want[k]}


@pytest.mark.parametrize("W,H,spp,depth", [
    (64, 64, 1, 2),       # the reference's frame in small
    (128, 128, 1, 0),     # primary rays only
    (96, 72, 2, 1),
    (128, 128, 3, 4),
    (100, 60, 2, 3),      # partial tiles
    (8, 8, 1, 2),         # one tile: one wave of a four-wave workgroup has work
    (328, 200, 1, 2),     # 1025 tiles: a last workgroup with one tile
    (96, 96, 2, 8),       # the deepest frame the kernel takes (albedo stack: seven LDS rows)
    (64, 48, 17, 3),      # more samples than a workgroup has waves: a pixel's paths run on many lanes, its sum is taken in sample order
    (40, 24, 64, 2),      # 15 tiles, 64 samples: one workgroup's queue handed out 64 times over
])
def test_frame_kernel_matches_oracle(procedural_region, blue_noise, W, H, spp, depth):
    mats, mine = procedural_region
    _check(mats, mine, blue_noise, _uniforms(seed=1), W, H, spp, depth)


@pytest.mark.parametrize("group_tiles", ["1", "3", "4", "7", "8", "13", "16"])
def test_frame_kernel_tiles_per_group_and_threshold_do_not_change_results(procedural_region, blue_noise, group_tiles, monkeypatch):
    """RT_FRAME_GROUP_TILES (tiles of a four-wave workgroup: fewer than waves — some waves skip the primary phase and only share the
    paths —, whole tiles per wave, or a ragged deal) and RT_FRAME_THRESHOLD (parked lanes per pass) are scheduling only.  704 x 400 =
    4400 tiles: the last workgroup's tile indices run past the end for most values."""
    monkeypatch.setenv("RT_FRAME_GROUP_TILES", group_tiles)
    monkeypatch.setenv("RT_FRAME_THRESHOLD", {"1": "1", "3": "64", "4": "17", "7": "40", "8": "5", "13": "33", "16": "44"}[group_tiles])
    mats, mine = procedural_region
    u = _uniforms(origin=(100.0, 100.0, 60.0), heading=-2.0, pitch=-0.1, sun=0.7, seed=23)
    _check(mats, mine, blue_noise, u, 704, 400, 2 if int(group_tiles) % 2 else 1, 3)


@pytest.mark.parametrize("pose", [
    dict(origin=(100.0, 100.0, 60.0), heading=-2.0, pitch=-0.1, sun=0.7),
    dict(origin=(-30.0, -200.0, 100.0), heading=np.pi / 2, pitch=-0.2, sun=-0.7),   # outside the region (raytrace.comp:311-315)
    dict(origin=(10.0, 10.0, 5.0), heading=1.0, pitch=0.3, sun=1.2),                 # camera inside solid ground (degenerate start)
    dict(origin=(0.0, 0.0, 120.0), heading=0.3, pitch=-1.2, sun=0.0),                # looking steeply down
    dict(origin=(300.0, 40.0, 90.0), heading=3.0, pitch=-0.1, sun=0.4),              # camera outside the region, rays enter it
])
def test_frame_kernel_poses(procedural_region, blue_noise, pose):
    mats, mine = procedural_region
    u = _uniforms(pose["origin"], pose["heading"], pose["pitch"], pose["sun"], seed=77)
    _check(mats, mine, blue_noise, u, 96, 96, 2, 3)


@pytest.mark.parametrize("scene", ["empty", "floor", "voxel", "stairs", "blocks"])
def test_frame_kernel_analytic_scenes(native_built, blue_noise, scene):
    ids = {"empty": scenes.empty_ids, "floor": scenes.floor_ids, "voxel": scenes.single_voxel_ids,
           "stairs": scenes.staircase_ids, "blocks": scenes.random_blocks_ids}[scene]()
    mats, mine = world.region_from_ids(ids)
    u = _uniforms(origin=(-40.0, -100.0, 90.0), heading=1.1, pitch=-0.5, sun=0.3, seed=5)
    _check(mats, mine, blue_noise, u, 80, 80, 2, 3)


@pytest.mark.parametrize("lr", [(16, 32, 0), (-48, 0, 32)])
def test_frame_kernel_scrolled_regions(native_built, blue_noise, lr):
    """lr != 0 (the TerrainUploadManager's render offset, pipeline.rs:203-207): the generic wrap / border instantiation."""
    mats, mine = world.toroidal_region(lr)
    u = _uniforms(origin=(-14.0 + lr[0], -100.0 + lr[1], 100.0 + lr[2]), seed=9, lr=lr)
    _check(mats, mine, blue_noise, u, 96, 96, 1, 2)
    _check(mats, mine, blue_noise, u, 104, 56, 3, 4)


def test_frame_kernel_seed_clamp_and_wrap(procedural_region, blue_noise):
    """Q4: seed / 512 beyond the noise height clamps to row 511; seeds wrap at RT_NOISE_BYTES (pipeline.rs:201)."""
    mats, mine = procedural_region
    _check(mats, mine, blue_noise, _uniforms(seed=abi.NOISE_BYTES - 1), 64, 64, 3, 2)


@pytest.mark.parametrize("region", [512, 1024])
def test_frame_kernel_on_the_larger_regions(native_built, blue_noise, region):
    mats, mine = world.generate_region(world.DEFAULT_SEED, region=region)
    s = region // 256
    u = po.camera_uniforms((-30.0 * s, -128.0 * s, 110.0 * s), np.pi / 2, -0.05, 0.2, 11, (0, 0, 0))
    _check(mats, mine, blue_noise, u, 104, 72, 2, 3, region=region)


def test_frame_kernel_far_window_reaches_the_loop_limit(native_built, blue_noise):
    """raytrace.comp:109: a window 2048 voxels from the origin, where floats are 2^-13 apart and a quarter of the rays stall until
    the 2048-iteration limit (quirk Q8) — the same frame tests/test_gpu_parity.py holds the persistent kernels to."""
    lr = (2048, 0, 0)
    mats, mine = world.toroidal_region(lr)
    u = _uniforms(origin=(lr[0] - 30.0, lr[1] - 128.0, lr[2] + 100.0), pitch=-0.2, sun=0.3, seed=3, lr=lr)
    W, H, spp, depth = 64, 40, 2, 3
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    assert ccn.limit_exits > 1000
    _check(mats, mine, blue_noise, u, W, H, spp, depth)


def test_the_reference_frame_runs_on_the_frame_kernel_by_default(procedural_region, blue_noise):
    """RT_KERNEL_DEFAULT: one-sample frames of fewer than 2.5 M pixels — the reference's 1024 x 1024 — and multi-sample frames of fewer
    than 1.5 M pixel-sample-levels run on k_frame, everything larger on the persistent kernels.  The whole 1024 x 1024 frame against the oracle
    (two frames drawn: the second must not depend on anything the first left behind), counters included."""
    mats, mine = procedural_region
    W = H = 1024
    u = po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, 0.0, 0.0, 1)
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, 1, 2)
    gpu, gcn = _render_frame_kernel(mats, mine, blue_noise, u, W, H, 1, 2, frames=2, kernel=abi.RT_KERNEL_DEFAULT)
    _compare(gpu, cpu)
    assert gcn.as_dict() == _cached_counters(mats, mine, blue_noise, u, W, H, 1, 2, ccn)
    for (w, h, spp, want) in ((2304, 1152, 1, abi.RT_KERNEL_PERSISTENT), (512, 512, 4, abi.RT_KERNEL_PERSISTENT), (256, 256, 2, abi.RT_KERNEL_FRAME),
                               (256, 256, 1, abi.RT_KERNEL_FRAME), (1920, 1080, 1, abi.RT_KERNEL_FRAME)):
        with render.Context(render.make_config(w, h, spp=spp, depth=2, flags=CACHE)) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            if want == abi.RT_KERNEL_FRAME:
                assert ctx.kernel_in_use() == want     # before the first frame: what it will run
            ctx.draw_frame(u)
            ctx.sync()
            assert ctx.kernel_in_use() == want


def test_c2_primary_rays_only_at_full_size(procedural_region, blue_noise):
    """BASELINE.json config 2 — 1920 x 1080, one sample, primary rays only — is a k_frame frame under RT_KERNEL_DEFAULT (phase A alone:
    no pixel enters a queue): the whole frame against the oracle, counters included, drawn twice."""
    mats, mine = procedural_region
    W, H = 1920, 1080
    u = _uniforms(seed=1)
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, 1, 0)
    gpu, gcn = _render_frame_kernel(mats, mine, blue_noise, u, W, H, 1, 0, flags=abi.RT_FLAG_COUNTERS, frames=2, kernel=abi.RT_KERNEL_DEFAULT)
    _compare(gpu, cpu, gcn, ccn)
    assert gcn.rays == gcn.rays_primary == W * H


def test_more_samples_than_the_light_records_hold_fall_back(procedural_region, blue_noise, monkeypatch):
    """With more than one sample per pixel k_frame parks every path's light in the lane's light-record array and adds a pixel's
    samples in order when its workgroup's paths have ended; a frame whose samples do not fit one launch's records (RT_PERSIST_BATCH
    bounds them here) runs on the persistent kernels — and is the same frame."""
    monkeypatch.setenv("RT_PERSIST_BATCH", "2")
    mats, mine = procedural_region
    u = _uniforms(seed=8)
    W, H, spp, depth = 88, 56, 5, 3
    cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    for want_spp, want in ((5, None), (2, abi.RT_KERNEL_FRAME)):
        cfg = render.make_config(W, H, spp=want_spp, depth=depth, kernel=abi.RT_KERNEL_FRAME, flags=CACHE)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            if want is None:
                assert ctx.kernel_in_use() in (abi.RT_KERNEL_PATHS, abi.RT_KERNEL_PERSISTENT)
                _compare(ctx.readback_all(), cpu)
            else:
                assert ctx.kernel_in_use() == want
                _compare(ctx.readback_all(), po.render(mats, mine, blue_noise, u, W, H, want_spp, depth)[0])


def test_one_sample_frames_need_no_cache_flag(procedural_region, blue_noise):
    """With one sample per pixel every kernel traces the primary ray once, so RT_KERNEL_DEFAULT runs such a frame on k_frame whether
    or not the host set RT_FLAG_CACHE_PRIMARY (the C++ mirror and a host bound against ABI 1.0 do not): planes and the oracle's own
    (un-cached) counters."""
    mats, mine = procedural_region
    u = _uniforms(seed=6)
    W, H = 136, 72
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, 1, 2)
    cfg = render.make_config(W, H, spp=1, depth=2, flags=abi.RT_FLAG_COUNTERS)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        assert ctx.kernel_in_use() == abi.RT_KERNEL_FRAME
        ctx.draw_frame(u)
        ctx.sync()
        assert ctx.kernel_in_use() == abi.RT_KERNEL_FRAME
        gpu, gcn = ctx.readback_all(), ctx.counters()
    _compare(gpu, cpu, gcn, ccn)


def test_frames_the_frame_kernel_does_not_cover_run_on_the_persistent_kernels(procedural_region, blue_noise):
    """RT_KERNEL_FRAME asked for a frame outside k_frame's range (depth 9: eight albedo-stack levels; or no primary cache): the
    context falls back to k_paths / k_persist, says so, and the frame is still the oracle's."""
    mats, mine = procedural_region
    u = _uniforms(seed=4)
    for depth, flags in ((9, CACHE), (2, 0)):
        cfg = render.make_config(72, 64, spp=2, depth=depth, kernel=abi.RT_KERNEL_FRAME, flags=flags)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            assert ctx.kernel_in_use() in (abi.RT_KERNEL_PATHS, abi.RT_KERNEL_PERSISTENT)
            gpu = ctx.readback_all()
        cpu, _ = po.render(mats, mine, blue_noise, u, 72, 64, 2, depth)
        _compare(gpu, cpu)


def test_frame_kernel_tile_split_reassembles(procedural_region, blue_noise):
    """Three ranks' shares (tile_rank / tile_world) rendered by k_frame and re-assembled equal the whole frame."""
    mats, mine = procedural_region
    W, H, spp, depth = 200, 120, 2, 3
    u = _uniforms(seed=31)
    cpu, _ = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    world_ = 3
    for rank in range(world_):
        cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_FRAME, flags=CACHE, tile_rank=rank, tile_world=world_)
        with render.Context(cfg) as ctx:
            ctx.upload_world(mats, mine)
            ctx.upload_noise(blue_noise)
            ctx.draw_frame(u)
            ctx.sync()
            assert ctx.kernel_in_use() == abi.RT_KERNEL_FRAME
            got = ctx.readback_all()
        n = tiles.tile_count(W, H, rank, world_) * 64
        inside = tiles.tile_major_from_frame(np.ones((H, W), dtype=np.uint8), rank, world_)[:n].astype(bool)
        for name in cpu:
            exp = tiles.tile_major_from_frame(cpu[name], rank, world_)
            px = got[name].reshape((-1,) + exp.shape[1:])[:n]
            assert np.array_equal(px[inside], exp[:n][inside], equal_nan=True), (name, rank)


def test_frame_kernel_two_frames_in_flight(procedural_region, blue_noise):
    """Six frames with different cameras, seeds and two sun angles enqueued back to back with RT_FLAG_FRAMES_IN_FLIGHT_2 (consecutive
    k_frame launches go to the context's two streams and frame slots): the last frame is the last uniforms' frame."""
    mats, mine = procedural_region
    W, H, spp, depth = 264, 136, 1, 2
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=abi.RT_KERNEL_FRAME, flags=CACHE | abi.RT_FLAG_FRAMES_IN_FLIGHT_2)
    us = [po.camera_uniforms((-30.0 + 3 * i, -128.0, 100.0 - 2 * i), np.pi / 2 + 0.05 * i, -0.02 * i, 0.0 if i % 3 else 0.7, 11 + 5 * i) for i in range(6)]
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        for u in us:
            ctx.draw_frame(u)
        ctx.sync()
        got = ctx.readback_all()
        assert ctx.kernel_in_use() == abi.RT_KERNEL_FRAME
    cpu, _ = po.render(mats, mine, blue_noise, us[-1], W, H, spp, depth)
    _compare(got, cpu)
