"""GPU parity: the HIP path (through the C ABI) against the CPU oracle on identical inputs.

Bar: bit-exact for every integer plane and — because both sides compile the same arithmetic contract
(include/rt_math.h) with contraction off — also for the fp32 planes; the stated tolerance of the path
(BASELINE.json: per-pixel RMS <= 1e-4) is asserted as well so a future relaxation of exactness still has a gate.
"""
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


@pytest.mark.parametrize("W,H,spp,depth", [(64, 64, 1, 2), (96, 72, 4, 0), (100, 60, 5, 3), (128, 128, 8, 4)])
def test_primary_cache_same_pixels(procedural_region, blue_noise, W, H, spp, depth):
    """RT_FLAG_CACHE_PRIMARY traces the seed-independent primary ray once per pixel: identical planes, fewer rays."""
    mats, mine = procedural_region
    u = _uniforms(seed=3)
    cpu, ccn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    gpu, gcn = _render_gpu(mats, mine, blue_noise, u, W, H, spp, depth, abi.RT_KERNEL_PERSISTENT,
                           flags=abi.RT_FLAG_COUNTERS | abi.RT_FLAG_CACHE_PRIMARY)
    _compare(gpu, cpu)
    assert gcn.rays_primary == W * H
    assert gcn.rays_shadow == ccn.rays_shadow and gcn.rays_diffuse == ccn.rays_diffuse
    assert gcn.rays == ccn.rays - (spp - 1) * W * H
