"""Known-answer tests of the CPU oracle, derived by hand from shaders/glsl/raytrace.comp.
The reference holds no vectors for this path, so these are what pins the oracle."""
import math

import numpy as np
import pytest

from raytrace_amd import world
from oracle import pyoracle as po
from tests import scenes


@pytest.fixture(scope="module")
def empty_region(native_built):
    return world.region_from_ids(scenes.empty_ids())


@pytest.fixture(scope="module")
def floor_region(native_built):
    return world.region_from_ids(scenes.floor_ids(0, material=2))


def test_k1_empty_region_is_all_sky(empty_region, blue_noise):
    mats, mine = empty_region
    assert (mine == 6).all() and (mats == 0).all()          # chunk.rs:154-161
    u = po.camera_uniforms((-30.0, -128.0, 100.0), math.pi / 2, 0.0, 0.0, 1)
    W = H = 32
    planes, cn = po.render(mats, mine, blue_noise, u, W, H, 1, 2)
    assert (planes["normal_r8"] == 16).all()                 # raytrace.comp:366-370
    assert (planes["depth_r16"] == 0xFFFF).all()             # :357
    assert (planes["albedo_rgba8"] == 255).all()             # :371-375 vec4(1.0)
    assert (planes["emission_rgba8"] == 0).all()             # :376-380 vec4(0.0)
    assert cn.rays == W * H and cn.rays_shadow == 0 and cn.hits == 0 and cn.sky_exits == W * H
    # lighting = sample_sky(dir, ..., true) / 16 and fog = sample_sky(dir, ..., false) / 2 for an independent direction
    fwd = np.array(u.forward[:], dtype=np.float32)
    up = np.array(u.up[:], dtype=np.float32)
    right = np.array(u.right[:], dtype=np.float32)
    for (px, py) in ((0, 0), (W // 2, H // 2), (W - 1, 3), (5, H - 1)):
        sx = np.float32(px) / np.float32(W) * np.float32(2) - np.float32(1)
        sy = np.float32(py) / np.float32(H) * np.float32(2) - np.float32(1)
        d = po.normalize(tuple(fwd + sx * right + sy * up))
        sky = po.sample_sky(d, 0.0, True)
        assert np.allclose(planes["lighting_f32"][py, px, :3], sky / 16.0, rtol=0, atol=1e-6)
        assert planes["lighting_f32"][py, px, 3] == 1.0 / 16.0
        fog = po.sample_sky(d, 0.0, False)
        assert np.allclose(planes["fog_f32"][py, px, :3], fog / 2.0, rtol=0, atol=1e-6)


def test_k1_iteration_count_of_the_centre_ray(empty_region):
    """Straight along +y from y=-128 through 32-cells.  The start sits exactly on a cell face, where
    mod(-(y+128), 32) = 0, so the first iteration only nudges the ray 1e-4 into the cell (raytrace.comp:119); then
    the faces y=-96..128 are crossed one per iteration: 1 + 8 = 9."""
    mats, mine = empty_region
    h = po.trace_ray(mats, mine, (-30.0, -128.0, 100.0), (0.0, 1.0, 0.0))
    assert h.air == 1 and h.iterations == 9 and h.normal == 3     # NORMAL_y + 1 (moving +y)
    # started strictly inside a cell the nudge disappears
    h = po.trace_ray(mats, mine, (-30.0, -120.0, 100.0), (0.0, 1.0, 0.0))
    assert h.air == 1 and h.iterations == 8
    assert abs(h.position[1] - 128.0) < 2e-3


def test_k2_solid_floor(floor_region, blue_noise):
    mats, mine = floor_region
    h = po.trace_ray(mats, mine, (0.0, 0.0, 10.0), (0.0, 0.0, -1.0))
    assert h.air == 0 and h.normal == 4                            # NORMAL_z, ray moving -z
    assert h.packed_material == world.material_pack(2)
    assert np.allclose(h.albedo[:], np.array([39, 110, 61]) / 127.0, atol=1e-7)
    assert abs(h.distance - 10.0) < 1e-3
    # full frame looking straight down: centre pixel depth = uint(|origin - pos| * 32) where pos is pushed 0.001 off the face
    u = po.camera_uniforms((0.0, 0.0, 10.0), math.pi / 2, -math.pi / 2, 0.0, 1)
    planes, _ = po.render(mats, mine, blue_noise, u, 16, 16, 1, 2)
    d = int(planes["depth_r16"][8, 8])
    assert d in (319, 320)
    assert d == int(planes["depth_f32"][8, 8])
    assert planes["normal_r8"][8, 8] == 4
    assert tuple(planes["albedo_rgba8"][8, 8]) == (78, 221, 122, 255)   # round(39,110,61 / 127 * 255)
    assert tuple(planes["emission_rgba8"][8, 8]) == (0, 0, 0, 255)


def test_k5_pixel_interleave_and_its_inverse():
    lib = po.lib()
    assert lib.rt_oracle_pixel_of(17, 1) == 145                    # raytrace.comp:291-294
    for p in range(3840):
        wg = lib.rt_oracle_workgroup_of(p)
        assert lib.rt_oracle_pixel_of(wg, (p % 128) // 16) == p
    # the forward map covers each of 256 coordinates exactly once over 32 workgroups x 8 locals
    seen = sorted(lib.rt_oracle_pixel_of(wg, l) for wg in range(32) for l in range(8))
    assert seen == list(range(256))


def test_k6_camera_uniforms():
    u = po.camera_uniforms((-30.0, -128.0, 100.0), math.pi / 2, 0.0, 0.25, 7, (1, 2, 3))
    assert abs(u.forward[0] + 4.371139e-8) < 1e-12 and u.forward[1] == 1.0 and u.forward[2] == 0.0
    assert abs(u.up[2] - 0.4) < 1e-7 and abs(u.up[0]) < 1e-7 and abs(u.up[1]) < 1e-7
    assert abs(u.right[0] - 0.4) < 1e-7 and abs(u.right[1]) < 1e-7 and abs(u.right[2]) < 1e-7
    assert u.seed == 7 and u.sun_angle == 0.25 and tuple(u.lr) == (1, 2, 3) and tuple(u.lso) == (1, 2, 3)
    assert tuple(u.origin) == (-30.0, -128.0, 100.0)


def test_k7_sun_at_angle_zero():
    a, c = po.sun(0.0)
    ref = np.array([0.25, 0.0, 1.0]) / math.sqrt(0.25 ** 2 + 1.0)
    assert np.allclose(a, ref, atol=1e-7)
    assert np.allclose(c, np.array([0.9647, 0.7843, 0.8824], dtype=np.float32) * 2, atol=1e-7)   # main colour
    # below the horizon (raytrace.comp:266-268): mix(sunset, vec3(0), sun_amount * 2) with sun_amount = 1 extrapolates
    # to -sunset — the shader really produces negative sunlight there
    a2, c2 = po.sun(-2.0)
    assert a2[2] < 0
    assert np.allclose(c2, -np.array([0.7412, 0.2157, 0.1686], dtype=np.float32) * 2, atol=1e-6)


def test_k8_noise_addressing(blue_noise):
    base, off, vt, val = po.noise_lookup(blue_noise, 1, 0, 0)
    assert base == (1, 0) and off == (168.0, 91.0) and vt == (168, 91)
    n = blue_noise.reshape(512, 512, 4)
    assert np.allclose(val, n[91, 168] / 255.0, atol=1e-7)
    # Q3: the offset is keyed by the owning workgroup: pixel 17 belongs to workgroup 1 -> +8
    _, off17, _, _ = po.noise_lookup(blue_noise, 1, 17, 0)
    assert off17 == (176.0, 91.0)
    # Q4: seed / 512 beyond row 511 clamps (CLAMP_TO_EDGE)
    base_hi, _, _, _ = po.noise_lookup(blue_noise, 512 * 600 + 5, 0, 0)
    assert base_hi == (5, 511)


def test_k9_axis_tie_break_order(empty_region):
    mats, mine = empty_region
    s = 1.0 / math.sqrt(3.0)
    h = po.trace_ray(mats, mine, (0.0, 0.0, 0.0), (s, s, s))
    assert h.air == 1 and h.normal == 5          # x = y = z tie -> z (raytrace.comp:120-136), moving +z
    s2 = 1.0 / math.sqrt(2.0)
    h = po.trace_ray(mats, mine, (0.0, 0.0, 0.5), (s2, s2, 0.0))
    assert h.air == 1 and h.normal == 3          # x = y tie -> y, moving +y
    h = po.trace_ray(mats, mine, (0.0, 0.0, 0.5), (-s2, -s2, 0.0))
    assert h.air == 1 and h.normal == 2


def test_ray_starting_inside_a_solid_voxel_has_the_defined_outcome(floor_region):
    """Q12: step_size 0 => mod(x, 0) = NaN => one iteration, border fetch, hit with material 0 at a NaN position."""
    mats, mine = floor_region
    h = po.trace_ray(mats, mine, (3.2, 4.7, -20.5), (0.3, 0.2, 0.9))
    assert h.air == 0 and h.iterations == 1 and h.packed_material == 0 and h.border_fetches == 1
    assert all(math.isnan(v) for v in h.position[:])
    assert h.normal == 5


def test_toroidal_wrap_with_scrolled_region(native_built):
    """lr != 0 (TerrainUploadManager render offset): positions below -128 stay inside the sky bounds and the fetch
    wraps mod 256 (raytrace.comp:137) to the opposite side of the texture.  (Q7's exact-256.0 border texel needs
    |pos+128| < 2^-17, which pos+128 cannot produce next to -128 where floats are 2^-16 apart; the border is reached
    through NaN coordinates instead — see the inside-solid test.)"""
    ids = scenes.empty_ids()
    ids[:, :, 250:256] = 4                       # slab at texel x 250..255 = world x 122..127
    mats, mine = world.region_from_ids(ids)
    h = po.trace_ray(mats, mine, (-120.5, 0.3, 0.2), (-1.0, 0.0, 0.0), lr=(-64, 0, 0))
    assert h.air == 0 and h.packed_material == world.material_pack(4) and h.normal == 0    # NORMAL_x, moving -x
    assert -128.1 < h.position[0] < -127.9
    # with lr = 0 the same ray leaves through the x = -128 face
    h0 = po.trace_ray(mats, mine, (-120.5, 0.3, 0.2), (-1.0, 0.0, 0.0))
    assert h0.air == 1


def test_diffuse_direction_is_unit_and_in_the_hemisphere():
    for normal, axis, sign in ((0, 0, 1), (1, 0, -1), (2, 1, 1), (3, 1, -1), (4, 2, 1), (5, 2, -1)):
        for rg in ((0.1, 0.2), (0.9, 0.7), (0.5, 0.5), (0.0, 0.3)):
            d = po.diffuse_direction(normal, rg)
            assert abs(np.linalg.norm(d.astype(np.float64)) - 1.0) < 1e-6
            assert d[axis] * sign >= -1e-6


def test_depth_levels_extension(floor_region, blue_noise):
    """depth=0 traces primaries only; more depth adds light on hit pixels; counters follow 1 + 2*levels."""
    mats, mine = floor_region
    u = po.camera_uniforms((0.0, 0.0, 10.0), math.pi / 2, -0.6, 0.4, 3)
    p0, c0 = po.render(mats, mine, blue_noise, u, 24, 24, 1, 0)
    p1, c1 = po.render(mats, mine, blue_noise, u, 24, 24, 1, 1)
    p2, c2 = po.render(mats, mine, blue_noise, u, 24, 24, 1, 2)
    assert c0.rays == 24 * 24 and c0.rays_shadow == 0
    assert c1.rays_shadow == c0.hits and c1.rays_diffuse == c0.hits
    hit = p0["normal_r8"] != 16
    assert (p0["lighting_f32"][hit][:, :3] == 0).all()
    assert (p1["lighting_f32"][..., :3] >= p0["lighting_f32"][..., :3]).all()
    assert (p2["lighting_f32"][..., :3] >= p1["lighting_f32"][..., :3] - 1e-7).all()
    for name in ("depth_r16", "normal_r8", "albedo_rgba8", "fog_rgba8"):
        assert np.array_equal(p0[name], p2[name])


def test_spp_is_the_mean_of_single_sample_frames(floor_region, blue_noise):
    mats, mine = floor_region
    u = po.camera_uniforms((0.0, 0.0, 10.0), math.pi / 2, -0.6, 0.4, 10)
    acc = np.zeros((16, 16, 3), dtype=np.float32)
    for k in range(3):
        uk = po.camera_uniforms((0.0, 0.0, 10.0), math.pi / 2, -0.6, 0.4, 10 + k)
        pk, _ = po.render(mats, mine, blue_noise, uk, 16, 16, 1, 2)
        acc = acc + pk["lighting_f32"][..., :3] * np.float32(16.0)
    p3, _ = po.render(mats, mine, blue_noise, u, 16, 16, 3, 2)
    assert np.array_equal(p3["lighting_f32"][..., :3], (acc / np.float32(3.0)) / np.float32(16.0))
