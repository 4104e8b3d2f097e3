"""Known-answer tests of the CPU oracle, derived by hand from shaders/glsl/raytrace.comp, and the
oracle against a second restatement of the shader written from the GLSL alone (K10-K14: tests/shader_formulas.py,
tests/shader_trace.py).  The reference holds no vectors for this path, so these are what pins the oracle."""
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


# ---- K10 / K11: the closed-form parts of the shader against a SECOND restatement (tests/shader_formulas.py: float64 numpy written
# from the GLSL alone).  The reference has no vectors for this path; two independent readings agreeing is the next
# best pin.  fp64 against the oracle's fp32: absolute tolerances, stated per comparison.
def test_k10_sun_sky_and_diffuse_formulas_agree_with_the_second_restatement():
    from tests import shader_formulas as sf
    rng = np.random.default_rng(10)
    worst_sky = 0.0
    for a in (-2.0, -0.3, 0.0, 0.2, 0.49, 1.0, 1.5, 2.5, 3.3):
        sv, sc = po.sun(a)
        v64 = sf.sun_vector(a)
        c64 = sf.sun_color(v64)
        assert np.allclose(sv, v64, rtol=0, atol=3e-7)
        assert np.allclose(sc, c64, rtol=0, atol=2e-5)        # (1 - |xy|) * 50 amplifies the vector's rounding
        for _ in range(60):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            if rng.random() < 0.25:                            # some directions near the sun: halo and disc terms
                d = v64 + rng.normal(size=3) * 0.08
                d /= np.linalg.norm(d)
            sun_amount = 1.0 - 0.5 * np.linalg.norm(v64 - d)
            if abs(sun_amount - 0.98) < 1e-4:                  # the disc's threshold: fp32 and fp64 may fall on either side
                continue
            d32 = d.astype(np.float32)
            for inc in (True, False):
                got = po.sample_sky(tuple(d32), a, inc)
                want = sf.sample_sky(d32.astype(np.float64), v64, c64, inc)
                worst_sky = max(worst_sky, float(np.abs(got - want).max()))
    assert worst_sky < 1e-4, worst_sky      # pow(x, 40) carries 40 x the fp32 rounding of log2(x); values are O(1)
    for normal in range(6):
        for r in (0, 1, 37, 128, 200, 254, 255):
            for g in (0, 1, 64, 127, 128, 254, 255):
                axis, sign = normal // 2, (1.0 if normal % 2 == 0 else -1.0)
                rg = (r / 255.0, g / 255.0)
                raw = np.array([math.sin(2 * math.pi * rg[0]) * math.sin(math.acos(1 - 2 * rg[1])),
                                math.cos(2 * math.pi * rg[0]) * math.sin(math.acos(1 - 2 * rg[1])), math.cos(math.acos(1 - 2 * rg[1]))])
                raw[axis] += sign
                if np.linalg.norm(raw) < 1e-3:                 # sphere point = -normal: 0 / 0 in the shader (hard part 1), not a formula check
                    continue
                p = sf.diffuse_direction(normal, rg)
                got = po.diffuse_direction(normal, rg)
                # the unnormalised vector's fp32 rounding (a few 1e-7 on components of size <= 2) is divided by its length
                assert np.allclose(got, p, rtol=0, atol=max(2e-6, 6e-7 / np.linalg.norm(raw))), (normal, r, g, got, p)


def test_k11_lit_floor_frame_against_the_second_restatement(floor_region, blue_noise):
    """Solid floor below z = 0, nothing else, sun high: every floor pixel's shadow ray and diffuse ray leave the region, so
    light = sunlight + sample_sky(diffuse_direction) (raytrace.comp:324-332) with the pixel's own noise texel (:298-304, :324);
    sky pixels are sample_sky(primary direction).  All of it from tests/shader_formulas.py."""
    from tests import shader_formulas as sf
    mats, mine = floor_region
    sun_angle, seed, W, H = 1.0, 7, 32, 32
    u = po.camera_uniforms((3.0, -20.0, 12.0), math.pi / 2, -0.15, sun_angle, seed)
    planes, cn = po.render(mats, mine, blue_noise, u, W, H, 1, 2)
    sv = sf.sun_vector(sun_angle)
    sc = sf.sun_color(sv)
    assert sv[2] > 0.3
    fwd, up, right = (np.array(x[:], dtype=np.float64) for x in (u.forward, u.up, u.right))
    floor_px = sky_px = 0
    worst = 0.0
    for py in range(H):
        for px in range(W):
            d = sf.primary_direction(fwd, up, right, px, py, W, H)
            nrm = int(planes["normal_r8"][py, px])
            if nrm == 16:
                want = sf.sample_sky(d, sv, sc, True)
                sky_px += 1
            else:
                assert nrm == 4 and d[2] < 0                   # the floor's top face
                n = sf.noise_value(blue_noise, seed, px, py)
                if n[1] == 1.0:                                # sphere point = -normal
                    continue
                assert sf.sun_ray_direction(sv, n[:2])[2] > 0  # towards the open sky
                d1 = sf.diffuse_direction(4, n[:2])
                assert d1[2] > 0
                want = sc + sf.sample_sky(d1, sv, sc, True)
                floor_px += 1
            got = planes["lighting_f32"][py, px, :3].astype(np.float64) * 16.0
            worst = max(worst, float(np.abs(got - want).max()))
    assert floor_px > 300 and sky_px > 100, (floor_px, sky_px)
    assert worst < 2e-4, worst                                 # O(1..4) values through fp32 sin/cos/acos/pow
    assert cn.rays == W * H + 2 * (W * H - sky_px) and cn.hits == W * H - sky_px


def test_k12_trace_ray_bit_for_bit_against_the_second_restatement(procedural_region, floor_region):
    """tests/shader_trace.py restates raytrace.comp:78-183 a second time (numpy float32 scalars, exact fused multiply-add);
    the oracle's trace_ray must agree with it in every field, to the bit, ray by ray: terrain, a floor, a scrolled window."""
    from tests import shader_trace as st
    rng = np.random.default_rng(12)

    def same(h, r, what):
        assert not h.limit_exit and not r["limit"], what
        assert bool(h.air) == r["air"] and h.normal == r["normal"] and h.iterations == r["iterations"], (what, h.air, r["air"], h.normal, r["normal"], h.iterations, r["iterations"])
        got = np.array(list(h.position[:]) + [h.distance], dtype=np.float32).view(np.uint32)
        want = np.array(r["position"] + [r["distance"]], dtype=np.float32).view(np.uint32)
        assert (got == want).all(), (what, h.position[:], r["position"], h.distance, r["distance"])
        if not r["air"]:
            assert h.packed_material == r["packed_material"], what
            assert (np.array(h.albedo[:], dtype=np.float32).view(np.uint32) == np.array(r["albedo"], dtype=np.float32).view(np.uint32)).all(), what

    def rand_dir():
        d = rng.normal(size=3)
        return tuple((d / np.linalg.norm(d)).astype(np.float32))

    mats, mine = procedural_region
    hits = skies = 0
    surface = []
    for i in range(140):
        o = (float(rng.uniform(-110, 110)), float(rng.uniform(-110, 110)), float(rng.uniform(40, 126)))
        d = rand_dir() if i % 3 else (float(rng.normal() * 0.3), float(rng.normal() * 0.3), -1.0)
        h = po.trace_ray(mats, mine, o, d)
        same(h, st.trace_ray(mats, mine, o, d), ("terrain", o, d))
        if h.air:
            skies += 1
        else:
            hits += 1
            surface.append((tuple(h.position[:]), h.normal))
    assert hits > 40 and skies > 30, (hits, skies)
    # second-level rays: from a hit point (0.001 off its face) into the hemisphere of the face, as the diffuse and shadow rays start
    for p, n in surface[:60]:
        d = np.array(rand_dir(), dtype=np.float64)
        d[n // 2] = abs(d[n // 2]) * (1.0 if n % 2 == 0 else -1.0)
        d = tuple(d.astype(np.float32))
        same(po.trace_ray(mats, mine, p, d), st.trace_ray(mats, mine, p, d), ("bounce", p, d))
    # grazing and axis-aligned directions: zero components (1 / 0 = inf: the axis never wins), exact ties
    for d in ((1.0, 0.0, 0.0), (0.0, -1.0, 0.0), (1.0, 1.0, 0.0), (1.0, 1.0, -1.0), (0.0, 0.70710678, -0.70710678), (1e-8, 1.0, -1e-3)):
        for o in ((0.5, 0.5, 100.0), (-30.0, -127.99, 100.0), (16.0, 32.0, 64.0)):
            same(po.trace_ray(mats, mine, o, d), st.trace_ray(mats, mine, o, d), ("special", o, d))
    # exact ties between axes (:120-136: x < y, then x < z or y < z — a tie goes to the later axis): equal |direction| components
    # from origins at the same phase of the 16- and 32-cells
    ties = 0
    for d, o in (((1.0, 0.5, -1.0), (-31.75, -30.0, 95.75)), ((1.0, 1.0, -1.0), (-31.75, -31.75, 95.75)), ((0.5, 1.0, -1.0), (-30.0, -31.75, 95.75)),
                 ((1.0, 1.0, -0.25), (-31.75, -31.75, 100.0)), ((-1.0, 0.5, 1.0), (31.75, 3.0, 64.25))):
        r = st.trace_ray(mats, mine, o, d)
        same(po.trace_ray(mats, mine, o, d), r, ("tie", o, d))
        dn = st.normalize3([np.float32(x) for x in d])
        q = [np.float32(np.float32(0.0001) + st.mod32(np.float32((np.float32(o[a]) + np.float32(128)) * (np.float32(-1) if dn[a] > 0 else np.float32(1))), np.float32(16)))
             * np.float32(np.float32(1) / abs(dn[a])) for a in range(3)]
        ties += int(len({float(x) for x in q}) < 3)
    assert ties >= 4, ties       # the first step of these rays really is a tie (step size 16 up there)
    fm, fmine = floor_region
    for _ in range(30):
        o = (float(rng.uniform(-100, 100)), float(rng.uniform(-100, 100)), float(rng.uniform(0.5, 100)))
        d = rand_dir()
        same(po.trace_ray(fm, fmine, o, d), st.trace_ray(fm, fmine, o, d), ("floor", o, d))
    # a scrolled window (terrain_upload.rs:84-275): positions are world coordinates around lr, the texture wraps
    lr = (48, -32, 16)
    for _ in range(40):
        o = (float(lr[0] + rng.uniform(-100, 100)), float(lr[1] + rng.uniform(-100, 100)), float(lr[2] + rng.uniform(30, 120)))
        d = rand_dir()
        same(po.trace_ray(mats, mine, o, d, lr=lr), st.trace_ray(mats, mine, o, d, lr=lr), ("scrolled", o, d))


def test_k13_primary_planes_of_a_terrain_frame_against_the_second_restatement(procedural_region, blue_noise):
    """Pixel -> primary ray (raytrace.comp:296-315) -> trace_ray -> the integer G-buffer planes (:357-375), all from the second
    restatement (tests/shader_trace.py, float32, exact), for every pixel of a small frame of the procedural terrain at the
    reference's default pose: depth, normal and albedo planes must be equal, not close."""
    from tests import shader_trace as st
    f32 = np.float32
    mats, mine = procedural_region
    W = H = 40
    for origin, heading, pitch in (((-30.0, -128.0, 100.0), math.pi / 2, 0.0), ((20.0, -150.0, 90.0), math.pi / 2 + 0.3, -0.25)):
        u = po.camera_uniforms(origin, heading, pitch, 0.0, 1)
        planes, _ = po.render(mats, mine, blue_noise, u, W, H, 1, 0)
        fwd, up, right = ([f32(v) for v in x[:]] for x in (u.forward, u.up, u.right))
        org = [f32(v) for v in u.origin[:3]]
        hit_px = 0
        for py in range(H):
            for px in range(W):
                sx = f32(f32(f32(f32(px) / f32(W)) * f32(2)) - f32(1))                      # :296-297
                sy = f32(f32(f32(f32(py) / f32(H)) * f32(2)) - f32(1))
                d = st.normalize3([f32(f32(fwd[a] + f32(right[a] * sx)) + f32(up[a] * sy)) for a in range(3)])   # :306-310
                start = list(org)
                if -start[1] > 128:                                                            # :311-314
                    space = f32(f32(-start[1]) - f32(128))
                    k = f32(f32(space / d[1]) + f32(0.0001))
                    start = [f32(start[a] + f32(d[a] * k)) for a in range(3)]
                r = st.trace_ray(mats, mine, start, d)
                assert not r["limit"]
                if r["air"]:
                    want = (0xFFFF, 16, (255, 255, 255, 255))
                else:
                    dist = st.length3([f32(org[a] - r["position"][a]) for a in range(3)])      # :358-360 (the uniform's origin)
                    d32 = f32(dist * f32(32))
                    assert d32 < 65535
                    alb = tuple(int(np.rint(f32(c * f32(255)))) for c in r["albedo"]) + (255,)   # UNORM8: round to nearest
                    want = (int(d32), r["normal"], alb)
                    hit_px += 1
                got = (int(planes["depth_r16"][py, px]), int(planes["normal_r8"][py, px]), tuple(int(v) for v in planes["albedo_rgba8"][py, px]))
                assert got == want, (origin, px, py, got, want)
        assert hit_px > W * H // 4, hit_px


def test_k14_two_level_light_of_a_terrain_frame_against_the_second_restatement(procedural_region, blue_noise):
    """raytrace.comp:319-350 on terrain, depth 2: which rays are cast from where, what each contributes, and how the second level
    is folded in (light2 *= albedo2; light += light2) — composed here a second time from pieces that are pinned on their own:
    tests/shader_trace.py for every ray (K12), the pixel's noise texel and the sky / sun colour from tests/shader_formulas.py
    (K10, K11), ray directions in float32 (the oracle's diffuse_direction, K10; the sun ray restated below).  Tolerance: the sky
    terms are fp64 here against the oracle's fp32 (2e-4 on O(1) values); the geometry is exact."""
    from tests import shader_formulas as sf, shader_trace as st
    f32 = np.float32
    mats, mine = procedural_region
    W = H = 24
    sun_angle, seed = 0.35, 3
    u = po.camera_uniforms((-30.0, -128.0, 100.0), math.pi / 2, -0.1, sun_angle, seed)
    planes, _ = po.render(mats, mine, blue_noise, u, W, H, 1, 2)
    sun32, col32 = po.sun(sun_angle)
    sv, sc = sun32.astype(np.float64), col32.astype(np.float64)
    fwd, up, right = ([f32(v) for v in x[:]] for x in (u.forward, u.up, u.right))
    org = [f32(v) for v in u.origin[:3]]

    def sun_ray(n):           # :185-187 in float32: direction + vec3(noise.rg, 0) * 0.05, normalised
        return st.normalize3([f32(sun32[0] + f32(f32(n[0]) * f32(0.05))), f32(sun32[1] + f32(f32(n[1]) * f32(0.05))), f32(sun32[2] + f32(f32(0.0) * f32(0.05)))])

    def level(pos, normal, n, last):
        light = np.zeros(3)
        if st.trace_ray(mats, mine, pos, sun_ray(n))["air"]:                    # :325-328 / :337-340
            light += sc
        d = [f32(v) for v in po.diffuse_direction(normal, (f32(n[0]), f32(n[1])))]   # :329 / :341
        dif = st.trace_ray(mats, mine, pos, d)                                   # :330 / :342
        if dif["air"]:
            light += sf.sample_sky(np.array(d, dtype=np.float64), sv, sc, True)  # :331-332 / :343-345
        elif not last:
            light2 = level(dif["position"], dif["normal"], n, True)             # :336: + 2/512 of a texel — the same texel (Q5)
            light2 = light2 * np.array(dif["albedo"], dtype=np.float64)          # :346
            light = light + light2                                               # :347-348 (emission is vec3(0))
        return light

    worst, lit, second = 0.0, 0, 0
    for py in range(H):
        for px in range(W):
            if planes["normal_r8"][py, px] == 16:
                continue
            sx = f32(f32(f32(f32(px) / f32(W)) * f32(2)) - f32(1))
            sy = f32(f32(f32(f32(py) / f32(H)) * f32(2)) - f32(1))
            d = st.normalize3([f32(f32(fwd[a] + f32(right[a] * sx)) + f32(up[a] * sy)) for a in range(3)])
            primary = st.trace_ray(mats, mine, org, d)
            assert not primary["air"]
            n8 = sf.noise_value(blue_noise, seed, px, py)
            n = (f32(f32(int(round(n8[0] * 255))) / f32(255)), f32(f32(int(round(n8[1] * 255))) / f32(255)))   # RGBA8 UNORM -> float32
            if int(round(n8[1] * 255)) == 255:
                continue
            want = level(primary["position"], primary["normal"], n, False)
            got = planes["lighting_f32"][py, px, :3].astype(np.float64) * 16.0
            worst = max(worst, float(np.abs(got - want).max()))
            lit += 1
            second += int(not st.trace_ray(mats, mine, primary["position"], [f32(v) for v in po.diffuse_direction(primary["normal"], n)])["air"])
    assert lit > 150 and second > 20, (lit, second)
    assert worst < 2e-4, worst
