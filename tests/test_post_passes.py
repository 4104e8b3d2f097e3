"""bilateral_denoise.comp x6 and finalize.comp: oracle KATs on CPU, HIP parity on GPU."""
import numpy as np
import pytest

from raytrace_amd import abi, render
from oracle import pyoracle as po


def _planes(h, w, seed=0):
    rng = np.random.default_rng(seed)
    lighting = rng.integers(0, 20000, size=(h, w, 4), dtype=np.uint16)
    lighting[..., 3] = 4096
    depth = rng.integers(100, 4000, size=(h, w), dtype=np.uint16)
    normal = rng.integers(0, 6, size=(h, w), dtype=np.uint8)
    return lighting, depth, normal


def test_denoise_leaves_a_uniform_field_and_the_sky_alone():
    h, w = 40, 48
    lighting = np.zeros((h, w, 4), dtype=np.uint16)
    lighting[...] = (12345, 2222, 40000, 4096)
    depth = np.full((h, w), 800, dtype=np.uint16)
    normal = np.full((h, w), 4, dtype=np.uint8)
    normal[:10] = 16                      # sky rows: normal 16 (raytrace.comp:369) -> copied through (denoise :90-92)
    depth[:10] = 0xFFFF
    out = po.denoise(lighting, depth, normal, faithful=False)
    assert np.array_equal(out[:10], lighting[:10])                         # sky untouched, alpha stays 1/16
    assert np.array_equal(out[10:, :, :3], lighting[10:, :, :3])           # weighted mean of equal values
    assert (out[10:, :, 3] == 65535).all()                                 # vec4(sum / total_weight, 1.0)


def test_denoise_is_edge_stopping():
    """A lighting step that coincides with a normal change survives the six passes (a differing normal divides the tap
    weight by 11, bilateral_denoise.comp:29-30), while the same step inside one surface is blurred away."""
    h, w = 32, 64
    lighting = np.zeros((h, w, 4), dtype=np.uint16)
    lighting[:, :32, :3] = 1000
    lighting[:, 32:, :3] = 30000
    depth = np.full((h, w), 1000, dtype=np.uint16)
    normal = np.zeros((h, w), dtype=np.uint8)
    normal[:, 32:] = 2
    out = po.denoise(lighting, depth, normal, faithful=False).astype(np.float64)
    assert out[:, 31, 0].mean() < 6000 and out[:, 32, 0].mean() > 25000          # the edge is kept
    flat = po.denoise(lighting, depth, np.zeros((h, w), dtype=np.uint8), faithful=False).astype(np.float64)
    assert flat[:, 32, 0].mean() - flat[:, 31, 0].mean() < 2000                   # same step, one surface: smoothed
    assert abs(flat[:, :, 0].mean() - 15500) < 800                                # energy roughly preserved


def test_denoise_pong_quirk():
    """descriptor_sets.rs:38-39: on the pong set the shader's depth binding reads the normal image and its normal
    binding reads the depth image, so the `center_normal < 16` test compares the DEPTH value: the odd passes only touch
    pixels closer than 16/32 voxel.  With all depths >= 16 the faithful result equals three ping passes (1, 4, 8)
    interleaved with copies, and differs from the consistent binding."""
    lighting, depth, normal = _planes(24, 40, seed=3)
    faithful = po.denoise(lighting, depth, normal, faithful=True)
    fixed = po.denoise(lighting, depth, normal, faithful=False)
    assert not np.array_equal(faithful, fixed)
    near = depth.copy()
    near[5:9, 5:9] = 3                          # a few very near pixels are filtered even on the pong set
    f2 = po.denoise(lighting, near, normal, faithful=True)
    assert not np.array_equal(f2[5:9, 5:9], faithful[5:9, 5:9])


def test_finalize_known_values(blue_noise):
    h, w = 4, 6
    albedo = np.zeros((h, w, 4), dtype=np.uint8); albedo[...] = (255, 128, 0, 255)
    emission = np.zeros((h, w, 4), dtype=np.uint8); emission[..., 3] = 255
    fog = np.zeros((h, w, 4), dtype=np.uint8); fog[...] = (100, 150, 200, 255)
    lighting = np.zeros((h, w, 4), dtype=np.uint16); lighting[...] = (4096, 4096, 4096, 4096)      # light = 1.0
    depth = np.full((h, w), 0xFFFF, dtype=np.uint16)                                                # sky: no fog
    out = po.finalize(albedo, emission, fog, lighting, depth, blue_noise)
    n = blue_noise.reshape(512, 512, 4)
    for (x, y) in ((0, 0), (5, 3), (2, 1)):
        exp = []
        for k, a in enumerate((255, 128, 0)):
            v = np.float32(a) / np.float32(255) * (np.float32(4096) / np.float32(65535) * np.float32(16))
            v = float(v)
            if v < 0.3: t = v * v
            elif v < 1.13333: t = v * 0.6 - 0.09
            elif v < 2.5: t = 1.0 - 0.219512195116 * (v - 2.5) ** 2
            else: t = 1.0
            t += n[y, x, k] / 255.0 / 128.0
            exp.append(int(np.floor(min(max(t, 0.0), 1.0) * 255 + 0.5)))
        got = out[h - 1 - y, x]                      # Y flip (finalize.comp:60-62); memory order B,G,R,A
        assert abs(int(got[2]) - exp[0]) <= 1 and abs(int(got[1]) - exp[1]) <= 1 and abs(int(got[0]) - exp[2]) <= 1
        assert got[3] == 255
    # full fog at depth >= 32768 replaces the colour by fog * 2 (finalize.comp:44-49)
    depth[:] = 40000
    out2 = po.finalize(albedo, emission, fog, lighting, depth, blue_noise)
    v = 100 / 255 * 2
    t = v * 0.6 - 0.09
    x, y = 1, 1
    assert abs(int(out2[h - 1 - y, x, 2]) - int(np.floor((t + n[y, x, 0] / 255 / 128) * 255 + 0.5))) <= 1


def test_post_passes_of_a_terrain_frame_against_the_second_restatement(procedural_region, blue_noise):
    """The oracle's six denoise dispatches and its finalize against tests/shader_post.py (float64, written from the two shaders):
    a rendered terrain frame (sky, terrain, silhouettes) and random planes; both descriptor bindings.  The restatement keeps
    real numbers where the shader stores UNORM texels, so a stored texel may differ from it by the rounding (0.5) plus what fp32
    arithmetic and, for the denoise, five intermediate quantisations add: 3 of 65535 per channel, 0.52 of 255 for finalize —
    a wrong weight, offset, binding or curve segment is off by hundreds."""
    from tests import shader_post as sp
    from raytrace_amd import world as rt_world
    mats, mine = procedural_region
    W, H = 72, 48
    u = po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.05, 0.3, 5)
    planes, _ = po.render(mats, mine, blue_noise, u, W, H, 1, 2)
    assert (planes["normal_r8"] == 16).any() and (planes["normal_r8"] < 16).any()
    cases = [(planes["lighting_rgba16"], planes["depth_r16"], planes["normal_r8"]), _planes(40, 56, seed=3)]
    for lighting, depth, normal in cases:
        for faithful in (True, False):
            got = po.denoise(lighting, depth, normal, faithful).astype(np.int64)
            want = sp.denoise(lighting, depth, normal, faithful).astype(np.int64)
            assert np.abs(got - want).max() <= 3, (faithful, np.abs(got - want).max())
            assert (got[..., 3] == want[..., 3]).all()
        # one dispatch on its own, real-valued: the stored texel is the rounding of it
        for size, swapped in ((1, False), (2, True), (16, True)):
            d, n = depth.astype(np.int64), normal.astype(np.int64)
            real, _ = sp.denoise_pass(lighting, n if swapped else d, d if swapped else n, size)
            assert np.isfinite(real).all()
    # the faithful chain differs from the fixed one (the quirk is observable on this frame)
    assert (po.denoise(*cases[0], True) != po.denoise(*cases[0], False)).any()
    den = po.denoise(planes["lighting_rgba16"], planes["depth_r16"], planes["normal_r8"], True)
    out = po.finalize(planes["albedo_rgba8"], planes["emission_rgba8"], planes["fog_rgba8"], den, planes["depth_r16"], blue_noise)
    real = sp.finalize(planes["albedo_rgba8"], planes["emission_rgba8"], planes["fog_rgba8"], den, planes["depth_r16"], blue_noise)
    got_rgb = out[..., [2, 1, 0]].astype(np.float64)          # memory order B, G, R, A
    assert np.abs(got_rgb - real).max() <= 0.52, np.abs(got_rgb - real).max()
    assert (out[..., 3] == 255).all()
    assert len(np.unique(out[..., :3])) > 50                   # a real picture: sky gradient, lit and fogged terrain


@pytest.mark.gpu
@pytest.mark.parametrize("faithful", [True, False])
@pytest.mark.parametrize("W,H,spp", [(96, 64, 1), (200, 120, 4)])
def test_gpu_post_passes_match_oracle(procedural_region, blue_noise, W, H, spp, faithful):
    mats, mine = procedural_region
    u = po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.1, 0.3, 5)
    cfg = render.make_config(W, H, spp=spp, depth=2, flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        g = ctx.readback_all()
        ctx.denoise(faithful=faithful)
        ctx.sync()
        den = ctx.readback(abi.RT_BUF_LIGHTING_RGBA16)
        ctx.finalize()
        ctx.sync()
        fin = ctx.readback(abi.RT_BUF_FINAL_BGRA8)
    exp_den = po.denoise(g["lighting_rgba16"], g["depth_r16"], g["normal_r8"], faithful=faithful)
    assert np.array_equal(den, exp_den), "denoise differs at %d values" % int(np.count_nonzero(den != exp_den))
    exp_fin = po.finalize(g["albedo_rgba8"], g["emission_rgba8"], g["fog_rgba8"], exp_den, g["depth_r16"], blue_noise)
    assert np.array_equal(fin, exp_fin), "finalize differs at %d bytes" % int(np.count_nonzero(fin != exp_fin))
    # near pixels exist in this pose only if the camera hugs terrain; make sure the quirk path ran at least as a copy
    assert (den[..., 3][g["normal_r8"] == 16] == 4096).all()


@pytest.mark.gpu
def test_gpu_post_passes_match_oracle_at_1080p(procedural_region, blue_noise):
    """The post passes at the benchmark's frame size (VERDICT r2 #2c): six denoise dispatches (LDS-tiled sizes 1 and 2, direct
    4, 8, 8, 16) and finalize at 1920x1080 against the oracle, with the reference's pong binding and with the consistent one."""
    mats, mine = procedural_region
    W, H = 1920, 1080
    u = po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.1, 0.3, 5)
    cfg = render.make_config(W, H, spp=4, depth=2, flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        for faithful in (True, False):
            ctx.draw_frame(u)
            ctx.sync()
            g = ctx.readback_all()
            ctx.denoise(faithful=faithful)
            ctx.sync()
            den = ctx.readback(abi.RT_BUF_LIGHTING_RGBA16)
            ctx.finalize()
            ctx.sync()
            fin = ctx.readback(abi.RT_BUF_FINAL_BGRA8)
            exp_den = po.denoise(g["lighting_rgba16"], g["depth_r16"], g["normal_r8"], faithful=faithful)
            assert np.array_equal(den, exp_den), "denoise (faithful=%s) differs at %d values" % (faithful, int(np.count_nonzero(den != exp_den)))
            exp_fin = po.finalize(g["albedo_rgba8"], g["emission_rgba8"], g["fog_rgba8"], exp_den, g["depth_r16"], blue_noise)
            assert np.array_equal(fin, exp_fin), "finalize (faithful=%s) differs at %d bytes" % (faithful, int(np.count_nonzero(fin != exp_fin)))


@pytest.mark.gpu
def test_denoise_division_is_exact_on_its_whole_domain(native_built):
    """The denoise passes replace the IEEE division of bilateral_denoise.comp:31 by a reciprocal and one residual correction;
    rt_selftest compares the two over EVERY value the expression can take (37 weights x {k/64 + 1, k/64 + 11 : k < 65536}) on
    the device the suite runs on: not one quotient may differ."""
    with render.Context(render.make_config(64, 64)) as ctx:
        assert ctx.selftest(1) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("W,H", [(333, 77), (40, 30)])
def test_gpu_denoise_tiled_and_direct_dispatches_agree_with_the_oracle(procedural_region, blue_noise, W, H, monkeypatch):
    """Frame sizes that are not multiples of the 32x8 tile, smaller than the size-8 halo on one axis, both descriptor-set
    modes: the LDS-tiled dispatches (sizes 1, 2, 4, 8), the direct one (RT_DENOISE_UNTILED) and the oracle give the same bits."""
    mats, mine = procedural_region
    u = po.camera_uniforms((-30.0, -128.0, 100.0), np.pi / 2, -0.25, 0.3, 9)
    cfg = render.make_config(W, H, spp=1, depth=2, flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        for faithful in (True, False):
            outs = []
            for untiled in (False, True):
                if untiled:
                    monkeypatch.setenv("RT_DENOISE_UNTILED", "1")
                else:
                    monkeypatch.delenv("RT_DENOISE_UNTILED", raising=False)
                ctx.draw_frame(u)
                ctx.sync()
                g = ctx.readback_all()
                ctx.denoise(faithful=faithful)
                ctx.sync()
                outs.append(ctx.readback(abi.RT_BUF_LIGHTING_RGBA16))
            exp = po.denoise(g["lighting_rgba16"], g["depth_r16"], g["normal_r8"], faithful=faithful)
            assert np.array_equal(outs[0], exp) and np.array_equal(outs[1], exp), faithful
