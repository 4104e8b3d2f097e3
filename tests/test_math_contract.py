"""include/rt_math.h — the arithmetic contract — against double-precision references."""
import numpy as np

from oracle import pyoracle as po


def test_sin_cos_accuracy():
    x = np.linspace(-7.0, 7.0, 400001, dtype=np.float32)
    xd = x.astype(np.float64)
    assert np.abs(po.math("sin", x) - np.sin(xd)).max() < 2e-7
    assert np.abs(po.math("cos", x) - np.cos(xd)).max() < 2e-7


def test_sincos_large_and_nonfinite_arguments_are_defined():
    x = np.array([1e5, -1e5, 3e7, np.inf, -np.inf, np.nan], dtype=np.float32)
    assert np.array_equal(po.math("sin", x), np.zeros(6, dtype=np.float32))
    assert np.array_equal(po.math("cos", x), np.ones(6, dtype=np.float32))


def test_acos_accuracy_and_clamp():
    a = np.linspace(-1.0, 1.0, 400001, dtype=np.float32)
    assert np.abs(po.math("acos", a) - np.arccos(a.astype(np.float64))).max() < 5e-7
    out = po.math("acos", np.array([-2.0, 2.0], dtype=np.float32))
    assert abs(out[0] - np.pi) < 1e-6 and out[1] == 0.0


def test_pow_accuracy_on_the_sky_model_domain():
    b = np.linspace(1e-6, 1.0, 200001, dtype=np.float32)
    for e in (1.0, 2.5, 5.0, 10.0, 25.0, 40.0):      # raytrace.comp:278,280,283 use exponents in [1, 40]
        got = po.math("pow", b, np.full_like(b, e)).astype(np.float64)
        ref = b.astype(np.float64) ** e
        assert np.abs(got - ref).max() < 2e-7


def test_pow_of_nonpositive_base_is_zero():
    # quirk Q11: GLSL leaves pow(x<=0, y) undefined; the contract defines 0
    x = np.array([0.0, -0.0, -1e-7, -1.0, np.nan, 1e-45], dtype=np.float32)
    assert np.array_equal(po.math("pow", x, np.full_like(x, 5.0)), np.zeros(6, dtype=np.float32))


def test_mod_is_glsl_mod():
    rng = np.random.default_rng(1)
    x = rng.uniform(-600, 600, 100000).astype(np.float32)
    for y in (1.0, 2.0, 32.0, 256.0, 512.0):
        got = po.math("mod", x, np.full_like(x, y))
        ref = x - np.float32(y) * np.floor(x / np.float32(y))
        assert np.array_equal(got, ref.astype(np.float32))
        assert (got >= 0).all() and (got <= y).all()
    # Q7: a tiny negative coordinate wraps to exactly 256.0 (border texel)
    assert po.math("mod", np.array([-1e-8], dtype=np.float32), np.array([256.0], dtype=np.float32))[0] == 256.0
    # step_size 0: mod(x, 0) is NaN (quirk Q12: ray starting inside a solid voxel)
    assert np.isnan(po.math("mod", np.array([3.0], dtype=np.float32), np.array([0.0], dtype=np.float32))[0])


def test_sqrt_and_reciprocal_are_correctly_rounded():
    rng = np.random.default_rng(2)
    x = rng.uniform(1e-6, 1e6, 100000).astype(np.float32)
    assert np.array_equal(po.math("sqrt", x), np.sqrt(x.astype(np.float64)).astype(np.float32))
    assert np.array_equal(po.math("rcp", x), (1.0 / x.astype(np.float64)).astype(np.float32))


def test_normalize_unit_length():
    for v in ((3.0, 4.0, 0.0), (1e-3, -2e-3, 5e-4), (-30.0, 128.0, 100.0)):
        n = po.normalize(v).astype(np.float64)
        assert abs(np.linalg.norm(n) - 1.0) < 2e-7
        assert np.allclose(n, np.array(v) / np.linalg.norm(v), atol=2e-7)
    assert np.isnan(po.normalize((0.0, 0.0, 0.0))).all()      # normalize(0) = 0 * inf


def test_unorm8_times_255_is_the_byte():
    """texture(blue_noise, p).r * 255.0 (raytrace.comp:302-303) is exactly the stored byte — the kernels rely on it."""
    v = np.arange(256, dtype=np.float32)
    assert np.array_equal((v / np.float32(255.0)) * np.float32(255.0), v)


def test_store_conversions():
    assert po.unorm(0.5, 255.0) == 128 and po.unorm(1.5, 255.0) == 255 and po.unorm(-0.1, 255.0) == 0
    assert po.unorm(float("nan"), 65535.0) == 0 and po.unorm(1.0 / 16.0, 65535.0) == 4096
    assert po.f2u16(319.97) == 319 and po.f2u16(-3.0) == 0 and po.f2u16(1e9) == 65535 and po.f2u16(float("nan")) == 0


def test_noise_value_texel_is_level_independent():
    """Q5: the per-level noise offset (level-1) * 2/512 never reaches the next texel, and noise_offset is an exact
    integer in float, so floor(mod(offset + add, 512)) == int(offset) & 511 for every level — the path kernel fetches the
    noise_value texel once per path.  Exhaustive over every offset a 16384-wide frame can produce."""
    import ctypes as C
    f = po.lib().rt_oracle_noise_level_texel
    f.restype = C.c_int32
    f.argtypes = [C.c_float, C.c_int]
    max_offset = 255 + 8 * ((16383 // 128) * 16 + 15)
    for level in (1, 2, 3, 8, 16):
        for off in range(0, max_offset + 1):
            assert f(float(off), level) == (off & 511), (off, level)
