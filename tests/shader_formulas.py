"""A second, independent restatement of the closed-form parts of shaders/glsl/raytrace.comp — float64 numpy, written from the
GLSL text and from nothing else in this repository (not from oracle/rt_oracle.cpp, which the same shader lines were restated
into first).  Test infrastructure: tests/test_oracle_kat.py holds the oracle's functions and whole frames of analytic scenes
against it, so a misreading of a formula would have to be made twice, in two languages, to go unnoticed.  fp64 against the
shader's fp32: comparisons carry a tolerance (stated where they are made); bit-exactness is the GPU-vs-oracle tests' business.

Each function cites the shader lines it restates."""
import math

import numpy as np

NOISE_SIZE = 512


def sun_vector(sun_angle):
    """raytrace.comp:317 — normalize(vec3(cos(a) * 0.5 + (a - 0.5) * 0.5, sin(a), cos(a)))."""
    a = float(sun_angle)
    v = np.array([math.cos(a) * 0.5 + (a - 0.5) * 0.5, math.sin(a), math.cos(a)])
    return v / np.linalg.norm(v)


def mix(x, y, a):
    """GLSL mix: x * (1 - a) + y * a."""
    return np.asarray(x) * (1.0 - a) + np.asarray(y) * a


def sun_color(sun_direction):
    """raytrace.comp:259-269."""
    horizon = math.hypot(sun_direction[0], sun_direction[1])
    sun_amount = min(1.0 - horizon, 0.02) * 50.0
    main_color = np.array([0.9647, 0.7843, 0.8824]) * 2.0
    sunset_color = np.array([0.7412, 0.2157, 0.1686]) * 2.0
    if sun_direction[2] >= 0.0:
        return mix(sunset_color, main_color, sun_amount)
    return mix(sunset_color, np.zeros(3), sun_amount * 2)


def sample_sky(direction, sun_direction, sunlight, include_sun):
    """raytrace.comp:271-288 (the `direction.z < 0` branch is empty in the shader)."""
    d = np.asarray(direction, dtype=np.float64)
    bright_color = np.array([0.5294, 0.8275, 0.9647])
    dark_color = np.array([0.0863, 0.1294, 0.2196])
    sunlight_amount = min(max((sunlight[0] + sunlight[1] + sunlight[2]) * 0.2 - 0.02, 0.0), 1.0)
    horizon = math.hypot(d[0], d[1]) ** float(mix(40.0, 10.0, sunlight_amount))
    sun_amount = 1.0 - 0.5 * np.linalg.norm(np.asarray(sun_direction) - d)
    sun_halo_amount = sun_amount ** float(mix(5.0, 1.0, sunlight_amount))
    bright_amount = min(horizon + sun_halo_amount * 0.5, 1.0)
    color = mix(dark_color, bright_color, bright_amount * max(sunlight_amount, 0.1))
    color = color + np.asarray(sunlight) * sun_amount ** 5.0 * 0.5
    if sun_amount > 0.98 and include_sun:
        color = color + np.asarray(sunlight)
    return color


def diffuse_direction(normal, noise_rg):
    """raytrace.comp:189-212 — a point on the unit sphere from (noise.r, noise.g) plus the face normal, normalised.
    Face ids: NORMAL_x = 0 (+x), 1 (-x), NORMAL_y = 2, 3, NORMAL_z = 4, 5."""
    r, g = float(noise_rg[0]), float(noise_rg[1])
    theta1 = math.pi * 2.0 * r
    theta2 = math.acos(1.0 - 2.0 * g)
    d = np.array([math.sin(theta1) * math.sin(theta2), math.cos(theta1) * math.sin(theta2), math.cos(theta2)])
    d[normal // 2] += 1.0 if normal % 2 == 0 else -1.0
    return d / np.linalg.norm(d)


def sun_ray_direction(sun_direction, noise_rg):
    """raytrace.comp:185-187 — normalize(direction + vec3(noise_value.rg, 0) * 0.05)."""
    v = np.asarray(sun_direction) + np.array([noise_rg[0], noise_rg[1], 0.0]) * 0.05
    return v / np.linalg.norm(v)


def workgroup_of_pixel(p):
    """Inverse of raytrace.comp:291-294 on one axis: pixel = (wg - wg % 16) * 8 + wg % 16 + local * 16, local in [0, 8).
    (wg - wg % 16) * 8 is 128 * (wg // 16); the rest, wg % 16 + 16 * local, is below 128."""
    return (p // 128) * 16 + p % 16


def noise_value(noise_rgba, seed, px, py):
    """raytrace.comp:298-304 and :324 — the texel of `noise_value` for a pixel: base = 255 * noise[seed % 512, seed / 512].rg
    (unnormalised NEAREST, CLAMP_TO_EDGE), offset by 8 * the pixel's workgroup id, modulo 512.  `noise_rgba` is [512, 512, 4],
    row = y.  Returns the RGBA8 texel / 255 (UNORM)."""
    n = np.asarray(noise_rgba).reshape(NOISE_SIZE, NOISE_SIZE, 4)
    bx, by = seed % NOISE_SIZE, min(seed // NOISE_SIZE, NOISE_SIZE - 1)
    base = n[by, bx, :2].astype(np.int64)          # (v / 255) * 255 is exactly v
    ox = (int(base[0]) + workgroup_of_pixel(px) * 8) % NOISE_SIZE
    oy = (int(base[1]) + workgroup_of_pixel(py) * 8) % NOISE_SIZE
    return n[oy, ox].astype(np.float64) / 255.0


def primary_direction(forward, up, right, px, py, width, height):
    """raytrace.comp:296-297, 306-310."""
    sx = px / width * 2.0 - 1.0
    sy = py / height * 2.0 - 1.0
    v = np.asarray(forward, dtype=np.float64) + sx * np.asarray(right, dtype=np.float64) + sy * np.asarray(up, dtype=np.float64)
    return v / np.linalg.norm(v)
