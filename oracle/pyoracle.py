"""ctypes binding of the CPU oracle (oracle/rt_oracle.cpp).  TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg — never by
raytrace_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from raytrace_amd.abi import RtCounters, RtUniforms

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "librt_oracle.so")
    srcs = [os.path.join(_HERE, "rt_oracle.cpp"), os.path.join(_HERE, "..", "include", "rt_math.h"),
            os.path.join(_HERE, "..", "include", "rt_abi.h")]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "librt_oracle.so"], stdout=subprocess.DEVNULL)
    return so


class RtOracleOut(C.Structure):
    _fields_ = [("lighting_rgba16", C.c_void_p), ("depth_r16", C.c_void_p), ("normal_r8", C.c_void_p),
                ("albedo_rgba8", C.c_void_p), ("emission_rgba8", C.c_void_p), ("fog_rgba8", C.c_void_p),
                ("lighting_f32", C.c_void_p), ("fog_f32", C.c_void_p), ("depth_f32", C.c_void_p)]


class RtOracleHit(C.Structure):
    _fields_ = [("albedo", C.c_float * 3), ("emission", C.c_float * 3), ("air", C.c_int32),
                ("distance", C.c_float), ("normal", C.c_uint32), ("position", C.c_float * 3),
                ("packed_material", C.c_uint32), ("iterations", C.c_uint32), ("border_fetches", C.c_uint32),
                ("limit_exit", C.c_uint32)]


def lib():
    global _LIB
    if _LIB is None:
        # RT_ORACLE_LIB: load another build of the same source (the sanitizer build of `make -C oracle asan`)
        _LIB = C.CDLL(os.environ.get("RT_ORACLE_LIB") or build())
        _LIB.rt_oracle_render.restype = C.c_int
        _LIB.rt_oracle_pixel_of.restype = C.c_uint32
        _LIB.rt_oracle_pixel_of.argtypes = [C.c_uint32, C.c_uint32]
        _LIB.rt_oracle_workgroup_of.restype = C.c_uint32
        _LIB.rt_oracle_workgroup_of.argtypes = [C.c_uint32]
        _LIB.rt_oracle_pack_material.restype = C.c_uint32
        _LIB.rt_oracle_pack_material.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        _LIB.rt_oracle_sun.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        _LIB.rt_oracle_sample_sky.argtypes = [C.c_void_p, C.c_float, C.c_int, C.c_void_p]
        _LIB.rt_oracle_camera_uniforms.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_uint32,
                                                   C.c_void_p, C.c_void_p]
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


PLANES = {"lighting_rgba16": (np.uint16, 4), "depth_r16": (np.uint16, 1), "normal_r8": (np.uint8, 1),
          "albedo_rgba8": (np.uint8, 4), "emission_rgba8": (np.uint8, 4), "fog_rgba8": (np.uint8, 4),
          "lighting_f32": (np.float32, 4), "fog_f32": (np.float32, 4), "depth_f32": (np.float32, 1)}


def render(materials, minefield, noise, uniforms, width, height, spp=1, depth=2, rows=None, threads=0, region=256):
    """Render with the oracle. Returns (dict of planes [H,W,(C)], RtCounters)."""
    materials = np.ascontiguousarray(materials, dtype=np.uint32).reshape(-1)
    minefield = np.ascontiguousarray(minefield, dtype=np.uint8).reshape(-1)
    noise = np.ascontiguousarray(noise, dtype=np.uint8).reshape(-1)
    assert materials.size == region ** 3 and minefield.size == region ** 3 and noise.size == 512 * 512 * 4
    y0, y1 = rows if rows is not None else (0, height)
    planes = {}
    out = RtOracleOut()
    for name, (dt, ch) in PLANES.items():
        shape = (height, width, ch) if ch > 1 else (height, width)
        planes[name] = np.zeros(shape, dtype=dt)
        setattr(out, name, planes[name].ctypes.data)
    cn = RtCounters()
    rc = lib().rt_oracle_render_region(_p(materials), _p(minefield), _p(noise), C.byref(uniforms), int(region), int(width),
                                       int(height), int(spp), int(depth), int(y0), int(y1), int(threads), C.byref(out), C.byref(cn))
    if rc != 0:
        raise RuntimeError("rt_oracle_render failed: %d" % rc)
    return planes, cn


def camera_uniforms(origin, heading, pitch, sun_angle=0.0, seed=1, lr=(0, 0, 0)):
    u = RtUniforms()
    o = (C.c_float * 3)(*origin)
    l = (C.c_int32 * 3)(*lr)
    lib().rt_oracle_camera_uniforms(o, float(heading), float(pitch), float(sun_angle), int(seed), l, C.byref(u))
    return u


def trace_ray(materials, minefield, origin, direction, lr=(0, 0, 0)):
    materials = np.ascontiguousarray(materials, dtype=np.uint32).reshape(-1)
    minefield = np.ascontiguousarray(minefield, dtype=np.uint8).reshape(-1)
    h = RtOracleHit()
    lib().rt_oracle_trace_ray(_p(materials), _p(minefield), (C.c_int32 * 3)(*lr), (C.c_float * 3)(*origin),
                              (C.c_float * 3)(*direction), C.byref(h))
    return h


def sun(sun_angle):
    a = (C.c_float * 3)()
    c = (C.c_float * 3)()
    lib().rt_oracle_sun(float(sun_angle), a, c)
    return np.array(a[:], dtype=np.float32), np.array(c[:], dtype=np.float32)


def sample_sky(direction, sun_angle, include_sun):
    o = (C.c_float * 3)()
    lib().rt_oracle_sample_sky((C.c_float * 3)(*direction), float(sun_angle), int(bool(include_sun)), o)
    return np.array(o[:], dtype=np.float32)


def diffuse_direction(normal, noise_rg):
    o = (C.c_float * 3)()
    lib().rt_oracle_diffuse_direction(C.c_uint32(int(normal)), (C.c_float * 2)(*noise_rg), o)
    return np.array(o[:], dtype=np.float32)


def noise_lookup(noise, seed, px, py):
    noise = np.ascontiguousarray(noise, dtype=np.uint8).reshape(-1)
    base = (C.c_int32 * 2)()
    off = (C.c_float * 2)()
    vt = (C.c_int32 * 2)()
    val = (C.c_float * 4)()
    lib().rt_oracle_noise_lookup(_p(noise), C.c_uint32(seed), C.c_uint32(px), C.c_uint32(py), base, off, vt, val)
    return tuple(base[:]), tuple(off[:]), tuple(vt[:]), tuple(val[:])


def pack_chunk(solid, packed):
    """UnpackedChunkData::pack_into restated (src/world/chunk.rs:125-184) on one 64^3 chunk."""
    solid = np.ascontiguousarray(solid, dtype=np.uint8).reshape(-1)
    packed = np.ascontiguousarray(packed, dtype=np.uint32).reshape(-1)
    assert solid.size == 64 ** 3 and packed.size == 64 ** 3
    mats = np.zeros(64 ** 3, dtype=np.uint32)
    mine = np.zeros(64 ** 3, dtype=np.uint8)
    lib().rt_oracle_pack_chunk(_p(solid), _p(packed), _p(mats), _p(mine))
    return mats, mine


_MATH_FN = {"sin": 0, "cos": 1, "acos": 2, "pow": 3, "mod": 4, "exp2": 5, "log2": 6, "sqrt": 7, "rcp": 8}


def math(fn, x, y=None):
    """Evaluate one function of the arithmetic contract (include/rt_math.h) elementwise on float32 arrays."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.ascontiguousarray(y if y is not None else np.zeros_like(x), dtype=np.float32)
    out = np.empty_like(x)
    lib().rt_oracle_math(C.c_int(_MATH_FN[fn]), _p(x), _p(y), _p(out), C.c_size_t(x.size))
    return out


def normalize(v):
    o = (C.c_float * 3)()
    lib().rt_oracle_normalize((C.c_float * 3)(*v), o)
    return np.array(o[:], dtype=np.float32)


def unorm(x, maxv):
    f = lib().rt_oracle_unorm
    f.restype = C.c_uint32
    f.argtypes = [C.c_float, C.c_float]
    return int(f(float(x), float(maxv)))


def f2u16(x):
    f = lib().rt_oracle_f2u16
    f.restype = C.c_uint32
    f.argtypes = [C.c_float]
    return int(f(float(x)))


def denoise(lighting_rgba16, depth_r16, normal_r8, faithful=True):
    """The six bilateral_denoise.comp dispatches (pipeline.rs:98-115) on [H,W,4] uint16 lighting; returns the result."""
    lighting = np.ascontiguousarray(lighting_rgba16, dtype=np.uint16).copy()
    depth = np.ascontiguousarray(depth_r16, dtype=np.uint16)
    normal = np.ascontiguousarray(normal_r8, dtype=np.uint8)
    h, w = depth.shape
    rc = lib().rt_oracle_denoise(_p(lighting), _p(depth), _p(normal), C.c_int(w), C.c_int(h), C.c_int(1 if faithful else 0))
    assert rc == 0
    return lighting


def finalize(albedo_rgba8, emission_rgba8, fog_rgba8, lighting_rgba16, depth_r16, noise):
    """finalize.comp -> [H,W,4] uint8 in B,G,R,A byte order, rows top-down."""
    a = np.ascontiguousarray(albedo_rgba8, dtype=np.uint8)
    e = np.ascontiguousarray(emission_rgba8, dtype=np.uint8)
    f = np.ascontiguousarray(fog_rgba8, dtype=np.uint8)
    l = np.ascontiguousarray(lighting_rgba16, dtype=np.uint16)
    d = np.ascontiguousarray(depth_r16, dtype=np.uint16)
    n = np.ascontiguousarray(noise, dtype=np.uint8).reshape(-1)
    h, w = d.shape
    out = np.zeros((h, w, 4), dtype=np.uint8)
    rc = lib().rt_oracle_finalize(_p(a), _p(e), _p(f), _p(l), _p(d), _p(n), C.c_int(w), C.c_int(h), _p(out))
    assert rc == 0
    return out
