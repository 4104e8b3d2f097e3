"""Python binding of the ray-trace path.

Two layers, both thin ctypes wrappers (all logic lives in the native libraries):

* `Context`  — the C ABI of include/rt_abi.h, one object per GPU context (rt_create ... rt_destroy).
* `Camera`, `Game`, `Pipeline`, `create_instance` — the C++ host mirror of the reference's public `render` API
  (src/render/mod.rs:20-43, src/render/pipeline/pipeline.rs:134-255, src/game/mod.rs:37-58), so that tests read
  like a user of the reference: `game = Game(); pipeline = create_instance(cfg, game); pipeline.draw_frame(game)`.
"""
import ctypes as C
import math

import numpy as np

from . import _lib
from .abi import (BUFFER_FORMATS, BUFFER_NAMES, RT_BUF_COUNT, RT_BUF_FINAL_BGRA8, RT_KERNEL_DEFAULT, RtConfig, RtCounters, RtInfo, RtTiming,
                  RtUniforms)


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("rt error %d: %s" % (code, message))
        self.code = code


def make_config(width, height, spp=1, depth=2, device=0, tile_rank=0, tile_world=1, kernel=RT_KERNEL_DEFAULT, flags=0,
                region=256):
    cfg = RtConfig()
    cfg.struct_size = C.sizeof(RtConfig)
    cfg.width, cfg.height, cfg.region = int(width), int(height), int(region)
    cfg.spp, cfg.depth, cfg.device = int(spp), int(depth), int(device)
    cfg.tile_rank, cfg.tile_world = int(tile_rank), int(tile_world)
    cfg.kernel, cfg.flags = int(kernel), int(flags)
    return cfg


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """Owns one RtContext*.  Use as a context manager or call destroy()."""

    def __init__(self, cfg=None, handle=None, owned=True):
        self._lib = _lib.amd()
        self._owned = owned
        if handle is not None:
            self._h = C.c_void_p(handle)
            self.cfg = cfg
            return
        h = C.c_void_p()
        rc = self._lib.rt_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(None).decode())
        self._h = h
        self.cfg = cfg

    # -- lifecycle ---------------------------------------------------------------------------------------
    def destroy(self):
        self._retire_staging_views()
        if self._h and self._owned:
            self._lib.rt_destroy(self._h)
        self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.destroy()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RtError(rc, self._lib.rt_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    # -- uploads -----------------------------------------------------------------------------------------
    def upload_world(self, materials, minefield):
        materials = np.ascontiguousarray(materials, dtype=np.uint32).reshape(-1)
        minefield = np.ascontiguousarray(minefield, dtype=np.uint8).reshape(-1)
        n = (self.cfg.region if self.cfg is not None else 256) ** 3
        if materials.size != n or minefield.size != n:
            raise ValueError("world arrays must hold region^3 voxels")
        self._check(self._lib.rt_upload_world(self._h, _p(materials), _p(minefield)))

    def slice_staging(self):
        """rt_slice_staging: (materials u32[16 R^2], minefield u8[16 R^2]) views of the library's pinned slab staging — fill them
        and pass them to upload_slice to skip the copy into the staging buffer.  The memory belongs to the context: the views are
        writable from this call until the next upload_slice (which hands the buffer to the device and makes them read-only: there
        are two staging sets, ask again for every slab) and dead after destroy() — do not keep them."""
        pm, pf = C.c_void_p(), C.c_void_p()
        self._check(self._lib.rt_slice_staging(self._h, C.byref(pm), C.byref(pf)))
        n = 16 * (self.cfg.region if self.cfg is not None else 256) ** 2
        mats = np.ctypeslib.as_array(C.cast(pm, C.POINTER(C.c_uint32)), shape=(n,))
        mine = np.ctypeslib.as_array(C.cast(pf, C.POINTER(C.c_uint8)), shape=(n,))
        self._staging_views = (mats, mine)
        return mats, mine

    def _retire_staging_views(self):
        for v in getattr(self, "_staging_views", None) or ():
            v.flags.writeable = False       # a write after the hand-over would race with the host-to-device copy
        self._staging_views = None

    def upload_slice(self, axis, texel_offset, materials, minefield):
        materials = np.ascontiguousarray(materials, dtype=np.uint32).reshape(-1)
        minefield = np.ascontiguousarray(minefield, dtype=np.uint8).reshape(-1)
        n = 16 * (self.cfg.region if self.cfg is not None else 256) ** 2
        if materials.size != n or minefield.size != n:
            raise ValueError("slice arrays must hold 16*region*region voxels")
        try:
            self._check(self._lib.rt_upload_slice(self._h, int(axis), int(texel_offset), _p(materials), _p(minefield)))
        finally:
            self._retire_staging_views()

    def upload_noise(self, rgba8):
        rgba8 = np.ascontiguousarray(rgba8, dtype=np.uint8).reshape(-1)
        if rgba8.size != 512 * 512 * 4:
            raise ValueError("noise must be 512x512 RGBA8")
        self._check(self._lib.rt_upload_noise(self._h, _p(rgba8)))

    # -- frames ------------------------------------------------------------------------------------------
    def draw_frame(self, uniforms):
        self._check(self._lib.rt_draw_frame(self._h, C.byref(uniforms)))

    def sync(self):
        self._check(self._lib.rt_sync(self._h))

    def set_stream(self, stream_ptr):
        self._check(self._lib.rt_set_stream(self._h, C.c_void_p(stream_ptr)))

    def buffer_bytes(self, buffer_id):
        return int(self._lib.rt_buffer_bytes(self._h, int(buffer_id)))

    def device_ptr(self, buffer_id):
        return self._lib.rt_device_ptr(self._h, int(buffer_id))

    def tile_count(self):
        return int(self._lib.rt_tile_count(self._h))

    def tile_capacity(self):
        return int(self._lib.rt_tile_capacity(self._h))

    def readback(self, buffer_id):
        """Returns the plane as numpy: [H,W(,C)] for whole-frame contexts, [capacity*64(,C)] for tile-split ones."""
        dt, ch = BUFFER_FORMATS[buffer_id]
        nbytes = self.buffer_bytes(buffer_id)
        out = np.empty(nbytes // np.dtype(dt).itemsize, dtype=dt)
        self._check(self._lib.rt_readback(self._h, int(buffer_id), _p(out), nbytes))
        if self.cfg is not None and self.cfg.tile_world == 1:
            shape = (self.cfg.height, self.cfg.width) + ((ch,) if ch > 1 else ())
        else:
            shape = (-1,) + ((ch,) if ch > 1 else ())
        return out.reshape(shape)

    def readback_all(self):
        """The nine planes the ray-trace dispatch writes (not the finalize output)."""
        return {BUFFER_NAMES[b]: self.readback(b) for b in range(RT_BUF_FINAL_BGRA8)}

    def untile(self, buffer_id, gathered_dev_ptr, world, frame_dev_ptr):
        self._check(self._lib.rt_untile(self._h, int(buffer_id), C.c_void_p(gathered_dev_ptr), int(world),
                                        C.c_void_p(frame_dev_ptr)))

    def gbuffer_ptr(self):
        return self._lib.rt_gbuffer_ptr(self._h)

    def gbuffer_bytes(self):
        return int(self._lib.rt_gbuffer_bytes(self._h))

    def gbuffer_offset(self, buffer_id):
        """Byte offset of a reference-format plane inside the contiguous G-buffer block."""
        return int(self._lib.rt_gbuffer_offset(self._h, int(buffer_id)))

    def untile_gbuffer(self, gathered_dev_ptr, world, frame_dev_ptrs):
        """Scatter `world` gathered G-buffer blocks into six row-major planes (device pointers, None to skip)."""
        arr = (C.c_void_p * 6)(*[C.c_void_p(p) if p else None for p in frame_dev_ptrs])
        self._check(self._lib.rt_untile_gbuffer(self._h, C.c_void_p(gathered_dev_ptr), int(world), arr))

    # -- multi-GPU frame assembly (rt_gather_gbuffer) ------------------------------------------------------
    def comm_init_rank(self, unique_id):
        """ncclCommInitRank(tile_world, unique_id, tile_rank) on this context's device; returns the communicator handle."""
        comm = C.c_void_p()
        buf = (C.c_char * len(unique_id)).from_buffer_copy(bytes(unique_id))
        self._check(self._lib.rt_comm_init_rank(self._h, buf, len(unique_id), C.byref(comm)))
        return comm.value

    def gather_gbuffer(self, comm, root, frame_dev_ptrs=None, overlapped=False):
        """Every rank's six reference-format planes -> `root` over RCCL on the context's stream, un-tiled there into the six
        row-major device planes frame_dev_ptrs (root only)."""
        arr = (C.c_void_p * 6)(*[C.c_void_p(p) if p else None for p in frame_dev_ptrs]) if frame_dev_ptrs else None
        self._check(self._lib.rt_gather_gbuffer(self._h, C.c_void_p(comm) if comm else None, int(root), arr, 1 if overlapped else 0))

    def selftest(self, which=1):
        """rt_selftest: RT_SELFTEST_DENOISE_DIVISION (1) -> number of inexact quotients over the denoise division's domain."""
        out = C.c_uint64(0)
        self._check(self._lib.rt_selftest(self._h, int(which), C.byref(out)))
        return int(out.value)

    def frame_ptr(self, buffer_id):
        return self._lib.rt_frame_ptr(self._h, int(buffer_id))

    def frame_readback(self, buffer_id):
        """Plane `buffer_id` of the frame rt_gather_gbuffer assembled in the library's own planes (frames_dev = NULL)."""
        dt, ch = BUFFER_FORMATS[buffer_id]
        W, H = self.cfg.width, self.cfg.height
        out = np.empty((H, W, ch) if ch > 1 else (H, W), dtype=dt)
        self._check(self._lib.rt_frame_readback(self._h, int(buffer_id), _p(out), out.nbytes))
        return out

    # -- post passes (pipeline.rs:98-123) ----------------------------------------------------------------
    def denoise(self, faithful=True):
        self._check(self._lib.rt_denoise(self._h, 1 if faithful else 0))

    def finalize(self):
        self._check(self._lib.rt_finalize(self._h))

    def denoise_planes(self, lighting_ptr, depth_ptr, normal_ptr, faithful=True):
        """The six denoise dispatches on caller-owned row-major device planes (e.g. the gathered frame on rank 0)."""
        self._check(self._lib.rt_denoise_planes(self._h, C.c_void_p(lighting_ptr), C.c_void_p(depth_ptr), C.c_void_p(normal_ptr),
                                                1 if faithful else 0))

    def finalize_planes(self, albedo_ptr, emission_ptr, fog_ptr, lighting_ptr, depth_ptr, out_ptr):
        self._check(self._lib.rt_finalize_planes(self._h, C.c_void_p(albedo_ptr), C.c_void_p(emission_ptr), C.c_void_p(fog_ptr),
                                                 C.c_void_p(lighting_ptr), C.c_void_p(depth_ptr), C.c_void_p(out_ptr)))

    # -- instrumentation ---------------------------------------------------------------------------------
    def kernel_in_use(self):
        """RtKernel the context runs (what RT_KERNEL_DEFAULT resolved to)."""
        return int(self._lib.rt_kernel_in_use(self._h))

    def counters(self):
        cn = RtCounters()
        self._check(self._lib.rt_get_counters(self._h, C.byref(cn)))
        return cn

    def reset_counters(self):
        self._check(self._lib.rt_reset_counters(self._h))

    def info(self):
        i = RtInfo()
        i.struct_size = C.sizeof(RtInfo)
        self._check(self._lib.rt_get_info(self._h, C.byref(i)))
        return i

    def timing(self):
        t = RtTiming()
        self._check(self._lib.rt_get_timing(self._h, C.byref(t)))
        return t

    def gather_timing(self):
        """rt_get_gather_timing: (sum of the rt_gather_gbuffer calls' device time in ms, number of calls) since the last call —
        events round the transfer + un-tile on the stream they run on; contexts with RT_FLAG_TIMING only (else (0.0, 0))."""
        ms, n = C.c_float(0.0), C.c_uint32(0)
        self._check(self._lib.rt_get_gather_timing(self._h, C.byref(ms), C.byref(n)))
        return float(ms.value), int(n.value)


# ---- host mirror of the reference API ------------------------------------------------------------------------

def compute_triple_euler_vector(heading, pitch):
    """src/util.rs:9-22 (C++ mirror). Returns (forward, up, right) as float32 arrays."""
    f = (C.c_float * 3)()
    u = (C.c_float * 3)()
    r = (C.c_float * 3)()
    _lib.host().rth_compute_triple_euler_vector(float(heading), float(pitch), f, u, r)
    return (np.array(f[:], dtype=np.float32), np.array(u[:], dtype=np.float32), np.array(r[:], dtype=np.float32))


class Camera:
    """render::Camera (src/render/mod.rs:20-34): view of the native Game's camera."""

    def __init__(self, game):
        self._game = game

    def _get(self):
        o = (C.c_float * 3)()
        h = C.c_float()
        p = C.c_float()
        _lib.host().rth_game_get_camera(self._game._h, o, C.byref(h), C.byref(p))
        return [o[0], o[1], o[2]], h.value, p.value

    @property
    def origin(self):
        return tuple(self._get()[0])

    @property
    def heading(self):
        return self._get()[1]

    @property
    def pitch(self):
        return self._get()[2]

    def set(self, origin=None, heading=None, pitch=None):
        o, h, p = self._get()
        if origin is not None:
            o = list(origin)
        if heading is not None:
            h = heading
        if pitch is not None:
            p = pitch
        _lib.host().rth_game_set_camera(self._game._h, (C.c_float * 3)(*o), float(h), float(p))


class Game:
    """game::Game (src/game/mod.rs:14-58): camera + sun angle + world.  `args` are the reference's six optional
    positional CLI floats `x y z heading pitch sun_angle` (mod.rs:45-52)."""

    def __init__(self, args=()):
        argv = [b"raytrace"] + [str(a).encode() for a in args]
        arr = (C.c_char_p * len(argv))(*argv)
        self._h = C.c_void_p(_lib.host().rth_game_new(len(argv), arr))
        if not self._h:
            raise MemoryError("Game")
        self.camera = Camera(self)

    def borrow_camera(self):
        return self.camera

    def get_sun_angle(self):
        return float(_lib.host().rth_game_get_sun_angle(self._h))

    def set_sun_angle(self, a):
        _lib.host().rth_game_set_sun_angle(self._h, float(a))

    def set_world(self, materials, minefield, region=256):
        materials = np.ascontiguousarray(materials, dtype=np.uint32).reshape(-1)
        minefield = np.ascontiguousarray(minefield, dtype=np.uint8).reshape(-1)
        if materials.size != region ** 3 or minefield.size != region ** 3:
            raise ValueError("world arrays must hold region^3 voxels")
        rc = _lib.host().rth_game_set_world_r(self._h, _p(materials), _p(minefield), int(region))
        if rc != 0:
            raise RtError(rc, "set_world: region must be 256, 512 or 1024")

    def generate_world(self, seed=0x5EED, region=256):
        rc = _lib.host().rth_game_generate_world_r(self._h, C.c_uint64(int(seed)), int(region))
        if rc != 0:
            raise RtError(rc, "generate_world: region must be 256, 512 or 1024")

    def close(self):
        if self._h:
            _lib.host().rth_game_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Pipeline:
    """render::Pipeline (src/render/pipeline/pipeline.rs).  Create with `create_instance`."""

    def __init__(self, handle, cfg):
        self._h = C.c_void_p(handle)
        self.cfg = cfg
        self.context = Context(cfg=cfg, handle=_lib.host().rth_pipeline_context(self._h), owned=False)

    def draw_frame(self, game):
        """pipeline.rs:134-255: derive uniforms from the camera, submit the ray-trace work (asynchronous)."""
        rc = _lib.host().rth_pipeline_draw_frame(self._h, game._h)
        if rc != 0:
            raise RtError(rc, _lib.host().rth_pipeline_last_error(self._h).decode())

    def wait(self):
        rc = _lib.host().rth_pipeline_wait(self._h)
        if rc != 0:
            raise RtError(rc, _lib.host().rth_pipeline_last_error(self._h).decode())

    def uniforms(self):
        u = RtUniforms()
        _lib.host().rth_pipeline_uniforms(self._h, C.byref(u))
        return u

    def set_seed(self, seed):
        _lib.host().rth_pipeline_set_seed(self._h, C.c_uint32(int(seed)))

    def enable_terrain_streaming(self, seed=0x5EED, storage_dir=""):
        """pipeline.rs:174-189: every draw_frame moves the TerrainUploadManager towards the camera (<= 1 slab per frame)."""
        _lib.host().rth_pipeline_enable_streaming(self._h, C.c_uint64(int(seed)), str(storage_dir).encode() if storage_dir else None)

    def enable_post_passes(self, faithful=True):
        """pipeline.rs:98-123: every draw_frame then enqueues ray trace -> six denoise dispatches -> finalize (the reference's one
        command buffer, :229-235); RT_BUF_FINAL_BGRA8 holds the frame's swapchain image.  Whole-frame contexts only."""
        rc = _lib.host().rth_pipeline_enable_post_passes(self._h, 1 if faithful else 0)
        if rc != 0:
            raise RtError(rc, "enable_post_passes: whole-frame contexts only (tile_world == 1)")

    def close(self):
        if self._h:
            _lib.host().rth_pipeline_free(self._h)   # impl Drop for Pipeline, pipeline.rs:258-277
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def create_instance(cfg, game, blue_noise_rgba8):
    """render::create_instance (src/render/mod.rs:36-43): builds the pipeline, uploads the game's world
    (generating the procedural one if the game has none) and the blue-noise table."""
    noise = np.ascontiguousarray(blue_noise_rgba8, dtype=np.uint8).reshape(-1)
    if noise.size != 512 * 512 * 4:
        raise ValueError("noise must be 512x512 RGBA8")
    err = C.create_string_buffer(512)
    h = _lib.host().rth_create_instance(C.byref(cfg), _p(noise), game._h, err, 512)
    if not h:
        raise RtError(-1, err.value.decode())
    return Pipeline(h, cfg)


def camera_uniforms(origin, heading, pitch, sun_angle=0.0, seed=1, lr=(0, 0, 0)):
    """The uniform fill of Pipeline::draw_frame (pipeline.rs:191-207) as a standalone helper (product-side;
    the oracle has its own restatement)."""
    fwd, up, right = compute_triple_euler_vector(heading, pitch)
    u = RtUniforms()
    u.sun_angle = float(sun_angle)
    u.seed = int(seed)
    for a in range(3):
        u.origin[a] = float(origin[a])
        u.forward[a] = float(fwd[a])
        u.up[a] = float(np.float32(up[a]) * np.float32(0.4))
        u.right[a] = float(np.float32(right[a]) * np.float32(0.4))
        u.lr[a] = int(lr[a])
        u.lso[a] = int(lr[a])
    return u


DEFAULT_POSE = dict(origin=(-30.0, -128.0, 100.0), heading=math.pi / 2, pitch=0.0, sun_angle=0.0)  # game/mod.rs:53-55


def comm_unique_id():
    """ncclGetUniqueId: 128 opaque bytes one rank creates and the host hands to every rank (rt_comm_unique_id)."""
    buf = (C.c_char * 128)()
    rc = _lib.amd().rt_comm_unique_id(buf, 128)
    if rc != 0:
        raise RtError(rc, _lib.amd().rt_last_error(None).decode())
    return bytes(buf)


def comm_destroy(comm):
    if comm:
        _lib.amd().rt_comm_destroy(C.c_void_p(comm))
