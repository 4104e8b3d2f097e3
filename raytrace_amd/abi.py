"""ctypes mirror of include/rt_abi.h (structs, enums, constants).

`RtUniforms` is byte-identical to the reference's `RaytraceUniformData`
(src/render/pipeline/structs.rs:3-31) and the GLSL block at shaders/glsl/raytrace.comp:25-35.
"""
import ctypes as C

ROOT_BLOCK_SIZE = 256      # src/render/constants.rs:27
CHUNK_SIZE = 64            # constants.rs:23
NOISE_SIZE = 512           # constants.rs:16-17
NOISE_BYTES = 512 * 512 * 4  # constants.rs:19 BLUE_NOISE_SIZE
MAX_DEPTH = 16

RT_OK = 0
RT_ERR_INVALID_ARG = -1
RT_ERR_NO_DEVICE = -2
RT_ERR_HIP = -3
RT_ERR_NOT_READY = -4
RT_ERR_UNIMPLEMENTED = -5
RT_ERR_OOM = -6

RT_FLAG_TRUSTED_WORLD = 0x10
RT_KERNEL_DEFAULT, RT_KERNEL_MEGA, RT_KERNEL_WAVEFRONT, RT_KERNEL_PERSISTENT, RT_KERNEL_PATHS, RT_KERNEL_FRAME = 0, 1, 2, 3, 5, 7   # 4 (PERSISTENT2) and 6 (SEQ): retired
RT_FLAG_FRAMES_IN_FLIGHT_2 = 0x20
RT_FLAG_COUNTERS = 0x1
RT_FLAG_CACHE_PRIMARY = 0x2
RT_FLAG_TIMING = 0x4
RT_FLAG_TIMING_ALL = 0xC

(RT_BUF_LIGHTING_RGBA16, RT_BUF_DEPTH_R16UI, RT_BUF_NORMAL_R8UI, RT_BUF_ALBEDO_RGBA8,
 RT_BUF_EMISSION_RGBA8, RT_BUF_FOG_RGBA8, RT_BUF_LIGHTING_F32, RT_BUF_FOG_F32,
 RT_BUF_DEPTH_F32, RT_BUF_FINAL_BGRA8, RT_BUF_COUNT) = range(11)

# (numpy dtype, channels) per output plane
BUFFER_FORMATS = {
    RT_BUF_LIGHTING_RGBA16: ("uint16", 4),
    RT_BUF_DEPTH_R16UI: ("uint16", 1),
    RT_BUF_NORMAL_R8UI: ("uint8", 1),
    RT_BUF_ALBEDO_RGBA8: ("uint8", 4),
    RT_BUF_EMISSION_RGBA8: ("uint8", 4),
    RT_BUF_FOG_RGBA8: ("uint8", 4),
    RT_BUF_LIGHTING_F32: ("float32", 4),
    RT_BUF_FOG_F32: ("float32", 4),
    RT_BUF_DEPTH_F32: ("float32", 1),
    RT_BUF_FINAL_BGRA8: ("uint8", 4),
}
BUFFER_NAMES = {
    RT_BUF_LIGHTING_RGBA16: "lighting_rgba16", RT_BUF_DEPTH_R16UI: "depth_r16", RT_BUF_NORMAL_R8UI: "normal_r8",
    RT_BUF_ALBEDO_RGBA8: "albedo_rgba8", RT_BUF_EMISSION_RGBA8: "emission_rgba8", RT_BUF_FOG_RGBA8: "fog_rgba8",
    RT_BUF_LIGHTING_F32: "lighting_f32", RT_BUF_FOG_F32: "fog_f32", RT_BUF_DEPTH_F32: "depth_f32",
    RT_BUF_FINAL_BGRA8: "final_bgra8",
}


class RtUniforms(C.Structure):
    _fields_ = [
        ("sun_angle", C.c_float), ("seed", C.c_uint32), ("_padding0", C.c_uint32 * 2),
        ("origin", C.c_float * 3), ("_padding1", C.c_uint32),
        ("forward", C.c_float * 3), ("_padding2", C.c_uint32),
        ("up", C.c_float * 3), ("_padding3", C.c_uint32),
        ("right", C.c_float * 3), ("_padding4", C.c_uint32),
        ("old_origin", C.c_float * 3), ("_padding5", C.c_uint32),
        ("old_transform_c0", C.c_float * 3), ("_padding6", C.c_uint32),
        ("old_transform_c1", C.c_float * 3), ("_padding7", C.c_uint32),
        ("old_transform_c2", C.c_float * 3), ("_padding8", C.c_uint32),
        ("region_offset", C.c_int32 * 3), ("_padding9", C.c_uint32),
        ("lr", C.c_int32 * 3), ("_padding10", C.c_uint32),
        ("lso", C.c_int32 * 3), ("_padding11", C.c_uint32),
    ]


assert C.sizeof(RtUniforms) == 192


class RtConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32), ("width", C.c_int32), ("height", C.c_int32), ("region", C.c_int32),
        ("spp", C.c_int32), ("depth", C.c_int32), ("device", C.c_int32), ("tile_rank", C.c_int32),
        ("tile_world", C.c_int32), ("kernel", C.c_int32), ("flags", C.c_uint32), ("reserved", C.c_int32 * 5),
    ]


class RtCounters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "rays", "rays_primary", "rays_shadow", "rays_diffuse", "iterations", "minefield_fetches",
        "material_fetches", "noise_fetches", "hits", "sky_exits", "limit_exits", "border_fetches",
        "pixels", "frames")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}

    def algorithmic_bytes(self):
        """B_alg of SURVEY.md 8(d) / BASELINE.md 5."""
        return self.minefield_fetches + 4 * self.material_fetches + 4 * self.noise_fetches + 23 * self.pixels


class RtInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("num_cus", C.c_int32), ("samples_per_launch", C.c_uint32), ("launches_in_flight", C.c_uint16), ("frames_in_flight", C.c_uint16),
                ("light_record_budget_bytes", C.c_uint64), ("light_record_bytes", C.c_uint64), ("device_bytes", C.c_uint64)]


class RtTiming(C.Structure):
    _fields_ = [("frame_ms", C.c_float), ("trace_ms", C.c_float), ("shade_ms", C.c_float),
                ("trace_launches", C.c_uint32), ("other_launches", C.c_uint32), ("rays_traced", C.c_uint64)]
