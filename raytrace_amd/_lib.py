"""Loads the in-tree native libraries.  There is no Python or CPU fallback: if librt_amd.so is missing the
import fails loudly (build it with `python -m raytrace_amd.build`)."""
import ctypes as C
import os

from .abi import RtConfig, RtCounters, RtInfo, RtTiming, RtUniforms

HERE = os.path.dirname(os.path.abspath(__file__))
# RT_AMD_LIB: load another build of the same library (same-box A/B timing of two kernel variants, tools/ab.sh)
LIB_AMD_PATH = os.environ.get("RT_AMD_LIB") or os.path.join(HERE, "librt_amd.so")
LIB_HOST_PATH = os.path.join(HERE, "librt_host.so")

# Every symbol include/rt_abi.h declares.
ABI_SYMBOLS = (
    "rt_create", "rt_destroy", "rt_last_error", "rt_upload_world", "rt_upload_slice", "rt_slice_staging", "rt_upload_noise",
    "rt_draw_frame", "rt_sync", "rt_readback", "rt_buffer_bytes", "rt_device_ptr", "rt_set_stream",
    "rt_tile_count", "rt_tile_capacity", "rt_untile", "rt_gbuffer_ptr", "rt_gbuffer_bytes", "rt_gbuffer_offset",
    "rt_untile_gbuffer", "rt_denoise", "rt_finalize", "rt_denoise_planes", "rt_finalize_planes", "rt_kernel_in_use", "rt_get_counters", "rt_reset_counters", "rt_get_timing",
    "rt_abi_version",
    "rt_comm_unique_id", "rt_comm_init_rank", "rt_comm_init_all", "rt_comm_destroy", "rt_gather_gbuffer", "rt_frame_ptr",
    "rt_frame_readback", "rt_selftest", "rt_get_info", "rt_samples_per_launch", "rt_get_gather_timing",
)

_amd = None
_host = None


class NativeLibraryMissing(ImportError):
    pass


def amd():
    """librt_amd.so: the HIP kernels behind the C ABI."""
    global _amd
    if _amd is None:
        if not os.path.exists(LIB_AMD_PATH):
            raise NativeLibraryMissing(
                "%s not found: the HIP extension is required (run `python -m raytrace_amd.build`); "
                "there is no CPU fallback" % LIB_AMD_PATH)
        lib = C.CDLL(LIB_AMD_PATH, mode=C.RTLD_GLOBAL)
        P = C.c_void_p
        lib.rt_create.argtypes = [C.POINTER(RtConfig), C.POINTER(P)]
        lib.rt_create.restype = C.c_int
        lib.rt_destroy.argtypes = [P]
        lib.rt_destroy.restype = None
        lib.rt_last_error.argtypes = [P]
        lib.rt_last_error.restype = C.c_char_p
        lib.rt_upload_world.argtypes = [P, P, P]
        lib.rt_upload_slice.argtypes = [P, C.c_int, C.c_int, P, P]
        lib.rt_slice_staging.argtypes = [P, C.POINTER(P), C.POINTER(P)]
        lib.rt_upload_noise.argtypes = [P, P]
        lib.rt_draw_frame.argtypes = [P, C.POINTER(RtUniforms)]
        lib.rt_sync.argtypes = [P]
        lib.rt_readback.argtypes = [P, C.c_int, P, C.c_size_t]
        lib.rt_buffer_bytes.argtypes = [P, C.c_int]
        lib.rt_buffer_bytes.restype = C.c_size_t
        lib.rt_device_ptr.argtypes = [P, C.c_int]
        lib.rt_device_ptr.restype = P
        lib.rt_set_stream.argtypes = [P, P]
        lib.rt_tile_count.argtypes = [P]
        lib.rt_tile_capacity.argtypes = [P]
        lib.rt_untile.argtypes = [P, C.c_int, P, C.c_int, P]
        lib.rt_gbuffer_ptr.argtypes = [P]
        lib.rt_gbuffer_ptr.restype = P
        lib.rt_gbuffer_bytes.argtypes = [P]
        lib.rt_gbuffer_bytes.restype = C.c_size_t
        lib.rt_gbuffer_offset.argtypes = [P, C.c_int]
        lib.rt_gbuffer_offset.restype = C.c_size_t
        lib.rt_untile_gbuffer.argtypes = [P, P, C.c_int, P]
        lib.rt_untile_gbuffer.restype = C.c_int
        lib.rt_denoise.argtypes = [P, C.c_int]
        lib.rt_finalize.argtypes = [P]
        lib.rt_denoise_planes.argtypes = [P, P, P, P, C.c_int]
        lib.rt_finalize_planes.argtypes = [P, P, P, P, P, P, P]
        lib.rt_kernel_in_use.argtypes = [P]
        lib.rt_kernel_in_use.restype = C.c_int
        lib.rt_get_counters.argtypes = [P, C.POINTER(RtCounters)]
        lib.rt_reset_counters.argtypes = [P]
        lib.rt_get_timing.argtypes = [P, C.POINTER(RtTiming)]
        lib.rt_get_info.argtypes = [P, C.POINTER(RtInfo)]
        lib.rt_get_info.restype = C.c_int
        lib.rt_abi_version.restype = C.c_uint32
        lib.rt_samples_per_launch.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64, C.c_char_p]
        lib.rt_samples_per_launch.restype = C.c_uint64
        lib.rt_get_gather_timing.argtypes = [P, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        lib.rt_get_gather_timing.restype = C.c_int
        lib.rt_comm_unique_id.argtypes = [P, C.c_size_t]
        lib.rt_comm_init_rank.argtypes = [P, P, C.c_size_t, C.POINTER(P)]
        lib.rt_comm_init_all.argtypes = [C.c_int, C.POINTER(C.c_int), C.POINTER(P)]
        lib.rt_comm_destroy.argtypes = [P]
        lib.rt_gather_gbuffer.argtypes = [P, P, C.c_int, P, C.c_int]
        lib.rt_frame_ptr.argtypes = [P, C.c_int]
        lib.rt_frame_ptr.restype = P
        lib.rt_frame_readback.argtypes = [P, C.c_int, P, C.c_size_t]
        lib.rt_frame_readback.restype = C.c_int
        lib.rt_selftest.argtypes = [P, C.c_int, C.POINTER(C.c_uint64)]
        lib.rt_selftest.restype = C.c_int
        for name in ("rt_comm_unique_id", "rt_comm_init_rank", "rt_comm_init_all", "rt_comm_destroy", "rt_gather_gbuffer"):
            getattr(lib, name).restype = C.c_int
        for name in ("rt_upload_world", "rt_upload_slice", "rt_slice_staging", "rt_upload_noise", "rt_draw_frame", "rt_sync", "rt_readback",
                     "rt_set_stream", "rt_tile_count", "rt_tile_capacity", "rt_untile", "rt_untile_gbuffer", "rt_denoise", "rt_finalize",
                     "rt_denoise_planes", "rt_finalize_planes", "rt_get_counters",
                     "rt_reset_counters", "rt_get_timing"):
            getattr(lib, name).restype = C.c_int
        _amd = lib
    return _amd


def host():
    """librt_host.so: C++ host mirror (world flattening, camera, Game, Pipeline)."""
    global _host
    if _host is None:
        amd()  # dependency; also enforces the loud failure
        if not os.path.exists(LIB_HOST_PATH):
            raise NativeLibraryMissing("%s not found (run `python -m raytrace_amd.build`)" % LIB_HOST_PATH)
        # RT_HOST_LIB: another build of the same sources (the CPU sanitizer build, `make -C oracle asan`)
        lib = C.CDLL(os.environ.get("RT_HOST_LIB") or LIB_HOST_PATH)
        P = C.c_void_p
        lib.rth_material_pack.argtypes = [C.c_int]
        lib.rth_material_pack.restype = C.c_uint32
        lib.rth_material_unpack.argtypes = [C.c_uint32, P, P]
        lib.rth_material_get.argtypes = [C.c_int, P, P, P]
        lib.rth_pack_chunk.argtypes = [P, P, P]
        lib.rth_generate_region.argtypes = [C.c_uint64, P, P]
        lib.rth_generate_region_r.argtypes = [C.c_uint64, C.c_int, P, P]
        lib.rth_region_from_ids.argtypes = [P, P, P]
        lib.rth_heightmap.argtypes = [C.c_long, C.c_long, C.c_uint64, P]
        lib.rth_copy_3d_u32.argtypes = [P] * 7
        lib.rth_copy_3d_auto_clip_u32.argtypes = [P, C.c_int, P, P, C.c_int]
        lib.rth_copy_3d_bounded_auto_clip_u32.argtypes = [P] * 7
        lib.rth_fill_slice_3d_auto_clip_u8.argtypes = [C.c_uint8, P, C.c_int, P, P]
        lib.rth_chunk_codec_available.restype = C.c_int
        lib.rth_chunk_file_name.argtypes = [C.c_long, C.c_long, C.c_long, P]
        lib.rth_chunk_write.argtypes = [C.c_char_p, P, P]
        lib.rth_chunk_read.argtypes = [C.c_char_p, P, P]
        lib.rth_chunk_storage_new.argtypes = [C.c_char_p, C.c_uint64]
        lib.rth_chunk_storage_new.restype = P
        lib.rth_chunk_storage_free.argtypes = [P]
        lib.rth_chunk_storage_free.restype = None
        lib.rth_chunk_storage_borrow.argtypes = [P, C.c_long, C.c_long, C.c_long, P, P]
        lib.rth_chunk_storage_stats.argtypes = [P, P, P]
        lib.rth_tum_new.argtypes = [C.c_uint64]
        lib.rth_tum_new.restype = P
        lib.rth_tum_new_r.argtypes = [C.c_uint64, C.c_int]
        lib.rth_tum_new_r.restype = P
        lib.rth_tum_free.argtypes = [P]
        lib.rth_tum_free.restype = None
        lib.rth_tum_request.argtypes = [P, C.c_int, C.c_int]
        lib.rth_tum_move_towards.argtypes = [P, P]
        lib.rth_tum_pending.argtypes = [P]
        lib.rth_tum_step.argtypes = [P]
        lib.rth_tum_render_offset.argtypes = [P, P]
        lib.rth_tum_region.argtypes = [P, P, P]
        lib.rth_pipeline_enable_streaming.argtypes = [P, C.c_uint64, C.c_char_p]
        lib.rth_pipeline_enable_streaming.restype = None
        lib.rth_compute_triple_euler_vector.argtypes = [C.c_float, C.c_float, P, P, P]
        lib.rth_game_new.argtypes = [C.c_int, P]
        lib.rth_game_new.restype = P
        lib.rth_game_free.argtypes = [P]
        lib.rth_game_free.restype = None
        lib.rth_game_set_camera.argtypes = [P, P, C.c_float, C.c_float]
        lib.rth_game_get_camera.argtypes = [P, P, P, P]
        lib.rth_game_set_sun_angle.argtypes = [P, C.c_float]
        lib.rth_game_get_sun_angle.argtypes = [P]
        lib.rth_game_get_sun_angle.restype = C.c_float
        lib.rth_game_set_world.argtypes = [P, P, P]
        lib.rth_game_set_world_r.argtypes = [P, P, P, C.c_int]
        lib.rth_game_generate_world.argtypes = [P, C.c_uint64]
        lib.rth_game_generate_world_r.argtypes = [P, C.c_uint64, C.c_int]
        lib.rth_create_instance.argtypes = [C.POINTER(RtConfig), P, P, P, C.c_size_t]
        lib.rth_create_instance.restype = P
        lib.rth_pipeline_free.argtypes = [P]
        lib.rth_pipeline_free.restype = None
        lib.rth_pipeline_draw_frame.argtypes = [P, P]
        lib.rth_pipeline_wait.argtypes = [P]
        lib.rth_pipeline_context.argtypes = [P]
        lib.rth_pipeline_context.restype = P
        lib.rth_pipeline_uniforms.argtypes = [P, C.POINTER(RtUniforms)]
        lib.rth_pipeline_set_seed.argtypes = [P, C.c_uint32]
        lib.rth_pipeline_enable_post_passes.argtypes = [P, C.c_int]
        lib.rth_pipeline_enable_post_passes.restype = C.c_int
        lib.rth_pipeline_last_error.argtypes = [P]
        lib.rth_pipeline_last_error.restype = C.c_char_p
        _host = lib
    return _host
