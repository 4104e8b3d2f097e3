"""The C ABI (include/rt_abi.h): the library loads, exports every declared symbol, struct layouts match the reference's
uniform block, and — without a GPU — fails loudly instead of falling back to anything."""
import ctypes as C
import os
import re

import pytest

from raytrace_amd import _lib, abi, render
from tests.conftest import ROOT

pytestmark = pytest.mark.usefixtures("native_built")


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.amd()
    declared = _declared_functions()
    assert len(declared) >= 21
    assert sorted(_lib.ABI_SYMBOLS) == declared
    for name in declared:
        assert hasattr(lib, name), "librt_amd.so does not export %s" % name
    assert lib.rt_abi_version() >> 16 == 1


def test_abi_minor_version_is_pinned_and_its_history_is_in_the_header():
    """the ABI changed in round 3 (new exports, RtKernel 4 rejected, asynchronous rt_upload_slice) without a version
    bump.  The minor version now moves with every such change and the header says which minor introduced what."""
    lib = _lib.amd()
    text = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    major = int(re.search(r"#define RT_ABI_VERSION_MAJOR (\d+)", text).group(1))
    minor = int(re.search(r"#define RT_ABI_VERSION_MINOR (\d+)", text).group(1))
    assert (major, minor) == (1, 3)
    assert lib.rt_abi_version() == (major << 16) | minor
    for needle in ("1.1  round 3: rt_slice_staging", "1.2  round 4: RT_FLAG_FRAMES_IN_FLIGHT_2", "rt_samples_per_launch", "rt_get_gather_timing",
                   "1.3  round 4: RtKernel value 7 (RT_KERNEL_FRAME)"):
        assert needle in text, needle
    assert abi.RT_KERNEL_FRAME == int(re.search(r"RT_KERNEL_FRAME = (\d+),", text).group(1)) == 7
    assert abi.RT_FLAG_FRAMES_IN_FLIGHT_2 == int(re.search(r"#define RT_FLAG_FRAMES_IN_FLIGHT_2 (0x[0-9a-f]+)u", text).group(1), 16)
    assert C.sizeof(abi.RtInfo) == 40 and abi.RtInfo.launches_in_flight.offset == 12 and abi.RtInfo.frames_in_flight.offset == 14


def test_launch_sizing_rule():
    """the samples one path launch covers (free memory, pixels, spp -> B) as a pure function of its inputs."""
    f = _lib.amd().rt_samples_per_launch
    GiB = 1 << 30
    npix_1080, npix_4k = 1920 * 1080, 3840 * 2160
    assert f(8 * GiB, npix_1080, 64, None) == 64                       # the headline frame: one launch per frame on either lane
    assert f(8 * GiB, npix_4k, 256, None) == 8 * GiB // (12 * npix_4k) == 86      # C4: 86 + 86 + 84 samples over two lanes
    assert f(1 * GiB, npix_4k, 256, None) == 10
    assert f(100, npix_4k, 256, None) == 1                             # never below one sample
    assert f(64 * GiB, npix_4k, 1024, None) == (1 << 31) // npix_4k    # path indices are 32-bit
    assert f(8 * GiB, 0, 4, None) == 4                                 # an empty share still sizes
    assert f(8 * GiB, npix_4k, 256, b"16") == 16                       # RT_PERSIST_BATCH lowers ...
    assert f(8 * GiB, npix_4k, 256, b"500") == 86                      # ... and never raises the bound


def test_uniform_block_layout_matches_the_reference():
    # src/render/pipeline/structs.rs:3-31; shaders/glsl/raytrace.comp:25-35 (std140)
    assert C.sizeof(abi.RtUniforms) == 192
    offsets = {"sun_angle": 0, "seed": 4, "origin": 16, "forward": 32, "up": 48, "right": 64, "old_origin": 80,
               "old_transform_c0": 96, "old_transform_c1": 112, "old_transform_c2": 128, "region_offset": 144,
               "lr": 160, "lso": 176}
    for name, off in offsets.items():
        assert getattr(abi.RtUniforms, name).offset == off, name


def test_struct_sizes_match_the_header():
    text = open(os.path.join(ROOT, "include", "rt_abi.h")).read()
    assert C.sizeof(abi.RtConfig) == 4 * 11 + 4 * 5
    assert C.sizeof(abi.RtCounters) == 8 * 14
    assert len(re.findall(r"uint64_t\s+\w+;", text.split("typedef struct RtCounters")[1].split("}")[0])) == 14
    assert C.sizeof(abi.RtTiming) == 32
    for name in ("RT_BUF_LIGHTING_RGBA16 = 0", "RT_BUF_DEPTH_F32       = 8", "RT_BUF_FINAL_BGRA8     = 9", "RT_BUF_COUNT           = 10"):
        assert name in text


def test_create_rejects_bad_configs_before_touching_a_device():
    lib = _lib.amd()
    h = C.c_void_p()
    cfg = render.make_config(64, 64)
    cfg.struct_size = 12
    assert lib.rt_create(C.byref(cfg), C.byref(h)) == abi.RT_ERR_INVALID_ARG and not h
    assert b"struct_size" in lib.rt_last_error(None)
    for bad in (dict(width=0), dict(height=-3), dict(spp=0), dict(depth=17), dict(tile_rank=2, tile_world=2),
                dict(kernel=9), dict(kernel=4), dict(kernel=6), dict(region=128)):
        cfg = render.make_config(64, 64)
        for k, v in bad.items():
            setattr(cfg, k, v)
        assert lib.rt_create(C.byref(cfg), C.byref(h)) == abi.RT_ERR_INVALID_ARG, bad
        assert not h
    assert lib.rt_create(None, C.byref(h)) == abi.RT_ERR_INVALID_ARG


def test_null_context_is_handled():
    lib = _lib.amd()
    lib.rt_destroy(None)
    assert lib.rt_sync(None) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_draw_frame(None, None) == abi.RT_ERR_INVALID_ARG
    assert lib.rt_buffer_bytes(None, 0) == 0
    assert lib.rt_device_ptr(None, 0) is None
    assert lib.rt_last_error(None) is not None


def test_no_gpu_means_no_context_and_no_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the loud-failure path is covered on the CPU-only runner")
    with pytest.raises(render.RtError) as e:
        render.Context(render.make_config(64, 64))
    assert e.value.code == abi.RT_ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under raytrace_amd/ may import, link or dlopen it."""
    pkg = os.path.join(ROOT, "raytrace_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "pyoracle" not in text and "rt_oracle" not in text and "librt_oracle" not in text, os.path.join(dirpath, f)
    out = os.popen("ldd %s" % _lib.LIB_AMD_PATH).read() + os.popen("ldd %s" % _lib.LIB_HOST_PATH).read()
    assert "oracle" not in out
