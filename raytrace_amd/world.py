"""Python binding of the C++ world mirror (raytrace_amd/host/world.cpp): material table, minefield builder,
region assembly and the deterministic terrain generator.  Mirrors src/world + src/render/GEN_MATERIALS.rs."""
import ctypes as C

import numpy as np

from . import _lib
from .abi import CHUNK_SIZE, ROOT_BLOCK_SIZE

REGION_VOLUME = ROOT_BLOCK_SIZE ** 3
CHUNK_VOLUME = CHUNK_SIZE ** 3
DEFAULT_SEED = 0x5EED


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def material_pack(material_id):
    """MATERIALS[id].pack() — GEN_MATERIALS.rs:44-51,70-106."""
    return int(_lib.host().rth_material_pack(int(material_id)))


def material_unpack(packed):
    """Material::unpack — GEN_MATERIALS.rs:53-68. Returns ((r,g,b), solid)."""
    alb = (C.c_uint16 * 3)()
    solid = C.c_int()
    _lib.host().rth_material_unpack(C.c_uint32(int(packed)), alb, C.byref(solid))
    return tuple(alb[:]), bool(solid.value)


def material(material_id):
    alb = (C.c_uint16 * 3)()
    emi = (C.c_uint16 * 3)()
    solid = C.c_int()
    rc = _lib.host().rth_material_get(int(material_id), alb, emi, C.byref(solid))
    if rc != 0:
        raise IndexError(material_id)
    return {"albedo": tuple(alb[:]), "emission": tuple(emi[:]), "solid": bool(solid.value)}


def pack_chunk(ids):
    """UnpackedChunkData::pack_into (src/world/chunk.rs:125-184) on a 64^3 array of material ids [z,y,x]."""
    ids = np.ascontiguousarray(ids, dtype=np.uint8).reshape(-1)
    assert ids.size == CHUNK_VOLUME
    mats = np.zeros(CHUNK_VOLUME, dtype=np.uint32)
    mine = np.zeros(CHUNK_VOLUME, dtype=np.uint8)
    rc = _lib.host().rth_pack_chunk(_p(ids), _p(mats), _p(mine))
    assert rc == 0
    return mats.reshape(64, 64, 64), mine.reshape(64, 64, 64)


def generate_region(seed=DEFAULT_SEED):
    """Procedural 256^3 region (render_data.rs:203-249 assembly over the deterministic generator).
    Returns (materials u32[256,256,256], minefield u8[256,256,256]) indexed [z,y,x], texel = world + 128."""
    mats = np.zeros(REGION_VOLUME, dtype=np.uint32)
    mine = np.zeros(REGION_VOLUME, dtype=np.uint8)
    rc = _lib.host().rth_generate_region(C.c_uint64(int(seed)), _p(mats), _p(mine))
    assert rc == 0
    return mats.reshape(256, 256, 256), mine.reshape(256, 256, 256)


def region_from_ids(ids):
    """Flatten a 256^3 array of material ids [z,y,x] (texel space) exactly as the reference would:
    split into 64^3 chunks, pack_into each, assemble."""
    ids = np.ascontiguousarray(ids, dtype=np.uint8).reshape(-1)
    assert ids.size == REGION_VOLUME
    mats = np.zeros(REGION_VOLUME, dtype=np.uint32)
    mine = np.zeros(REGION_VOLUME, dtype=np.uint8)
    rc = _lib.host().rth_region_from_ids(_p(ids), _p(mats), _p(mine))
    assert rc == 0
    return mats.reshape(256, 256, 256), mine.reshape(256, 256, 256)


def heightmap(chunk_x, chunk_y, seed=DEFAULT_SEED):
    out = np.zeros(64 * 64, dtype=np.int64)
    _lib.host().rth_heightmap(int(chunk_x), int(chunk_y), C.c_uint64(int(seed)), _p(out))
    return out.reshape(64, 64)


def _i3(v):
    return (C.c_int * 3)(*[int(x) for x in v])


def _l3(v):
    return (C.c_long * 3)(*[int(x) for x in v])


def copy_3d(size, source, source_dims, source_start, target, target_dims, target_position):
    """util::copy_3d (src/util.rs:380-415); arrays are flat u32, x fastest. Raises where the reference panics."""
    assert source.dtype == np.uint32 and target.dtype == np.uint32
    rc = _lib.host().rth_copy_3d_u32(_i3(size), _p(source), _i3(source_dims), _i3(source_start), _p(target),
                                     _i3(target_dims), _i3(target_position))
    if rc != 0:
        raise ValueError("copy_3d out of bounds")


def copy_3d_auto_clip(source, source_stride, source_offset, target, target_stride):
    """util::copy_3d_auto_clip (src/util.rs:440-494)."""
    assert source.dtype == np.uint32 and target.dtype == np.uint32
    _lib.host().rth_copy_3d_auto_clip_u32(_p(source), int(source_stride), _l3(source_offset), _p(target), int(target_stride))


def copy_3d_bounded_auto_clip(size, source, source_dims, source_start, target, target_dims, target_start):
    """util::copy_3d_bounded_auto_clip (src/util.rs:507-583)."""
    assert source.dtype == np.uint32 and target.dtype == np.uint32
    _lib.host().rth_copy_3d_bounded_auto_clip_u32(_i3(size), _p(source), _i3(source_dims), _i3(source_start), _p(target),
                                                  _i3(target_dims), _l3(target_start))


def fill_slice_3d_auto_clip(value, target, target_stride, slice_start, slice_size):
    """util::fill_slice_3d_auto_clip (src/util.rs:636-668) on a flat u8 array."""
    assert target.dtype == np.uint8
    _lib.host().rth_fill_slice_3d_auto_clip_u8(C.c_uint8(int(value)), _p(target), int(target_stride), _l3(slice_start),
                                               _i3(slice_size))
