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


def generate_region(seed=DEFAULT_SEED, region=256):
    """Procedural R^3 region (render_data.rs:203-249 assembly over the deterministic generator; R = 256 in the reference,
    512 / 1024 are the build's extension).  Returns (materials u32[R,R,R], minefield u8[R,R,R]) indexed [z,y,x],
    texel = world + R/2."""
    n = int(region) ** 3
    mats = np.zeros(n, dtype=np.uint32)
    mine = np.zeros(n, dtype=np.uint8)
    rc = _lib.host().rth_generate_region_r(C.c_uint64(int(seed)), int(region), _p(mats), _p(mine))
    assert rc == 0
    return mats.reshape(region, region, region), mine.reshape(region, region, region)


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


# ---- chunk disk cache (src/world/chunk_storage.rs) --------------------------------------------------------------

def chunk_codec_available():
    return bool(_lib.host().rth_chunk_codec_available())


def chunk_file_name(cx, cy, cz):
    """chunk_storage.rs:37-40: three {:016X} of the isize chunk coordinates."""
    buf = C.create_string_buffer(49)
    _lib.host().rth_chunk_file_name(int(cx), int(cy), int(cz), buf)
    return buf.value.decode()


def write_chunk_file(path, materials, minefield):
    """write_packed_chunk_data (chunk_storage.rs:42-55): LZ4 frame (level 4) of materials u32[64^3] then minefield u8[64^3]."""
    materials = np.ascontiguousarray(materials, dtype=np.uint32).reshape(-1)
    minefield = np.ascontiguousarray(minefield, dtype=np.uint8).reshape(-1)
    assert materials.size == CHUNK_VOLUME and minefield.size == CHUNK_VOLUME
    if _lib.host().rth_chunk_write(str(path).encode(), _p(materials), _p(minefield)) != 0:
        raise IOError("cannot write %s" % path)


def read_chunk_file(path):
    """read_into_packed_chunk_data (chunk_storage.rs:57-68)."""
    mats = np.zeros(CHUNK_VOLUME, dtype=np.uint32)
    mine = np.zeros(CHUNK_VOLUME, dtype=np.uint8)
    if _lib.host().rth_chunk_read(str(path).encode(), _p(mats), _p(mine)) != 0:
        raise IOError("cannot read %s" % path)
    return mats.reshape(64, 64, 64), mine.reshape(64, 64, 64)


class ChunkStorage:
    """world::ChunkStorage (chunk_storage.rs:13-152): generate-on-miss chunk cache with optional LZ4 files on disk."""

    def __init__(self, storage_dir="", seed=DEFAULT_SEED):
        self._h = C.c_void_p(_lib.host().rth_chunk_storage_new(str(storage_dir).encode() if storage_dir else None, C.c_uint64(seed)))

    def borrow_packed_chunk_data(self, cx, cy, cz):
        mats = np.zeros(CHUNK_VOLUME, dtype=np.uint32)
        mine = np.zeros(CHUNK_VOLUME, dtype=np.uint8)
        _lib.host().rth_chunk_storage_borrow(self._h, int(cx), int(cy), int(cz), _p(mats), _p(mine))
        return mats.reshape(64, 64, 64), mine.reshape(64, 64, 64)

    def stats(self):
        g, l = C.c_size_t(), C.c_size_t()
        _lib.host().rth_chunk_storage_stats(self._h, C.byref(g), C.byref(l))
        return {"generated": g.value, "loaded": l.value}

    def close(self):
        if self._h:
            _lib.host().rth_chunk_storage_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostTerrainUploadManager:
    """TerrainUploadManager (terrain_upload.rs:49-368) driving a HOST copy of the toroidal region — what the pipeline
    does on the device through rt_upload_slice, for CPU tests."""

    def __init__(self, seed=DEFAULT_SEED, region=256):
        self.region_size = int(region)
        self._h = C.c_void_p(_lib.host().rth_tum_new_r(C.c_uint64(seed), self.region_size))
        if not self._h:
            raise ValueError("region must be 256, 512 or 1024")

    def request_increase(self, axis):
        _lib.host().rth_tum_request(self._h, int(axis), 1)

    def request_decrease(self, axis):
        _lib.host().rth_tum_request(self._h, int(axis), 0)

    def request_move_towards(self, center):
        _lib.host().rth_tum_move_towards(self._h, (C.c_long * 3)(*[int(c) for c in center]))

    def pending(self):
        return int(_lib.host().rth_tum_pending(self._h))

    def setup_next_request(self):
        rc = _lib.host().rth_tum_step(self._h)
        assert rc == 0

    def get_render_offset(self):
        o = (C.c_long * 3)()
        _lib.host().rth_tum_render_offset(self._h, o)
        return tuple(o[:])

    def region(self):
        R = self.region_size
        mats = np.zeros(R ** 3, dtype=np.uint32)
        mine = np.zeros(R ** 3, dtype=np.uint8)
        _lib.host().rth_tum_region(self._h, _p(mats), _p(mine))
        return mats.reshape(R, R, R), mine.reshape(R, R, R)

    def close(self):
        if self._h:
            _lib.host().rth_tum_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def toroidal_region(render_offset, seed=DEFAULT_SEED, region=256):
    """Expected texture content for a render offset (multiples of 16): texel t on each axis holds the world voxel v with
    (v + R/2) mod R == t inside the window [offset-R/2, offset+R/2) (R = 256 in the reference).  Built from whole chunks,
    independently of the TerrainUploadManager."""
    R, half = int(region), int(region) // 2
    cs = ChunkStorage("", seed)
    mats = np.zeros((R, R, R), dtype=np.uint32)
    mine = np.zeros((R, R, R), dtype=np.uint8)
    lo = [int(o) - half for o in render_offset]
    cr = [range(l // 64, (l + R - 1) // 64 + 1) for l in lo]
    for cz in cr[2]:
        for cy in cr[1]:
            for cx in cr[0]:
                m, f = cs.borrow_packed_chunk_data(cx, cy, cz)
                c0 = (cx * 64, cy * 64, cz * 64)
                s = [slice(max(c0[a], lo[a]) - c0[a], min(c0[a] + 64, lo[a] + R) - c0[a]) for a in range(3)]
                if any(x.stop <= x.start for x in s):
                    continue
                t = [(c0[a] + s[a].start + half) % R for a in range(3)]
                d = [slice(t[a], t[a] + (s[a].stop - s[a].start)) for a in range(3)]
                mats[d[2], d[1], d[0]] = m[s[2], s[1], s[0]]
                mine[d[2], d[1], d[0]] = f[s[2], s[1], s[0]]
    cs.close()
    return mats, mine
