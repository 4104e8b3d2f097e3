"""the chunk disk format (src/world/chunk_storage.rs) and the TerrainUploadManager
(src/render/pipeline/terrain_upload.rs), C++ mirrors under raytrace_amd/host/."""
import os

import numpy as np
import pytest

from raytrace_amd import abi, render, world
from oracle import pyoracle as po

pytestmark = pytest.mark.usefixtures("native_built")


def test_chunk_file_names():
    # chunk_storage.rs:37-40 — {:016X} of isize: two's complement for negative coordinates
    assert world.chunk_file_name(0, 0, 0) == "0" * 48
    assert world.chunk_file_name(1, 2, 255) == "0000000000000001" "0000000000000002" "00000000000000FF"
    assert world.chunk_file_name(-1, -2, 3) == "FFFFFFFFFFFFFFFF" "FFFFFFFFFFFFFFFE" "0000000000000003"


def test_chunk_file_round_trip_and_frame_format(tmp_path):
    assert world.chunk_codec_available()
    rng = np.random.default_rng(0)
    ids = np.where(rng.random((64, 64, 64)) < 0.2, rng.integers(1, 7, size=(64, 64, 64)), 0).astype(np.uint8)
    mats, mine = world.pack_chunk(ids)
    path = tmp_path / world.chunk_file_name(3, -4, 0)
    world.write_chunk_file(path, mats, mine)
    raw = open(path, "rb").read()
    assert raw[:4] == bytes([0x04, 0x22, 0x4D, 0x18])            # LZ4 frame magic number (little-endian 0x184D2204)
    assert len(raw) < 64 ** 3 * 5                                  # compressed payload: 1 MiB materials + 256 KiB minefield
    m2, f2 = world.read_chunk_file(path)
    assert np.array_equal(m2, mats) and np.array_equal(f2, mine)
    # a truncated file is a read error (read_exact, chunk_storage.rs:63-66)
    open(path, "wb").write(raw[: len(raw) // 2])
    with pytest.raises(IOError):
        world.read_chunk_file(path)


def test_chunk_storage_generates_on_miss_and_loads_on_hit(tmp_path):
    a = world.ChunkStorage(tmp_path, seed=0x5EED)
    m1, f1 = a.borrow_packed_chunk_data(0, -1, 0)
    assert a.stats() == {"generated": 1, "loaded": 0}
    assert os.path.exists(tmp_path / world.chunk_file_name(0, -1, 0))
    a.close()
    b = world.ChunkStorage(tmp_path, seed=12345)          # different seed: must come from disk, not from the generator
    m2, f2 = b.borrow_packed_chunk_data(0, -1, 0)
    assert b.stats() == {"generated": 0, "loaded": 1}
    assert np.array_equal(m1, m2) and np.array_equal(f1, f2)
    # a corrupt file falls back to generation (chunk_storage.rs:131-138)
    open(tmp_path / world.chunk_file_name(5, 5, 0), "wb").write(b"not an lz4 frame")
    b.borrow_packed_chunk_data(5, 5, 0)
    assert b.stats()["generated"] == 1
    b.close()
    # chunk (c) of the region assembly is world chunk (c - 2): the same data the storage hands out
    mats, mine = world.generate_region(0x5EED)
    c = world.ChunkStorage("", seed=0x5EED)
    m, f = c.borrow_packed_chunk_data(0, -1, 0)
    assert np.array_equal(mats[128:192, 64:128, 128:192], m) and np.array_equal(mine[128:192, 64:128, 128:192], f)
    c.close()


def test_terrain_upload_manager_positions_and_requests():
    t = world.HostTerrainUploadManager()
    assert t.get_render_offset() == (0, 0, 0)                       # Position::default, terrain_upload.rs:39-47
    t.request_move_towards((10, 0, 0))                              # within one slab: nothing to do (:347-367)
    assert t.pending() == 0
    t.request_move_towards((100, 0, 40))                            # x first, one request per call
    assert t.pending() == 1
    t.setup_next_request()
    assert t.get_render_offset() == (16, 0, 0)
    t.request_move_towards((16, 0, 40))                             # x satisfied -> y (0) satisfied -> z
    t.setup_next_request()
    assert t.get_render_offset() == (16, 0, 16)
    t.request_move_towards((16, 0, -40))
    t.setup_next_request()
    assert t.get_render_offset() == (16, 0, 0)
    t.close()


@pytest.mark.parametrize("moves", [
    [(0, 1)] * 3,                                   # three slabs along +x
    [(1, 0)] * 2,                                   # two slabs along -y (wraps the slab count below zero)
    [(0, 1)] * 17,                                  # more than a whole region along +x: origin advances by 4 chunks
    [(0, 1), (2, 1), (0, 1), (1, 0), (2, 0), (0, 0)],   # mixed axes and directions
])
def test_terrain_upload_manager_region_content(moves):
    """After any sequence of slab requests the toroidal region equals the world voxels of the shifted window."""
    t = world.HostTerrainUploadManager()
    for axis, inc in moves:
        (t.request_increase if inc else t.request_decrease)(axis)
    while t.pending():
        t.setup_next_request()
    off = t.get_render_offset()
    exp = [16 * (sum(1 for a, i in moves if a == ax and i) - sum(1 for a, i in moves if a == ax and not i)) for ax in range(3)]
    assert list(off) == exp
    mats, mine = t.region()
    emats, emine = world.toroidal_region(off)
    assert np.array_equal(mats, emats) and np.array_equal(mine, emine)
    t.close()


def test_terrain_upload_manager_region_512():
    """The same manager on the 512^3 extension (8 chunks, 32 slabs per region edge): slab requests across the +x wrap and on
    -z, content against the independently assembled toroidal window."""
    t = world.HostTerrainUploadManager(region=512)
    assert t.get_render_offset() == (0, 0, 0)
    moves = [(0, 1)] * 2 + [(2, 0)]
    for axis, inc in moves:
        (t.request_increase if inc else t.request_decrease)(axis)
    while t.pending():
        t.setup_next_request()
    off = t.get_render_offset()
    assert off == (32, 0, -16)
    mats, mine = t.region()
    emats, emine = world.toroidal_region(off, region=512)
    assert np.array_equal(mats, emats) and np.array_equal(mine, emine)
    t.close()


@pytest.mark.gpu
def test_streaming_pipeline_region_512(blue_noise):
    """Pipeline::draw_frame with terrain streaming on a 512^3 region: three frames, one slab each through the incremental
    rt_upload_slice, every frame against the oracle fed the expected toroidal window."""
    g = render.Game(args=(120, -200, 140, 1.6, -0.2, 0.3))
    g.generate_world(world.DEFAULT_SEED, region=512)
    cfg = render.make_config(64, 48, spp=1, depth=2, region=512)
    p = render.create_instance(cfg, g, blue_noise)
    p.enable_terrain_streaming(world.DEFAULT_SEED)
    seen = []
    for frame in range(3):
        p.draw_frame(g)
        p.wait()
        u = p.uniforms()
        seen.append(tuple(u.lr))
        mats, mine = world.toroidal_region(tuple(u.lr), region=512)
        cpu, _ = po.render(mats, mine, blue_noise, u, 64, 48, 1, 2, region=512)
        gpu = p.context.readback_all()
        for name in cpu:
            assert np.array_equal(gpu[name], cpu[name], equal_nan=True), (frame, name)
    assert seen == [(16, 0, 0), (32, 0, 0), (48, 0, 0)]
    p.close()
    g.close()


@pytest.mark.gpu
def test_streaming_pipeline_follows_the_camera(blue_noise):
    """Pipeline::draw_frame with the TerrainUploadManager (pipeline.rs:174-207): the camera starts 60 voxels from the
    region centre on x; each frame uploads one slab through rt_upload_slice and `lr` follows; every frame equals the
    oracle fed the expected toroidal region."""
    g = render.Game(args=(60, -100, 70, 1.6, -0.2, 0.3))
    g.generate_world(world.DEFAULT_SEED)
    cfg = render.make_config(64, 48, spp=1, depth=2)
    p = render.create_instance(cfg, g, blue_noise)
    p.enable_terrain_streaming(world.DEFAULT_SEED)
    seen = []
    for frame in range(5):
        p.draw_frame(g)
        p.wait()
        u = p.uniforms()
        seen.append(tuple(u.lr))
        mats, mine = world.toroidal_region(tuple(u.lr))
        cpu, _ = po.render(mats, mine, blue_noise, u, 64, 48, 1, 2)
        gpu = p.context.readback_all()
        for name in cpu:
            assert np.array_equal(gpu[name], cpu[name], equal_nan=True), (frame, name)
    # x advances while the camera (x = 60) is more than one slab (16) ahead: 0 -> 16 -> 32 -> 48, then z towards 70
    assert seen == [(16, 0, 0), (32, 0, 0), (48, 0, 0), (48, 0, 16), (48, 0, 32)]
    p.close()
    g.close()
