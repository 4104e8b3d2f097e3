// world.hpp — host-side scene flattening for the ray-trace path (C++ mirror of the reference's
// src/world + the material table of src/render/GEN_MATERIALS.rs).
//
// Produces exactly the two arrays the traversal kernel consumes — packed materials u32[256^3] and the
// "minefield" u8[256^3], x-fastest, texel = world + 128 — from voxel material ids.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../include/rt_abi.h"

namespace rt::world {

constexpr int kChunk = RT_CHUNK_SIZE;                       // constants.rs:23
constexpr int kChunkVolume = kChunk * kChunk * kChunk;      // constants.rs:24
constexpr int kRegion = RT_ROOT_BLOCK_SIZE;                 // constants.rs:27
constexpr size_t kRegionVolume = (size_t)kRegion * kRegion * kRegion;
constexpr int kRegionChunks = RT_ROOT_CHUNK_SIZE;           // constants.rs:26

// Material — src/render/GEN_MATERIALS.rs:2-7.  Albedo channels are 0..127 (build.rs:205-210 halves the CSV).
struct Material {
    uint16_t albedo[3];
    uint16_t emission[3];
    bool solid;
    // Material::pack — GEN_MATERIALS.rs:44-51.  The solid flag (bit 15) overlaps bit 1 of red (quirk Q10).
    uint32_t pack() const;
    // Material::unpack — GEN_MATERIALS.rs:53-68.
    static Material unpack(uint32_t packed);
};

// MATERIALS — GEN_MATERIALS.rs:70-106 (generated from misc/materials.csv:1-8).
constexpr int kMaterialCount = 7;
extern const Material MATERIALS[kMaterialCount];

// PackedChunkData — src/world/chunk.rs:53-57
struct PackedChunkData {
    std::vector<uint8_t> minefield;
    std::vector<uint32_t> materials;
    PackedChunkData() : minefield(kChunkVolume, 0), materials(kChunkVolume, 0) {}
};

// UnpackedChunkData — src/world/chunk.rs:104-123; voxels are stored as ids into MATERIALS.
struct UnpackedChunkData {
    std::vector<uint8_t> ids;
    UnpackedChunkData() : ids(kChunkVolume, 0) {}
    void set_block(int x, int y, int z, uint8_t id) { ids[((size_t)z * kChunk + y) * kChunk + x] = id; }
    void fill(uint8_t id) { ids.assign(kChunkVolume, id); }
    // pack_into — src/world/chunk.rs:125-184: per voxel 0 if solid, else the smallest L in 1..6 whose aligned
    // 2^L cube holds a solid voxel; an all-empty chunk becomes minefield = 6, materials = 0.
    void pack_into(PackedChunkData& out) const;
};

// Heightmap — src/world/heightmap.rs:4-17
struct Heightmap {
    std::vector<long> data;
    Heightmap() : data((size_t)kChunk * kChunk, 0) {}
    long get(int x, int y) const { return data[(size_t)y * kChunk + x]; }
};

// Deterministic stand-in for src/world/generate.rs + functions.rs.  The reference's terrain depends on the
// un-vendored `noise 0.6.0` crate and on rand::thread_rng() (OS-seeded), so it is not reproducible even by the
// reference itself; this generator keeps the SHAPE (fBm -> slope erosion -> pow 2.6 -> height = n*120+10;
// z-banded materials 2/5/6 with dither) with its own hash noise and a counter-based RNG keyed by `seed`.
// PARITY UNPINNED for terrain content — voxel arrays are the parity input of the render path.
void generate_heightmap(Heightmap& out, long chunk_x, long chunk_y, uint64_t seed);             // generate.rs:19-33
void generate_chunk(UnpackedChunkData& out, long cx, long cy, long cz, const Heightmap& hm,
                    uint64_t seed);                                                              // generate.rs:53-85
double mountain_noise2(double x, double y, uint64_t seed);                                       // functions.rs:69-99

// Region assembly — RenderData::make_world_upload_buffers (render_data.rs:203-249): region chunk c in [0,4)^3 holds
// world chunk c-2 and lands at texel offset c*64 (copy_materials/copy_minefield, chunk.rs:66-94).
// `region` = edge R of the target arrays: 256 in the reference (ROOT_BLOCK_SIZE); 512 / 1024 are the build's extension
// (R/64 chunks per axis, centred on the origin like the reference's 4).
void assemble_region_procedural(uint64_t seed, uint32_t* materials, uint8_t* minefield, int region = kRegion);
// Same assembly for caller-provided voxel ids (u8[256^3], texel space, x fastest).
void assemble_region_from_ids(const uint8_t* ids, uint32_t* materials, uint8_t* minefield);

// ---- 3-D array copies (src/util.rs:375-668).  X is the fastest axis (util.rs:104-106). ----
struct Dims3 { int x, y, z; };
struct Off3 { long x, y, z; };
// copy_3d — util.rs:380-415: copy a `size` block from `src` (dims sd) at so to `dst` (dims dd) at dof.
// Returns false (and copies nothing) where the reference would panic on its bounds asserts (util.rs:391-395).
template <typename T>
bool copy_3d(Dims3 size, const T* src, Dims3 sd, Dims3 so, T* dst, Dims3 dd, Dims3 dof);
// copy_3d_auto_clip — util.rs:440-494: cubic source placed at signed `offset` inside a cubic target; copies the overlap.
template <typename T>
void copy_3d_auto_clip(const T* src, int src_stride, Off3 offset, T* dst, int dst_stride);
// copy_3d_bounded_auto_clip — util.rs:507-583: `size` block starting at so lands at signed dof; clipped to both arrays.
template <typename T>
void copy_3d_bounded_auto_clip(Dims3 size, const T* src, Dims3 sd, Dims3 so, T* dst, Dims3 dd, Off3 dof);
// fill_slice_3d_auto_clip — util.rs:636-668
template <typename T>
void fill_slice_3d_auto_clip(T value, T* dst, int dst_stride, Off3 start, Dims3 size);

}  // namespace rt::world
