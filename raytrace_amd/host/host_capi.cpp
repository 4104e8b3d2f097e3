// host_capi.cpp — plain-C exports of the C++ host mirror (librt_host.so) so the Python tests and bench.py can
// drive it through ctypes.  Nothing here touches the GPU except the rth_pipeline_* functions, which forward to
// the C ABI of include/rt_abi.h.
#include <cstring>
#include <new>

#include <vector>

#include "chunk_storage.hpp"
#include "render.hpp"
#include "terrain_upload.hpp"
#include "world.hpp"

using namespace rt;

extern "C" {

// ---- world -------------------------------------------------------------------------------------------
uint32_t rth_material_pack(int id) {
    if (id < 0 || id >= world::kMaterialCount) return 0;
    return world::MATERIALS[id].pack();
}

void rth_material_unpack(uint32_t packed, uint16_t* albedo3, int* solid) {
    world::Material m = world::Material::unpack(packed);
    albedo3[0] = m.albedo[0]; albedo3[1] = m.albedo[1]; albedo3[2] = m.albedo[2];
    *solid = m.solid ? 1 : 0;
}

int rth_material_get(int id, uint16_t* albedo3, uint16_t* emission3, int* solid) {
    if (id < 0 || id >= world::kMaterialCount) return RT_ERR_INVALID_ARG;
    const world::Material& m = world::MATERIALS[id];
    for (int c = 0; c < 3; c++) { albedo3[c] = m.albedo[c]; emission3[c] = m.emission[c]; }
    *solid = m.solid ? 1 : 0;
    return RT_OK;
}

// UnpackedChunkData::pack_into on one 64^3 chunk of material ids.
int rth_pack_chunk(const uint8_t* ids, uint32_t* materials, uint8_t* minefield) {
    if (!ids || !materials || !minefield) return RT_ERR_INVALID_ARG;
    world::UnpackedChunkData uc;
    std::memcpy(uc.ids.data(), ids, world::kChunkVolume);
    world::PackedChunkData pc;
    uc.pack_into(pc);
    std::memcpy(materials, pc.materials.data(), sizeof(uint32_t) * world::kChunkVolume);
    std::memcpy(minefield, pc.minefield.data(), world::kChunkVolume);
    return RT_OK;
}

int rth_generate_region(uint64_t seed, uint32_t* materials, uint8_t* minefield) {
    if (!materials || !minefield) return RT_ERR_INVALID_ARG;
    world::assemble_region_procedural(seed, materials, minefield);
    return RT_OK;
}

int rth_generate_region_r(uint64_t seed, int region, uint32_t* materials, uint8_t* minefield) {
    if (!materials || !minefield || (region != 256 && region != 512 && region != 1024)) return RT_ERR_INVALID_ARG;
    world::assemble_region_procedural(seed, materials, minefield, region);
    return RT_OK;
}

int rth_region_from_ids(const uint8_t* ids, uint32_t* materials, uint8_t* minefield) {
    if (!ids || !materials || !minefield) return RT_ERR_INVALID_ARG;
    world::assemble_region_from_ids(ids, materials, minefield);
    return RT_OK;
}

int rth_heightmap(long chunk_x, long chunk_y, uint64_t seed, long* out64x64) {
    world::Heightmap hm;
    world::generate_heightmap(hm, chunk_x, chunk_y, seed);
    std::memcpy(out64x64, hm.data.data(), sizeof(long) * hm.data.size());
    return RT_OK;
}

// 3-D copies on u32 arrays (the element type the reference's tests use, util.rs:417-435,496-505,585-603).
int rth_copy_3d_u32(const int* size3, const uint32_t* src, const int* sdims3, const int* soff3, uint32_t* dst,
                    const int* ddims3, const int* doff3) {
    bool ok = world::copy_3d<uint32_t>({size3[0], size3[1], size3[2]}, src, {sdims3[0], sdims3[1], sdims3[2]},
                                       {soff3[0], soff3[1], soff3[2]}, dst, {ddims3[0], ddims3[1], ddims3[2]},
                                       {doff3[0], doff3[1], doff3[2]});
    return ok ? RT_OK : RT_ERR_INVALID_ARG;
}
void rth_copy_3d_auto_clip_u32(const uint32_t* src, int src_stride, const long* off3, uint32_t* dst, int dst_stride) {
    world::copy_3d_auto_clip<uint32_t>(src, src_stride, {off3[0], off3[1], off3[2]}, dst, dst_stride);
}
void rth_copy_3d_bounded_auto_clip_u32(const int* size3, const uint32_t* src, const int* sdims3, const int* soff3,
                                       uint32_t* dst, const int* ddims3, const long* doff3) {
    world::copy_3d_bounded_auto_clip<uint32_t>({size3[0], size3[1], size3[2]}, src, {sdims3[0], sdims3[1], sdims3[2]},
                                               {soff3[0], soff3[1], soff3[2]}, dst, {ddims3[0], ddims3[1], ddims3[2]},
                                               {doff3[0], doff3[1], doff3[2]});
}
void rth_fill_slice_3d_auto_clip_u8(uint8_t value, uint8_t* dst, int dst_stride, const long* start3, const int* size3) {
    world::fill_slice_3d_auto_clip<uint8_t>(value, dst, dst_stride, {start3[0], start3[1], start3[2]},
                                            {size3[0], size3[1], size3[2]});
}

// ---- chunk disk cache (chunk_storage.rs) ------------------------------------------------------------
int rth_chunk_codec_available() { return world::ChunkStorage::codec_available() ? 1 : 0; }
void rth_chunk_file_name(long cx, long cy, long cz, char* out49) {
    std::string n = world::ChunkStorage::file_name(cx, cy, cz);
    std::strncpy(out49, n.c_str(), 49);
}
int rth_chunk_write(const char* path, const uint32_t* materials, const uint8_t* minefield) {
    world::PackedChunkData pc;
    std::memcpy(pc.materials.data(), materials, sizeof(uint32_t) * world::kChunkVolume);
    std::memcpy(pc.minefield.data(), minefield, world::kChunkVolume);
    return world::ChunkStorage::write_packed_chunk_data(path, pc) ? RT_OK : RT_ERR_INVALID_ARG;
}
int rth_chunk_read(const char* path, uint32_t* materials, uint8_t* minefield) {
    world::PackedChunkData pc;
    if (!world::ChunkStorage::read_into_packed_chunk_data(path, pc)) return RT_ERR_INVALID_ARG;
    std::memcpy(materials, pc.materials.data(), sizeof(uint32_t) * world::kChunkVolume);
    std::memcpy(minefield, pc.minefield.data(), world::kChunkVolume);
    return RT_OK;
}
void* rth_chunk_storage_new(const char* dir, uint64_t seed) { return new (std::nothrow) world::ChunkStorage(dir ? dir : "", seed); }
void rth_chunk_storage_free(void* s) { delete static_cast<world::ChunkStorage*>(s); }
int rth_chunk_storage_borrow(void* s, long cx, long cy, long cz, uint32_t* materials, uint8_t* minefield) {
    const world::PackedChunkData& pc = static_cast<world::ChunkStorage*>(s)->borrow_packed_chunk_data(cx, cy, cz);
    std::memcpy(materials, pc.materials.data(), sizeof(uint32_t) * world::kChunkVolume);
    std::memcpy(minefield, pc.minefield.data(), world::kChunkVolume);
    return RT_OK;
}
void rth_chunk_storage_stats(void* s, size_t* generated, size_t* loaded) {
    *generated = static_cast<world::ChunkStorage*>(s)->generated();
    *loaded = static_cast<world::ChunkStorage*>(s)->loaded();
}

// ---- TerrainUploadManager against a host-side toroidal region (CPU tests) -----------------------------
struct HostTum {
    render::TerrainUploadManager tum;
    world::ChunkStorage chunks;
    std::vector<uint32_t> materials;
    std::vector<uint8_t> minefield;
    int region;
    HostTum(uint64_t seed, int r) : tum(r), chunks("", seed), materials((size_t)r * r * r), minefield((size_t)r * r * r), region(r) {
        world::assemble_region_procedural(seed, materials.data(), minefield.data(), r);
    }
};
void* rth_tum_new(uint64_t seed) { return new (std::nothrow) HostTum(seed, world::kRegion); }
void* rth_tum_new_r(uint64_t seed, int region) {
    if (region != 256 && region != 512 && region != 1024) return nullptr;
    return new (std::nothrow) HostTum(seed, region);
}
void rth_tum_free(void* t) { delete static_cast<HostTum*>(t); }
void rth_tum_request(void* t, int axis, int increase) {
    auto* h = static_cast<HostTum*>(t);
    if (increase) h->tum.request_increase((render::Axis)axis); else h->tum.request_decrease((render::Axis)axis);
}
void rth_tum_move_towards(void* t, const long* center3) { static_cast<HostTum*>(t)->tum.request_move_towards(center3); }
int rth_tum_pending(void* t) { return (int)static_cast<HostTum*>(t)->tum.pending(); }
// Consumes one request; the slab is applied to the host region exactly like rt_upload_slice applies it on the device.
int rth_tum_step(void* t) {
    auto* h = static_cast<HostTum*>(t);
    return h->tum.setup_next_request(h->chunks, [h](int axis, int off, const uint32_t* m, const uint8_t* f) {
        const int R = h->region, S = RT_SLICE_SIZE;
        world::Dims3 shape{R, R, R};
        (axis == 0 ? shape.x : (axis == 1 ? shape.y : shape.z)) = S;
        world::Dims3 at{0, 0, 0};
        (axis == 0 ? at.x : (axis == 1 ? at.y : at.z)) = off;
        world::copy_3d(shape, m, shape, {0, 0, 0}, h->materials.data(), {R, R, R}, at);
        world::copy_3d(shape, f, shape, {0, 0, 0}, h->minefield.data(), {R, R, R}, at);
        return (int)RT_OK;
    });
}
void rth_tum_render_offset(void* t, long* out3) { static_cast<HostTum*>(t)->tum.get_render_offset(out3); }
void rth_tum_region(void* t, uint32_t* materials, uint8_t* minefield) {
    auto* h = static_cast<HostTum*>(t);
    std::memcpy(materials, h->materials.data(), sizeof(uint32_t) * h->materials.size());
    std::memcpy(minefield, h->minefield.data(), h->minefield.size());
}

// ---- camera / uniforms ------------------------------------------------------------------------------
void rth_compute_triple_euler_vector(float heading, float pitch, float* forward3, float* up3, float* right3) {
    render::TripleEulerVector v = render::compute_triple_euler_vector(heading, pitch);
    std::memcpy(forward3, v.forward, 12); std::memcpy(up3, v.up, 12); std::memcpy(right3, v.right, 12);
}

// ---- Game + Pipeline ---------------------------------------------------------------------------------
void* rth_game_new(int argc, const char* const* argv) { return new (std::nothrow) game::Game(argc, argv); }
void rth_game_free(void* g) { delete static_cast<game::Game*>(g); }
void rth_game_set_camera(void* g, const float* origin3, float heading, float pitch) {
    auto* gm = static_cast<game::Game*>(g);
    std::memcpy(gm->camera.origin, origin3, 12);
    gm->camera.heading = heading;
    gm->camera.pitch = pitch;
}
void rth_game_get_camera(void* g, float* origin3, float* heading, float* pitch) {
    auto* gm = static_cast<game::Game*>(g);
    std::memcpy(origin3, gm->camera.origin, 12);
    *heading = gm->camera.heading;
    *pitch = gm->camera.pitch;
}
void rth_game_set_sun_angle(void* g, float a) { static_cast<game::Game*>(g)->sun_angle = a; }
float rth_game_get_sun_angle(void* g) { return static_cast<game::Game*>(g)->get_sun_angle(); }
// Replace the game's world with caller-provided region arrays (tests) instead of the procedural one.
int rth_game_set_world(void* g, const uint32_t* materials, const uint8_t* minefield) {
    return static_cast<game::Game*>(g)->set_world(materials, minefield);
}
int rth_game_set_world_r(void* g, const uint32_t* materials, const uint8_t* minefield, int region) {
    return static_cast<game::Game*>(g)->set_world(materials, minefield, region);
}
int rth_game_generate_world(void* g, uint64_t seed) { return static_cast<game::Game*>(g)->generate_world(seed); }
int rth_game_generate_world_r(void* g, uint64_t seed, int region) { return static_cast<game::Game*>(g)->generate_world(seed, region); }

void* rth_create_instance(const RtConfig* cfg, const uint8_t* blue_noise_rgba8, void* g, char* err, size_t err_len) {
    std::string msg;
    render::Pipeline* p = render::create_instance(*cfg, blue_noise_rgba8, *static_cast<game::Game*>(g), &msg);
    if (!p && err && err_len) { std::strncpy(err, msg.c_str(), err_len - 1); err[err_len - 1] = 0; }
    return p;
}
void rth_pipeline_free(void* p) { delete static_cast<render::Pipeline*>(p); }
int rth_pipeline_draw_frame(void* p, void* g) {
    return static_cast<render::Pipeline*>(p)->draw_frame(*static_cast<game::Game*>(g));
}
int rth_pipeline_wait(void* p) { return static_cast<render::Pipeline*>(p)->wait(); }
RtContext* rth_pipeline_context(void* p) { return static_cast<render::Pipeline*>(p)->context(); }
void rth_pipeline_uniforms(void* p, RtUniforms* out) { *out = static_cast<render::Pipeline*>(p)->uniforms(); }
void rth_pipeline_set_seed(void* p, uint32_t seed) { static_cast<render::Pipeline*>(p)->set_seed(seed); }
void rth_pipeline_enable_streaming(void* p, uint64_t seed, const char* dir) {
    static_cast<render::Pipeline*>(p)->enable_terrain_streaming(seed, dir ? dir : "");
}
int rth_pipeline_enable_post_passes(void* p, int faithful) { return static_cast<render::Pipeline*>(p)->enable_post_passes(faithful != 0); }
const char* rth_pipeline_last_error(void* p) { return static_cast<render::Pipeline*>(p)->last_error(); }

}  // extern "C"
