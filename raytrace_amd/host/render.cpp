// render.cpp — see render.hpp.
#include "render.hpp"

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "world.hpp"

namespace rt::render {

TripleEulerVector compute_triple_euler_vector(float heading, float pitch) {
    const float half_pi = 1.57079632679489661923f;  // std::f32::consts::FRAC_PI_2
    TripleEulerVector v;
    v.forward[0] = std::cos(heading) * std::cos(pitch);
    v.forward[1] = std::sin(heading) * std::cos(pitch);
    v.forward[2] = std::sin(pitch);
    v.up[0] = std::cos(heading) * std::cos(pitch + half_pi);
    v.up[1] = std::sin(heading) * std::cos(pitch + half_pi);
    v.up[2] = std::sin(pitch + half_pi);
    // forward.cross(up), cgmath convention
    v.right[0] = v.forward[1] * v.up[2] - v.forward[2] * v.up[1];
    v.right[1] = v.forward[2] * v.up[0] - v.forward[0] * v.up[2];
    v.right[2] = v.forward[0] * v.up[1] - v.forward[1] * v.up[0];
    return v;
}

Pipeline* create_instance(const RtConfig& cfg, const uint8_t* blue_noise_rgba8, game::Game& game, std::string* error) {
    auto fail = [&](const char* what, RtContext* ctx) -> Pipeline* {
        if (error) *error = std::string(what) + ": " + rt_last_error(ctx);
        if (ctx) rt_destroy(ctx);
        return nullptr;
    };
    if (!blue_noise_rgba8) { if (error) *error = "blue noise table is required"; return nullptr; }
    if (!game.has_world()) game.generate_world(0x5EED, cfg.region);
    if (game.world_region() != cfg.region) {   // a world the caller set is never replaced behind their back
        if (error) *error = "create_instance: the game's world has region " + std::to_string(game.world_region()) + " but the config asks for " +
                            std::to_string(cfg.region);
        return nullptr;
    }
    RtContext* ctx = nullptr;
    if (rt_create(&cfg, &ctx) != RT_OK) return fail("rt_create", nullptr);
    if (rt_upload_world(ctx, game.world_materials(), game.world_minefield()) != RT_OK) return fail("rt_upload_world", ctx);
    if (rt_upload_noise(ctx, blue_noise_rgba8) != RT_OK) return fail("rt_upload_noise", ctx);
    Pipeline* p = new Pipeline();
    p->ctx_ = ctx;
    p->spp_ = cfg.spp > 0 ? cfg.spp : 1;
    p->region_ = cfg.region;
    p->tile_world_ = cfg.tile_world;
    std::memset(&p->uniforms_, 0, sizeof(p->uniforms_));
    p->uniforms_.lr[0] = -64; p->uniforms_.lr[1] = -64;    // create_raytrace_uniform_data, render_data.rs:146-147;
    p->uniforms_.lso[0] = -64; p->uniforms_.lso[1] = -64;  // overwritten on the first draw_frame (pipeline.rs:203-207)
    return p;
}

Pipeline::~Pipeline() { rt_destroy(ctx_); }

void Pipeline::enable_terrain_streaming(uint64_t seed, const std::string& storage_dir) {
    chunks_.reset(new world::ChunkStorage(storage_dir, seed));
    tum_.reset(new TerrainUploadManager(region_));
}

int Pipeline::enable_post_passes(bool faithful) {
    if (tile_world_ != 1) return RT_ERR_UNIMPLEMENTED;   // a 48-pixel halo is needed: gather the tiles, then rt_denoise_planes / rt_finalize_planes
    post_ = true;
    post_faithful_ = faithful;
    return RT_OK;
}

const char* Pipeline::last_error() const { return rt_last_error(ctx_); }

int Pipeline::wait() { return rt_sync(ctx_); }

static bool invert3(const float c0[3], const float c1[3], const float c2[3], float out[3][3]) {
    // columns c0,c1,c2 -> inverse, returned as columns out[0..2]
    const float a = c0[0], b = c1[0], c = c2[0], d = c0[1], e = c1[1], f = c2[1], g = c0[2], h = c1[2], i = c2[2];
    const float det = a * (e * i - f * h) - b * (d * i - f * g) + c * (d * h - e * g);
    if (det == 0.0f) return false;
    const float r = 1.0f / det;
    out[0][0] = (e * i - f * h) * r; out[0][1] = (f * g - d * i) * r; out[0][2] = (d * h - e * g) * r;
    out[1][0] = (c * h - b * i) * r; out[1][1] = (a * i - c * g) * r; out[1][2] = (b * g - a * h) * r;
    out[2][0] = (b * f - c * e) * r; out[2][1] = (c * d - a * f) * r; out[2][2] = (a * e - b * d) * r;
    return true;
}

int Pipeline::draw_frame(game::Game& game) {
    int rc = rt_sync(ctx_);                                          // pipeline.rs:162-172
    if (rc != RT_OK) return rc;
    const Camera& camera = game.borrow_camera();
    if (tum_) {                                                      // pipeline.rs:174-189
        const long towards[3] = {(long)camera.origin[0], 0, (long)camera.origin[2]};   // (x, literal 0, z) — :175-179
        tum_->request_move_towards(towards);
        if (tum_->pending() > 0) {   // the slab is assembled in the library's pinned staging (the reference's mapped upload buffers, terrain_upload.rs:65-82)
            uint32_t* sm = nullptr; uint8_t* sf = nullptr;
            rc = rt_slice_staging(ctx_, &sm, &sf);
            if (rc != RT_OK) return rc;
            tum_->bind_upload_buffers(sm, sf);
        }
        rc = tum_->setup_next_request(*chunks_, [this](int axis, int off, const uint32_t* m, const uint8_t* f) {
            return rt_upload_slice(ctx_, axis, off, m, f);
        });
        if (rc != RT_OK) return rc;
        long off[3];
        tum_->get_render_offset(off);
        for (int a = 0; a < 3; a++) render_offset_[a] = (int)off[a];
    }
    TripleEulerVector v = compute_triple_euler_vector(camera.heading, camera.pitch);   // :191-193
    RtUniforms& u = uniforms_;
    for (int a = 0; a < 3; a++) {
        u.origin[a] = camera.origin[a];                              // :196
        u.forward[a] = v.forward[a];                                 // :197
        u.up[a] = v.up[a] * 0.4f;                                    // :198
        u.right[a] = v.right[a] * 0.4f;                              // :199
        u.lr[a] = render_offset_[a];                                 // :203-206
        u.lso[a] = render_offset_[a];
    }
    u.seed = (u.seed + 1) % (uint32_t)RT_NOISE_BYTES;                // :201
    u.sun_angle = game.get_sun_angle();                              // :202
    rc = rt_draw_frame(ctx_, &u);                                    // :209-211; the dispatch recorded at :86-90
    if (rc == RT_OK && post_) {                                      // the same command buffer goes on (:98-123), one submit (:229-235)
        rc = rt_denoise(ctx_, post_faithful_ ? 1 : 0);               // six bilateral_denoise.comp dispatches, sizes 1,2,4,8,8,16
        if (rc == RT_OK) rc = rt_finalize(ctx_);                     // finalize.comp -> the swapchain image
    }
    // spp > 1 consumes seeds seed..seed+spp-1; leave the counter on the last one used.
    u.seed = (u.seed + (uint32_t)(spp_ - 1)) % (uint32_t)RT_NOISE_BYTES;
    // :213-227 — written after the upload so it only affects the next frame; the shader never reads it (Q9).
    std::memcpy(u.old_origin, u.origin, 12);
    float c0[3], c1[3], inv[3][3];
    for (int a = 0; a < 3; a++) { c0[a] = v.right[a] * 0.4f; c1[a] = v.up[a] * 0.4f; }
    if (invert3(c0, c1, v.forward, inv)) {
        std::memcpy(u.old_transform_c0, inv[0], 12);
        std::memcpy(u.old_transform_c1, inv[1], 12);
        std::memcpy(u.old_transform_c2, inv[2], 12);
    }
    return rc;
}

}  // namespace rt::render

namespace rt::game {

Game::Game(int argc, const char* const* argv) {
    if (argc > 6) {  // mod.rs:45-52
        camera.origin[0] = std::strtof(argv[1], nullptr);
        camera.origin[1] = std::strtof(argv[2], nullptr);
        camera.origin[2] = std::strtof(argv[3], nullptr);
        camera.heading = std::strtof(argv[4], nullptr);
        camera.pitch = std::strtof(argv[5], nullptr);
        sun_angle = std::strtof(argv[6], nullptr);
    } else {         // mod.rs:53-55
        camera.origin[0] = -30.0f;
        camera.origin[1] = -128.0f;
        camera.origin[2] = 100.0f;
    }
}

int Game::generate_world(uint64_t seed, int region) {
    if (region != 256 && region != 512 && region != 1024) return RT_ERR_INVALID_ARG;
    const size_t n = (size_t)region * region * region;
    materials_.assign(n, 0);
    minefield_.assign(n, 0);
    world::assemble_region_procedural(seed, materials_.data(), minefield_.data(), region);
    region_ = region;
    return RT_OK;
}

int Game::set_world(const uint32_t* materials, const uint8_t* minefield, int region) {
    if (!materials || !minefield) return RT_ERR_INVALID_ARG;
    if (region != 256 && region != 512 && region != 1024) return RT_ERR_INVALID_ARG;
    const size_t n = (size_t)region * region * region;
    materials_.assign(materials, materials + n);
    minefield_.assign(minefield, minefield + n);
    region_ = region;
    return RT_OK;
}

}  // namespace rt::game
