// render.hpp — C++ host mirror of the reference's `render` public API for the ray-trace path:
// Camera (src/render/mod.rs:20-34), create_instance (mod.rs:36-43), Pipeline::{new, draw_frame, drop}
// (src/render/pipeline/pipeline.rs:36-76,134-255,258-277), and the Game state it reads
// (src/game/mod.rs:14-58,103-125).  Everything GPU-side goes through the C ABI in include/rt_abi.h.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "terrain_upload.hpp"

namespace rt::render {

// Positive Y (angle PI/2) is forward, positive X is right, positive Z is up (mod.rs:14-18).
struct Camera {
    float origin[3] = {0.0f, 0.0f, 0.0f};
    float heading = 3.14159265358979323846f * 0.5f;  // Camera::new, mod.rs:27-33
    float pitch = 0.0f;
};

struct TripleEulerVector { float forward[3], up[3], right[3]; };
// src/util.rs:9-22
TripleEulerVector compute_triple_euler_vector(float heading, float pitch);

}  // namespace rt::render

namespace rt::game {

// Game — src/game/mod.rs:14-58.  Interactive controls (control.rs, tick) are out of scope; the world is the
// flattened 256^3 region rather than a ChunkStorage (disk cache is out of scope).
class Game {
 public:
    // Game::new (mod.rs:37-58): six optional positional floats `x y z heading pitch sun_angle`
    // (argv[1..6]); otherwise the default pose (-30, -128, 100), heading PI/2, pitch 0, sun 0.
    Game(int argc, const char* const* argv);
    const render::Camera& borrow_camera() const { return camera; }   // mod.rs:111-113
    float get_sun_angle() const { return sun_angle; }                // mod.rs:123-125
    int generate_world(uint64_t seed, int region = RT_ROOT_BLOCK_SIZE);   // region: 256 = the reference; 512 / 1024 = extension
    int world_region() const { return region_; }
    int set_world(const uint32_t* materials, const uint8_t* minefield, int region = RT_ROOT_BLOCK_SIZE);   // region^3 voxels each
    bool has_world() const { return !materials_.empty(); }
    const uint32_t* world_materials() const { return materials_.data(); }
    const uint8_t* world_minefield() const { return minefield_.data(); }

    render::Camera camera;
    float sun_angle = 0.0f;

 private:
    std::vector<uint32_t> materials_;
    std::vector<uint8_t> minefield_;
    int region_ = RT_ROOT_BLOCK_SIZE;
};

}  // namespace rt::game

namespace rt::render {

class Pipeline {
 public:
    ~Pipeline();                       // impl Drop for Pipeline, pipeline.rs:258-277
    // pipeline.rs:134-255: wait for the previous frame, derive the 192-byte uniform block from the camera,
    // submit the ray-trace dispatch.  Returns an RtStatus instead of panicking.
    int draw_frame(game::Game& game);
    int wait();                        // the fence wait at pipeline.rs:162-172, callable on its own
    RtContext* context() const { return ctx_; }
    const RtUniforms& uniforms() const { return uniforms_; }
    void set_seed(uint32_t seed) { uniforms_.seed = seed; }
    const char* last_error() const;
    // Terrain streaming (pipeline.rs:174-189): when enabled, every draw_frame asks the TerrainUploadManager to move towards
    // the camera and uploads at most one slab; the render offset becomes the uniform block's `lr`.  Off by default
    // (static region).  `storage_dir` empty = no disk cache.
    void enable_terrain_streaming(uint64_t seed, const std::string& storage_dir);
    TerrainUploadManager* terrain_upload_manager() { return tum_.get(); }
    // The rest of the reference's per-frame command buffer (pipeline.rs:98-123): after the ray-trace dispatch, the six
    // bilateral_denoise.comp dispatches and finalize.comp, submitted together (:229-235).  When enabled, draw_frame enqueues
    // ray trace -> rt_denoise -> rt_finalize on the context's stream and RT_BUF_FINAL_BGRA8 holds the swapchain image of the
    // frame.  `faithful` = the reference's pong descriptor set with its swapped bindings (descriptor_sets.rs:38-39).
    // Off by default (the G-buffer planes are then the ray-trace dispatch's own output, which the parity tests compare);
    // whole-frame contexts only (RT_ERR_UNIMPLEMENTED on a tile-split one: the passes need a halo, gather first).
    int enable_post_passes(bool faithful);
    bool post_passes() const { return post_; }

 private:
    friend Pipeline* create_instance(const RtConfig&, const uint8_t*, game::Game&, std::string*);
    Pipeline() = default;
    RtContext* ctx_ = nullptr;
    RtUniforms uniforms_{};            // RenderData::raytrace_uniform_data, render_data.rs:134-162
    int spp_ = 1;
    int region_ = RT_ROOT_BLOCK_SIZE;
    int render_offset_[3] = {0, 0, 0}; // TerrainUploadManager::get_render_offset (terrain_upload.rs:30-47)
    int tile_world_ = 1;
    bool post_ = false, post_faithful_ = true;
    std::unique_ptr<TerrainUploadManager> tum_;
    std::unique_ptr<world::ChunkStorage> chunks_;
};

// render::create_instance (mod.rs:36-43) -> Pipeline::new (pipeline.rs:36-76): creates the device context, uploads
// the game's world (RenderData::initialize, render_data.rs:269-301) and the blue-noise table
// (render_data.rs:110-133).  Returns nullptr and fills *error on failure.
Pipeline* create_instance(const RtConfig& cfg, const uint8_t* blue_noise_rgba8, game::Game& game, std::string* error);

}  // namespace rt::render
