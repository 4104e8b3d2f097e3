// terrain_upload.hpp — C++ mirror of TerrainUploadManager (src/render/pipeline/terrain_upload.rs:49-368):
// the region texture is toroidal; as the camera moves, 16-thick slabs of the neighbouring region replace the slabs that
// fell behind, one slab per frame, and the render offset (`lr` in the shader) follows.
#pragma once
#include <deque>
#include <functional>
#include <vector>

#include "chunk_storage.hpp"

namespace rt::render {

enum class Axis { X = 0, Y = 1, Z = 2 };

// Position — terrain_upload.rs:21-47: chunk coordinate of the region's minimum corner and how many 16-voxel slabs of the
// NEXT region are already loaded on each axis.
struct Position {
    long origin[3] = {-RT_ROOT_CHUNK_SIZE / 2, -RT_ROOT_CHUNK_SIZE / 2, -RT_ROOT_CHUNK_SIZE / 2};
    int num_loaded_slices[3] = {0, 0, 0};
    int region_chunks = RT_ROOT_CHUNK_SIZE;    // chunks per region edge: 4 in the reference (ROOT_CHUNK_SIZE); R / 64 for the larger regions
    void render_offset(long out[3]) const;     // :29-36
};

// Where a finished slab goes: (axis, texel offset along it, materials, minefield) -> RtStatus.  The pipeline binds this to
// rt_upload_slice; tests bind it to a host-side toroidal array.
using SliceSink = std::function<int(int, int, const uint32_t*, const uint8_t*)>;

class TerrainUploadManager {
 public:
    // region: edge of the toroidal region texture, 256 in the reference (ROOT_BLOCK_SIZE); 512 and 1024 are the build's extension
    explicit TerrainUploadManager(int region = RT_ROOT_BLOCK_SIZE);
    int region() const { return region_; }
    void request_increase(Axis axis);                          // :289-318
    void request_decrease(Axis axis);                          // :320-345
    void request_move_towards(const long desired_center[3]);   // :347-367
    // Consumes at most one queued request (:275-287): builds the slab from the world chunks and hands it to `sink`.
    // Returns RT_OK (also when the queue is empty) or the sink's error.
    int setup_next_request(world::ChunkStorage& chunks, const SliceSink& sink);
    void get_render_offset(long out[3]) const { gpu_position_.render_offset(out); }   // :285-287
    // The reference's upload buffers are host-visible mapped device buffers (:65-82) that upload_slice fills in place; a host
    // that has such memory (rt_slice_staging: pinned) binds it here for the NEXT request, otherwise the slab is built in the
    // manager's own vectors.  Each must hold 16 * R * R elements.
    void bind_upload_buffers(uint32_t* materials, uint8_t* minefield) { ext_materials_ = materials; ext_minefield_ = minefield; }
    size_t pending() const { return queue_.size(); }
    const Position& cpu_position() const { return cpu_position_; }

 private:
    struct Request { long origin[3]; int num_slices[3]; Axis axis; Position new_position; };   // :11-19
    int upload_slice(world::ChunkStorage& chunks, const SliceSink& sink, const Request& request);   // :84-275
    int region_, slices_per_region_, region_chunks_;
    std::deque<Request> queue_;
    Position cpu_position_, gpu_position_;
    std::vector<uint32_t> material_upload_buffer_;   // one slab: 16 x R x R (:65-82)
    std::vector<uint8_t> minefield_upload_buffer_;
    uint32_t* ext_materials_ = nullptr;
    uint8_t* ext_minefield_ = nullptr;
};

}  // namespace rt::render
