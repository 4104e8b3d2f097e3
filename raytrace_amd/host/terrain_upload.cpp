// terrain_upload.cpp — see terrain_upload.hpp.
#include "terrain_upload.hpp"

namespace rt::render {

namespace {
constexpr int kSlice = RT_SLICE_SIZE;                          // 16
constexpr int kChunk = RT_CHUNK_SIZE;                          // 64
inline long floor_div(long a, long b) { long q = a / b; return (a % b != 0 && ((a < 0) != (b < 0))) ? q - 1 : q; }
}  // namespace

void Position::render_offset(long out[3]) const {
    for (int a = 0; a < 3; a++) out[a] = (origin[a] + region_chunks / 2) * kChunk + (long)num_loaded_slices[a] * kSlice;
}

TerrainUploadManager::TerrainUploadManager(int region)
    : region_(region), slices_per_region_(region / kSlice), region_chunks_(region / kChunk),
      material_upload_buffer_((size_t)kSlice * region * region), minefield_upload_buffer_((size_t)kSlice * region * region) {
    for (Position* p : {&cpu_position_, &gpu_position_}) {
        p->region_chunks = region_chunks_;
        for (int a = 0; a < 3; a++) p->origin[a] = -region_chunks_ / 2;     // Position::default, :39-47
    }
}

void TerrainUploadManager::request_increase(Axis axis) {
    // the slab at index num_loaded_slices is replaced by the same slab of the NEXT region (origin + 4 chunks), then the count grows
    const int m = (int)axis;
    Request r{};
    for (int a = 0; a < 3; a++) { r.origin[a] = cpu_position_.origin[a]; r.num_slices[a] = cpu_position_.num_loaded_slices[a]; }
    r.origin[m] += region_chunks_;
    r.axis = axis;
    cpu_position_.num_loaded_slices[m] += 1;
    if (cpu_position_.num_loaded_slices[m] == slices_per_region_) {
        cpu_position_.num_loaded_slices[m] = 0;
        cpu_position_.origin[m] += region_chunks_;
    }
    r.new_position = cpu_position_;
    queue_.push_back(r);
}

void TerrainUploadManager::request_decrease(Axis axis) {
    // step the count back first, then reload that slab from the CURRENT region
    const int m = (int)axis;
    if (cpu_position_.num_loaded_slices[m] == 0) {
        cpu_position_.num_loaded_slices[m] = slices_per_region_;
        cpu_position_.origin[m] -= region_chunks_;
    }
    cpu_position_.num_loaded_slices[m] -= 1;
    Request r{};
    for (int a = 0; a < 3; a++) { r.origin[a] = cpu_position_.origin[a]; r.num_slices[a] = cpu_position_.num_loaded_slices[a]; }
    r.axis = axis;
    r.new_position = cpu_position_;
    queue_.push_back(r);
}

void TerrainUploadManager::request_move_towards(const long desired_center[3]) {
    long off[3];
    cpu_position_.render_offset(off);
    for (int a = 0; a < 3; a++) {       // first axis that is more than one slab away wins (x, then y, then z)
        const long delta = desired_center[a] - off[a];
        if (delta > kSlice) { request_increase((Axis)a); return; }
        if (-delta > kSlice) { request_decrease((Axis)a); return; }
    }
}

int TerrainUploadManager::setup_next_request(world::ChunkStorage& chunks, const SliceSink& sink) {
    if (queue_.empty()) return RT_OK;
    Request r = queue_.front();
    queue_.pop_front();
    return upload_slice(chunks, sink, r);
}

// The slab holds, on the main axis m, the 16 world voxels starting at request.origin[m]*64 + num_slices[m]*16, and on
// each other axis a the R-voxel window starting at request.origin[a]*64 + num_slices[a]*16, every voxel stored at texel
// (voxel - origin*64) mod R — the texture's toroidal addressing.  Chunk pieces never straddle the wrap because R is a
// multiple of the chunk size.
int TerrainUploadManager::upload_slice(world::ChunkStorage& chunks, const SliceSink& sink, const Request& rq) {
    const int m = (int)rq.axis;
    const int kRegion = region_;
    world::Dims3 shape{kRegion, kRegion, kRegion};
    (m == 0 ? shape.x : (m == 1 ? shape.y : shape.z)) = kSlice;
    const world::Dims3 cdims{kChunk, kChunk, kChunk};
    long win[3];
    for (int a = 0; a < 3; a++) win[a] = rq.origin[a] * kChunk + (long)rq.num_slices[a] * kSlice;
    uint32_t* const mat_buf = ext_materials_ ? ext_materials_ : material_upload_buffer_.data();
    uint8_t* const mine_buf = ext_minefield_ ? ext_minefield_ : minefield_upload_buffer_.data();
    // chunk ranges per axis
    long c0[3], c1[3];
    for (int a = 0; a < 3; a++) {
        const long len = a == m ? kSlice : kRegion;
        c0[a] = floor_div(win[a], kChunk);
        c1[a] = floor_div(win[a] + len - 1, kChunk);
    }
    for (long cz = c0[2]; cz <= c1[2]; cz++)
        for (long cy = c0[1]; cy <= c1[1]; cy++)
            for (long cx = c0[0]; cx <= c1[0]; cx++) {
                const long cc[3] = {cx, cy, cz};
                int src[3], dst[3], size[3];
                for (int a = 0; a < 3; a++) {
                    const long len = a == m ? kSlice : kRegion;
                    const long lo = std::max(cc[a] * kChunk, win[a]), hi = std::min(cc[a] * kChunk + kChunk, win[a] + len);
                    size[a] = (int)(hi - lo);
                    src[a] = (int)(lo - cc[a] * kChunk);
                    // texel inside the slab: main axis counts from the slab start; other axes use the toroidal address
                    dst[a] = a == m ? (int)(lo - win[a]) : (int)(((lo - rq.origin[a] * kChunk) % kRegion + kRegion) % kRegion);
                }
                if (size[0] <= 0 || size[1] <= 0 || size[2] <= 0) continue;
                const world::PackedChunkData& pc = chunks.borrow_packed_chunk_data(cx, cy, cz);
                world::copy_3d({size[0], size[1], size[2]}, pc.materials.data(), cdims, {src[0], src[1], src[2]},
                               mat_buf, shape, {dst[0], dst[1], dst[2]});
                world::copy_3d({size[0], size[1], size[2]}, pc.minefield.data(), cdims, {src[0], src[1], src[2]},
                               mine_buf, shape, {dst[0], dst[1], dst[2]});
            }
    const int axis_offset = rq.num_slices[m] * kSlice;          // :224-230
    int rc = sink(m, axis_offset, mat_buf, mine_buf);
    if (rc == RT_OK) gpu_position_ = rq.new_position;           // :273
    return rc;
}

}  // namespace rt::render
