// chunk_storage.hpp — C++ mirror of src/world/chunk_storage.rs: the on-disk chunk cache.
//
// File format (chunk_storage.rs:42-68): one LZ4 *frame* (lz4 crate 1.23.1 -> liblz4 frame API, compression level 4)
// whose payload is materials: [u32 LE; 64^3] (1 MiB) followed by minefield: [u8; 64^3] (256 KiB).  File name
// (chunk_storage.rs:37-40): three `{:016X}` of the isize chunk coordinates (two's complement for negatives).
// The frame codec is the system's liblz4.so.1 (loaded at run time; the image ships no lz4 headers, so the few
// prototypes used are declared in chunk_storage.cpp).
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <tuple>

#include "world.hpp"

namespace rt::world {

class ChunkStorage {
 public:
    // `storage_dir` empty => no disk cache (generate every miss).  The reference uses
    // dirs::config_dir()/raytrace/world (chunk_storage.rs:22-27).
    ChunkStorage(std::string storage_dir, uint64_t seed);
    // borrow_packed_chunk_data (chunk_storage.rs:147-152): read the chunk file if present (falling back to generation on a
    // read error, :131-138), otherwise generate, pack and store it (:74-93; a write error only warns, :84-90).
    const PackedChunkData& borrow_packed_chunk_data(long cx, long cy, long cz);
    bool has_chunk(long cx, long cy, long cz) const;                                   // :70-72
    static std::string file_name(long cx, long cy, long cz);                           // :37-40
    std::string path_for(long cx, long cy, long cz) const;
    static bool write_packed_chunk_data(const std::string& path, const PackedChunkData& data);       // :42-55
    static bool read_into_packed_chunk_data(const std::string& path, PackedChunkData& data);          // :57-68
    static bool codec_available();
    size_t generated() const { return generated_; }
    size_t loaded() const { return loaded_; }

 private:
    std::string dir_;
    uint64_t seed_;
    std::map<std::tuple<long, long, long>, PackedChunkData> cache_;   // stands in for the 256-buffer pool (:11,:28-34)
    size_t generated_ = 0, loaded_ = 0;
};

}  // namespace rt::world
