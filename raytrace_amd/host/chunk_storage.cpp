// chunk_storage.cpp — see chunk_storage.hpp.
#include "chunk_storage.hpp"

#include <dlfcn.h>
#include <sys/stat.h>

#include <cstdio>
#include <cstring>
#include <vector>

namespace rt::world {

namespace {

// ---- the part of liblz4's frame API that is used (lz4frame.h of lz4 1.9.x; stable ABI) ------------------------------
struct Lz4fFrameInfo {
    int blockSizeID; int blockMode; int contentChecksumFlag; int frameType;
    unsigned long long contentSize; unsigned dictID; int blockChecksumFlag;
};
struct Lz4fPreferences {
    Lz4fFrameInfo frameInfo; int compressionLevel; unsigned autoFlush; unsigned favorDecSpeed; unsigned reserved[3];
};
struct Lz4Api {
    void* handle = nullptr;
    unsigned (*isError)(size_t) = nullptr;
    size_t (*compressFrameBound)(size_t, const Lz4fPreferences*) = nullptr;
    size_t (*compressFrame)(void*, size_t, const void*, size_t, const Lz4fPreferences*) = nullptr;
    size_t (*createDctx)(void**, unsigned) = nullptr;
    size_t (*freeDctx)(void*) = nullptr;
    size_t (*decompress)(void*, void*, size_t*, const void*, size_t*, const void*) = nullptr;
    bool ok = false;
};

const Lz4Api& lz4() {
    static Lz4Api api = [] {
        Lz4Api a;
        a.handle = dlopen("liblz4.so.1", RTLD_NOW | RTLD_LOCAL);
        if (!a.handle) return a;
        a.isError = (unsigned (*)(size_t))dlsym(a.handle, "LZ4F_isError");
        a.compressFrameBound = (size_t(*)(size_t, const Lz4fPreferences*))dlsym(a.handle, "LZ4F_compressFrameBound");
        a.compressFrame = (size_t(*)(void*, size_t, const void*, size_t, const Lz4fPreferences*))dlsym(a.handle, "LZ4F_compressFrame");
        a.createDctx = (size_t(*)(void**, unsigned))dlsym(a.handle, "LZ4F_createDecompressionContext");
        a.freeDctx = (size_t(*)(void*))dlsym(a.handle, "LZ4F_freeDecompressionContext");
        a.decompress = (size_t(*)(void*, void*, size_t*, const void*, size_t*, const void*))dlsym(a.handle, "LZ4F_decompress");
        a.ok = a.isError && a.compressFrameBound && a.compressFrame && a.createDctx && a.freeDctx && a.decompress;
        return a;
    }();
    return api;
}

constexpr size_t kPayload = (size_t)kChunkVolume * 4 + (size_t)kChunkVolume;   // materials then minefield

}  // namespace

bool ChunkStorage::codec_available() { return lz4().ok; }

ChunkStorage::ChunkStorage(std::string storage_dir, uint64_t seed) : dir_(std::move(storage_dir)), seed_(seed) {
    if (!dir_.empty()) mkdir(dir_.c_str(), 0755);   // create_dir_all of the leaf (chunk_storage.rs:27)
}

std::string ChunkStorage::file_name(long cx, long cy, long cz) {
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%016lX%016lX%016lX", (unsigned long)cx, (unsigned long)cy, (unsigned long)cz);
    return buf;
}

std::string ChunkStorage::path_for(long cx, long cy, long cz) const { return dir_ + "/" + file_name(cx, cy, cz); }

bool ChunkStorage::has_chunk(long cx, long cy, long cz) const {
    if (dir_.empty()) return false;
    struct stat st;
    return stat(path_for(cx, cy, cz).c_str(), &st) == 0;
}

bool ChunkStorage::write_packed_chunk_data(const std::string& path, const PackedChunkData& data) {
    const Lz4Api& z = lz4();
    if (!z.ok) return false;
    std::vector<uint8_t> raw(kPayload);
    std::memcpy(raw.data(), data.materials.data(), (size_t)kChunkVolume * 4);          // native (little-endian) u32s, :45-50
    std::memcpy(raw.data() + (size_t)kChunkVolume * 4, data.minefield.data(), kChunkVolume);
    Lz4fPreferences prefs{};
    prefs.frameInfo.contentChecksumFlag = 1;   // lz4 crate EncoderBuilder default: ContentChecksum::ChecksumEnabled
    prefs.compressionLevel = 4;                // EncoderBuilder::new().level(4), chunk_storage.rs:44
    std::vector<uint8_t> out(z.compressFrameBound(raw.size(), &prefs));
    size_t n = z.compressFrame(out.data(), out.size(), raw.data(), raw.size(), &prefs);
    if (z.isError(n)) return false;
    FILE* f = std::fopen(path.c_str(), "wb");
    if (!f) return false;
    bool ok = std::fwrite(out.data(), 1, n, f) == n;
    ok = std::fclose(f) == 0 && ok;
    return ok;
}

bool ChunkStorage::read_into_packed_chunk_data(const std::string& path, PackedChunkData& data) {
    const Lz4Api& z = lz4();
    if (!z.ok) return false;
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) return false;
    std::vector<uint8_t> in;
    uint8_t buf[1 << 16];
    size_t got;
    while ((got = std::fread(buf, 1, sizeof(buf), f)) > 0) in.insert(in.end(), buf, buf + got);
    std::fclose(f);
    void* dctx = nullptr;
    if (z.isError(z.createDctx(&dctx, 100))) return false;
    std::vector<uint8_t> raw(kPayload);
    size_t ipos = 0, opos = 0;
    bool ok = true;
    while (ipos < in.size() && opos < raw.size()) {
        size_t dst = raw.size() - opos, src = in.size() - ipos;
        size_t r = z.decompress(dctx, raw.data() + opos, &dst, in.data() + ipos, &src, nullptr);
        if (z.isError(r)) { ok = false; break; }
        ipos += src; opos += dst;
        if (r == 0) break;             // frame complete
        if (src == 0 && dst == 0) { ok = false; break; }
    }
    z.freeDctx(dctx);
    if (!ok || opos != raw.size()) return false;   // read_exact of both arrays (:63-66)
    std::memcpy(data.materials.data(), raw.data(), (size_t)kChunkVolume * 4);
    std::memcpy(data.minefield.data(), raw.data() + (size_t)kChunkVolume * 4, kChunkVolume);
    return true;
}

const PackedChunkData& ChunkStorage::borrow_packed_chunk_data(long cx, long cy, long cz) {
    auto key = std::make_tuple(cx, cy, cz);
    auto it = cache_.find(key);
    if (it != cache_.end()) return it->second;
    PackedChunkData pc;
    bool have = false;
    if (has_chunk(cx, cy, cz)) {
        have = read_into_packed_chunk_data(path_for(cx, cy, cz), pc);
        if (have) loaded_++;
        else std::fprintf(stderr, "WARNING: Failed to read chunk data for (%ld, %ld, %ld).\n", cx, cy, cz);   // :109-115
    }
    if (!have) {
        Heightmap hm;
        generate_heightmap(hm, cx, cy, seed_);
        UnpackedChunkData uc;
        generate_chunk(uc, cx, cy, cz, hm, seed_);
        uc.pack_into(pc);
        generated_++;
        if (!dir_.empty() && !write_packed_chunk_data(path_for(cx, cy, cz), pc))
            std::fprintf(stderr, "WARNING: Failed to write chunk data for (%ld, %ld, %ld).\n", cx, cy, cz);   // :84-90
    }
    if (cache_.size() > 512) cache_.clear();
    return cache_.emplace(key, std::move(pc)).first->second;
}

}  // namespace rt::world
