// world.cpp — see world.hpp.
#include "world.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace rt::world {

// ---- materials (GEN_MATERIALS.rs:44-106; misc/materials.csv) --------------------------------------------
const Material MATERIALS[kMaterialCount] = {
    {{0, 0, 0}, {0, 0, 0}, false},          // 0 air
    {{127, 0, 127}, {0, 0, 0}, true},       // 1 debug magenta
    {{39, 110, 61}, {0, 0, 0}, true},       // 2 grass
    {{51, 38, 25}, {320, 154, 76}, true},   // 3 lamp (emission never reaches the shader: raytrace.comp:155)
    {{51, 51, 51}, {0, 0, 0}, true},        // 4 dark stone
    {{62, 27, 22}, {0, 0, 0}, true},        // 5 dirt
    {{110, 116, 115}, {0, 0, 0}, true},     // 6 snow / rock
};

uint32_t Material::pack() const {
    uint32_t rgb = (uint32_t)albedo[0] << 14 | (uint32_t)albedo[1] << 7 | (uint32_t)albedo[2];
    return rgb | (solid ? 1u << 15 : 0u);
}

Material Material::unpack(uint32_t packed) {
    Material m{};
    m.albedo[0] = (uint16_t)(packed >> 14 & 0x7F);
    m.albedo[1] = (uint16_t)(packed >> 7 & 0x7F);
    m.albedo[2] = (uint16_t)(packed & 0x7F);
    m.solid = (packed >> 15 & 1u) != 0;
    return m;
}

// ---- minefield builder (chunk.rs:125-184) ---------------------------------------------------------------
// Built bottom-up: occupancy pyramid occ[L] over aligned 2^L cubes (OR of the 8 children), then every empty
// voxel takes the first level whose cube is occupied.  occ[L] here is lods[L-1] of the reference.
void UnpackedChunkData::pack_into(PackedChunkData& out) const {
    std::vector<uint8_t> occ[RT_MAX_CHUNK_LOD + 1];
    occ[0].resize(kChunkVolume);
    for (int i = 0; i < kChunkVolume; i++) {
        const Material& m = MATERIALS[ids[i] < kMaterialCount ? ids[i] : 0];
        occ[0][i] = m.solid ? 1 : 0;
        out.materials[i] = m.pack();
    }
    for (int L = 1; L <= RT_MAX_CHUNK_LOD; L++) {
        const int n = kChunk >> L, pn = n * 2;
        occ[L].assign((size_t)n * n * n, 0);
        const uint8_t* child = occ[L - 1].data();
        for (int z = 0; z < n; z++)
            for (int y = 0; y < n; y++)
                for (int x = 0; x < n; x++) {
                    uint8_t any = 0;
                    for (int k = 0; k < 8; k++)
                        any |= child[((size_t)(2 * z + (k >> 2)) * pn + (2 * y + (k >> 1 & 1))) * pn + 2 * x + (k & 1)];
                    occ[L][((size_t)z * n + y) * n + x] = any;
                }
    }
    if (!occ[RT_MAX_CHUNK_LOD][0]) {  // chunk.rs:154-161
        std::fill(out.materials.begin(), out.materials.end(), MATERIALS[0].pack());
        std::fill(out.minefield.begin(), out.minefield.end(), (uint8_t)RT_MAX_CHUNK_LOD);
        return;
    }
    for (int z = 0; z < kChunk; z++)
        for (int y = 0; y < kChunk; y++)
            for (int x = 0; x < kChunk; x++) {
                size_t i = ((size_t)z * kChunk + y) * kChunk + x;
                uint8_t v = 0;
                if (!occ[0][i]) {
                    for (int L = 1; L <= RT_MAX_CHUNK_LOD; L++) {
                        const int n = kChunk >> L;
                        if (occ[L][((size_t)(z >> L) * n + (y >> L)) * n + (x >> L)]) { v = (uint8_t)L; break; }
                    }
                }
                out.minefield[i] = v;
            }
}

// ---- deterministic terrain (shape of functions.rs:69-99 and generate.rs:11-85) --------------------------
namespace {

inline uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline uint64_t hash3(uint64_t seed, long a, long b, long c) {
    uint64_t h = mix64(seed ^ 0xA5A5A5A5DEADBEEFull);
    h = mix64(h ^ (uint64_t)a);
    h = mix64(h ^ (uint64_t)b * 0x9E3779B97F4A7C15ull);
    h = mix64(h ^ (uint64_t)c * 0xC2B2AE3D27D4EB4Full);
    return h;
}

// 2-D gradient noise in [-1, 1] with 16 hashed gradient directions and quintic fade.
double gradient_noise(double x, double y, uint64_t seed) {
    static const double kDir[16][2] = {
        {1.0, 0.0}, {0.9238795325, 0.3826834324}, {0.7071067812, 0.7071067812}, {0.3826834324, 0.9238795325},
        {0.0, 1.0}, {-0.3826834324, 0.9238795325}, {-0.7071067812, 0.7071067812}, {-0.9238795325, 0.3826834324},
        {-1.0, 0.0}, {-0.9238795325, -0.3826834324}, {-0.7071067812, -0.7071067812}, {-0.3826834324, -0.9238795325},
        {0.0, -1.0}, {0.3826834324, -0.9238795325}, {0.7071067812, -0.7071067812}, {0.9238795325, -0.3826834324}};
    double fx = std::floor(x), fy = std::floor(y);
    long ix = (long)fx, iy = (long)fy;
    double tx = x - fx, ty = y - fy;
    auto corner = [&](long cx, long cy, double dx, double dy) {
        const double* g = kDir[hash3(seed, cx, cy, 0) & 15];
        return g[0] * dx + g[1] * dy;
    };
    auto fade = [](double t) { return t * t * t * (t * (t * 6.0 - 15.0) + 10.0); };
    double n00 = corner(ix, iy, tx, ty), n10 = corner(ix + 1, iy, tx - 1.0, ty);
    double n01 = corner(ix, iy + 1, tx, ty - 1.0), n11 = corner(ix + 1, iy + 1, tx - 1.0, ty - 1.0);
    double u = fade(tx), v = fade(ty);
    double a = n00 + (n10 - n00) * u, b = n01 + (n11 - n01) * u;
    return (a + (b - a) * v) * 1.4142135623730951;  // scale to roughly [-1, 1]
}

// 6 octaves, frequency 2, lacunarity 2, persistence 0.5 (the defaults the reference configures at
// functions.rs:73-80), each octave with its own hash stream.
double basic_multi(double x, double y, uint64_t seed) {
    double px = x * 2.0, py = y * 2.0, amp = 1.0, sum = 0.0;
    for (int o = 0; o < 6; o++) {
        sum += gradient_noise(px, py, seed + (uint64_t)o * 0x632BE59BD9B4E019ull) * amp;
        px *= 2.0; py *= 2.0; amp *= 0.5;
    }
    return sum * 0.5;
}

inline double get_noise(double x, double y, uint64_t seed) { return basic_multi(x, y, seed) * 0.5 + 0.5; }  // functions.rs:83-85

constexpr double kScale = 600.0;  // generate.rs:11

long terrain_height(long x, long y, uint64_t seed) {  // generate.rs:13-15
    return (long)(mountain_noise2((double)x / kScale, (double)y / kScale, seed) * kScale * 0.2 + 10.0);
}

uint8_t material_for_height(uint64_t seed, long x, long y, long z) {  // generate.rs:31-51
    auto roll = [&](uint32_t span) { return (uint32_t)(hash3(seed ^ 0x51ED270B7F4A7C15ull, x, y, z) >> 16) % span; };
    if (z < 20) return 2;
    if (z < 80) return roll(80 - 20) < (uint32_t)(z - 20) ? 5 : 2;
    if (z < 160) return roll(160 - 80) < (uint32_t)(z - 80) ? 6 : 5;
    return 6;
}

}  // namespace

double mountain_noise2(double x, double y, uint64_t seed) {  // functions.rs:87-98
    const double d = 0.2;
    double left = get_noise(x - d, y, seed), right = get_noise(x + d, y, seed);
    double up = get_noise(x, y - d, seed), down = get_noise(x, y + d, seed);
    double dx = (right - left) / (d * 2.0), dy = (down - up) / (d * 2.0);
    double slope = std::sqrt(dx * dx + dy * dy);
    double base = get_noise(x, y, seed);
    double eroded = base + (1.0 - slope) * 0.7;
    if (eroded < 0.0) eroded = 0.0;  // powf of a negative base is NaN in the reference; keep heights finite
    return std::pow(eroded / 1.5, 2.6);
}

void generate_heightmap(Heightmap& out, long chunk_x, long chunk_y, uint64_t seed) {
    const long ox = chunk_x * kChunk, oy = chunk_y * kChunk;
    for (int y = 0; y < kChunk; y++)
        for (int x = 0; x < kChunk; x++) out.data[(size_t)y * kChunk + x] = terrain_height(ox + x, oy + y, seed);
}

void generate_chunk(UnpackedChunkData& out, long cx, long cy, long cz, const Heightmap& hm, uint64_t seed) {
    const long ox = cx * kChunk, oy = cy * kChunk, oz = cz * kChunk;
    if (oz + kChunk < 12) {  // generate.rs:63-64: deep chunks are solid grass whatever the heightmap says
        out.fill(2);
        return;
    }
    for (int y = 0; y < kChunk; y++)
        for (int x = 0; x < kChunk; x++) {
            const long h = hm.get(x, y);
            for (int lz = 0; lz < kChunk; lz++) {
                const long z = oz + lz;
                uint8_t id = 0;
                if (h >= oz && z < h) id = material_for_height(seed, ox + x, oy + y, z);  // generate.rs:66-82
                out.set_block(x, y, lz, id);
            }
        }
}

// ---- 3-D copies ------------------------------------------------------------------------------------------
template <typename T>
bool copy_3d(Dims3 size, const T* src, Dims3 sd, Dims3 so, T* dst, Dims3 dd, Dims3 dof) {
    auto fits = [](int off, int len, int dim) { return off >= 0 && len >= 0 && off + len <= dim; };
    if (!fits(so.x, size.x, sd.x) || !fits(so.y, size.y, sd.y) || !fits(so.z, size.z, sd.z)) return false;
    if (!fits(dof.x, size.x, dd.x) || !fits(dof.y, size.y, dd.y) || !fits(dof.z, size.z, dd.z)) return false;
    for (int z = 0; z < size.z; z++)
        for (int y = 0; y < size.y; y++) {
            const T* s = src + ((size_t)(so.z + z) * sd.y + (so.y + y)) * sd.x + so.x;
            T* d = dst + ((size_t)(dof.z + z) * dd.y + (dof.y + y)) * dd.x + dof.x;
            std::memcpy(d, s, sizeof(T) * (size_t)size.x);
        }
    return true;
}

namespace {
// One axis of a clipped copy: a run of `len` source cells starting at source index s0 is to land at signed
// target index t0; clip to [0,sdim) and [0,ddim).  Returns the surviving length.
int clip_axis(int len, int sdim, int ddim, int& s0, long t0, int& t) {
    long lo = std::max<long>(0, -t0);            // cells cut at the front
    long s = s0 + lo, tt = t0 + lo;
    long n = std::min<long>({(long)len - lo, (long)sdim - s, (long)ddim - tt});
    s0 = (int)s; t = (int)tt;
    return n > 0 ? (int)n : 0;
}
}  // namespace

template <typename T>
void copy_3d_bounded_auto_clip(Dims3 size, const T* src, Dims3 sd, Dims3 so, T* dst, Dims3 dd, Off3 dof) {
    Dims3 n, t;
    n.x = clip_axis(size.x, sd.x, dd.x, so.x, dof.x, t.x);
    n.y = clip_axis(size.y, sd.y, dd.y, so.y, dof.y, t.y);
    n.z = clip_axis(size.z, sd.z, dd.z, so.z, dof.z, t.z);
    if (n.x == 0 || n.y == 0 || n.z == 0) return;
    copy_3d(n, src, sd, so, dst, dd, t);
}

template <typename T>
void copy_3d_auto_clip(const T* src, int src_stride, Off3 offset, T* dst, int dst_stride) {
    Dims3 sd{src_stride, src_stride, src_stride}, dd{dst_stride, dst_stride, dst_stride};
    copy_3d_bounded_auto_clip(sd, src, sd, Dims3{0, 0, 0}, dst, dd, offset);
}

template <typename T>
void fill_slice_3d_auto_clip(T value, T* dst, int dst_stride, Off3 start, Dims3 size) {
    auto clip = [&](long s, int len, int& t0) {
        long lo = std::max<long>(0, s), hi = std::min<long>(dst_stride, s + len);
        t0 = (int)lo;
        return hi > lo ? (int)(hi - lo) : 0;
    };
    int x0, y0, z0;
    int nx = clip(start.x, size.x, x0), ny = clip(start.y, size.y, y0), nz = clip(start.z, size.z, z0);
    for (int z = 0; z < nz; z++)
        for (int y = 0; y < ny; y++)
            std::fill_n(dst + ((size_t)(z0 + z) * dst_stride + (y0 + y)) * dst_stride + x0, nx, value);
}

template bool copy_3d<uint32_t>(Dims3, const uint32_t*, Dims3, Dims3, uint32_t*, Dims3, Dims3);
template bool copy_3d<uint8_t>(Dims3, const uint8_t*, Dims3, Dims3, uint8_t*, Dims3, Dims3);
template void copy_3d_auto_clip<uint32_t>(const uint32_t*, int, Off3, uint32_t*, int);
template void copy_3d_auto_clip<uint8_t>(const uint8_t*, int, Off3, uint8_t*, int);
template void copy_3d_bounded_auto_clip<uint32_t>(Dims3, const uint32_t*, Dims3, Dims3, uint32_t*, Dims3, Off3);
template void copy_3d_bounded_auto_clip<uint8_t>(Dims3, const uint8_t*, Dims3, Dims3, uint8_t*, Dims3, Off3);
template void fill_slice_3d_auto_clip<uint32_t>(uint32_t, uint32_t*, int, Off3, Dims3);
template void fill_slice_3d_auto_clip<uint8_t>(uint8_t, uint8_t*, int, Off3, Dims3);

// ---- region assembly (render_data.rs:203-249) -------------------------------------------------------------
namespace {
void place_chunk(const PackedChunkData& pc, int cx, int cy, int cz, uint32_t* materials, uint8_t* minefield, int region = kRegion) {
    Off3 at{(long)cx * kChunk, (long)cy * kChunk, (long)cz * kChunk};  // scale_coord_3d(&chunk_coord, CHUNK_SIZE)
    copy_3d_auto_clip(pc.materials.data(), kChunk, at, materials, region);
    copy_3d_auto_clip(pc.minefield.data(), kChunk, at, minefield, region);
}
}  // namespace

void assemble_region_procedural(uint64_t seed, uint32_t* materials, uint8_t* minefield, int region) {
    const int nchunks = region / kChunk, half = nchunks / 2;
    UnpackedChunkData uc;
    PackedChunkData pc;
    Heightmap hm;
    for (int cy = 0; cy < nchunks; cy++)
        for (int cx = 0; cx < nchunks; cx++) {
            generate_heightmap(hm, cx - half, cy - half, seed);
            for (int cz = 0; cz < nchunks; cz++) {
                generate_chunk(uc, cx - half, cy - half, cz - half, hm, seed);
                uc.pack_into(pc);
                place_chunk(pc, cx, cy, cz, materials, minefield, region);
            }
        }
}

void assemble_region_from_ids(const uint8_t* ids, uint32_t* materials, uint8_t* minefield) {
    UnpackedChunkData uc;
    PackedChunkData pc;
    const Dims3 rd{kRegion, kRegion, kRegion}, cd{kChunk, kChunk, kChunk};
    for (int cz = 0; cz < kRegionChunks; cz++)
        for (int cy = 0; cy < kRegionChunks; cy++)
            for (int cx = 0; cx < kRegionChunks; cx++) {
                copy_3d(cd, ids, rd, Dims3{cx * kChunk, cy * kChunk, cz * kChunk}, uc.ids.data(), cd, Dims3{0, 0, 0});
                uc.pack_into(pc);
                place_chunk(pc, cx, cy, cz, materials, minefield);
            }
}

}  // namespace rt::world
