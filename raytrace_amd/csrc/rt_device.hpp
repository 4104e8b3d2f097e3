// rt_device.hpp — device-side building blocks shared by the gfx950 kernels in rt_kernels.hip.
//
// What the reference does in shaders/glsl/raytrace.comp is re-expressed here for CDNA4: raw buffer loads with
// explicit integer addressing instead of samplers (both reference samplers are NEAREST), a 4^3-brick-swizzled
// voxel layout, and a per-brick nibble map that the traversal kernel keeps in LDS.  All fp32 arithmetic that
// decides where a ray goes uses the operations of include/rt_math.h with contraction off, so results are
// bit-identical to the CPU oracle.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_abi.h"
#include "../../include/rt_math.h"

namespace rtd {

// Region edge R = 1 << logr.  The reference's ROOT_BLOCK_WIDTH is 256 (logr = 8); 512 and 1024 are the build's extension
// (SURVEY 8d, config C5).  The nibble map always has 64^3 entries = 128 KiB (it has to fit LDS), so one entry covers a
// cube of edge R/64: a 4^3 brick at R = 256, 8^3 at 512, 16^3 at 1024.
constexpr int kCoarsePerAxis = 64;
constexpr int kCoarseCount = kCoarsePerAxis * kCoarsePerAxis * kCoarsePerAxis;  // 262144
constexpr int kCoarseWords = kCoarseCount / 8;   // 32768 u32 = 128 KiB: one nibble per coarse cube
constexpr uint32_t kNibMixed = 15;               // cube holds differing values (or a value > 14): read the byte
constexpr uint32_t kMaxStepValue = 30;           // minefield values above this are rejected at upload

// Voxel (ix,iy,iz) in [0,R)^3 -> index into the brick-swizzled arrays: 64 consecutive entries per 4^3 brick, so one
// 64-byte line of the minefield (256 B of the materials) is one brick.  lb = logr - 2 = log2(bricks per axis).
__device__ __forceinline__ uint32_t swizzled_index(int ix, int iy, int iz, int lb) {
    const uint32_t brick = ((uint32_t)(iz >> 2) << (2 * lb)) | ((uint32_t)(iy >> 2) << lb) | (uint32_t)(ix >> 2);
    return (brick << 6) | ((uint32_t)(iz & 3) << 4) | ((uint32_t)(iy & 3) << 2) | (uint32_t)(ix & 3);
}
// Voxel -> nibble-map entry: the top 6 bits of each coordinate.
__device__ __forceinline__ uint32_t coarse_index(int ix, int iy, int iz, int logr) {
    const int sh = logr - 6;
    return ((uint32_t)(iz >> sh) << 12) | ((uint32_t)(iy >> sh) << 6) | (uint32_t)(ix >> sh);
}

struct Scene {
    const uint8_t* mine;      // u8[256^3]  brick-swizzled minefield (raytrace.comp binding 1)
    const uint32_t* mat;      // u32[256^3] brick-swizzled packed materials (binding 0)
    const uint32_t* coarse;   // u32[32768] nibble per coarse cube (edge R/64; the 4^3 brick at R = 256): common value 0..14, or 15 = mixed
    const uint8_t* brick;     // R > 256 (round 4): the same nibble per 4^3 BRICK, (R/4)^3 of them (1 MiB at 512, 8 MiB at 1024), in global
                              // memory — consulted when the cube's entry says "mixed", so that only bricks whose 64 voxels really
                              // differ cost a byte from the 1 GiB array (one 64-byte line each).  Entry of swizzled voxel index v:
                              // nibble (v >> 6) & 1 of byte v >> 7.  At R = 256 it IS the coarse map (same pointer, never consulted).
    const uint32_t* noise;    // u32[512*512] RGBA8 blue noise (binding 9), R in the low byte
};

// Per-frame constants; passed by value as a kernel argument (scalar registers).  Mirrors the live fields of
// the uniform block (raytrace.comp:25-35) plus values the shader recomputes per thread although they are
// uniform (sunangle, sunlight: raytrace.comp:317-318), which the host evaluates once with the same rt_math.h.
struct Frame {
    float origin[3], forward[3], up[3], right[3];
    float lr[3];
    float sunangle[3], sunlight[3];
    uint32_t seed;         // seed of sample 0
    int width, height;
    int tiles_x, tiles_y;  // 8x8-pixel tiles
    int tile_rank, tile_world;
    int ntiles_local;      // tiles this context renders
    int spp, depth;
    int lr_zero;           // lr == (0,0,0): enables the sky-first shortcut in the traversal loop
    int logr;              // log2 of the region edge R (8, 9, 10)
    float region;          // R as float (raytrace.comp:37 ROOT_BLOCK_WIDTH)
};

struct vec3 { float x, y, z; };
__device__ __forceinline__ vec3 v3(float x, float y, float z) { vec3 v = {x, y, z}; return v; }
__device__ __forceinline__ vec3 vadd(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ vec3 vsub(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ vec3 vscale(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ vec3 vmul(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ vec3 vnormalize(vec3 v) { rtm_vec3 o = rtm_normalize3({v.x, v.y, v.z}); return v3(o.x, o.y, o.z); }
__device__ __forceinline__ float vlength(vec3 v) { return rtm_length3({v.x, v.y, v.z}); }
__device__ __forceinline__ vec3 vmix(vec3 a, vec3 b, float t) { return v3(rtm_mix(a.x, b.x, t), rtm_mix(a.y, b.y, t), rtm_mix(a.z, b.z, t)); }
__device__ __forceinline__ vec3 ld3(const float* p) { return v3(p[0], p[1], p[2]); }

// ---- pixel <-> path mapping ---------------------------------------------------------------------------
// Paths are laid out tile-major: 64 consecutive paths are one 8x8-pixel tile (one wave), so a wave's primary rays
// are as coherent as possible.  Local tile j of this context is global tile tile_rank + j*tile_world (row-major
// over tiles_x x tiles_y).
struct PixelId { int px, py; bool inside; uint32_t out_index; };
__device__ __forceinline__ PixelId pixel_of_local(const Frame& f, uint32_t local_pixel) {
    uint32_t j = local_pixel >> 6, l = local_pixel & 63;
    uint32_t t = (uint32_t)f.tile_rank + j * (uint32_t)f.tile_world;
    PixelId p;
    p.px = (int)(t % (uint32_t)f.tiles_x) * 8 + (int)(l & 7);
    p.py = (int)(t / (uint32_t)f.tiles_x) * 8 + (int)(l >> 3);
    p.inside = p.px < f.width && p.py < f.height && j < (uint32_t)f.ntiles_local;
    // whole-frame contexts write row-major planes (row 0 = bottom of the view); tile-split contexts write
    // tile-major planes that rt_untile scatters after the gather.
    p.out_index = f.tile_world == 1 ? (uint32_t)p.py * (uint32_t)f.width + (uint32_t)p.px : local_pixel;
    return p;
}

// Inverse of the reference's thread->pixel interleave (raytrace.comp:291-294): workgroup that owns coordinate p.
__device__ __forceinline__ uint32_t owning_workgroup(uint32_t p) { return (p / 128u) * 16u + (p % 16u); }

// ---- blue noise (raytrace.comp:298-304,324,336; sampler render_data.rs:110-133) -------------------------
// Unnormalised coordinates, NEAREST, CLAMP_TO_EDGE.
__device__ __forceinline__ uint32_t noise_texel(const Scene& sc, float cx, float cy) {
    float fx = rtm_floor(cx), fy = rtm_floor(cy);
    int ix = fx < 0.0f ? 0 : (fx > 511.0f ? 511 : (int)fx);
    int iy = fy < 0.0f ? 0 : (fy > 511.0f ? 511 : (int)fy);
    if (!(fx == fx)) ix = 0;
    if (!(fy == fy)) iy = 0;
    return sc.noise[iy * RT_NOISE_SIZE + ix];
}
__device__ __forceinline__ float unorm8(uint32_t texel, int channel) { return (float)((texel >> (8 * channel)) & 0xFFu) / 255.0f; }

struct NoiseOffset { float x, y; };
// noise_offset of raytrace.comp:298-304 for pixel (px,py) and `seed`.
__device__ __forceinline__ NoiseOffset noise_offset_of(const Scene& sc, uint32_t seed, int px, int py) {
    uint32_t base = noise_texel(sc, (float)(seed % RT_NOISE_SIZE), (float)(seed / RT_NOISE_SIZE));
    NoiseOffset o;
    o.x = unorm8(base, 0) * 255.0f + (float)(owning_workgroup((uint32_t)px) * RT_SHADER_GROUP_SIZE);
    o.y = unorm8(base, 1) * 255.0f + (float)(owning_workgroup((uint32_t)py) * RT_SHADER_GROUP_SIZE);
    return o;
}
// noise_value for surface level `level` (1-based): raytrace.comp:324 (level 1) and :336 (level 2, + 2/512).
__device__ __forceinline__ uint32_t noise_value_texel(const Scene& sc, NoiseOffset o, int level) {
    float add = (float)(level - 1) * (2.0f / (float)RT_NOISE_SIZE);
    return noise_texel(sc, rtm_mod(o.x + add, (float)RT_NOISE_SIZE), rtm_mod(o.y + add, (float)RT_NOISE_SIZE));
}

// ---- sky / sun (raytrace.comp:259-288) ------------------------------------------------------------------
__device__ __forceinline__ vec3 sample_sky(vec3 direction, vec3 sun_direction, vec3 sunlight, bool include_sun) {
    const vec3 bright_color = v3(0.5294f, 0.8275f, 0.9647f);
    const vec3 dark_color = v3(0.0863f, 0.1294f, 0.2196f);
    float sunlight_amount = rtm_clamp((sunlight.x + sunlight.y + sunlight.z) * 0.2f - 0.02f, 0.0f, 1.0f);
    float horizon = rtm_pow(rtm_length2(direction.x, direction.y), rtm_mix(40.0f, 10.0f, sunlight_amount));
    float sun_amount = 1.0f - 0.5f * vlength(vsub(sun_direction, direction));
    float sun_halo_amount = rtm_pow(sun_amount, rtm_mix(5.0f, 1.0f, sunlight_amount));
    float bright_amount = rtm_min(horizon + sun_halo_amount * 0.5f, 1.0f);
    vec3 color = vmix(dark_color, bright_color, bright_amount * rtm_max(sunlight_amount, 0.1f));
    color = vadd(color, vscale(vscale(sunlight, rtm_pow(sun_amount, 5.0f)), 0.5f));
    if (sun_amount > 0.98f && include_sun) color = vadd(color, sunlight);
    return color;
}

// diffuse_direction — raytrace.comp:189-212
__device__ __forceinline__ vec3 diffuse_direction(uint32_t normal, float noise_r, float noise_g) {
    float theta1 = RTM_PI * 2.0f * noise_r;
    float theta2 = rtm_acos(1.0f - 2.0f * noise_g);
    float s1, c1, s2, c2;
    rtm_sincos(theta1, &s1, &c1);
    rtm_sincos(theta2, &s2, &c2);
    vec3 d = v3(s1 * s2, c1 * s2, c2);
    if (normal == 0) d.x += 1.0f;
    else if (normal == 1) d.x -= 1.0f;
    else if (normal == 2) d.y += 1.0f;
    else if (normal == 3) d.y -= 1.0f;
    else if (normal == 4) d.z += 1.0f;
    else if (normal == 5) d.z -= 1.0f;
    return vnormalize(d);
}

// Direction handed to trace_ray by trace_sun — raytrace.comp:185-187
__device__ __forceinline__ vec3 sun_ray_direction(vec3 sunangle, float noise_r, float noise_g) {
    return vnormalize(v3(sunangle.x + noise_r * 0.05f, sunangle.y + noise_g * 0.05f, sunangle.z + 0.0f * 0.05f));
}

// Primary ray of a pixel — raytrace.comp:296-297,306-315
__device__ __forceinline__ void primary_ray(const Frame& f, int px, int py, vec3* start, vec3* dir) {
    float sx = ((float)px / (float)f.width) * 2.0f - 1.0f;
    float sy = ((float)py / (float)f.height) * 2.0f - 1.0f;
    vec3 ray_start = ld3(f.origin);
    vec3 d = vnormalize(vadd(vadd(ld3(f.forward), vscale(ld3(f.right), sx)), vscale(ld3(f.up), sy)));
    if (-ray_start.y > f.region / 2.0f) {
        float space = -ray_start.y - (f.region / 2.0f);
        ray_start = vadd(ray_start, vscale(d, space / d.y + 0.0001f));
    }
    *start = ray_start;
    *dir = d;
}

__device__ __forceinline__ vec3 albedo_of(uint32_t packed) {   // raytrace.comp:156-158
    return v3((float)(packed >> 14 & 0x7Fu) / 127.0f, (float)(packed >> 7 & 0x7Fu) / 127.0f,
              (float)(packed & 0x7Fu) / 127.0f);
}

// ---- texel addressing with the reference's border semantics ------------------------------------------------
// mod(pos + R/2, R) per axis, then NEAREST + CLAMP_TO_BORDER (raytrace.comp:79,106,137; render_data.rs:90-101).
// Returns false when any coordinate lands outside [0,R) or is NaN (border colour 0).
__device__ __forceinline__ bool wrap_texel(vec3 pos, float W, int* ix, int* iy, int* iz) {
    float cx = rtm_mod(pos.x + W / 2, W), cy = rtm_mod(pos.y + W / 2, W), cz = rtm_mod(pos.z + W / 2, W);
    bool ok = (cx >= 0.0f && cx < W) && (cy >= 0.0f && cy < W) && (cz >= 0.0f && cz < W);
    *ix = ok ? (int)cx : 0; *iy = ok ? (int)cy : 0; *iz = ok ? (int)cz : 0;
    return ok;
}
// textureLod(world, mod((pos+R/2)/R, 1.0), 0) — raytrace.comp:150-154 (normalised coordinates).
__device__ __forceinline__ uint32_t fetch_material(const Scene& sc, vec3 pos, float W, int lb) {
    float ux = rtm_mod((pos.x + W / 2) / W, 1.0f) * W, uy = rtm_mod((pos.y + W / 2) / W, 1.0f) * W,
          uz = rtm_mod((pos.z + W / 2) / W, 1.0f) * W;
    bool ok = (ux >= 0.0f && ux < W) && (uy >= 0.0f && uy < W) && (uz >= 0.0f && uz < W);
    if (!ok) return 0u;
    return sc.mat[swizzled_index((int)ux, (int)uy, (int)uz, lb)];
}

// ---- one complete ray, generic form (any lr, byte minefield straight from memory) ---------------------------
// Used by the one-thread-per-pixel kernel; the wavefront kernel has its own restructured loop.
struct Hit {
    vec3 position;
    uint32_t material;
    uint32_t normal;
    bool air;
    uint32_t iterations, border, limit_exit;
};

__device__ __forceinline__ uint32_t fetch_step_global(const Scene& sc, vec3 pos, float W, int lb, uint32_t* border) {
    int ix, iy, iz;
    if (!wrap_texel(pos, W, &ix, &iy, &iz)) { (*border)++; return 0u; }
    return sc.mine[swizzled_index(ix, iy, iz, lb)];
}

__device__ inline Hit trace_ray_generic(const Scene& sc, const Frame& f, vec3 origin, vec3 direction) {
    direction = vnormalize(direction);                                                  // raytrace.comp:83
    Hit h;
    h.position = origin; h.material = 0; h.normal = 0; h.air = false; h.iterations = 0; h.border = 0; h.limit_exit = 0;
    vec3 len = v3(1.0f / rtm_abs(direction.x), 1.0f / rtm_abs(direction.y), 1.0f / rtm_abs(direction.z));   // :88
    uint32_t nx = direction.x > 0.0f ? 1u : 0u, ny = direction.y > 0.0f ? 3u : 2u, nz = direction.z > 0.0f ? 5u : 4u;
    vec3 muls = v3(direction.x > 0.0f ? -1.0f : 1.0f, direction.y > 0.0f ? -1.0f : 1.0f, direction.z > 0.0f ? -1.0f : 1.0f);
    const float half = f.region / 2;
    const int lb = f.logr - 2;
    uint32_t step = fetch_step_global(sc, h.position, f.region, lb, &h.border);           // :106
    uint32_t step_size = (1u << (step & 31u)) / 2u;                                        // :107
    bool done = false;
    for (uint32_t limit = RT_TRACE_LIMIT; limit > 0; limit--) {                            // :109-113
        h.iterations++;
        float ss = (float)step_size;
        vec3 q = vmul(vadd(h.position, v3(half, half, half)), muls);
        float lx = (0.0001f + rtm_mod(q.x, ss)) * len.x, ly = (0.0001f + rtm_mod(q.y, ss)) * len.y,
              lz = (0.0001f + rtm_mod(q.z, ss)) * len.z;                                   // :119
        float t; uint32_t n;
        if (lx < ly) { if (lx < lz) { t = lx; n = nx; } else { t = lz; n = nz; } }
        else         { if (ly < lz) { t = ly; n = ny; } else { t = lz; n = nz; } }          // :120-136
        h.position = v3(rtm_fma(direction.x, t, h.position.x), rtm_fma(direction.y, t, h.position.y),
                        rtm_fma(direction.z, t, h.position.z));   // fused (rt_math.h contract)
        h.normal = n;
        step = fetch_step_global(sc, h.position, f.region, lb, &h.border);                 // :137
        if (rtm_abs(h.position.x - f.lr[0]) >= half || rtm_abs(h.position.y - f.lr[1]) >= half ||
            rtm_abs(h.position.z - f.lr[2]) >= half) {                                      // :138-145
            h.air = true; done = true; break;
        } else if (step == 0u) {                                                            // :146-160
            h.material = fetch_material(sc, h.position, f.region, lb);
            done = true; break;
        }
        step_size = (1u << (step & 31u)) / 2u;                                              // :161
    }
    if (!done) h.limit_exit = 1;   // Q8: defined as a non-air hit with material 0
    const float off = 0.001f;                                                               // :166-180
    if (h.normal == 0) h.position.x += off; else if (h.normal == 1) h.position.x -= off;
    else if (h.normal == 2) h.position.y += off; else if (h.normal == 3) h.position.y -= off;
    else if (h.normal == 4) h.position.z += off; else if (h.normal == 5) h.position.z -= off;
    return h;
}

// ---- G-buffer stores (raytrace.comp:352-385; formats render_data.rs:166-189) --------------------------------
struct Planes {
    uint16_t* lighting_rgba16; uint16_t* depth_r16; uint8_t* normal_r8; uint32_t* albedo_rgba8;
    uint32_t* emission_rgba8; uint32_t* fog_rgba8; float* lighting_f32; float* fog_f32; float* depth_f32;
};
__device__ __forceinline__ uint32_t pack_rgba8(float r, float g, float b, float a) {
    return rtm_unorm(r, 255.0f) | (rtm_unorm(g, 255.0f) << 8) | (rtm_unorm(b, 255.0f) << 16) | (rtm_unorm(a, 255.0f) << 24);
}
// Everything except lighting: depends on the primary hit only.
__device__ __forceinline__ void store_primary_planes(const Planes& pl, uint32_t i, const Frame& f, vec3 ray_direction,
                                                     bool air, uint32_t normal, uint32_t material, vec3 position) {
    vec3 sunangle = ld3(f.sunangle), sunlight = ld3(f.sunlight);
    float depth_f = air ? 65535.0f : vlength(vsub(ld3(f.origin), position)) * 32.0f;    // :356-359
    pl.depth_f32[i] = depth_f;
    pl.depth_r16[i] = (uint16_t)(air ? RT_DEPTH_AIR : rtm_f2u16(depth_f));
    pl.normal_r8[i] = (uint8_t)(air ? RT_NORMAL_AIR : normal);                          // :366-370
    vec3 alb = air ? v3(1.0f, 1.0f, 1.0f) : albedo_of(material);                        // :371-375
    pl.albedo_rgba8[i] = pack_rgba8(alb.x, alb.y, alb.z, 1.0f);
    pl.emission_rgba8[i] = air ? 0u : pack_rgba8(0.0f, 0.0f, 0.0f, 1.0f);               // :376-380 (emission is vec3(0), :155)
    vec3 fog = vscale(sample_sky(ray_direction, sunangle, sunlight, false), 0.5f);      // :381-385
    float4 fv = make_float4(fog.x, fog.y, fog.z, 1.0f);
    reinterpret_cast<float4*>(pl.fog_f32)[i] = fv;
    pl.fog_rgba8[i] = pack_rgba8(fog.x, fog.y, fog.z, 1.0f);
}
// Lighting: vec4(light, 1) / LIGHTING_SCALE (:352-356) where light = sum / spp.
__device__ __forceinline__ void store_lighting(const Planes& pl, uint32_t i, vec3 sum, int spp) {
    float n = (float)spp;
    float4 lv = make_float4((sum.x / n) / RT_LIGHTING_SCALE, (sum.y / n) / RT_LIGHTING_SCALE,
                            (sum.z / n) / RT_LIGHTING_SCALE, 1.0f / RT_LIGHTING_SCALE);
    reinterpret_cast<float4*>(pl.lighting_f32)[i] = lv;
    ushort4 q;
    q.x = (uint16_t)rtm_unorm(lv.x, 65535.0f); q.y = (uint16_t)rtm_unorm(lv.y, 65535.0f);
    q.z = (uint16_t)rtm_unorm(lv.z, 65535.0f); q.w = (uint16_t)rtm_unorm(lv.w, 65535.0f);
    reinterpret_cast<ushort4*>(pl.lighting_rgba16)[i] = q;
}

// ---- light unwinding shared by all kernels ----------------------------------------------------------------
// L_j = [sun_j] S + (dif_j air ? sky : (j < D ? L_{j+1} * albedo_{j+1} + emission : 0))  — the body of
// raytrace.comp:324-349 generalised to `depth` levels; evaluated innermost-first so the fp32 operation order is
// exactly the shader's (light2 *= albedo2; light2 += emission; light += light2).
template <typename AlbedoAt>
__device__ __forceinline__ vec3 unwind_light(int K, uint32_t sunbits, bool terminal_sky, vec3 sky, vec3 sunlight,
                                             AlbedoAt albedo_at /* level j in 1..K-1 -> packed material of surface j+1 */) {
    vec3 L = v3(0.0f, 0.0f, 0.0f);
    if (sunbits >> (K - 1) & 1u) L = vadd(L, sunlight);
    if (terminal_sky) L = vadd(L, sky);
    for (int j = K - 1; j >= 1; j--) {
        vec3 light2 = vmul(L, albedo_of(albedo_at(j)));
        light2 = vadd(light2, v3(0.0f, 0.0f, 0.0f));     // + dif.emission, always vec3(0) (raytrace.comp:155)
        vec3 acc = v3(0.0f, 0.0f, 0.0f);
        if (sunbits >> (j - 1) & 1u) acc = vadd(acc, sunlight);
        L = vadd(acc, light2);
    }
    return L;
}

// ---- exact integer counters ------------------------------------------------------------------------------
struct DevCounters {
    unsigned long long rays, rays_primary, rays_shadow, rays_diffuse, iterations, minefield_fetches,
        material_fetches, noise_fetches, hits, sky_exits, limit_exits, border_fetches, pixels, frames;
    // kernel-structure statistics of counting builds (not part of RtCounters; RT_DEBUG_STATS=1 prints them)
    unsigned long long dbg_loop_iters, dbg_s_execs, dbg_f_execs, dbg_s_lanes, dbg_f_lanes, dbg_passes, dbg_pass_lanes, dbg_sky_lanes;
};
// Sum `v` over the wave and let one lane add it.
__device__ __forceinline__ void wave_add(unsigned long long* dst, unsigned long long v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    if ((threadIdx.x & 63) == 0 && v) atomicAdd(dst, v);
}

}  // namespace rtd
