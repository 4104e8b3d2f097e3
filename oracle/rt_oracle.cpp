/*
 * rt_oracle.cpp — CPU ORACLE for the ray-trace hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (raytrace_amd/, librt_amd.so) never links, imports or calls it.
 *
 * What it is: a scalar fp32 restatement of the reference's GLSL compute shader
 * shaders/glsl/raytrace.comp (the whole file), plus the host-side pieces that feed it
 * (uniform derivation, voxel packing, minefield builder, region assembly).  Every function
 * cites the reference file:line it follows.  GLSL built-ins follow the GLSL 4.50 spec:
 * mod(x,y) = x - y*floor(x/y); mix(x,y,a) = x*(1-a)+y*a; NEAREST texel = floor(coord).
 * Elementary fp32 functions (normalize, sin, cos, acos, pow) come from include/rt_math.h,
 * the arithmetic contract both this oracle and the HIP kernels compile.
 *
 * PARITY UNPINNED BY THE REFERENCE: the reference holds no golden vectors, known-answer
 * tests or fixtures for this path (its 4 #[test]s cover 3-D array copies and a no-assert
 * generation smoke test: src/util.rs:417-435,496-505,585-603; src/world/chunk_storage.rs:154-183),
 * and the reference itself (Rust + Vulkan + glslc) cannot be built or run in this image.
 * This restatement is pinned only by known-answer tests derived by hand from the shader text
 * (tests/test_oracle_kat.py) and by agreeing — trace_ray and the primary planes bit for
 * bit — with a second restatement written from the GLSL alone (tests/shader_trace.py, tests/shader_formulas.py:
 * K10-K14).  Neither is the reference's own output.
 *
 * Build: see oracle/Makefile (g++ -O2 -ffp-contract=off -fopenmp).
 */
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/rt_abi.h"
#include "../include/rt_math.h"

namespace {

constexpr int kDefaultRegion = RT_ROOT_BLOCK_SIZE;   // raytrace.comp:37 ROOT_BLOCK_WIDTH (256); 512/1024 are the build's extension
constexpr uint32_t NORMAL_x = 0, NORMAL_y = 2, NORMAL_z = 4;  // raytrace.comp:45-47

struct vec3 { float x, y, z; };
inline vec3 operator+(vec3 a, vec3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline vec3 operator-(vec3 a, vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline vec3 operator*(vec3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline vec3 operator*(vec3 a, vec3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline vec3 normalize(vec3 v) { rtm_vec3 o = rtm_normalize3({v.x, v.y, v.z}); return {o.x, o.y, o.z}; }
inline float length(vec3 v) { return rtm_length3({v.x, v.y, v.z}); }
inline vec3 mix3(vec3 a, vec3 b, float t) { return {rtm_mix(a.x, b.x, t), rtm_mix(a.y, b.y, t), rtm_mix(a.z, b.z, t)}; }

struct Counters {
    uint64_t rays = 0, rays_primary = 0, rays_shadow = 0, rays_diffuse = 0, iterations = 0,
             minefield_fetches = 0, material_fetches = 0, noise_fetches = 0, hits = 0, sky_exits = 0,
             limit_exits = 0, border_fetches = 0;
    void add(const Counters& o) {
        rays += o.rays; rays_primary += o.rays_primary; rays_shadow += o.rays_shadow;
        rays_diffuse += o.rays_diffuse; iterations += o.iterations;
        minefield_fetches += o.minefield_fetches; material_fetches += o.material_fetches;
        noise_fetches += o.noise_fetches; hits += o.hits; sky_exits += o.sky_exits;
        limit_exits += o.limit_exits; border_fetches += o.border_fetches;
    }
};

struct Scene {
    const uint32_t* materials;  // u32[R^3], x fastest (src/util.rs:104-106), texel = world + R/2
    const uint8_t* minefield;   // u8[R^3]
    const uint8_t* noise;       // RGBA8 512x512
    int R = kDefaultRegion;     // ROOT_BLOCK_WIDTH
};

// HitResult — raytrace.comp:62-69
struct HitResult {
    vec3 albedo{0, 0, 0};
    vec3 emission{0, 0, 0};
    bool air = false;
    float distance = 0;
    uint32_t normal = 0;
    vec3 position{0, 0, 0};
    uint32_t packed_material = 0;  // not in the shader struct; kept for tests
    uint32_t iterations = 0;
};

// texel index of an unnormalised NEAREST lookup on one axis; -1 = outside (CLAMP_TO_BORDER) or NaN.
// Sampler: render_data.rs:90-101 (minefield), structures.rs:382-461.
inline int border_texel(float c, int R) {
    if (!(c >= 0.0f && c < (float)R)) return -1;
    return (int)c;  // floor for c >= 0
}

// Optional fetch statistics for kernel design studies (single-threaded use): per fetched value, how many fetches
// landed in a 4^3 brick whose 64 values are all equal.
static const uint8_t* g_uniform4 = nullptr;   // [64^3] 1 = uniform brick
static uint64_t g_fetch_hist[2][32];

// get_step — raytrace.comp:78-80.  Border colour INT_OPAQUE_BLACK => 0 (render_data.rs:97-98).
inline uint32_t get_step(const Scene& sc, vec3 tex_pos, Counters& cn) {
    cn.minefield_fetches++;
    const int R = sc.R;
    int ix = border_texel(tex_pos.x, R), iy = border_texel(tex_pos.y, R), iz = border_texel(tex_pos.z, R);
    if ((ix | iy | iz) < 0) { cn.border_fetches++; return 0; }
    const uint8_t v = sc.minefield[((size_t)iz * R + iy) * R + ix];
    if (g_uniform4 && R == 256) g_fetch_hist[g_uniform4[((size_t)(iz >> 2) * 64 + (iy >> 2)) * 64 + (ix >> 2)] ? 1 : 0][v & 31]++;
    return v;
}

// textureLod(world, mod((pos+off)/256, 1.0), 0).r — raytrace.comp:150-154; sampler render_data.rs:62-73
// (normalised coordinates, NEAREST, CLAMP_TO_BORDER): texel = floor(u * 256).
inline uint32_t get_material(const Scene& sc, vec3 p, Counters& cn) {
    cn.material_fetches++;
    const int R = sc.R;
    float u[3] = {rtm_mod(p.x / (float)R, 1.0f), rtm_mod(p.y / (float)R, 1.0f), rtm_mod(p.z / (float)R, 1.0f)};
    int t[3];
    for (int a = 0; a < 3; a++) t[a] = border_texel(u[a] * (float)R, R);
    if ((t[0] | t[1] | t[2]) < 0) return 0;
    return sc.materials[((size_t)t[2] * R + t[1]) * R + t[0]];
}

// (1 << current_step) / 2 — raytrace.comp:107,161.  Shifts >= 32 are undefined in GLSL; the build
// defines the shift amount as (step & 31).
inline uint32_t step_size_of(uint32_t step) { return (1u << (step & 31u)) / 2u; }

// trace_ray — raytrace.comp:82-183
HitResult trace_ray(const Scene& sc, const int32_t lr[3], vec3 origin, vec3 direction, Counters& cn) {
    cn.rays++;
    direction = normalize(direction);                                        // :83
    HitResult result;
    result.position = origin;                                                // :85
    vec3 length_per_axis = {1.0f / rtm_abs(direction.x), 1.0f / rtm_abs(direction.y),
                            1.0f / rtm_abs(direction.z)};                    // :88
    uint32_t normals[3] = {direction.x > 0 ? NORMAL_x + 1 : NORMAL_x,
                           direction.y > 0 ? NORMAL_y + 1 : NORMAL_y,
                           direction.z > 0 ? NORMAL_z + 1 : NORMAL_z};        // :89-93
    vec3 muls = {direction.x > 0 ? -1.0f : 1.0f, direction.y > 0 ? -1.0f : 1.0f,
                 direction.z > 0 ? -1.0f : 1.0f};                             // :94-98
    vec3 current_rotation = {(float)lr[0], (float)lr[1], (float)lr[2]};       // :104
    const float W = (float)sc.R;
    vec3 pos_offset = {W / 2, W / 2, W / 2};                                  // :105
    auto texpos = [&](vec3 p) {
        vec3 q = p + pos_offset;
        return vec3{rtm_mod(q.x, W), rtm_mod(q.y, W), rtm_mod(q.z, W)};
    };
    uint32_t current_step = get_step(sc, texpos(result.position), cn);       // :106
    uint32_t step_size = step_size_of(current_step);                          // :107
    bool terminated = false;
    for (uint32_t limit = RT_TRACE_LIMIT; limit > 0; limit--) {               // :109-113
        cn.iterations++;
        result.iterations++;
        float ss = (float)step_size;
        vec3 q = (result.position + pos_offset) * muls;
        vec3 l = {(0.0001f + rtm_mod(q.x, ss)) * length_per_axis.x,
                  (0.0001f + rtm_mod(q.y, ss)) * length_per_axis.y,
                  (0.0001f + rtm_mod(q.z, ss)) * length_per_axis.z};          // :119
        // result.position += direction * length: the multiply-add is fused (rt_math.h contract; GLSL allows it)
        auto advance = [&](float t) {
            result.position = {rtm_fma(direction.x, t, result.position.x), rtm_fma(direction.y, t, result.position.y),
                               rtm_fma(direction.z, t, result.position.z)};
        };
        if (l.x < l.y) {                                                      // :120-136
            if (l.x < l.z) { advance(l.x); result.normal = normals[0]; }
            else           { advance(l.z); result.normal = normals[2]; }
        } else {
            if (l.y < l.z) { advance(l.y); result.normal = normals[1]; }
            else           { advance(l.z); result.normal = normals[2]; }
        }
        current_step = get_step(sc, texpos(result.position), cn);            // :137
        if (rtm_abs(result.position.x - current_rotation.x) >= W / 2 ||
            rtm_abs(result.position.y - current_rotation.y) >= W / 2 ||
            rtm_abs(result.position.z - current_rotation.z) >= W / 2) {       // :138-145
            result.air = true;
            cn.sky_exits++;
            terminated = true;
            break;
        } else if (current_step <= 0) {                                       // :146-160
            result.air = false;
            uint32_t pm = get_material(sc, result.position + pos_offset, cn);
            result.packed_material = pm;
            result.emission = {0, 0, 0};
            result.albedo = {(float)(pm >> 14 & 0x7F) / 127.0f, (float)(pm >> 7 & 0x7F) / 127.0f,
                             (float)(pm >> 0 & 0x7F) / 127.0f};
            cn.hits++;
            terminated = true;
            break;
        }
        step_size = step_size_of(current_step);                               // :161
    }
    if (!terminated) {
        // Q8: the shader leaves air/albedo undefined when the limit is reached; the build defines
        // air = false, albedo = 0, emission = 0 and counts the event.
        result.air = false;
        cn.limit_exits++;
    }
    result.distance = length(origin - result.position);                       // :164
    const float offset_amount = 0.001f;                                       // :166-180
    if (result.normal == NORMAL_x) result.position.x += offset_amount;
    else if (result.normal == NORMAL_x + 1) result.position.x -= offset_amount;
    else if (result.normal == NORMAL_y) result.position.y += offset_amount;
    else if (result.normal == NORMAL_y + 1) result.position.y -= offset_amount;
    else if (result.normal == NORMAL_z) result.position.z += offset_amount;
    else if (result.normal == NORMAL_z + 1) result.position.z -= offset_amount;
    return result;
}

struct vec4 { float r, g, b, a; };

// texture(blue_noise, coord): unnormalised coordinates, NEAREST, CLAMP_TO_EDGE, R8G8B8A8_UNORM
// (render_data.rs:110-133).
inline vec4 noise_texel(const Scene& sc, float cx, float cy, Counters& cn) {
    cn.noise_fetches++;
    float fx = rtm_floor(cx), fy = rtm_floor(cy);
    int ix = fx < 0 ? 0 : (fx > 511 ? 511 : (int)fx);
    int iy = fy < 0 ? 0 : (fy > 511 ? 511 : (int)fy);
    if (!(fx == fx)) ix = 0;
    if (!(fy == fy)) iy = 0;
    const uint8_t* t = sc.noise + ((size_t)iy * RT_NOISE_SIZE + ix) * 4;
    return {t[0] / 255.0f, t[1] / 255.0f, t[2] / 255.0f, t[3] / 255.0f};
}

// trace_sun — raytrace.comp:185-187
HitResult trace_sun(const Scene& sc, const int32_t lr[3], const HitResult& from, vec3 direction, vec4 noise_value,
                    Counters& cn) {
    vec3 d = {direction.x + noise_value.r * 0.05f, direction.y + noise_value.g * 0.05f, direction.z + 0.0f * 0.05f};
    cn.rays_shadow++;
    return trace_ray(sc, lr, from.position, normalize(d), cn);
}

// diffuse_direction — raytrace.comp:189-212
vec3 diffuse_direction(uint32_t normal, vec4 noise_value) {
    float theta1 = RTM_PI * 2.0f * noise_value.r;
    float theta2 = rtm_acos(1.0f - 2.0f * noise_value.g);
    float s1, c1, s2, c2;
    rtm_sincos(theta1, &s1, &c1);
    rtm_sincos(theta2, &s2, &c2);
    vec3 direction = {s1 * s2, c1 * s2, c2};
    if (normal == NORMAL_x) direction.x += 1;
    else if (normal == NORMAL_x + 1) direction.x -= 1;
    else if (normal == NORMAL_y) direction.y += 1;
    else if (normal == NORMAL_y + 1) direction.y -= 1;
    else if (normal == NORMAL_z) direction.z += 1;
    else if (normal == NORMAL_z + 1) direction.z -= 1;
    return normalize(direction);
}

// sun_color — raytrace.comp:259-269
vec3 sun_color(vec3 sun_direction) {
    float horizon = rtm_length2(sun_direction.x, sun_direction.y);
    float sun_amount = rtm_min(1.0f - horizon, 0.02f) * 50.0f;
    vec3 main_color = vec3{0.9647f, 0.7843f, 0.8824f} * 2.0f;
    vec3 sunset_color = vec3{0.7412f, 0.2157f, 0.1686f} * 2.0f;
    if (sun_direction.z >= 0.0f) return mix3(sunset_color, main_color, sun_amount);
    return mix3(sunset_color, vec3{0, 0, 0}, sun_amount * 2);
}

// sample_sky — raytrace.comp:271-288
vec3 sample_sky(vec3 direction, vec3 sun_direction, vec3 sunlight, bool include_sun) {
    vec3 bright_color = {0.5294f, 0.8275f, 0.9647f};
    vec3 dark_color = {0.0863f, 0.1294f, 0.2196f};
    float sunlight_amount = rtm_clamp((sunlight.x + sunlight.y + sunlight.z) * 0.2f - 0.02f, 0.0f, 1.0f);
    float horizon = rtm_pow(rtm_length2(direction.x, direction.y), rtm_mix(40.0f, 10.0f, sunlight_amount));
    float sun_amount = 1.0f - 0.5f * length(sun_direction - direction);
    float sun_halo_amount = rtm_pow(sun_amount, rtm_mix(5.0f, 1.0f, sunlight_amount));
    float bright_amount = rtm_min(horizon + sun_halo_amount * 0.5f, 1.0f);
    vec3 color = mix3(dark_color, bright_color, bright_amount * rtm_max(sunlight_amount, 0.1f));
    color = color + sunlight * rtm_pow(sun_amount, 5.0f) * 0.5f;
    if (sun_amount > 0.98f && include_sun) color = color + sunlight;
    return color;
}

// vec3 sunangle = normalize(...) — raytrace.comp:317
vec3 sun_vector(float a) {
    float s, c;
    rtm_sincos(a, &s, &c);
    return normalize(vec3{c * 0.5f + (a - 0.5f) * 0.5f, s, c});
}

// Inverse of the thread -> pixel interleave (raytrace.comp:291-294): the workgroup that
// owns pixel coordinate p on one axis is (p/128)*16 + p%16.
inline uint32_t owning_workgroup(uint32_t p) { return (p / 128u) * 16u + p % 16u; }

struct FrameConsts {
    vec3 sunangle, sunlight;
};

// Light reaching surface level `level` (1-based): the body of raytrace.comp:324-349 generalised to
// `depth` levels.  depth == 2 is exactly the shader: level 1 is :324-333, level 2 is :334-348.
vec3 level_light(const Scene& sc, const RtUniforms& u, const FrameConsts& fc, const HitResult& surface,
                 float noise_off_x, float noise_off_y, int level, int depth, Counters& cn) {
    // :324 / :336 — the second lookup adds 2.0/NOISE_SIZE; level j adds (j-1)*2/NOISE_SIZE.
    float add = (float)(level - 1) * (2.0f / (float)RT_NOISE_SIZE);
    vec4 noise_value = noise_texel(sc, rtm_mod(noise_off_x + add, (float)RT_NOISE_SIZE),
                                   rtm_mod(noise_off_y + add, (float)RT_NOISE_SIZE), cn);
    vec3 light = {0, 0, 0};
    HitResult sun = trace_sun(sc, u.lr, surface, fc.sunangle, noise_value, cn);       // :325 / :337
    if (sun.air) light = light + fc.sunlight;                                           // :326-328
    vec3 dif_dir = diffuse_direction(surface.normal, noise_value);                      // :329 / :341
    cn.rays_diffuse++;
    HitResult dif = trace_ray(sc, u.lr, surface.position, dif_dir, cn);                 // :330 / :342
    if (dif.air) {
        light = light + sample_sky(dif_dir, fc.sunangle, fc.sunlight, true);            // :331-332 / :343-345
    } else if (level < depth) {
        vec3 light2 = level_light(sc, u, fc, dif, noise_off_x, noise_off_y, level + 1, depth, cn);
        light2 = light2 * dif.albedo;                                                   // :346
        light2 = light2 + dif.emission;                                                 // :347
        light = light + light2;                                                         // :348
    }
    return light;
}

struct PixelOut {
    vec3 light;       // one sample
    bool air;
    uint32_t normal;
    vec3 albedo, emission, fog;
    float depth_f;
};

// main — raytrace.comp:290-385 for one pixel and one seed.
PixelOut shade_pixel(const Scene& sc, const RtUniforms& u, const FrameConsts& fc, int W, int H, int px, int py,
                     uint32_t seed, int depth, Counters& cn) {
    // :296-297  screen_pos = pixel / vec2(imageSize) * 2 - 1
    float sx = ((float)px / (float)W) * 2 - 1.0f;
    float sy = ((float)py / (float)H) * 2 - 1.0f;
    // :298-304
    float lookup_x = (float)(seed % RT_NOISE_SIZE), lookup_y = (float)(seed / RT_NOISE_SIZE);
    vec4 base = noise_texel(sc, lookup_x, lookup_y, cn);
    cn.noise_fetches--;  // the shader samples this texel twice (:302,:303); B_alg counts it once (4 B uniform)
    float noise_off_x = base.r * 255.0f + (float)(owning_workgroup((uint32_t)px) * RT_SHADER_GROUP_SIZE);
    float noise_off_y = base.g * 255.0f + (float)(owning_workgroup((uint32_t)py) * RT_SHADER_GROUP_SIZE);
    // :306-315
    vec3 origin = {u.origin[0], u.origin[1], u.origin[2]};
    vec3 forward = {u.forward[0], u.forward[1], u.forward[2]};
    vec3 up = {u.up[0], u.up[1], u.up[2]}, right = {u.right[0], u.right[1], u.right[2]};
    vec3 ray_start = origin;
    vec3 ray_direction = normalize(forward + right * sx + up * sy);
    if (-ray_start.y > (float)sc.R / 2.0f) {
        float space = -ray_start.y - ((float)sc.R / 2.0f);
        ray_start = ray_start + ray_direction * (space / ray_direction.y + 0.0001f);
    }
    PixelOut o;
    vec3 light = {0, 0, 0};
    cn.rays_primary++;
    HitResult primary = trace_ray(sc, u.lr, ray_start, ray_direction, cn);              // :320
    if (primary.air) {
        light = sample_sky(ray_direction, fc.sunangle, fc.sunlight, true);              // :321-322
    } else if (depth >= 1) {
        light = light + level_light(sc, u, fc, primary, noise_off_x, noise_off_y, 1, depth, cn);
    }
    o.light = light;
    o.air = primary.air;
    o.normal = primary.air ? RT_NORMAL_AIR : primary.normal;                            // :366-370
    o.albedo = primary.air ? vec3{1, 1, 1} : primary.albedo;                            // :371-375
    o.emission = primary.air ? vec3{0, 0, 0} : primary.emission * (1.0f / 4.0f);        // :376-380 (x/4 == x*0.25 exactly)
    o.depth_f = primary.air ? 65535.0f : length(origin - primary.position) * 32;        // :356-359
    o.fog = sample_sky(ray_direction, fc.sunangle, fc.sunlight, false) * 0.5f;          // :381-385 (x/2 == x*0.5 exactly)
    return o;
}

}  // namespace

extern "C" {

struct RtOracleOut {        // any pointer may be NULL; planes are W*H row-major, row 0 = bottom
    uint16_t* lighting_rgba16;
    uint16_t* depth_r16;
    uint8_t* normal_r8;
    uint8_t* albedo_rgba8;
    uint8_t* emission_rgba8;
    uint8_t* fog_rgba8;
    float* lighting_f32;    // 4 floats / pixel
    float* fog_f32;         // 4 floats / pixel
    float* depth_f32;       // 1 float / pixel
};

// Render rows [y0, y1) of a W x H frame with `spp` samples and `depth` levels.  threads <= 0 => OpenMP default.
int rt_oracle_render_region(const uint32_t* materials, const uint8_t* minefield, const uint8_t* noise,
                            const RtUniforms* u, int region, int W, int H, int spp, int depth, int y0, int y1, int threads,
                            RtOracleOut* out, RtCounters* counters) {
    if (!materials || !minefield || !noise || !u || !out || W <= 0 || H <= 0 || spp < 1 || depth < 0 ||
        depth > RT_MAX_DEPTH || y0 < 0 || y1 > H || y0 > y1 || (region != 256 && region != 512 && region != 1024))
        return RT_ERR_INVALID_ARG;
    Scene sc{materials, minefield, noise, region};
    FrameConsts fc;
    fc.sunangle = sun_vector(u->sun_angle);       // raytrace.comp:317
    fc.sunlight = sun_color(fc.sunangle);         // raytrace.comp:318
    Counters total;
#ifdef _OPENMP
#pragma omp parallel num_threads(threads > 0 ? threads : omp_get_max_threads())
#endif
    {
        Counters cn;
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
        for (int py = y0; py < y1; py++) {
            for (int px = 0; px < W; px++) {
                vec3 sum = {0, 0, 0};
                PixelOut first{};
                for (int s = 0; s < spp; s++) {
                    uint32_t seed = (u->seed + (uint32_t)s) % (uint32_t)RT_NOISE_BYTES;  // pipeline.rs:201
                    PixelOut o = shade_pixel(sc, *u, fc, W, H, px, py, seed, depth, cn);
                    if (s == 0) first = o;
                    sum = sum + o.light;
                }
                float n = (float)spp;
                vec3 light = {sum.x / n, sum.y / n, sum.z / n};
                size_t i = (size_t)py * W + px;
                float lv[4] = {light.x / RT_LIGHTING_SCALE, light.y / RT_LIGHTING_SCALE,
                               light.z / RT_LIGHTING_SCALE, 1.0f / RT_LIGHTING_SCALE};   // :352-356
                float fv[4] = {first.fog.x, first.fog.y, first.fog.z, 1.0f};
                if (out->lighting_f32) memcpy(out->lighting_f32 + i * 4, lv, 16);
                if (out->fog_f32) memcpy(out->fog_f32 + i * 4, fv, 16);
                if (out->depth_f32) out->depth_f32[i] = first.depth_f;
                if (out->lighting_rgba16)
                    for (int c = 0; c < 4; c++) out->lighting_rgba16[i * 4 + c] = (uint16_t)rtm_unorm(lv[c], 65535.0f);
                if (out->depth_r16) out->depth_r16[i] = first.air ? RT_DEPTH_AIR : (uint16_t)rtm_f2u16(first.depth_f);
                if (out->normal_r8) out->normal_r8[i] = (uint8_t)first.normal;
                if (out->albedo_rgba8) {
                    float av[4] = {first.albedo.x, first.albedo.y, first.albedo.z, 1.0f};
                    for (int c = 0; c < 4; c++) out->albedo_rgba8[i * 4 + c] = (uint8_t)rtm_unorm(av[c], 255.0f);
                }
                if (out->emission_rgba8) {
                    float ev[4] = {first.emission.x, first.emission.y, first.emission.z, first.air ? 0.0f : 1.0f};
                    for (int c = 0; c < 4; c++) out->emission_rgba8[i * 4 + c] = (uint8_t)rtm_unorm(ev[c], 255.0f);
                }
                if (out->fog_rgba8)
                    for (int c = 0; c < 4; c++) out->fog_rgba8[i * 4 + c] = (uint8_t)rtm_unorm(fv[c], 255.0f);
            }
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        total.add(cn);
    }
    if (counters) {
        memset(counters, 0, sizeof(*counters));
        counters->rays = total.rays; counters->rays_primary = total.rays_primary;
        counters->rays_shadow = total.rays_shadow; counters->rays_diffuse = total.rays_diffuse;
        counters->iterations = total.iterations; counters->minefield_fetches = total.minefield_fetches;
        counters->material_fetches = total.material_fetches;
        counters->noise_fetches = total.noise_fetches + (uint64_t)spp;  // + one seed-base texel per frame (:302-303)
        counters->hits = total.hits; counters->sky_exits = total.sky_exits;
        counters->limit_exits = total.limit_exits; counters->border_fetches = total.border_fetches;
        counters->pixels = (uint64_t)(y1 - y0) * W; counters->frames = 1;
    }
    return RT_OK;
}

void rt_oracle_fetch_stats_begin(const uint8_t* uniform4) { g_uniform4 = uniform4; memset(g_fetch_hist, 0, sizeof(g_fetch_hist)); }
void rt_oracle_fetch_stats_end(uint64_t* out64) { memcpy(out64, g_fetch_hist, sizeof(g_fetch_hist)); g_uniform4 = nullptr; }

}  // extern "C"

// =====================================================================================================
// bilateral_denoise.comp and finalize.comp — restated for the parity tests of the
// post-passes.  Images are UNORM / UINT storage images: imageLoad returns u16/65535, u8/255 or the integer.
// =====================================================================================================
namespace {
struct Tap { int dx, dy; float w; };
// the 36 SAMPLE(...) lines of bilateral_denoise.comp:45-88, in source order
const Tap kTaps[36] = {
    {0, 1, 0.092566f}, {0, -1, 0.092566f}, {1, 0, 0.092566f}, {-1, 0, 0.092566f},
    {1, 1, 0.058434f}, {-1, 1, 0.058434f}, {-1, -1, 0.058434f}, {1, -1, 0.058434f},
    {2, 0, 0.023205f}, {-2, 0, 0.023205f}, {0, 2, 0.023205f}, {0, -2, 0.023205f},
    {2, 2, 0.003672f}, {-2, 2, 0.003672f}, {-2, -2, 0.003672f}, {2, -2, 0.003672f},
    {2, 1, 0.014648f}, {-2, 1, 0.014648f}, {-2, -1, 0.014648f}, {2, -1, 0.014648f},
    {1, 2, 0.014648f}, {-1, 2, 0.014648f}, {-1, -2, 0.014648f}, {1, -2, 0.014648f},
    {3, 0, 0.002289f}, {-3, 0, 0.002289f}, {0, 3, 0.002289f}, {0, -3, 0.002289f},
    {3, 1, 0.001445f}, {-3, 1, 0.001445f}, {-3, -1, 0.001445f}, {3, -1, 0.001445f},
    {1, 3, 0.001445f}, {-1, 3, 0.001445f}, {-1, -3, 0.001445f}, {1, -3, 0.001445f}};

// One dispatch of bilateral_denoise.comp.  `depth_img` / `normal_img` are what the shader's bindings 1 and 2 return:
// on the "pong" descriptor set the reference binds them SWAPPED (descriptor_sets.rs:38-39 vs :31-32), so binding 1
// (declared r16ui) reads the normal image and binding 2 (declared r8ui) reads the depth image; the loads are taken
// to return the bound image's own integer value.
void denoise_pass(const uint16_t* lin, const uint32_t* depth_img, const uint32_t* normal_img, int W, int H, int size,
                  uint16_t* lout) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t c = (size_t)y * W + x;
            float center_distance = (float)depth_img[c] / 256.0f;                                  // :36
            uint32_t center_normal = normal_img[c];                                                // :37
            if (center_normal < 16) {                                                              // :39
                float total_weight = 0.146634f;                                                    // :40
                float sum[3];
                for (int k = 0; k < 3; k++) sum[k] = ((float)lin[c * 4 + k] / 65535.0f) * total_weight;   // :41
                for (const Tap& t : kTaps) {                                                       // SAMPLE, :23-33
                    int px = x + t.dx * size, py = y + t.dy * size;                                // sampleAt, :14-21
                    if (px < 0) px = 0;
                    if (py < 0) py = 0;
                    if (px >= W) px = W - 1;
                    if (py >= H) py = H - 1;
                    const size_t i = (size_t)py * W + px;
                    float dist = (float)depth_img[i] / 256.0f;
                    float distance_difference = 4.0f * rtm_abs(center_distance - dist);
                    float normal_difference = normal_img[i] == center_normal ? 0.0f : 10.0f;
                    float weight = t.w / (distance_difference + normal_difference + 1.0f);
                    total_weight += weight;
                    for (int k = 0; k < 3; k++) sum[k] = rtm_fma((float)lin[i * 4 + k] / 65535.0f, weight, sum[k]);
                }
                for (int k = 0; k < 3; k++) lout[c * 4 + k] = (uint16_t)rtm_unorm(sum[k] / total_weight, 65535.0f);   // :89
                lout[c * 4 + 3] = 65535;
            } else {
                for (int k = 0; k < 4; k++) lout[c * 4 + k] = lin[c * 4 + k];                      // :91
            }
        }
}

// filmic_curve — finalize.comp:21-31
inline float filmic_curve(float x) {
    if (x < 0.3f) return x * x;
    if (x < 1.13333f) return rtm_fma(x, 0.6f, -0.09f);
    if (x < 2.5f) return rtm_fma(-0.219512195116f * (x - 2.5f), x - 2.5f, 1.0f);
    return 1.0f;
}
}  // namespace

extern "C" {
// The six denoise dispatches of pipeline.rs:98-115 (sizes 1,2,4,8,8,16; ping / pong descriptor sets alternate).
// lighting is updated in place (the sixth pass writes lighting_buffer).  faithful != 0 reproduces the swapped
// bindings of the pong set; faithful == 0 binds depth and normal the same way in every pass.
int rt_oracle_denoise(uint16_t* lighting_rgba16, const uint16_t* depth_r16, const uint8_t* normal_r8, int W, int H, int faithful) {
    const size_t n = (size_t)W * H;
    std::vector<uint32_t> depth(n), normal(n);
    for (size_t i = 0; i < n; i++) { depth[i] = depth_r16[i]; normal[i] = normal_r8[i]; }
    std::vector<uint16_t> pong(n * 4);
    const int sizes[6] = {1, 2, 4, 8, 8, 16};
    for (int pass = 0; pass < 6; pass++) {
        const bool odd = pass % 2 == 1;
        const uint16_t* in = odd ? pong.data() : lighting_rgba16;
        uint16_t* out = odd ? lighting_rgba16 : pong.data();
        const bool swapped = odd && faithful;
        denoise_pass(in, swapped ? normal.data() : depth.data(), swapped ? depth.data() : normal.data(), W, H, sizes[pass], out);
    }
    return RT_OK;
}

// finalize.comp:33-63.  Output is the swapchain image: B8G8R8A8_UNORM (core_builder.rs:557-568), rows top-down
// (the shader flips Y, :60-62).
int rt_oracle_finalize(const uint8_t* albedo_rgba8, const uint8_t* emission_rgba8, const uint8_t* fog_rgba8,
                       const uint16_t* lighting_rgba16, const uint16_t* depth_r16, const uint8_t* noise, int W, int H,
                       uint8_t* out_bgra8) {
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const size_t c = (size_t)y * W + x;
            float final_color[3];
            for (int k = 0; k < 3; k++) {
                float albedo = (float)albedo_rgba8[c * 4 + k] / 255.0f;
                float emission = ((float)emission_rgba8[c * 4 + k] / 255.0f) * 4.0f;                   // :37
                float light = ((float)lighting_rgba16[c * 4 + k] / 65535.0f) * RT_LIGHTING_SCALE;       // :39
                final_color[k] = rtm_fma(albedo, light, emission);                                      // :40
            }
            uint32_t depth = depth_r16[c];
            if (depth < 0xFFFFu) {                                                                      // :44-49
                float fog_amount = (float)depth / (32.0f * 128.0f * 8.0f);
                if (fog_amount > 1.0f) fog_amount = 1.0f;
                for (int k = 0; k < 3; k++) {
                    float fog = ((float)fog_rgba8[c * 4 + k] / 255.0f) * 2.0f;
                    final_color[k] = rtm_mix(final_color[k], fog, fog_amount);
                }
            }
            const uint8_t* nt = noise + ((size_t)(y % RT_NOISE_SIZE) * RT_NOISE_SIZE + (x % RT_NOISE_SIZE)) * 4;   // :55-57
            uint8_t rgb[3];
            for (int k = 0; k < 3; k++) {
                float v = filmic_curve(final_color[k]) + ((float)nt[k] / 255.0f) / 128.0f;              // :51-58
                rgb[k] = (uint8_t)rtm_unorm(v, 255.0f);
            }
            uint8_t* o = out_bgra8 + ((size_t)(H - y - 1) * W + x) * 4;                                 // :60-62
            o[0] = rgb[2]; o[1] = rgb[1]; o[2] = rgb[0]; o[3] = 255;
        }
    return RT_OK;
}
}  // extern "C"

extern "C" {
// ---- arithmetic-contract probes (tests/test_math_contract.py): fn 0 sin, 1 cos, 2 acos, 3 pow(x,y), 4 mod(x,y),
// 5 exp2, 6 log2, 7 sqrt, 8 1/x ----
void rt_oracle_math(int fn, const float* x, const float* y, float* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        switch (fn) {
            case 0: out[i] = rtm_sin(x[i]); break;
            case 1: out[i] = rtm_cos(x[i]); break;
            case 2: out[i] = rtm_acos(x[i]); break;
            case 3: out[i] = rtm_pow(x[i], y[i]); break;
            case 4: out[i] = rtm_mod(x[i], y[i]); break;
            case 5: out[i] = rtm_exp2(x[i]); break;
            case 6: out[i] = rtm_log2_pos(x[i]); break;
            case 7: out[i] = rtm_sqrt(x[i]); break;
            default: out[i] = 1.0f / x[i]; break;
        }
    }
}
void rt_oracle_normalize(const float* v3, float* out3) {
    rtm_vec3 o = rtm_normalize3({v3[0], v3[1], v3[2]});
    out3[0] = o.x; out3[1] = o.y; out3[2] = o.z;
}
// Texel coordinate (one axis) of the level-`level` noise_value lookup for a given noise_offset component
// (raytrace.comp:324,336), exactly as level_light() computes it.
int32_t rt_oracle_noise_level_texel(float noise_offset, int level) {
    float add = (float)(level - 1) * (2.0f / (float)RT_NOISE_SIZE);
    float c = rtm_floor(rtm_mod(noise_offset + add, (float)RT_NOISE_SIZE));
    return c < 0 ? 0 : (c > 511 ? 511 : (int32_t)c);
}
uint32_t rt_oracle_unorm(float x, float maxv) { return rtm_unorm(x, maxv); }
uint32_t rt_oracle_f2u16(float x) { return rtm_f2u16(x); }

// The reference's region size (ROOT_BLOCK_WIDTH = 256).
int rt_oracle_render(const uint32_t* materials, const uint8_t* minefield, const uint8_t* noise, const RtUniforms* u, int W,
                     int H, int spp, int depth, int y0, int y1, int threads, RtOracleOut* out, RtCounters* counters) {
    return rt_oracle_render_region(materials, minefield, noise, u, kDefaultRegion, W, H, spp, depth, y0, y1, threads, out, counters);
}

// ---- single-function entry points for the known-answer tests ---------------------------------
struct RtOracleHit {
    float albedo[3]; float emission[3]; int32_t air; float distance; uint32_t normal; float position[3];
    uint32_t packed_material; uint32_t iterations; uint32_t border_fetches; uint32_t limit_exit;
};

int rt_oracle_trace_ray(const uint32_t* materials, const uint8_t* minefield, const int32_t* lr,
                        const float* origin, const float* direction, RtOracleHit* out) {
    Scene sc{materials, minefield, nullptr};
    Counters cn;
    HitResult h = trace_ray(sc, lr, {origin[0], origin[1], origin[2]}, {direction[0], direction[1], direction[2]}, cn);
    out->albedo[0] = h.albedo.x; out->albedo[1] = h.albedo.y; out->albedo[2] = h.albedo.z;
    out->emission[0] = h.emission.x; out->emission[1] = h.emission.y; out->emission[2] = h.emission.z;
    out->air = h.air; out->distance = h.distance; out->normal = h.normal;
    out->position[0] = h.position.x; out->position[1] = h.position.y; out->position[2] = h.position.z;
    out->packed_material = h.packed_material; out->iterations = h.iterations;
    out->border_fetches = (uint32_t)cn.border_fetches; out->limit_exit = (uint32_t)cn.limit_exits;
    return RT_OK;
}

void rt_oracle_sun(float sun_angle, float* sunangle3, float* sunlight3) {
    vec3 a = sun_vector(sun_angle), c = sun_color(a);
    sunangle3[0] = a.x; sunangle3[1] = a.y; sunangle3[2] = a.z;
    sunlight3[0] = c.x; sunlight3[1] = c.y; sunlight3[2] = c.z;
}

void rt_oracle_sample_sky(const float* dir3, float sun_angle, int include_sun, float* out3) {
    vec3 a = sun_vector(sun_angle), c = sun_color(a);
    vec3 s = sample_sky({dir3[0], dir3[1], dir3[2]}, a, c, include_sun != 0);
    out3[0] = s.x; out3[1] = s.y; out3[2] = s.z;
}

void rt_oracle_diffuse_direction(uint32_t normal, const float* noise_rg, float* out3) {
    vec3 d = diffuse_direction(normal, {noise_rg[0], noise_rg[1], 0, 0});
    out3[0] = d.x; out3[1] = d.y; out3[2] = d.z;
}

// Forward thread -> pixel map of raytrace.comp:291-294 for one axis, and its inverse.
uint32_t rt_oracle_pixel_of(uint32_t workgroup, uint32_t local) {
    uint32_t p = workgroup - workgroup % RT_PIXEL_SPREAD;
    p *= RT_SHADER_GROUP_SIZE;
    p += workgroup % RT_PIXEL_SPREAD;
    p += local * RT_PIXEL_SPREAD;
    return p;
}
uint32_t rt_oracle_workgroup_of(uint32_t pixel) { return owning_workgroup(pixel); }

// noise_offset and noise_value texel coordinates for a pixel/seed (raytrace.comp:298-304,324).
void rt_oracle_noise_lookup(const uint8_t* noise, uint32_t seed, uint32_t px, uint32_t py, int32_t* base_texel2,
                            float* noise_offset2, int32_t* value_texel2, float* noise_value4) {
    Scene sc{nullptr, nullptr, noise};
    Counters cn;
    uint32_t bx = seed % RT_NOISE_SIZE, by = seed / RT_NOISE_SIZE;
    if (by > 511) by = 511;
    base_texel2[0] = (int32_t)bx; base_texel2[1] = (int32_t)by;
    vec4 base = noise_texel(sc, (float)(seed % RT_NOISE_SIZE), (float)(seed / RT_NOISE_SIZE), cn);
    float ox = base.r * 255.0f + (float)(owning_workgroup(px) * RT_SHADER_GROUP_SIZE);
    float oy = base.g * 255.0f + (float)(owning_workgroup(py) * RT_SHADER_GROUP_SIZE);
    noise_offset2[0] = ox; noise_offset2[1] = oy;
    float mx = rtm_mod(ox, 512.0f), my = rtm_mod(oy, 512.0f);
    value_texel2[0] = (int32_t)rtm_floor(mx); value_texel2[1] = (int32_t)rtm_floor(my);
    vec4 v = noise_texel(sc, mx, my, cn);
    noise_value4[0] = v.r; noise_value4[1] = v.g; noise_value4[2] = v.b; noise_value4[3] = v.a;
}

// compute_triple_euler_vector (src/util.rs:9-22) + the uniform fill of Pipeline::draw_frame
// (src/render/pipeline/pipeline.rs:191-207).  Rust's f32::cos/sin lower to the platform libm.
void rt_oracle_camera_uniforms(const float* origin3, float heading, float pitch, float sun_angle, uint32_t seed,
                               const int32_t* lr3, RtUniforms* u) {
    memset(u, 0, sizeof(*u));
    const float half_pi = 1.57079632679489661923f;  // std::f32::consts::FRAC_PI_2
    float fwd[3] = {cosf(heading) * cosf(pitch), sinf(heading) * cosf(pitch), sinf(pitch)};
    float up[3] = {cosf(heading) * cosf(pitch + half_pi), sinf(heading) * cosf(pitch + half_pi), sinf(pitch + half_pi)};
    // cgmath Vector3::cross: (a.y*b.z - a.z*b.y, a.z*b.x - a.x*b.z, a.x*b.y - a.y*b.x)
    float right[3] = {fwd[1] * up[2] - fwd[2] * up[1], fwd[2] * up[0] - fwd[0] * up[2], fwd[0] * up[1] - fwd[1] * up[0]};
    u->sun_angle = sun_angle;
    u->seed = seed;
    for (int a = 0; a < 3; a++) {
        u->origin[a] = origin3[a];
        u->forward[a] = fwd[a];
        u->up[a] = up[a] * 0.4f;        // pipeline.rs:198
        u->right[a] = right[a] * 0.4f;  // pipeline.rs:199
        u->lr[a] = lr3[a];              // pipeline.rs:204-207 (rotation == space_offset == render_offset)
        u->lso[a] = lr3[a];
    }
}

// Material::pack — src/render/GEN_MATERIALS.rs:44-51
uint32_t rt_oracle_pack_material(uint32_t r, uint32_t g, uint32_t b, int solid) {
    uint32_t albedo = r << 14 | g << 7 | b;
    return ((solid ? 1u : 0u) << 15) | albedo;
}

// UnpackedChunkData::pack_into — src/world/chunk.rs:125-184, on one 64^3 chunk.
// solid: u8[64^3] (x fastest), packed_in: u32[64^3] packed materials of the unpacked chunk.
void rt_oracle_pack_chunk(const uint8_t* solid, const uint32_t* packed_in, uint32_t* materials_out,
                          uint8_t* minefield_out) {
    const int C = RT_CHUNK_SIZE, V = C * C * C;
    std::vector<std::vector<uint8_t>> lods;                                   // :126-131
    for (int vol = V / 8; vol > 0; vol /= 8) lods.emplace_back((size_t)vol, 0);
    for (int index = 0; index < V; index++) {                                 // :133-152
        if (solid[index]) {
            int x = index % C, y = index / C % C, z = index / C / C;
            int lx = x / 2, ly = y / 2, lz = z / 2, stride = C / 2;
            for (auto& lod : lods) {
                size_t li = ((size_t)lz * stride + ly) * stride + lx;
                if (lod[li]) break;
                lod[li] = 1;
                lx /= 2; ly /= 2; lz /= 2; stride /= 2;
            }
        }
        materials_out[index] = packed_in[index];
    }
    if (!lods[RT_MAX_CHUNK_LOD - 1][0]) {                                     // :154-161
        for (int index = 0; index < V; index++) { materials_out[index] = 0; minefield_out[index] = RT_MAX_CHUNK_LOD; }
        return;
    }
    for (int index = 0; index < V; index++) {                                 // :163-183
        if (solid[index]) { minefield_out[index] = 0; continue; }
        int x = index % C, y = index / C % C, z = index / C / C;
        int lx = x / 2, ly = y / 2, lz = z / 2, stride = C / 2;
        uint8_t current = 1;
        for (auto& lod : lods) {
            size_t li = ((size_t)lz * stride + ly) * stride + lx;
            if (lod[li]) { minefield_out[index] = current; break; }
            lx /= 2; ly /= 2; lz /= 2; stride /= 2;
            current++;
        }
    }
}

}  // extern "C"
