// rt_post.hip — the two compute passes that follow the ray-trace dispatch in the reference's command buffer
// (src/render/pipeline/pipeline.rs:98-123): bilateral_denoise.comp (six dispatches) and finalize.comp.
// Streaming stencil / point kernels over the G-buffer planes; one thread per pixel, row-major, coalesced.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "rt_device.hpp"
#include "rt_kernels.hpp"

namespace rtd {

struct DenoiseTap { int dx, dy; float w; };
// the 36 SAMPLE(...) lines of bilateral_denoise.comp:45-88, in source order (the sum order is part of the result)
constexpr DenoiseTap kDenoiseTaps[36] = {
    {0, 1, 0.092566f}, {0, -1, 0.092566f}, {1, 0, 0.092566f}, {-1, 0, 0.092566f},
    {1, 1, 0.058434f}, {-1, 1, 0.058434f}, {-1, -1, 0.058434f}, {1, -1, 0.058434f},
    {2, 0, 0.023205f}, {-2, 0, 0.023205f}, {0, 2, 0.023205f}, {0, -2, 0.023205f},
    {2, 2, 0.003672f}, {-2, 2, 0.003672f}, {-2, -2, 0.003672f}, {2, -2, 0.003672f},
    {2, 1, 0.014648f}, {-2, 1, 0.014648f}, {-2, -1, 0.014648f}, {2, -1, 0.014648f},
    {1, 2, 0.014648f}, {-1, 2, 0.014648f}, {-1, -2, 0.014648f}, {1, -2, 0.014648f},
    {3, 0, 0.002289f}, {-3, 0, 0.002289f}, {0, 3, 0.002289f}, {0, -3, 0.002289f},
    {3, 1, 0.001445f}, {-3, 1, 0.001445f}, {-3, -1, 0.001445f}, {3, -1, 0.001445f},
    {1, 3, 0.001445f}, {-1, 3, 0.001445f}, {-1, -3, 0.001445f}, {1, -3, 0.001445f}};

// The six dispatches work on a 16-byte working pixel: the three lighting channels as the floats the shader's imageLoad
// returns (u16 / 65535, bilateral_denoise.comp:27,41) and one word of guide bits, depth | normal << 16 | computed << 24.
// A pixel is read by up to 37 taps per pass, so converting it once where it is produced (instead of once per tap)
// removes three IEEE divisions and two loads from every tap; the values are the same bit for bit.
constexpr uint32_t kDnComputed = 1u << 24;   // the pixel went through the filter branch at least once (alpha = 1.0, :89)

__global__ __launch_bounds__(256) void k_denoise_prepare(const ushort4* __restrict__ lin, const uint16_t* __restrict__ depth,
                                                         const uint8_t* __restrict__ normal, uint32_t n, uint4* __restrict__ out) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const ushort4 l = lin[i];
    uint4 o;
    o.x = __builtin_bit_cast(uint32_t, (float)l.x / 65535.0f);
    o.y = __builtin_bit_cast(uint32_t, (float)l.y / 65535.0f);
    o.z = __builtin_bit_cast(uint32_t, (float)l.z / 65535.0f);
    o.w = (uint32_t)depth[i] | (uint32_t)normal[i] << 16;
    out[i] = o;
}

// weight / (distance_difference + normal_difference + 1.0) (bilateral_denoise.comp:31) — the IEEE quotient, in 4 instructions
// instead of the 11 of the generic correctly-rounded sequence: one v_rcp_f32 and ONE residual correction.  That is exact on this
// kernel's domain — nine tap weights over the denominators k/64 + 1 and k/64 + 11, k = 0..65535 (depth differences are
// multiples of 1/256, the normal term is 0 or 10) — which k_selftest_dn_div checks exhaustively against the `/` operator on the
// device it runs on (tests/test_post_passes.py::test_denoise_division_is_exact_on_its_whole_domain).
__device__ __forceinline__ float dn_div(float w, float den) {
    const float rc = __builtin_amdgcn_rcpf(den);
    const float q0 = w * rc;
    const float r = rtm_fma(-q0, den, w);
    return rtm_fma(r, rc, q0);
}

// all (weight, k, normal term) triples of the domain above: counts the quotients that differ from IEEE division
__global__ __launch_bounds__(256) void k_selftest_dn_div(unsigned long long* mismatches) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;   // grid covers exactly 37 * 2 * 65536
    const uint32_t k = i & 0xFFFFu, c = (i >> 16) & 1u, t = i >> 17;
    const float w = t < 36u ? kDenoiseTaps[t].w : 0.146634f;
    const float den = (float)k / 64.0f + (c ? 10.0f : 0.0f) + 1.0f;
    const bool bad = __builtin_bit_cast(uint32_t, dn_div(w, den)) != __builtin_bit_cast(uint32_t, w / den);
    const uint64_t m = __ballot(bad);
    if ((threadIdx.x & 63u) == 0u && m) atomicAdd(mismatches, (unsigned long long)__popcll(m));
}

// One dispatch of bilateral_denoise.comp.  SWAPPED = the "pong" descriptor set, on which the reference binds the normal
// image to the shader's depth binding and the depth image to its normal binding (descriptor_sets.rs:38-39 vs :31-32).
// LAST = the sixth dispatch: also stores the RGBA16 lighting image the later passes read.
template <bool SWAPPED, bool LAST>
__global__ __launch_bounds__(256) void k_denoise_pass(const uint4* __restrict__ in, int W, int H, int size, uint4* __restrict__ out,
                                                      ushort4* __restrict__ lighting) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t c = (size_t)y * W + x;
    auto depth_binding = [](uint32_t g) -> uint32_t { return SWAPPED ? (g >> 16 & 0xFFu) : (g & 0xFFFFu); };
    auto normal_binding = [](uint32_t g) -> uint32_t { return SWAPPED ? (g & 0xFFFFu) : (g >> 16 & 0xFFu); };
    uint4 pc = in[c];
    const float center_distance = (float)depth_binding(pc.w) / 256.0f;                               // :36
    const uint32_t center_normal = normal_binding(pc.w);                                             // :37
    if (center_normal < 16u) {                                                                       // :39
        float total_weight = 0.146634f;                                                              // :40
        float sr = __builtin_bit_cast(float, pc.x) * total_weight, sg = __builtin_bit_cast(float, pc.y) * total_weight,
              sb = __builtin_bit_cast(float, pc.z) * total_weight;                                   // :41
#pragma unroll
        for (int t = 0; t < 36; t++) {                                                               // SAMPLE, :23-33
            int px = x + kDenoiseTaps[t].dx * size, py = y + kDenoiseTaps[t].dy * size;             // sampleAt, :14-21
            px = px < 0 ? 0 : (px >= W ? W - 1 : px);
            py = py < 0 ? 0 : (py >= H ? H - 1 : py);
            const uint4 l = in[(size_t)py * W + px];
            const float dist = (float)depth_binding(l.w) / 256.0f;
            const float distance_difference = 4.0f * rtm_abs(center_distance - dist);
            const float normal_difference = normal_binding(l.w) == center_normal ? 0.0f : 10.0f;
            const float weight = dn_div(kDenoiseTaps[t].w, distance_difference + normal_difference + 1.0f);
            total_weight += weight;
            sr = rtm_fma(__builtin_bit_cast(float, l.x), weight, sr);
            sg = rtm_fma(__builtin_bit_cast(float, l.y), weight, sg);
            sb = rtm_fma(__builtin_bit_cast(float, l.z), weight, sb);
        }
        const uint32_t qr = rtm_unorm(sr / total_weight, 65535.0f), qg = rtm_unorm(sg / total_weight, 65535.0f),
                       qb = rtm_unorm(sb / total_weight, 65535.0f);                                  // imageStore to RGBA16_UNORM, :89
        pc.x = __builtin_bit_cast(uint32_t, (float)qr / 65535.0f);
        pc.y = __builtin_bit_cast(uint32_t, (float)qg / 65535.0f);
        pc.z = __builtin_bit_cast(uint32_t, (float)qb / 65535.0f);
        pc.w |= kDnComputed;
        if (LAST) { ushort4 o; o.x = (uint16_t)qr; o.y = (uint16_t)qg; o.z = (uint16_t)qb; o.w = 65535; lighting[c] = o; }
    } else if (LAST) {                                                                               // :91 (copy)
        // u16 -> float -> u16 is the identity (q/65535 rounds back to q); alpha of a never-filtered pixel is the original one
        ushort4 o;
        o.x = (uint16_t)rtm_unorm(__builtin_bit_cast(float, pc.x), 65535.0f);
        o.y = (uint16_t)rtm_unorm(__builtin_bit_cast(float, pc.y), 65535.0f);
        o.z = (uint16_t)rtm_unorm(__builtin_bit_cast(float, pc.z), 65535.0f);
        o.w = (pc.w & kDnComputed) ? (uint16_t)65535 : lighting[c].w;
        lighting[c] = o;
    }
    if (!LAST) out[c] = pc;
}

// The same dispatch for tap spacings S = 1 and 2 with the workgroup's 32x8 output tile and its 3 S halo staged in LDS: a
// working pixel is fetched from L2 once per workgroup instead of once per tap (37x), the edge clamp of sampleAt (:14-21) is
// applied once while the tile is filled, and a tap is one ds_read_b128 at an immediate offset — no address arithmetic.  A
// workgroup none of whose pixels takes the filter branch (the pong dispatches, which test depth < 16) skips the tile.
// (Measured at 3840x2160: S = 1 89 us against 118 us direct; S = 4 and 8 lose — 116 and 355 us — because the halo outgrows the
// tile, so those spacings stay on k_denoise_pass.)
template <bool SWAPPED, bool LAST, int S>
__global__ __launch_bounds__(256) void k_denoise_tiled(const uint4* __restrict__ in, int W, int H, uint4* __restrict__ out,
                                                       ushort4* __restrict__ lighting) {
    constexpr int TW = 32, THt = 8, HALO = 3 * S, PW = TW + 2 * HALO, PH = THt + 2 * HALO;
    __shared__ uint4 tile[PH * PW];
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * THt;
    auto depth_binding = [](uint32_t g) -> uint32_t { return SWAPPED ? (g >> 16 & 0xFFu) : (g & 0xFFFFu); };
    auto normal_binding = [](uint32_t g) -> uint32_t { return SWAPPED ? (g & 0xFFFFu) : (g >> 16 & 0xFFu); };
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int x = x0 + tx, y = y0 + ty;
    const bool inside = x < W && y < H;
    const size_t c = (size_t)y * W + x;
    uint4 pc = make_uint4(0, 0, 0, 0);
    if (inside) pc = in[c];
    const uint32_t center_normal = normal_binding(pc.w);                                             // :37
    const bool filter = inside && center_normal < 16u;                                               // :39
    if (!__syncthreads_or(filter)) {   // the whole tile takes the copy branch (:91)
        if (!inside) return;
        if (LAST) {
            ushort4 o;
            o.x = (uint16_t)rtm_unorm(__builtin_bit_cast(float, pc.x), 65535.0f);
            o.y = (uint16_t)rtm_unorm(__builtin_bit_cast(float, pc.y), 65535.0f);
            o.z = (uint16_t)rtm_unorm(__builtin_bit_cast(float, pc.z), 65535.0f);
            o.w = (pc.w & kDnComputed) ? (uint16_t)65535 : lighting[c].w;
            lighting[c] = o;
        } else {
            out[c] = pc;
        }
        return;
    }
    for (int i = threadIdx.x; i < PW * PH; i += 256) {
        int gx = x0 + i % PW - HALO, gy = y0 + i / PW - HALO;
        gx = gx < 0 ? 0 : (gx >= W ? W - 1 : gx);
        gy = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
        tile[i] = in[(size_t)gy * W + gx];
    }
    __syncthreads();
    if (!inside) return;
    const uint4* ctr = tile + (ty + HALO) * PW + tx + HALO;
    const float center_distance = (float)depth_binding(pc.w) / 256.0f;                               // :36
    if (filter) {
        float total_weight = 0.146634f;                                                              // :40
        float sr = __builtin_bit_cast(float, pc.x) * total_weight, sg = __builtin_bit_cast(float, pc.y) * total_weight,
              sb = __builtin_bit_cast(float, pc.z) * total_weight;                                   // :41
#pragma unroll
        for (int t = 0; t < 36; t++) {                                                               // SAMPLE, :23-33
            const uint4 l = ctr[kDenoiseTaps[t].dy * S * PW + kDenoiseTaps[t].dx * S];
            const float dist = (float)depth_binding(l.w) / 256.0f;
            const float distance_difference = 4.0f * rtm_abs(center_distance - dist);
            const float normal_difference = normal_binding(l.w) == center_normal ? 0.0f : 10.0f;
            const float weight = dn_div(kDenoiseTaps[t].w, distance_difference + normal_difference + 1.0f);
            total_weight += weight;
            sr = rtm_fma(__builtin_bit_cast(float, l.x), weight, sr);
            sg = rtm_fma(__builtin_bit_cast(float, l.y), weight, sg);
            sb = rtm_fma(__builtin_bit_cast(float, l.z), weight, sb);
        }
        const uint32_t qr = rtm_unorm(sr / total_weight, 65535.0f), qg = rtm_unorm(sg / total_weight, 65535.0f),
                       qb = rtm_unorm(sb / total_weight, 65535.0f);                                  // imageStore to RGBA16_UNORM, :89
        pc.x = __builtin_bit_cast(uint32_t, (float)qr / 65535.0f);
        pc.y = __builtin_bit_cast(uint32_t, (float)qg / 65535.0f);
        pc.z = __builtin_bit_cast(uint32_t, (float)qb / 65535.0f);
        pc.w |= kDnComputed;
        if (LAST) { ushort4 o; o.x = (uint16_t)qr; o.y = (uint16_t)qg; o.z = (uint16_t)qb; o.w = 65535; lighting[c] = o; }
    } else if (LAST) {                                                                               // :91 (copy)
        ushort4 o;
        o.x = (uint16_t)rtm_unorm(__builtin_bit_cast(float, pc.x), 65535.0f);
        o.y = (uint16_t)rtm_unorm(__builtin_bit_cast(float, pc.y), 65535.0f);
        o.z = (uint16_t)rtm_unorm(__builtin_bit_cast(float, pc.z), 65535.0f);
        o.w = (pc.w & kDnComputed) ? (uint16_t)65535 : lighting[c].w;
        lighting[c] = o;
    }
    if (!LAST) out[c] = pc;
}

__device__ __forceinline__ float filmic_curve(float x) {   // finalize.comp:21-31
    if (x < 0.3f) return x * x;
    if (x < 1.13333f) return rtm_fma(x, 0.6f, -0.09f);
    if (x < 2.5f) return rtm_fma(-0.219512195116f * (x - 2.5f), x - 2.5f, 1.0f);
    return 1.0f;
}

// finalize.comp:33-63 -> the swapchain image, B8G8R8A8_UNORM (core_builder.rs:557-568), rows top-down.
__global__ __launch_bounds__(256) void k_finalize(const uint32_t* __restrict__ albedo, const uint32_t* __restrict__ emission,
                                                  const uint32_t* __restrict__ fog, const ushort4* __restrict__ lighting,
                                                  const uint16_t* __restrict__ depth, const uint32_t* __restrict__ noise, int W, int H,
                                                  uint32_t* __restrict__ out_bgra8) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t c = (size_t)y * W + x;
    const uint32_t a = albedo[c], e = emission[c], fg = fog[c], d = depth[c];
    const ushort4 l = lighting[c];
    const uint32_t nt = noise[(y % RT_NOISE_SIZE) * RT_NOISE_SIZE + (x % RT_NOISE_SIZE)];           // :55-57
    const uint32_t lv[3] = {l.x, l.y, l.z};
    float fog_amount = (float)d / (32.0f * 128.0f * 8.0f);                                           // :47
    if (fog_amount > 1.0f) fog_amount = 1.0f;
    uint32_t out = 0xFF000000u;
    for (int k = 0; k < 3; k++) {
        const float alb = unorm8(a, k), emi = unorm8(e, k) * 4.0f;                                   // :36-37
        const float light = ((float)lv[k] / 65535.0f) * RT_LIGHTING_SCALE;                           // :39
        float v = rtm_fma(alb, light, emi);                                                          // :40
        if (d < 0xFFFFu) v = rtm_mix(v, unorm8(fg, k) * 2.0f, fog_amount);                           // :44-49
        v = filmic_curve(v) + unorm8(nt, k) / 128.0f;                                                // :51-58
        out |= rtm_unorm(v, 255.0f) << (8 * (2 - k));                                                // B in the low byte
    }
    out_bgra8[(size_t)(H - y - 1) * W + x] = out;                                                    // :60-62 (Y flip)
}

hipError_t launch_denoise_prepare(const void* lighting, const void* depth, const void* normal, int W, int H, void* work,
                                  hipStream_t st) {
    const uint32_t n = (uint32_t)W * (uint32_t)H;
    hipLaunchKernelGGL(k_denoise_prepare, dim3((n + 255u) / 256u), dim3(256), 0, st, (const ushort4*)lighting, (const uint16_t*)depth,
                       (const uint8_t*)normal, n, (uint4*)work);
    return hipGetLastError();
}

hipError_t launch_denoise(const void* work_in, int W, int H, int size, bool swapped, bool last, void* work_out, void* lighting,
                          hipStream_t st) {
    dim3 block(256);
    if ((size == 1 || size == 2) && !getenv("RT_DENOISE_UNTILED")) {
        // LDS-tiled dispatch (the reference runs size 1 on the ping set and size 2 on the pong set, pipeline.rs:103; every
        // combination is instantiated so the entry point stays general)
        dim3 grid((W + 31) / 32, (H + 7) / 8);
#define RT_LAUNCH_DT(SW, L, S) hipLaunchKernelGGL((k_denoise_tiled<SW, L, S>), grid, block, 0, st, (const uint4*)work_in, W, H, (uint4*)work_out, (ushort4*)lighting)
#define RT_LAUNCH_DT_S(SW, L) do { if (size == 1) RT_LAUNCH_DT(SW, L, 1); else RT_LAUNCH_DT(SW, L, 2); } while (0)
        if (swapped) { if (last) RT_LAUNCH_DT_S(true, true); else RT_LAUNCH_DT_S(true, false); }
        else { if (last) RT_LAUNCH_DT_S(false, true); else RT_LAUNCH_DT_S(false, false); }
#undef RT_LAUNCH_DT_S
#undef RT_LAUNCH_DT
        return hipGetLastError();
    }
    dim3 grid((W + 63) / 64, (H + 3) / 4);
#define RT_LAUNCH_DN(S, L) hipLaunchKernelGGL((k_denoise_pass<S, L>), grid, block, 0, st, (const uint4*)work_in, W, H, size, (uint4*)work_out, (ushort4*)lighting)
    if (swapped) { if (last) RT_LAUNCH_DN(true, true); else RT_LAUNCH_DN(true, false); }
    else { if (last) RT_LAUNCH_DN(false, true); else RT_LAUNCH_DN(false, false); }
#undef RT_LAUNCH_DN
    return hipGetLastError();
}

hipError_t launch_selftest_dn_div(unsigned long long* mismatches_dev, hipStream_t st) {
    hipLaunchKernelGGL(k_selftest_dn_div, dim3(37 * 2 * 65536 / 256), dim3(256), 0, st, mismatches_dev);
    return hipGetLastError();
}

hipError_t launch_finalize(const void* albedo, const void* emission, const void* fog, const void* lighting, const void* depth,
                           const uint32_t* noise, int W, int H, void* out_bgra8, hipStream_t st) {
    dim3 grid((W + 63) / 64, (H + 3) / 4), block(256);
    hipLaunchKernelGGL(k_finalize, grid, block, 0, st, (const uint32_t*)albedo, (const uint32_t*)emission, (const uint32_t*)fog,
                       (const ushort4*)lighting, (const uint16_t*)depth, noise, W, H, (uint32_t*)out_bgra8);
    return hipGetLastError();
}

}  // namespace rtd
