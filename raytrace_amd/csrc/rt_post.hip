// rt_post.hip — the two compute passes that follow the ray-trace dispatch in the reference's command buffer
// (src/render/pipeline/pipeline.rs:98-123): bilateral_denoise.comp (six dispatches) and finalize.comp.
// Streaming stencil / point kernels over the G-buffer planes; one thread per pixel, row-major, coalesced.
#include <hip/hip_runtime.h>

#include "rt_device.hpp"
#include "rt_kernels.hpp"

namespace rtd {

struct DenoiseTap { int dx, dy; float w; };
// the 36 SAMPLE(...) lines of bilateral_denoise.comp:45-88, in source order (the sum order is part of the result)
__constant__ DenoiseTap kDenoiseTaps[36] = {
    {0, 1, 0.092566f}, {0, -1, 0.092566f}, {1, 0, 0.092566f}, {-1, 0, 0.092566f},
    {1, 1, 0.058434f}, {-1, 1, 0.058434f}, {-1, -1, 0.058434f}, {1, -1, 0.058434f},
    {2, 0, 0.023205f}, {-2, 0, 0.023205f}, {0, 2, 0.023205f}, {0, -2, 0.023205f},
    {2, 2, 0.003672f}, {-2, 2, 0.003672f}, {-2, -2, 0.003672f}, {2, -2, 0.003672f},
    {2, 1, 0.014648f}, {-2, 1, 0.014648f}, {-2, -1, 0.014648f}, {2, -1, 0.014648f},
    {1, 2, 0.014648f}, {-1, 2, 0.014648f}, {-1, -2, 0.014648f}, {1, -2, 0.014648f},
    {3, 0, 0.002289f}, {-3, 0, 0.002289f}, {0, 3, 0.002289f}, {0, -3, 0.002289f},
    {3, 1, 0.001445f}, {-3, 1, 0.001445f}, {-3, -1, 0.001445f}, {3, -1, 0.001445f},
    {1, 3, 0.001445f}, {-1, 3, 0.001445f}, {-1, -3, 0.001445f}, {1, -3, 0.001445f}};

// One dispatch of bilateral_denoise.comp.  SWAPPED = the "pong" descriptor set, on which the reference binds the normal
// image to the shader's depth binding and the depth image to its normal binding (descriptor_sets.rs:38-39 vs :31-32).
template <bool SWAPPED>
__global__ __launch_bounds__(256) void k_denoise_pass(const ushort4* __restrict__ lin, const uint16_t* __restrict__ depth,
                                                      const uint8_t* __restrict__ normal, int W, int H, int size,
                                                      ushort4* __restrict__ lout) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t c = (size_t)y * W + x;
    auto depth_binding = [&](size_t i) -> uint32_t { return SWAPPED ? (uint32_t)normal[i] : (uint32_t)depth[i]; };
    auto normal_binding = [&](size_t i) -> uint32_t { return SWAPPED ? (uint32_t)depth[i] : (uint32_t)normal[i]; };
    const float center_distance = (float)depth_binding(c) / 256.0f;                                  // :36
    const uint32_t center_normal = normal_binding(c);                                                // :37
    const ushort4 lc = lin[c];
    if (center_normal < 16u) {                                                                       // :39
        float total_weight = 0.146634f;                                                              // :40
        float sr = ((float)lc.x / 65535.0f) * total_weight, sg = ((float)lc.y / 65535.0f) * total_weight,
              sb = ((float)lc.z / 65535.0f) * total_weight;                                          // :41
#pragma unroll 4
        for (int t = 0; t < 36; t++) {                                                               // SAMPLE, :23-33
            int px = x + kDenoiseTaps[t].dx * size, py = y + kDenoiseTaps[t].dy * size;             // sampleAt, :14-21
            px = px < 0 ? 0 : (px >= W ? W - 1 : px);
            py = py < 0 ? 0 : (py >= H ? H - 1 : py);
            const size_t i = (size_t)py * W + px;
            const float dist = (float)depth_binding(i) / 256.0f;
            const float distance_difference = 4.0f * rtm_abs(center_distance - dist);
            const float normal_difference = normal_binding(i) == center_normal ? 0.0f : 10.0f;
            const float weight = kDenoiseTaps[t].w / (distance_difference + normal_difference + 1.0f);
            total_weight += weight;
            const ushort4 l = lin[i];
            sr = rtm_fma((float)l.x / 65535.0f, weight, sr);
            sg = rtm_fma((float)l.y / 65535.0f, weight, sg);
            sb = rtm_fma((float)l.z / 65535.0f, weight, sb);
        }
        ushort4 o;
        o.x = (uint16_t)rtm_unorm(sr / total_weight, 65535.0f); o.y = (uint16_t)rtm_unorm(sg / total_weight, 65535.0f);
        o.z = (uint16_t)rtm_unorm(sb / total_weight, 65535.0f); o.w = 65535;                         // :89
        lout[c] = o;
    } else {
        lout[c] = lc;                                                                                // :91
    }
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

hipError_t launch_denoise(const void* lighting_in, const void* depth, const void* normal, int W, int H, int size, bool swapped,
                          void* lighting_out, hipStream_t st) {
    dim3 grid((W + 63) / 64, (H + 3) / 4), block(256);
    if (swapped) hipLaunchKernelGGL(k_denoise_pass<true>, grid, block, 0, st, (const ushort4*)lighting_in, (const uint16_t*)depth,
                                    (const uint8_t*)normal, W, H, size, (ushort4*)lighting_out);
    else hipLaunchKernelGGL(k_denoise_pass<false>, grid, block, 0, st, (const ushort4*)lighting_in, (const uint16_t*)depth,
                            (const uint8_t*)normal, W, H, size, (ushort4*)lighting_out);
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
