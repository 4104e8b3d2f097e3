"""A second restatement of the two post shaders — shaders/glsl/bilateral_denoise.comp and finalize.comp — in vectorised float64
numpy, written from the GLSL text (and pipeline.rs:98-115 for the six dispatches), not from oracle/rt_oracle.cpp.  Test
infrastructure: tests/test_post_passes.py holds the oracle against it.  fp64 here against the shaders' fp32, so stores are
compared as real numbers against the stored integers (tolerances where the comparisons are made)."""
import numpy as np

# bilateral_denoise.comp:44-88 — (dx, dy, weight), in the shader's order
TAPS = ([(0, 1), (0, -1), (1, 0), (-1, 0)], 0.092566), ([(1, 1), (-1, 1), (-1, -1), (1, -1)], 0.058434), \
       ([(2, 0), (-2, 0), (0, 2), (0, -2)], 0.023205), ([(2, 2), (-2, 2), (-2, -2), (2, -2)], 0.003672), \
       ([(2, 1), (-2, 1), (-2, -1), (2, -1), (1, 2), (-1, 2), (-1, -2), (1, -2)], 0.014648), \
       ([(3, 0), (-3, 0), (0, 3), (0, -3)], 0.002289), \
       ([(3, 1), (-3, 1), (-3, -1), (3, -1), (1, 3), (-1, 3), (-1, -3), (1, -3)], 0.001445)
CENTER_WEIGHT = 0.146634
SIZES = (1, 2, 4, 8, 8, 16)          # pipeline.rs:103


def denoise_pass(lighting_u16, binding1, binding2, size):
    """One dispatch.  lighting_u16: [H, W, 4] RGBA16_UNORM texels; binding1 / binding2: the integer images the shader reads as
    `depth_buffer` and `normal_buffer` (which image that is, is the caller's business: the pong set swaps them).  Returns the real-
    valued RGB the shader hands to imageStore, times 65535, and the alpha texel; sky pixels (binding2 >= 16) are copied."""
    H, W = binding1.shape
    light = lighting_u16[..., :3].astype(np.float64) / 65535.0
    b1 = binding1.astype(np.float64)
    yy, xx = np.mgrid[0:H, 0:W]
    center_distance = b1 / 256.0
    total = np.full((H, W), CENTER_WEIGHT)
    acc = light * CENTER_WEIGHT
    for offsets, w in TAPS:
        for dx, dy in offsets:
            sx = np.clip(xx + dx * size, 0, W - 1)          # sampleAt (:14-21)
            sy = np.clip(yy + dy * size, 0, H - 1)
            dist = b1[sy, sx] / 256.0
            distance_difference = 4.0 * np.abs(center_distance - dist)
            normal_difference = np.where(binding2[sy, sx] == binding2, 0.0, 10.0)
            weight = w / (distance_difference + normal_difference + 1.0)
            total += weight
            acc += light[sy, sx] * weight[..., None]
    out = acc / total[..., None] * 65535.0
    sky = binding2 >= 16                                   # :41 center_normal < 16, else the texel is copied (:89-91)
    out[sky] = lighting_u16[..., :3][sky].astype(np.float64)
    alpha = np.where(sky, lighting_u16[..., 3], 65535)
    return out, alpha


def denoise(lighting_u16, depth_u16, normal_u8, faithful=True):
    """The six dispatches with the image quantised to RGBA16_UNORM in between.  faithful: odd dispatches use the pong descriptor
    set, which binds the normal image at binding 1 and the depth image at binding 2 (descriptor_sets.rs:31-32 against :38-39)."""
    cur = np.array(lighting_u16, dtype=np.uint16)
    d, n = depth_u16.astype(np.int64), normal_u8.astype(np.int64)
    for i, size in enumerate(SIZES):
        swapped = faithful and i % 2 == 1
        real, alpha = denoise_pass(cur, n if swapped else d, d if swapped else n, size)
        nxt = np.empty_like(cur)
        nxt[..., :3] = np.rint(np.clip(real, 0, 65535)).astype(np.uint16)
        nxt[..., 3] = alpha
        cur = nxt
    return cur


def filmic_curve(x):
    """finalize.comp:21-31."""
    return np.where(x < 0.3, x * x, np.where(x < 1.13333, x * 0.6 - 0.09, np.where(x < 2.5, 1.0 - 0.219512195116 * (x - 2.5) * (x - 2.5), 1.0)))


def finalize(albedo_rgba8, emission_rgba8, fog_rgba8, lighting_rgba16, depth_r16, noise_rgba):
    """finalize.comp:33-63 — the real-valued colour handed to imageStore, times 255, [H, W, 3] in R, G, B order, rows as stored
    (row H - 1 - y holds pixel y: :60-62)."""
    H, W = depth_r16.shape
    albedo = albedo_rgba8[..., :3].astype(np.float64) / 255.0
    emission = emission_rgba8[..., :3].astype(np.float64) / 255.0 * 4.0
    light = lighting_rgba16[..., :3].astype(np.float64) / 65535.0 * 16.0
    final = albedo * light + emission
    depth = depth_r16.astype(np.float64)
    fog_color = fog_rgba8[..., :3].astype(np.float64) / 255.0 * 2.0
    fog_amount = np.minimum(depth / (32.0 * 128.0 * 8.0), 1.0)[..., None]
    fogged = final * (1.0 - fog_amount) + fog_color * fog_amount
    final = np.where((depth_r16 < 0xFFFF)[..., None], fogged, final)
    final = filmic_curve(final)
    n = np.asarray(noise_rgba).reshape(512, 512, 4)
    yy, xx = np.mgrid[0:H, 0:W]
    final = final + n[yy % 512, xx % 512, :3].astype(np.float64) / 255.0 / 128.0
    return (np.clip(final, 0.0, 1.0) * 255.0)[::-1]
