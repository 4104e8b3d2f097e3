// rt_pslot.hpp — the ray slot and the branch-free DDA step shared by the path kernels of rt_paths.hip (k_paths, k_seq).
#pragma once
#include <hip/hip_runtime.h>

#include "rt_device.hpp"

namespace rtd {
namespace pslot {

// nk = iterations LEFT before the loop limit (raytrace.comp:109; 2048 for a fresh ray) | flags:
//   K_END   the ray has ended (or the slot is empty)
//   K_AIR   ... by leaving the region (:138-145)
//   K_DEAD  ... in p_arm: NaN direction or first texel outside the texture (one iteration, border fetch, Q12)
// A ray in flight has 1 <= nk <= 2048; nk == 0 is a ray at the loop limit (its next iteration ends it either way).
enum : uint32_t { K_END = 1u << 16, K_AIR = 1u << 17, K_DEAD = 1u << 18 };
constexpr uint32_t kFreshInvalid = 1u << 24;   // counting builds: the ray's first texel was outside the texture
// one swizzle table of a 2^LOGR region: 2 R entries (0..R used, R = the wrap to texel 0; an index is masked, never clamped)
template <int LOGR> constexpr uint32_t swz_bytes() { return 8u << LOGR; }
constexpr uint32_t kSwzBytes = swz_bytes<8>();
// Scrolled regions (lr != 0): table entry R — mod(p + R/2, R) came out as R itself, the sampler's border texel, value 0
// (Q7) — carries this flag instead of a texel's bits; it survives the OR of the three words (see p_advance, ps_border).
constexpr uint32_t kSwzBorder = 0x80000000u;

typedef __attribute__((address_space(3))) uint32_t lds_u32;

// One ray of trace_ray (raytrace.comp:82-183).  Direction negated (see rt_dda.hpp), (sx, sy, sz) = swizzle-table words of the
// current texel: its swizzled voxel index is sx | sy | sz.  axis = axis of the last step (tracked for diffuse rays only: a
// shadow ray's result is one bit).
struct PSlot {
    float px, py, pz, ndx, ndy, ndz, lx, ly, lz;
    uint32_t sx, sy, sz, nk, axis;
};
__device__ __forceinline__ uint32_t ps_vox(const PSlot& r) { return r.sx | r.sy | r.sz; }
__device__ __forceinline__ bool ps_border(uint32_t vox) { return (vox & kSwzBorder) != 0u; }
// nibble-map entry of a swizzled voxel index: byte address in the map and bit offset (0 or 4) of the nibble.  At R = 256 a coarse
// cube IS the 4^3 brick (entry = vox >> 6); larger regions take the top six bits of each brick coordinate.
template <int LOGR, bool LRZ = true>
__device__ __forceinline__ void ps_nibble_of(uint32_t vox, uint32_t* byte, uint32_t* shift) {
    if (LOGR == 8) { *byte = LRZ ? vox >> 7 : (vox >> 7) & 0x1FFFFu; *shift = (vox >> 4) & 4u; return; }   // the mask: kSwzBorder
    constexpr int LB = LOGR - 2, SUB = LOGR - 8;
    const uint32_t c = ((vox >> (6 + SUB)) & 63u) | ((vox >> (LB + SUB)) & (63u << 6)) | ((vox >> (2 * LB + SUB - 6)) & (63u << 12));
    *byte = c >> 1; *shift = (c & 1u) << 2;
}
__device__ __forceinline__ uint32_t ps_axis_of_code(uint32_t code) { return (code & 2u) ? ((code & 1u) ? 0u : 1u) : 2u; }   // p_advance<.., 2>
__device__ __forceinline__ bool ps_running(uint32_t nk) { return nk - 1u < (uint32_t)RT_TRACE_LIMIT; }   // 1 <= nk <= 2048

// Path state of one context.  st = shadow bits (bit j-1: the shadow ray of level j reached the sky) | PP_FINAL | level << 20
// (level 0 = no path); PP_FINAL (the same bit as K_AIR) = the level in flight is the path's last (level == depth), or there is
// no path: (F.nk | st) & K_AIR says whether the parked context needs pass_finish or pass_continue.
// ent = noise_value bytes (r, g) of the path | face id whose diffuse-table entry sits in F's direction registers << 16 (7 = none).
constexpr uint32_t PP_FINAL = K_AIR;
struct PPath { uint32_t st, item, ent; };

__device__ __forceinline__ float f_bits(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t u_bits(float f) { return __builtin_bit_cast(uint32_t, f); }

// One loop iteration (raytrace.comp:109-161) with the fetched value `st` of the slot's current texel, for the lanes whose
// ray is in flight and `enable`d; every other lane leaves the slot as it is.  No branches: a lane that does not move
// advances by t = 0 (fma(-nd, 0, p) == p), so its position, and with it the recomputed table words, stay put.
// GENERIC_Q: q for u of either sign (only a ray's first step can see u < 0 when lr = 0; p_arm takes that step).
// swz = LDS byte address of the three swizzle tables (2 KiB aligned).
// AXIS: 0 = the slot's axis word is not maintained (shadow rays), 1 = axis of the last step (0, 1, 2), 2 = the same as a
// two-bit code built with integer subtractions instead of compares and selects (ps_axis_of_code decodes it).
// LRZ = false: a scrolled region, lr = (lrx, lry, lrz) != 0 (terrain_upload.rs:84-275).  Positions then lie anywhere in
// [lr - R/2, lr + R/2): q needs the generic form, the sky test its subtraction, and the texel the shader's own
// mod(p + R/2, R) = fma(-R, floor(x / R), x) — whose one rounding can land on R itself (the border texel: kSwzBorder).
template <bool GENERIC_Q, int AXIS, int LOGR = 8, bool LRZ = true>
__device__ __forceinline__ void p_advance(PSlot& r, uint32_t st, bool enable, uint32_t swz, float lrx = 0.0f, float lry = 0.0f, float lrz = 0.0f) {
    constexpr float half = (float)(1 << (LOGR - 1));
    const uint32_t nk = r.nk;
    // in flight and below the loop limit (:109), on a value > 0 (:146): the ray moves.  Otherwise it has ended (or does so now:
    // hit, limit, or a fresh ray on a 0 — the pass tells them apart).
    const bool go = enable && ps_running(nk) && st != 0u;
    const uint32_t sb = (st << 23) + (126u << 23);               // float((1 << st) / 2)
    const float sz = f_bits(sb), is = f_bits(0x7F000000u - sb);  // is == 1 / sz exactly
    const float ux = r.px + half, uy = r.py + half, uz = r.pz + half;
    float qx, qy, qz;                                            // (pos + 128) * muls, :94-98,119
    if (GENERIC_Q) {
        qx = r.ndx < 0.0f ? -ux : ux; qy = r.ndy < 0.0f ? -uy : uy; qz = r.ndz < 0.0f ? -uz : uz;
    } else if (!LRZ) {   // u of either sign on every step: flip it where nd is negative (nd = -0: see rt_dda.hpp, that axis never wins)
        qx = f_bits(u_bits(ux) ^ (u_bits(r.ndx) & 0x80000000u));
        qy = f_bits(u_bits(uy) ^ (u_bits(r.ndy) & 0x80000000u));
        qz = f_bits(u_bits(uz) ^ (u_bits(r.ndz) & 0x80000000u));
    } else {
        qx = f_bits((u_bits(ux) & 0x7FFFFFFFu) | (u_bits(r.ndx) & 0x80000000u));
        qy = f_bits((u_bits(uy) & 0x7FFFFFFFu) | (u_bits(r.ndy) & 0x80000000u));
        qz = f_bits((u_bits(uz) & 0x7FFFFFFFu) | (u_bits(r.ndz) & 0x80000000u));
    }
    const float mx = __builtin_fmaf(-sz, rtm_floor(qx * is), qx);   // mod(q, sz): both products exact
    const float my = __builtin_fmaf(-sz, rtm_floor(qy * is), qy);
    const float mz = __builtin_fmaf(-sz, rtm_floor(qz * is), qz);
    const float tx = (0.0001f + mx) * r.lx, ty = (0.0001f + my) * r.ly, tz = (0.0001f + mz) * r.lz;   // :119
    // :120-136 — the smallest of the three (ties: z before y before x).  The t's of a ray in flight are positive and never
    // NaN (0.0001 + mod >= 0.0001, 1/|d| > 0), so min3 returns the value the shader's compare chain selects.
    const float t = __builtin_fminf(__builtin_fminf(tx, ty), tz);
    if (AXIS == 1) {
        const uint32_t ax = tz == t ? 2u : (tx < ty ? 0u : 1u);
        r.axis = go ? ax : r.axis;
    }
    if (AXIS == 2) {
        // the t's are positive floats, so their order is the order of their bit patterns: (a - b) >> 31 is "a < b"
        const uint32_t notz = (u_bits(t) - u_bits(tz)) >> 31;      // t <= tz always: 1 iff the step is not along z
        const uint32_t xlt = (u_bits(tx) - u_bits(ty)) >> 31;      // tx < ty
        const uint32_t code = notz + notz + xlt;
        r.axis = go ? code : r.axis;
    }
    const float te = go ? t : 0.0f;
    r.px = __builtin_fmaf(-r.ndx, te, r.px); r.py = __builtin_fmaf(-r.ndy, te, r.py); r.pz = __builtin_fmaf(-r.ndz, te, r.pz);
    // sky test (:138-145; with lr = 0 the subtraction p - lr is the identity).  max ignores a NaN operand like the three
    // compares would.
    const bool sky = LRZ ? __builtin_fmaxf(__builtin_fmaxf(rtm_abs(r.px), rtm_abs(r.py)), rtm_abs(r.pz)) >= half
                         : __builtin_fmaxf(__builtin_fmaxf(rtm_abs(r.px - lrx), rtm_abs(r.py - lry)), rtm_abs(r.pz - lrz)) >= half;
    const uint32_t moved = nk + (sky ? (K_AIR | K_END) - 1u : 0xFFFFFFFFu);
    r.nk = go ? moved : (nk | K_END);
    // table words of the next fetch's texel (:137): a position inside the bounds has mod(p + 128, 256) floor-identical to
    // (int)(p + 128), 256 = the wrap to texel 0.  4 * (p + 128) is fma(p, 4, 512) bit for bit (scaling by 4 commutes with the
    // rounding), its integer part with the low two bits masked is the byte offset of table entry (int)(p + 128) — and the mask
    // keeps the index of a ray that left the region (or of a lane with garbage) inside the 512-entry table.
    constexpr float four_half = 4.0f * half;
    constexpr uint32_t kMask = swz_bytes<LOGR>() - 4u, kTab = swz_bytes<LOGR>();     // R = 256: 0x7FC, 2048
    float x4 = __builtin_fmaf(r.px, 4.0f, four_half), y4 = __builtin_fmaf(r.py, 4.0f, four_half), z4 = __builtin_fmaf(r.pz, 4.0f, four_half);
    if (!LRZ) {   // 4 * mod(p + R/2, R), bit for bit (the scaling commutes with both roundings); in [0, 4 R]
        constexpr float four_r = 8.0f * half, inv_four_r = 1.0f / four_r;
        x4 = __builtin_fmaf(-four_r, rtm_floor(x4 * inv_four_r), x4);
        y4 = __builtin_fmaf(-four_r, rtm_floor(y4 * inv_four_r), y4);
        z4 = __builtin_fmaf(-four_r, rtm_floor(z4 * inv_four_r), z4);
    }
    const uint32_t ix = (uint32_t)(int)x4, iy = (uint32_t)(int)y4, iz = (uint32_t)(int)z4;
    r.sx = *(lds_u32*)(uintptr_t)((ix & kMask) | swz);
    r.sy = *(lds_u32*)(uintptr_t)((iy & kMask) | (swz + kTab));
    r.sz = *(lds_u32*)(uintptr_t)((iz & kMask) | (swz + 2u * kTab));
}

}  // namespace pslot
}  // namespace rtd
