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

// ---------------------------------------------------------------------------------------------------------------------------
// Round 3: the ray of k_paths.  What the step loop needs to know about a ray besides its numbers — is it in flight, along which
// axis did it step last — are WAVE MASKS (one bit per lane, an SGPR pair each) next to the ray, not flag bits inside a register:
// testing, merging and keeping them costs scalar instructions, where the flag word cost every slot and step a subtract, two
// compares, an OR and three selects (VALU issue is what bounds the loop: DESIGN.md 5).  (Per-lane `bool`s carried round the
// loop do not stay masks: the compiler turns them into 0/1 registers with a v_cndmask to write and a v_cmp to read each.)
//   run      the ray is in flight: armed and neither on a value 0 (hit), nor outside the region (sky), nor at the loop limit
//   az, axy  (diffuse rays) the last step went along z / had tx < ty: the face of the hit (raytrace.comp:120-136,166-180)
// nk = iterations LEFT before the loop limit (2048 for a fresh ray), a plain counter (counting builds: | kFreshInvalid).
// A ray that never moved (nk == RT_TRACE_LIMIT when it is consumed) is "special" (Q12): a fresh ray on a 0, a NaN direction, a
// first texel outside the texture.  Whether an ended ray reached the sky is read off its position when the pass consumes it.
struct PRay {
    float px, py, pz, ndx, ndy, ndz, lx, ly, lz;
    uint32_t sx, sy, sz, nk;
};
typedef uint64_t lanemask;
__device__ __forceinline__ bool lm_lane(lanemask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }   // this lane's bit (an SGPR-to-VCC copy)
__device__ __forceinline__ uint32_t pr_vox(const PRay& r) { return r.sx | r.sy | r.sz; }
template <int LOGR, bool LRZ>
__device__ __forceinline__ bool pr_outside(float px, float py, float pz, float lrx, float lry, float lrz) {   // :138-145; a NaN is not outside
    constexpr float half = (float)(1 << (LOGR - 1));
    return LRZ ? __builtin_fmaxf(__builtin_fmaxf(rtm_abs(px), rtm_abs(py)), rtm_abs(pz)) >= half
               : __builtin_fmaxf(__builtin_fmaxf(rtm_abs(px - lrx), rtm_abs(py - lry)), rtm_abs(pz - lrz)) >= half;
}

// One loop iteration (raytrace.comp:109-161) of ray r with the fetched value `st` of its current texel.  The arithmetic is that
// of p_advance (same q, mod, t, position update and table index — see there); a lane whose ray does not move keeps its numbers.
// CAREFUL: the wave has a ray within reach of the loop limit (never on terrain: a ray crosses a unit plane per iteration, <= 771
// at R = 256; a 1024^3 region filled with value 1 gets there) and tests nk itself; otherwise the counter only counts.
template <bool DIFFUSE, int LOGR, bool LRZ, bool CAREFUL>
__device__ __forceinline__ void p_step(PRay& r, lanemask& run, lanemask& az, lanemask& axy, uint32_t st, uint32_t swz, float lrx, float lry, float lrz) {
    constexpr float half = (float)(1 << (LOGR - 1));
    // (the markers keep the compiler from folding the careful and the plain step group of k_paths into shared blocks with the
    // counter test in both: two comments in the assembly, no instructions)
    if (CAREFUL) asm volatile("; careful step" ::: "memory");
    lanemask go = run & __ballot(st != 0u);
    if (CAREFUL) go &= __ballot((r.nk & 0xFFFFu) != 0u);
    uint32_t sb;                                                 // float((1 << st) / 2) = (st << 23) + bits(0.5), in one instruction
    asm("v_lshl_add_u32 %0, %1, 23, 0.5" : "=v"(sb) : "v"(st));  // (the compiler shifts once and adds twice: 7.3 issue cycles against 5.4)
    const float sz = f_bits(sb), is = f_bits(0x7F000000u - sb);  // is == 1 / sz exactly
    const float ux = r.px + half, uy = r.py + half, uz = r.pz + half;
    float qx, qy, qz;                                            // (pos + R/2) * muls, :94-98,119
    if (!LRZ) {
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
    const float t = __builtin_fminf(__builtin_fminf(tx, ty), tz);  // :120-136 (see p_advance)
    if (DIFFUSE) { az = (go & __ballot(tz == t)) | (az & ~go); axy = (go & __ballot(tx < ty)) | (axy & ~go); }
#ifdef RT_PSTEP_TE_SELECT   // diagnostic variant: selects instead of the exec mask
    {
        const bool gol = lm_lane(go);
        const float te = gol ? t : 0.0f;
        r.px = __builtin_fmaf(-r.ndx, te, r.px); r.py = __builtin_fmaf(-r.ndy, te, r.py); r.pz = __builtin_fmaf(-r.ndz, te, r.pz);
        r.nk -= gol ? 1u : 0u;
    }
#else
    // pos += dir * t (:121-135) and the iteration count, for the lanes that move: under their exec mask instead of a select per
    // value — two scalar instructions around four full-rate VALU ones.  (The compiler would emit v_cndmask t / 0 and 0 / 1.)
    {
        lanemask saved;
        asm("s_and_saveexec_b64 %[sv], %[go]\n\t"
            "v_fma_f32 %[px], -%[dx], %[t], %[px]\n\t"
            "v_fma_f32 %[py], -%[dy], %[t], %[py]\n\t"
            "v_fma_f32 %[pz], -%[dz], %[t], %[pz]\n\t"
            "v_add_u32_e32 %[nk], -1, %[nk]\n\t"
            "s_mov_b64 exec, %[sv]"
            : [px] "+v"(r.px), [py] "+v"(r.py), [pz] "+v"(r.pz), [nk] "+v"(r.nk), [sv] "=&s"(saved)
            : [go] "s"(go), [t] "v"(t), [dx] "v"(r.ndx), [dy] "v"(r.ndy), [dz] "v"(r.ndz)
            : "scc");
    }
#endif
    run = go & ~__ballot(pr_outside<LOGR, LRZ>(r.px, r.py, r.pz, lrx, lry, lrz));
    // table words of the next fetch's texel (:137), as in p_advance
    constexpr float four_half = 4.0f * half;
    constexpr uint32_t kMask = swz_bytes<LOGR>() - 4u, kTab = swz_bytes<LOGR>();
    float x4 = __builtin_fmaf(r.px, 4.0f, four_half), y4 = __builtin_fmaf(r.py, 4.0f, four_half), z4 = __builtin_fmaf(r.pz, 4.0f, four_half);
    if (!LRZ) {
        constexpr float four_r = 8.0f * half, inv_four_r = 1.0f / four_r;
        x4 = __builtin_fmaf(-four_r, rtm_floor(x4 * inv_four_r), x4);
        y4 = __builtin_fmaf(-four_r, rtm_floor(y4 * inv_four_r), y4);
        z4 = __builtin_fmaf(-four_r, rtm_floor(z4 * inv_four_r), z4);
    }
    const uint32_t ix = (uint32_t)(int)x4, iy = (uint32_t)(int)y4, iz = (uint32_t)(int)z4;
    r.sx = *(lds_u32*)(uintptr_t)((ix & kMask) | swz);
    r.sy = *(lds_u32*)(uintptr_t)((iy & kMask) | (swz + kTab));
    r.sz = *(lds_u32*)(uintptr_t)((iz & kMask) | (swz + 2u * kTab));
    if (CAREFUL) asm volatile("; careful step end" ::: "memory");
}

}  // namespace pslot
}  // namespace rtd
