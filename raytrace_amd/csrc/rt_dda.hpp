// rt_dda.hpp — one ray of trace_ray (raytrace.comp:82-183) in the form the persistent kernels step it: the minefield value
// of the ray's current texel is looked up by the caller (nibble map in LDS, byte array behind it), dda_advance takes one
// loop iteration (:109-161) with that value.  Values are those of the shader under the rt_math.h contract; only the
// bookkeeping differs (see RaySlot2).
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rt_device.hpp"

namespace rtd {

enum : uint32_t { PX_AIR = 0, PX_HIT = 1, PX_LIMIT = 2, PX_SPECIAL = 3 };   // how a ray ended

// Ray state in registers.
// * The direction is kept NEGATED: with lr = 0, q = (d > 0 ? -u : u) (raytrace.comp:94-98,119) is then one v_bfi of u with
//   the sign of nd (u = p + half > 0 inside the region), the position update fma(d, t, p) is fma(-nd, t, p) with a free
//   source modifier, and no per-axis sign words are needed.  (For d = +0 the sign of q differs from the shader's; that
//   axis has 1/|d| = inf, so its boundary distance is inf either way and is never the minimum.)
// * nk = iteration count | kind << 16 (kind is written when the ray ends); a fresh ray has nk == 0.
struct RaySlot2 {
    float px, py, pz, ndx, ndy, ndz, lx, ly, lz, ux, uy, uz;
    uint32_t vox, cidx, nk, axis;   // vox: swizzled voxel index of the current texel; cidx: its nibble-map entry (R > 256)
    bool tracing, valid, fresh_invalid;   // valid: only the lr != 0 build steps with it; fresh_invalid: counting build only
};
__device__ __forceinline__ uint32_t r2_kind(const RaySlot2& r) { return r.nk >> 16; }

// Minefield value of the ray's current texel: nibble map first (entry b is nibble (b & 1) of byte b >> 1), byte array for
// cubes whose voxels differ.  The step loops inline this so that the loads of several rays overlap.
template <int LOGR>
__device__ __forceinline__ uint32_t dda_lookup(const RaySlot2& r, const uint8_t* s_nib, const Scene& sc) {
    const uint32_t b = LOGR == 8 ? r.vox >> 6 : r.cidx;   // at R = 256 a coarse cube IS the 4^3 brick
    uint32_t st = (s_nib[b >> 1] >> ((b & 1u) << 2)) & 15u;
    if (st == kNibMixed) st = sc.mine[r.vox];
    return st;
}

// One loop iteration (:109-161) with the fetched value `step` of the current texel.  GENERIC_Q: q for u of either sign
// (always used when lr != 0; with lr = 0 only a ray's first step can see u < 0 — dda_arm takes that step itself).
// s_swz (R = 256 with lr = 0 only, else unused): LDS tables [3][kSwzStride] of the swizzled-index contribution of each
// coordinate value 0..256 (256 = the wrap to texel 0), so the index of the next texel is three table reads OR-ed
// together instead of 14 shift/mask operations.
constexpr int kSwzStride = 260;
template <int LOGR, bool LRZ>
constexpr bool dda_uses_swz() { return LOGR == 8 && LRZ; }
__device__ __forceinline__ void dda_fill_swz(uint32_t* s_swz, uint32_t tid, uint32_t nthreads) {
    for (uint32_t i = tid; i < 3u * 257u; i += nthreads) {
        const uint32_t a = i / 257u, v = (i % 257u) & 255u;
        s_swz[a * kSwzStride + i % 257u] = ((v & 3u) << (2u * a)) | ((v >> 2) << (6u + 6u * a));
    }
}
template <int LOGR, bool LRZ, bool COUNT, bool GENERIC_Q, bool SWZ = dda_uses_swz<LOGR, LRZ>()>
__device__ __forceinline__ void dda_advance(RaySlot2& r, uint32_t step, const Frame& f, float half, unsigned long long& c_border,
                                            const uint32_t* s_swz) {
    constexpr int R = 1 << LOGR, LB = LOGR - 2;
    if (!LRZ && !r.valid) step = 0u;
    if (step == 0u) {
        // a fresh ray on a 0 has step_size 0 => mod(x,0) = NaN (defined outcome), otherwise a hit (:146-160)
        r.nk = r.nk == 0u ? (1u | PX_SPECIAL << 16) : (r.nk | PX_HIT << 16);
        r.tracing = false;
    } else if (r.nk == (uint32_t)RT_TRACE_LIMIT) {
        r.nk |= PX_LIMIT << 16; r.tracing = false;                                                  // :109 (Q8)
    } else {
        const uint32_t sb = (step << 23) + (126u << 23);          // float((1 << step) / 2)
        const float sz = __builtin_bit_cast(float, sb);
        const float is = __builtin_bit_cast(float, 0x7F000000u - sb);   // exactly 1/sz
        float qx, qy, qz;
        if (GENERIC_Q || !LRZ) {
            qx = r.ndx < 0.0f ? -r.ux : r.ux; qy = r.ndy < 0.0f ? -r.uy : r.uy; qz = r.ndz < 0.0f ? -r.uz : r.uz;
        } else {
            qx = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, r.ux) & 0x7FFFFFFFu) | (__builtin_bit_cast(uint32_t, r.ndx) & 0x80000000u));
            qy = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, r.uy) & 0x7FFFFFFFu) | (__builtin_bit_cast(uint32_t, r.ndy) & 0x80000000u));
            qz = __builtin_bit_cast(float, (__builtin_bit_cast(uint32_t, r.uz) & 0x7FFFFFFFu) | (__builtin_bit_cast(uint32_t, r.ndz) & 0x80000000u));
        }
        const float mx = __builtin_fmaf(-sz, rtm_floor(qx * is), qx);   // == q - sz*floor(q/sz): both products exact
        const float my = __builtin_fmaf(-sz, rtm_floor(qy * is), qy);
        const float mz = __builtin_fmaf(-sz, rtm_floor(qz * is), qz);
        const float tx = (0.0001f + mx) * r.lx, ty = (0.0001f + my) * r.ly, tz = (0.0001f + mz) * r.lz;   // :119
        const bool xy = tx < ty;                                                                        // :120-136
        const float m1 = xy ? tx : ty;
        const bool useZ = !(m1 < tz);
        const float t = useZ ? tz : m1;
        r.axis = useZ ? 2u : (xy ? 0u : 1u);
        r.px = __builtin_fmaf(-r.ndx, t, r.px); r.py = __builtin_fmaf(-r.ndy, t, r.py); r.pz = __builtin_fmaf(-r.ndz, t, r.pz);   // fused (rt_math.h contract)
        r.nk++;
        r.ux = r.px + half; r.uy = r.py + half; r.uz = r.pz + half;
        // sky test (:138-145), then the address of the next fetch (:137).  With lr = 0 the subtraction p - lr is the identity,
        // and a position inside the bounds has mod(p + half, R) floor-identical to (int)(p + half) & (R - 1).
        const bool sky = LRZ ? (rtm_abs(r.px) >= half || rtm_abs(r.py) >= half || rtm_abs(r.pz) >= half)
                             : (rtm_abs(r.px - f.lr[0]) >= half || rtm_abs(r.py - f.lr[1]) >= half || rtm_abs(r.pz - f.lr[2]) >= half);
        if (sky) {
            r.nk |= PX_AIR << 16; r.tracing = false;
        } else if (SWZ) {
            r.vox = s_swz[(int)r.ux] | s_swz[kSwzStride + (int)r.uy] | s_swz[2 * kSwzStride + (int)r.uz];   // u in [0, 256]
        } else if (LRZ) {
            const int ix = (int)r.ux & (R - 1), iy = (int)r.uy & (R - 1), iz = (int)r.uz & (R - 1);
            r.vox = swizzled_index(ix, iy, iz, LB);
            if (LOGR != 8) r.cidx = coarse_index(ix, iy, iz, LOGR);
        } else {
            int ix, iy, iz;
            r.valid = wrap_texel(v3(r.px, r.py, r.pz), (float)R, &ix, &iy, &iz);
            if (COUNT && !r.valid) c_border++;
            r.vox = swizzled_index(ix, iy, iz, LB);
            if (LOGR != 8) r.cidx = coarse_index(ix, iy, iz, LOGR);
        }
    }
}

// Head of trace_ray (:83-107) for a ray with direction d (already normalized, :83) from origin ro whose first texel is
// (vox0, cidx0), ok = that texel is inside the texture.  r.l* (1/|d|, :88) must be set by the caller.  DIRECT (k_frame): there
// is no nibble map in LDS (s_nib unused), the value is the byte of the array.
template <int LOGR, bool LRZ, bool COUNT, bool DIRECT = false, bool SWZ = dda_uses_swz<LOGR, LRZ>()>
__device__ __forceinline__ void dda_arm(RaySlot2& r, float dx, float dy, float dz, float rox, float roy, float roz, bool ok,
                                        uint32_t vox0, uint32_t cidx0, const Frame& f, float half, const uint8_t* s_nib,
                                        const Scene& sc, unsigned long long& c_border, const uint32_t* s_swz) {
    r.px = rox; r.py = roy; r.pz = roz;
    r.ndx = -dx; r.ndy = -dy; r.ndz = -dz;
    r.ux = rox + half; r.uy = roy + half; r.uz = roz + half;
    r.valid = true; r.fresh_invalid = !ok; r.vox = vox0; r.cidx = cidx0;
    r.nk = 0; r.axis = 2; r.tracing = true;
    // NaN direction, or a first texel outside the texture (border value 0: step_size 0 on a fresh ray): the ray ends at once
    if (dx != dx || dy != dy || dz != dz || !ok) { r.nk = 1u | PX_SPECIAL << 16; r.tracing = false; }
    if (LRZ && __builtin_expect(r.tracing && (r.ux < 0.0f || r.uy < 0.0f || r.uz < 0.0f), 0))   // rare (origin outside the region): see dda_advance
        dda_advance<LOGR, LRZ, COUNT, true, SWZ>(r, DIRECT ? (uint32_t)sc.mine[r.vox] : dda_lookup<LOGR>(r, s_nib, sc), f, half, c_border, s_swz);
}

// Exact counters of one finished ray (the counting build's share of SURVEY 8d's integers).
struct RayTally { unsigned long long iter = 0, hits = 0, sky = 0, limit = 0, border = 0; };
template <int LOGR>
__device__ __forceinline__ void dda_tally(const RaySlot2& r, RayTally& c) {
    const uint32_t kind = r2_kind(r);
    c.iter += r.nk & 0xFFFFu;
    if (kind == PX_AIR) {
        c.sky++;
        int tx, ty, tz;   // the fetch the shader makes before its sky test may hit the border
        if (!wrap_texel(v3(r.px, r.py, r.pz), (float)(1 << LOGR), &tx, &ty, &tz)) c.border++;
    } else if (kind == PX_LIMIT) c.limit++;
    else c.hits++;
    if (kind == PX_SPECIAL) c.border += 1u + (r.fresh_invalid ? 1u : 0u);
    else if (r.fresh_invalid) c.border++;
}

}  // namespace rtd
