// rt_persist.hip — the primary prepass, the direction tables, the ordered accumulation, and round 1's path kernel
// (k_persist).  The default path kernel of big launches is k_paths (rt_paths.hip); k_persist runs the small ones,
// frames with lr != 0 and frames without the primary cache.
//
//   k_primary2 : primary prepass.  One 1024-thread workgroup per CU, nibble map of the scene in LDS; a wave walks whole
//                8x8 tiles (lane = pixel, so the wave's rays are coherent): primary ray, the five primary-only G-buffer
//                planes, finished lighting for sky pixels, and a __ballot-compacted worklist of the pixels that still
//                need shadow/diffuse rays together with their primary hits (SoA in HBM).  (k_primary: the same with
//                one thread per pixel over the byte array, RT_PRIMARY_V=1.)
//   k_persist  : persistent wave64 path kernel.  One 1024-thread workgroup per CU stays resident for the frame and
//                shares a 128 KiB nibble map of the scene in LDS.  A lane owns one path (pixel, sample) at a time and
//                keeps its whole state in registers: level, shadow bits, and the level's TWO rays (shadow + diffuse)
//                which it walks together.  All lanes run the same DDA step loop.  A lane whose rays have ended parks;
//                when `threshold` lanes of the wave are parked (__ballot) the wave runs ONE transition pass for all of
//                them — consume the results, shade, start the next level (directions come from tables), or finish
//                the path and pull the next one from the wave's 128-path chunk of its XCD group's cursor (one atomicAdd per
//                chunk, paths dealt out ballot-ranked) — so shading runs on a well-filled wave and the step loop on
//                compacted work.  Per path one 12-byte light record goes to HBM; k_accumulate_paths adds a pixel's
//                samples in order.
//   (k_persist2, round 1's regrouping with two paths per lane and one slot each, was retired in round 3: k_paths and k_seq
//   supersede it on every axis.)
//
// The primary ray does not depend on the seed (raytrace.comp:306-320 reads no noise), so with RT_FLAG_CACHE_PRIMARY
// it is traced once per pixel (prepass) and every sample starts at its first shadow ray.  Without the flag the path
// kernels (CACHE=false) walk every pixel themselves and re-trace the primary ray for each sample (the reference's
// ray count; used for counter parity).
//
// Values are those of raytrace.comp; only the grouping of the work differs (see the loop note in rt_kernels.hip).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rt_dda.hpp"
#include "rt_device.hpp"
#include "rt_kernels.hpp"

namespace rtd {

// Unit-sphere point of diffuse_direction (raytrace.comp:190-197) for every (noise.r, noise.g) byte pair.
__global__ __launch_bounds__(256) void k_build_sphere_lut(float4* __restrict__ lut) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;   // grid covers exactly 65536
    const float nr = (float)(i & 255u) / 255.0f, ng = (float)(i >> 8) / 255.0f;
    float theta1 = RTM_PI * 2.0f * nr;
    float theta2 = rtm_acos(1.0f - 2.0f * ng);
    float s1, c1, s2, c2;
    rtm_sincos(theta1, &s1, &c1);
    rtm_sincos(theta2, &s2, &c2);
    lut[i] = make_float4(s1 * s2, c1 * s2, c2, 0.0f);
}

// Whole head of trace_ray for a diffuse ray, per (face id, noise byte pair).  One 64-byte line per entry:
// 4i = diffuse_direction (the sample_sky argument, raytrace.comp:331), 4i+1 = normalize of it (:83),
// 4i+2 = length_per_axis (:88), 4i+3 = padding.
__global__ __launch_bounds__(256) void k_build_dif_lut(const float4* __restrict__ sphere, float4* __restrict__ lut) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;   // grid covers exactly 6 * 65536
    const uint32_t normal = i >> 16;
    const float4 p = sphere[i & 0xFFFFu];
    vec3 d = v3(p.x, p.y, p.z);
    if (normal == 0) d.x += 1.0f;
    else if (normal == 1) d.x -= 1.0f;
    else if (normal == 2) d.y += 1.0f;
    else if (normal == 3) d.y -= 1.0f;
    else if (normal == 4) d.z += 1.0f;
    else d.z -= 1.0f;
    const vec3 dd = vnormalize(d);          // diffuse_direction's return value (:211)
    const vec3 d2 = vnormalize(dd);         // trace_ray's own normalize (:83)
    lut[4 * i] = make_float4(dd.x, dd.y, dd.z, 0.0f);
    lut[4 * i + 1] = make_float4(d2.x, d2.y, d2.z, 0.0f);
    lut[4 * i + 2] = make_float4(1.0f / rtm_abs(d2.x), 1.0f / rtm_abs(d2.y), 1.0f / rtm_abs(d2.z), 0.0f);
    lut[4 * i + 3] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}

// =====================================================================================================
// k_primary
// =====================================================================================================
template <bool COUNT>
__global__ __launch_bounds__(256) void k_primary(Scene sc, Frame f, Planes pl, PrimaryArgs a) {
    const uint32_t lp = blockIdx.x * 256u + threadIdx.x;
    PixelId pix = pixel_of_local(f, lp);
    bool queue = false;
    vec3 qpos = v3(0, 0, 0);
    uint32_t qnormal = 0;
    unsigned long long c_prim = 0, c_iter = 0, c_hits = 0, c_sky = 0, c_limit = 0, c_border = 0, c_pix = 0;
    if (pix.inside) {
        vec3 start, dir;
        primary_ray(f, pix.px, pix.py, &start, &dir);
        Hit h = trace_ray_generic(sc, f, start, dir);
        if (COUNT) {
            c_prim++; c_pix++; c_iter += h.iterations; c_border += h.border; c_limit += h.limit_exit;
            if (h.air) c_sky++; else if (!h.limit_exit) c_hits++;
        }
        store_primary_planes(pl, pix.out_index, f, dir, h.air, h.normal, h.material, h.position);
        if (h.air || f.depth < 1) {
            // every sample of this pixel has the same light (no noise is read): sum it spp times like the shader's
            // spp frames would, then store
            vec3 light = v3(0.0f, 0.0f, 0.0f);
            if (h.air) light = sample_sky(dir, ld3(f.sunangle), ld3(f.sunlight), true);            // raytrace.comp:321-322
            vec3 sum = v3(0.0f, 0.0f, 0.0f);
            for (int s = 0; s < f.spp; s++) sum = vadd(sum, light);
            store_lighting(pl, pix.out_index, sum, f.spp);
        } else {
            queue = true;
            qpos = h.position; qnormal = h.normal;
        }
    }
    // worklist append: one atomic per wave, ballot-ranked slots; the record of slot w is SoA over w so that the path
    // kernel's item fetch is coalesced and needs no pixel arithmetic
    const uint64_t m = __ballot(queue);
    if (m) {
        const uint32_t lane = threadIdx.x & 63u;
        uint32_t base = 0;
        if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(a.wl_count, (uint32_t)__popcll(m));
        base = __shfl(base, __builtin_ctzll(m), 64);
        if (queue) {
            const uint32_t w = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            a.worklist[w] = lp;
            // face id, and the noise_offset terms gl_WorkGroupID.xy * 8 of the pixel (raytrace.comp:304)
            const uint32_t info = (qnormal << 28) | (owning_workgroup((uint32_t)pix.py) * RT_SHADER_GROUP_SIZE) << 14 |
                                  (owning_workgroup((uint32_t)pix.px) * RT_SHADER_GROUP_SIZE);
            a.phit[w] = make_float4(qpos.x, qpos.y, qpos.z, __uint_as_float(info));
        }
    }
    if (COUNT) {
        DevCounters* cn = a.counters;
        wave_add(&cn->rays, c_prim); wave_add(&cn->rays_primary, c_prim); wave_add(&cn->iterations, c_iter);
        wave_add(&cn->minefield_fetches, c_prim + c_iter); wave_add(&cn->hits, c_hits); wave_add(&cn->material_fetches, c_hits);
        wave_add(&cn->sky_exits, c_sky); wave_add(&cn->limit_exits, c_limit); wave_add(&cn->border_fetches, c_border);
        wave_add(&cn->pixels, c_pix);
    }
}

// =====================================================================================================
// k_primary2 — the same prepass on the persistent machinery
// =====================================================================================================
// One 1024-thread workgroup per CU with the nibble map in LDS; every wave walks whole 8x8 tiles (lane = pixel, so the
// wave's rays are coherent and end within a few steps of each other) with the stepped ray of rt_dda.hpp: ~70 % of the
// fetches are answered from LDS instead of a dependent global byte load per step.  Results as k_primary.
template <int LOGR, bool LRZ, bool COUNT>
__global__ __launch_bounds__(1024) void k_primary2(Scene sc, Frame f, Planes pl, PrimaryArgs a) {
    __shared__ uint32_t s_coarse[kCoarseWords];
    __shared__ uint32_t s_swz[dda_uses_swz<LOGR, LRZ>() ? 3 * kSwzStride : 1];
    __shared__ uint32_t s_cnt[16], s_off[17];   // worklist append: pixels queued by each wave this round, their slot offsets
    // what later launches need zeroed — the frame slot's other worklist counter (the slot's next frame) and the path cursors of
    // the lane this frame's first path launch runs on (idle until then) — is cleared here instead of by memsets of their own in
    // front of every frame (two fill kernels, 9 us of a 230 us frame at 1024^2)
    if (blockIdx.x == 0u) {
        if (a.zero_words != nullptr) for (uint32_t i = threadIdx.x; i < a.zero_count; i += 1024u) a.zero_words[i] = 0u;
        if (a.zero_words2 != nullptr) for (uint32_t i = threadIdx.x; i < a.zero_count2; i += 1024u) a.zero_words2[i] = 0u;
    }
    {
        const uint4* src = reinterpret_cast<const uint4*>(sc.coarse);
        uint4* dst = reinterpret_cast<uint4*>(s_coarse);
        for (uint32_t i = threadIdx.x; i < kCoarseWords / 4; i += 1024u) dst[i] = src[i];
        if (dda_uses_swz<LOGR, LRZ>()) dda_fill_swz(s_swz, threadIdx.x, 1024u);
    }
    __syncthreads();
    const uint8_t* s_nib = reinterpret_cast<const uint8_t*>(s_coarse);
    constexpr int R = 1 << LOGR, LB = LOGR - 2;
    const float half = (float)R / 2;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * 1024u + threadIdx.x) >> 6, nwaves = gridDim.x * 16u;
    unsigned long long c_prim = 0, c_pix = 0, c_border = 0;
    RayTally tl;
    // All waves of the workgroup run the same number of rounds (the append below has two barriers per round); a wave whose
    // tile index is past the end carries no pixels in that round.
    const uint32_t nrounds = ((uint32_t)f.ntiles_local + nwaves - 1u) / nwaves;
    for (uint32_t round = 0; round < nrounds; round++) {
        const uint32_t tile = round * nwaves + wave;
        const uint32_t lp = tile * 64u + lane;
        PixelId pix = pixel_of_local(f, lp);
        if (tile >= (uint32_t)f.ntiles_local) pix.inside = false;
        RaySlot2 r;
        r.tracing = false; r.valid = true; r.fresh_invalid = false; r.nk = PX_HIT << 16; r.axis = 2u; r.vox = r.cidx = 0u;
        r.px = r.py = r.pz = r.ndx = r.ndy = r.ndz = r.lx = r.ly = r.lz = r.ux = r.uy = r.uz = 0.0f;
        vec3 start = v3(0, 0, 0), dir = v3(0, 0, 1);
        if (pix.inside) {
            primary_ray(f, pix.px, pix.py, &start, &dir);
            const vec3 d = vnormalize(dir);                                                          // raytrace.comp:83
            r.lx = 1.0f / rtm_abs(d.x); r.ly = 1.0f / rtm_abs(d.y); r.lz = 1.0f / rtm_abs(d.z);       // :88
            int ix, iy, iz;
            const bool ok = wrap_texel(start, (float)R, &ix, &iy, &iz);
            dda_arm<LOGR, LRZ, COUNT>(r, d.x, d.y, d.z, start.x, start.y, start.z, ok, swizzled_index(ix, iy, iz, LB),
                                      coarse_index(ix, iy, iz, LOGR), f, half, s_nib, sc, c_border, s_swz);
        }
        while (__ballot(r.tracing)) {
            if (r.tracing) dda_advance<LOGR, LRZ, COUNT, false>(r, dda_lookup<LOGR>(r, s_nib, sc), f, half, c_border, s_swz);
        }
        bool queue = false;
        vec3 qpos = v3(0, 0, 0);
        uint32_t qnormal = 0;
        if (pix.inside) {
            const uint32_t kind = r2_kind(r);
            const bool air = kind == PX_AIR;
            const uint32_t nrm = r.axis == 0 ? (r.ndx < 0.0f ? 1u : 0u) : (r.axis == 1 ? (r.ndy < 0.0f ? 3u : 2u) : (r.ndz < 0.0f ? 5u : 4u));
            uint32_t material = 0;
            if (kind == PX_HIT && (LRZ || r.valid)) material = sc.mat[r.vox];   // the hit texel is the texel of the last fetch (:150-154)
            float hx = r.px, hy = r.py, hz = r.pz;
            if (kind == PX_SPECIAL) { hx = hy = hz = __builtin_nanf(""); }
            const float off = 0.001f;                                           // :166-180
            if (nrm == 0) hx += off; else if (nrm == 1) hx -= off;
            else if (nrm == 2) hy += off; else if (nrm == 3) hy -= off;
            else if (nrm == 4) hz += off; else hz -= off;
            if (COUNT) { c_prim++; c_pix++; dda_tally<LOGR>(r, tl); }
            store_primary_planes(pl, pix.out_index, f, dir, air, nrm, material, v3(hx, hy, hz));
            if (air || f.depth < 1) {
                // every sample of this pixel has the same light (no noise is read): sum it spp times like the shader's
                // spp frames would, then store
                vec3 light = v3(0.0f, 0.0f, 0.0f);
                if (air) light = sample_sky(dir, ld3(f.sunangle), ld3(f.sunlight), true);            // raytrace.comp:321-322
                vec3 sum = v3(0.0f, 0.0f, 0.0f);
                for (int s = 0; s < f.spp; s++) sum = vadd(sum, light);
                store_lighting(pl, pix.out_index, sum, f.spp);
            } else {
                queue = true;
                qpos = v3(hx, hy, hz); qnormal = nrm;
            }
        }
        // worklist append: ONE atomic per workgroup and round (a single counter word saturates near 90 returning atomics
        // per microsecond — one per wave made the append, not the tracing, the bound of this kernel), slots ballot-ranked
        const uint64_t m = __ballot(queue);
        const uint32_t wiw = threadIdx.x >> 6;
        if (lane == 0) s_cnt[wiw] = (uint32_t)__popcll(m);
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t total = 0;
            for (uint32_t i = 0; i < 16u; i++) { s_off[i] = total; total += s_cnt[i]; }
            const uint32_t base = total ? atomicAdd(a.wl_count, total) : 0u;
            for (uint32_t i = 0; i < 16u; i++) s_off[i] += base;
        }
        __syncthreads();
        if (queue) {
            const uint32_t w = s_off[wiw] + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            a.worklist[w] = lp;
            const uint32_t info = (qnormal << 28) | (owning_workgroup((uint32_t)pix.py) * RT_SHADER_GROUP_SIZE) << 14 |
                                  (owning_workgroup((uint32_t)pix.px) * RT_SHADER_GROUP_SIZE);
            a.phit[w] = make_float4(qpos.x, qpos.y, qpos.z, __uint_as_float(info));
        }
    }
    if (COUNT) {
        DevCounters* cn = a.counters;
        c_border += tl.border;
        wave_add(&cn->rays, c_prim); wave_add(&cn->rays_primary, c_prim); wave_add(&cn->iterations, tl.iter);
        wave_add(&cn->minefield_fetches, c_prim + tl.iter); wave_add(&cn->hits, tl.hits); wave_add(&cn->material_fetches, tl.hits);
        wave_add(&cn->sky_exits, tl.sky); wave_add(&cn->limit_exits, tl.limit); wave_add(&cn->border_fetches, c_border);
        wave_add(&cn->pixels, c_pix);
    }
}

// =====================================================================================================
// k_persist
// =====================================================================================================
enum : uint32_t { PH_EMPTY = 0, PH_PRIMARY = 1, PH_SUN = 2, PH_DIF = 3 };

// Shadow rays depend only on the frame's sun vector and the (noise.r, noise.g) byte pair (raytrace.comp:185-187), so
// the whole head of trace_ray for them — normalize (trace_sun), normalize again (:83), 1/|d| (:88) — is tabulated once
// per frame: entry 2i = direction, 2i+1 = length_per_axis.
__global__ __launch_bounds__(256) void k_build_sun_lut(Frame f, float4* __restrict__ lut) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;   // grid covers exactly 65536
    const float nr = (float)(i & 255u) / 255.0f, ng = (float)(i >> 8) / 255.0f;
    const vec3 d = vnormalize(sun_ray_direction(ld3(f.sunangle), nr, ng));
    lut[2 * i] = make_float4(d.x, d.y, d.z, 0.0f);
    lut[2 * i + 1] = make_float4(1.0f / rtm_abs(d.x), 1.0f / rtm_abs(d.y), 1.0f / rtm_abs(d.z), 0.0f);
}

// One ray in flight.  A lane carries two: slot S walks the level's shadow ray, slot F its diffuse ray (or, with
// CACHE=false, the primary ray).  The two rays of a level are independent (raytrace.comp:325 and :330 both start from
// the same surface), so stepping them together doubles the memory-level parallelism of the dependent fetch chain and
// halves the number of transition passes.
// (The slot itself is RaySlot2 of rt_dda.hpp.)

// sample_sky(diffuse_direction, ..., true) (raytrace.comp:331-332 / :343-345) for every entry of the diffuse table, written
// once per frame into the entry's fourth slot: a path that ends on a sky exit reads its sky light instead of
// evaluating three pow() per lane in the transition pass.
__global__ __launch_bounds__(256) void k_build_sky_lut(Frame f, float4* __restrict__ dif_lut) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;   // grid covers exactly 6 * 65536
    const float4 dd = dif_lut[4 * i];
    const vec3 sky = sample_sky(v3(dd.x, dd.y, dd.z), ld3(f.sunangle), ld3(f.sunlight), true);
    dif_lut[4 * i + 3] = make_float4(sky.x, sky.y, sky.z, 0.0f);
}

template <int LOGR, bool LRZ, bool COUNT, bool CACHE>
__global__ __launch_bounds__(1024, 4) void k_persist(Scene sc, Frame f, Planes pl, PersistArgs a) {
    __shared__ uint32_t s_coarse[kCoarseWords];
    __shared__ float s_albedo[128];            // (packed >> k & 0x7F) / 127.0 (raytrace.comp:156-158), exact quotients
    __shared__ uint32_t s_swz[dda_uses_swz<LOGR, LRZ>() ? 3 * kSwzStride : 1];   // swizzle tables (rt_dda.hpp)
    // albedo stack of the lane's path (packed material of surface j+1 at level j) for depths up to kLdsStack + 1: the LDS the
    // nibble map leaves holds exactly seven levels for 1024 threads; deeper frames use the global stack
    constexpr uint32_t kLdsStack = 7;
    __shared__ uint32_t s_stack[kLdsStack][1024];
    // work items: path r = sample_in_batch * nwork + w, w = worklist slot (CACHE) or local pixel (CACHE=false)
    const uint32_t nwork = CACHE ? *a.wl_count : a.npix_pad;
    const uint32_t nitems = nwork * a.nsamples;
    if (nitems == 0u) return;
    {
        const uint4* src = reinterpret_cast<const uint4*>(sc.coarse);
        uint4* dst = reinterpret_cast<uint4*>(s_coarse);
        for (uint32_t i = threadIdx.x; i < kCoarseWords / 4; i += 1024u) dst[i] = src[i];
        if (threadIdx.x < 128u) s_albedo[threadIdx.x] = (float)threadIdx.x / 127.0f;
        if (dda_uses_swz<LOGR, LRZ>()) dda_fill_swz(s_swz, threadIdx.x, 1024u);
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gtid = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t threshold = a.threshold;
    constexpr int R = 1 << LOGR, LB = LOGR - 2;
    const float half = (float)R / 2;
    const vec3 sunangle = ld3(f.sunangle), sunlight = ld3(f.sunlight);
    const uint32_t D = (uint32_t)f.depth;

    RaySlot2 S, F;
    S.px = S.py = S.pz = S.ndx = S.ndy = S.lx = S.ly = S.lz = S.ux = S.uy = S.uz = 0.0f; S.ndz = -1.0f;
    S.vox = S.cidx = 0u; S.nk = PX_HIT << 16; S.axis = 2u;
    S.tracing = false; S.valid = true; S.fresh_invalid = false;
    F = S;
    // ---- path state ----
    uint32_t phase = PH_EMPTY;                // PH_EMPTY, PH_PRIMARY (F only, CACHE=false), PH_DIF (= a level: S and F)
    uint32_t item = 0, lp = 0, samp = 0, level = 0, sunbits = 0;
    uint32_t dif_entry = 0xFFFFFFFFu;         // diffuse-table entry held by F's direction registers (its slot 3 = sample_sky of that direction)
    uint32_t sun_entry = 0xFFFFFFFFu;         // shadow-table entry held by S's direction registers
    uint32_t nvtex = 0;                       // noise_value texel of the path (raytrace.comp:324,336)
    bool exhausted = false;
    // paths per cursor atomic (RT_PERSIST_CHUNK overrides).  Measured with the eight per-XCD cursors: 128 is best or equal from
    // a 1-sample 1024^2 frame to the spp-64 headline frame; 64 costs 2-3 % there (atomic rate), 256 lengthens small frames' tails
    const uint32_t kChunk = a.chunk ? a.chunk : 128u;
    uint32_t chunk_next = 0, chunk_end = 0;   // wave-uniform: the wave's current chunk of an XCD group's share of the paths
    uint32_t chunk_sb = 0, chunk_w = 0;       // (sample-in-batch, slot within the share) of path chunk_next
    uint32_t chunk_w0 = 0, chunk_nw = 1;      // the share's slot range
    const uint32_t home_grp = blockIdx.x & 7u;   // workgroups b and b + 8 share an XCD (round-robin dispatch; speed only)
    uint32_t grp_tries = 0;                   // shares found empty so far, starting with the own group's

    unsigned long long c_prim = 0, c_shadow = 0, c_dif = 0, c_iter = 0, c_hits = 0, c_sky = 0, c_limit = 0, c_border = 0,
                       c_noise = 0, c_pix = 0;
    unsigned long long d_iters = 0, d_sx = 0, d_fx = 0, d_sl = 0, d_fl = 0, d_pass = 0, d_pl = 0, d_sky = 0;   // wave-uniform

    // ---- the ray machinery of rt_dda.hpp bound to this kernel's constants ---------------------------------------------
    const uint8_t* s_nib = reinterpret_cast<const uint8_t*>(s_coarse);
    // head of trace_ray (:83-107) from origin ro (shared by the two rays of a level: same surface point); the slot's direction
    // registers (nd*, l*) are already set
    auto arm = [&](RaySlot2& r, vec3 ro, bool ok, uint32_t vox0, uint32_t cidx0) {
        dda_arm<LOGR, LRZ, COUNT>(r, -r.ndx, -r.ndy, -r.ndz, ro.x, ro.y, ro.z, ok, vox0, cidx0, f, half, s_nib, sc, c_border, s_swz);
    };
    auto set_dir = [&](RaySlot2& r, vec3 rd) {
        const vec3 d = vnormalize(rd);                                                               // :83
        r.ndx = -d.x; r.ndy = -d.y; r.ndz = -d.z;
        r.lx = 1.0f / rtm_abs(d.x); r.ly = 1.0f / rtm_abs(d.y); r.lz = 1.0f / rtm_abs(d.z);           // :88
    };
    auto advance = [&](RaySlot2& r, uint32_t step) { dda_advance<LOGR, LRZ, COUNT, false>(r, step, f, half, c_border, s_swz); };
    auto tally = [&](const RaySlot2& r) {   // exact counters of one finished ray
        RayTally t;
        dda_tally<LOGR>(r, t);
        c_iter += t.iter; c_hits += t.hits; c_sky += t.sky; c_limit += t.limit; c_border += t.border;
    };

    for (;;) {
        // Lanes whose rays have all ended park; when `threshold` of them are parked the wave runs one transition pass.
        const uint64_t m_busy = __ballot(S.tracing || F.tracing);
        const uint64_t m_wait = __ballot(!(S.tracing || F.tracing) && (phase != PH_EMPTY || !exhausted));
        const uint32_t n_busy = (uint32_t)__popcll(m_busy), n_wait = (uint32_t)__popcll(m_wait);
        if (n_wait < threshold && n_busy != 0u) {
            // ---- step loop: run until enough further lanes have parked -----------------------------------------
            const uint32_t need = threshold - n_wait;   // >= 1
            const uint32_t target = n_busy > need ? n_busy - need : 0u;
            do {
                // fetches of both slots first (:106 for a fresh ray, :137 otherwise), so their latencies overlap
                // nibble-map entry: at R = 256 a coarse cube IS the 4^3 brick, so the entry index is vox >> 6
                const uint32_t bS = LOGR == 8 ? S.vox >> 6 : S.cidx, bF = LOGR == 8 ? F.vox >> 6 : F.cidx;
                // byte reads: entry b is nibble (b & 1) of byte b >> 1 (little-endian words, k_build_coarse)
                const uint32_t wS = s_nib[bS >> 1], wF = s_nib[bF >> 1];
                uint32_t stS = (wS >> ((bS & 1u) << 2)) & 15u, stF = (wF >> ((bF & 1u) << 2)) & 15u;
                const bool gS = S.tracing && stS == kNibMixed, gF = F.tracing && stF == kNibMixed;
                uint8_t byS = 0, byF = 0;
                if (gS) byS = sc.mine[S.vox];
                if (gF) byF = sc.mine[F.vox];
                if (gS) stS = byS;
                if (gF) stF = byF;
                if (COUNT) {
                    const uint32_t nS = (uint32_t)__popcll(__ballot(S.tracing)), nF = (uint32_t)__popcll(__ballot(F.tracing));
                    d_iters++; if (nS) { d_sx++; d_sl += nS; } if (nF) { d_fx++; d_fl += nF; }
                }
                if (S.tracing) advance(S, stS);
                if (F.tracing) advance(F, stF);
            } while ((uint32_t)__popcll(__ballot(S.tracing || F.tracing)) > target);
            // falls through into the pass
        } else if (n_wait == 0u) {
            break;   // nothing in flight, nothing parked, no paths left
        }

        // =========================== transition pass ===========================================================
        const bool mine = !(S.tracing || F.tracing) && phase != PH_EMPTY;
        if (COUNT) { d_pass++; d_pl += (uint32_t)__popcll(__ballot(mine)); d_sky += (uint32_t)__popcll(__ballot(mine && phase == PH_DIF && r2_kind(F) == PX_AIR)); }
        bool path_done = false, begin_level = false, need_primary = false;
        vec3 light = v3(0, 0, 0);
        float sfx = 0, sfy = 0, sfz = 0;      // surface the next level stands on
        uint32_t snormal = 0;
        if (mine) {
            // Diffuse / primary result.  The hit texel is the texel of the last fetch (same position, same wrap:
            // mod((p+128)/256,1)*256 and mod(p+128,256) agree bit for bit), so the material is mat[vox] (:150-154); the
            // position gets the 0.001 face offset (:166-180).
            const uint32_t fkind = r2_kind(F);
            const bool air = fkind == PX_AIR;
            const uint32_t nrm = F.axis == 0 ? (F.ndx < 0.0f ? 1u : 0u) : (F.axis == 1 ? (F.ndy < 0.0f ? 3u : 2u) : (F.ndz < 0.0f ? 5u : 4u));
            uint32_t material = 0;
            if (fkind == PX_HIT && (LRZ || F.valid)) material = sc.mat[F.vox];
            float hx = F.px, hy = F.py, hz = F.pz;
            if (fkind == PX_SPECIAL) { hx = hy = hz = __builtin_nanf(""); }
            const float off = 0.001f;
            if (nrm == 0) hx += off; else if (nrm == 1) hx -= off;
            else if (nrm == 2) hy += off; else if (nrm == 3) hy -= off;
            else if (nrm == 4) hz += off; else hz -= off;
            if (COUNT) tally(F);
            if (!CACHE && phase == PH_PRIMARY) {
                PixelId pix = pixel_of_local(f, lp);
                vec3 pstart, pdir;
                primary_ray(f, pix.px, pix.py, &pstart, &pdir);
                if (samp == 0u) {
                    store_primary_planes(pl, pix.out_index, f, pdir, air, nrm, material, v3(hx, hy, hz));
                    if (COUNT) c_pix++;
                }
                if (air) {
                    light = sample_sky(pdir, sunangle, sunlight, true);                            // :321-322
                    path_done = true;
                } else if (D < 1u) {
                    path_done = true;
                } else {
                    sfx = hx; sfy = hy; sfz = hz; snormal = nrm;
                    level = 1; sunbits = 0; begin_level = true;
                }
            } else {
                // a level ended: shadow result (:326-328 / :338-340), then the diffuse result
                if (COUNT) tally(S);
                if (r2_kind(S) == PX_AIR) sunbits |= 1u << (level - 1);
                if (air || level == D) {
                    vec3 sky = v3(0, 0, 0);
                    if (air) { const float4 t = a.dif_lut[dif_entry + 3u]; sky = v3(t.x, t.y, t.z); }   // :331-332 / :343-345, tabulated
                    // L_j = [sun_j] S + L_{j+1} * albedo_{j+1} + emission, innermost first (raytrace.comp:346-348)
                    vec3 L = v3(0.0f, 0.0f, 0.0f);
                    if (sunbits >> (level - 1) & 1u) L = vadd(L, sunlight);
                    if (air) L = vadd(L, sky);
                    for (uint32_t j = level - 1; j >= 1u; j--) {
                        const uint32_t pm = D - 1u <= kLdsStack ? s_stack[j - 1][threadIdx.x] : a.stack[(size_t)(j - 1) * a.nthreads + gtid];
                        vec3 light2 = vmul(L, v3(s_albedo[pm >> 14 & 0x7Fu], s_albedo[pm >> 7 & 0x7Fu], s_albedo[pm & 0x7Fu]));
                        light2 = vadd(light2, v3(0.0f, 0.0f, 0.0f));      // + dif.emission, always vec3(0) (:155)
                        vec3 acc = v3(0.0f, 0.0f, 0.0f);
                        if (sunbits >> (j - 1) & 1u) acc = vadd(acc, sunlight);
                        L = vadd(acc, light2);
                    }
                    light = vadd(v3(0.0f, 0.0f, 0.0f), L);
                    path_done = true;
                } else {
                    if (D - 1u <= kLdsStack) s_stack[level - 1][threadIdx.x] = material;   // albedo of surface level+1
                    else a.stack[(size_t)(level - 1) * a.nthreads + gtid] = material;
                    sfx = hx; sfy = hy; sfz = hz; snormal = nrm;
                    level++; begin_level = true;
                }
            }
            if (path_done) {   // the path's light; k_accumulate_paths adds the samples of a pixel in order
                if (a.direct) {   // one sample per pixel: the sum is 0 + light and the pixel is finished (k_accumulate_paths' arithmetic)
                    const PixelId dp = pixel_of_local(f, CACHE ? a.worklist[lp] : lp);   // (lp: worklist slot with cached primaries)
                    if (dp.inside) store_lighting(pl, dp.out_index, v3(0.0f + light.x, 0.0f + light.y, 0.0f + light.z), f.spp);
                } else {
                    a.pl[item] = PathLight{light.x, light.y, light.z};
                }
                phase = PH_EMPTY;
            }
        }
        // empty lanes pull the next paths: the wave owns a chunk of kChunk consecutive paths (one atomicAdd on the global
        // cursor per chunk — a single word saturates near 90 returning atomics/us) and deals them out ballot-ranked
        if (!exhausted) {
            const uint64_t want = __ballot(phase == PH_EMPTY);
            const uint32_t nwant = (uint32_t)__popcll(want);
            if (nwant) {
                if (chunk_next >= chunk_end) {
                    // next chunk: from the share of this workgroup's XCD group first (worklist slots [w0, w0 + nw) of every
                    // sample: one band of the image, so an XCD's L2 keeps seeing the same part of the scene), then from the
                    // other groups' shares; eight cursor words also lift the ~90 atomics/us limit of a single one
                    for (;;) {
                        if (grp_tries == 8u) { exhausted = true; chunk_next = chunk_end = 0u; break; }
                        const uint32_t g = (home_grp + grp_tries) & 7u;
                        const uint32_t w0 = (uint32_t)((uint64_t)nwork * g >> 3), nw = (uint32_t)((uint64_t)nwork * (g + 1u) >> 3) - w0;
                        const uint32_t ng = nw * a.nsamples;
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(a.cursor + 32u * g, kChunk);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base < ng) {
                            chunk_next = base; chunk_end = base + kChunk < ng ? base + kChunk : ng;
                            chunk_w0 = w0; chunk_nw = nw;
                            chunk_sb = base / nw; chunk_w = base - chunk_sb * nw;   // once per chunk
                            break;
                        }
                        grp_tries++;   // that group's share is handed out for good (its cursor only grows)
                    }
                }
                const uint32_t take = min(nwant, chunk_end - chunk_next);
                chunk_next += take;
                if (phase == PH_EMPTY) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                    if (rank < take) {
                        // (sample-in-batch, slot) of path first+rank, stepped from the chunk's running position (no division)
                        uint32_t sb = chunk_sb, w = chunk_w + rank;
                        while (w >= chunk_nw) { w -= chunk_nw; sb++; }
                        w += chunk_w0;   // worklist slot (CACHE) / local pixel
                        uint32_t wgx8 = 0, wgy8 = 0;
                        bool ok = true;
                        if (CACHE) {
                            const float4 ph = a.phit[w];
                            const uint32_t info = __float_as_uint(ph.w);
                            sfx = ph.x; sfy = ph.y; sfz = ph.z;
                            snormal = info >> 28; wgx8 = info & 0x3FFFu; wgy8 = (info >> 14) & 0x3FFFu;
                        } else {
                            PixelId pix = pixel_of_local(f, w);
                            ok = pix.inside;           // padding pixels of partial tiles carry no path
                            wgx8 = owning_workgroup((uint32_t)pix.px) * RT_SHADER_GROUP_SIZE;
                            wgy8 = owning_workgroup((uint32_t)pix.py) * RT_SHADER_GROUP_SIZE;
                        }
                        if (ok) {
                            item = sb * nwork + w; lp = w; samp = a.sample0 + sb;
                            // noise_offset of this path (:298-304) and its noise_value texel (:324, :336).  The bytes are
                            // exact integers in float (texture().r * 255.0 == the byte) and the per-level offset
                            // (level-1) * 2/512 never reaches the next texel, so one integer lookup serves every level
                            // (tests/test_math_contract.py::test_noise_value_texel_is_level_independent).
                            const uint32_t seed = (f.seed + samp) % (uint32_t)RT_NOISE_BYTES;
                            const uint32_t by = seed / RT_NOISE_SIZE;
                            const uint32_t nb = sc.noise[(by > 511u ? 511u : by) * RT_NOISE_SIZE + seed % RT_NOISE_SIZE];
                            const uint32_t tx = ((nb & 0xFFu) + wgx8) & 511u, ty = (((nb >> 8) & 0xFFu) + wgy8) & 511u;
                            nvtex = sc.noise[ty * RT_NOISE_SIZE + tx];
                            if (CACHE) { level = 1; sunbits = 0; begin_level = true; }
                            else need_primary = true;
                        }
                    }
                }
                chunk_w += take;
                while (chunk_w >= chunk_nw) { chunk_w -= chunk_nw; chunk_sb++; }
            }
        }
        // both rays of a level (:324-330 / :336-342): noise_value, shadow ray and diffuse ray from the tables
        if (begin_level) {
            if (COUNT) { c_noise++; c_shadow++; c_dif++; }
            const vec3 ro = v3(sfx, sfy, sfz);
            int ix, iy, iz;
            const bool ok = wrap_texel(ro, (float)R, &ix, &iy, &iz);
            const uint32_t vox0 = swizzled_index(ix, iy, iz, LB), cidx0 = coarse_index(ix, iy, iz, LOGR);
            // Stepping never touches a slot's direction registers, so they still hold the table entry of the previous
            // level: the shadow entry depends on the path's noise texel only (same for all its levels, Q5) and the
            // diffuse entry repeats whenever the next surface has the same face — skip those table reads.
            const uint32_t se = nvtex & 0xFFFFu;
            if (se != sun_entry) {
                const float4 sd = a.sun_lut[2u * se], sl = a.sun_lut[2u * se + 1u];
                S.ndx = -sd.x; S.ndy = -sd.y; S.ndz = -sd.z; S.lx = sl.x; S.ly = sl.y; S.lz = sl.z;
                sun_entry = se;
            }
            arm(S, ro, ok, vox0, cidx0);
            const uint32_t di = 4u * ((snormal << 16) | se);
            if (di != dif_entry) {
                const float4 d2 = a.dif_lut[di + 1u], dl = a.dif_lut[di + 2u];
                F.ndx = -d2.x; F.ndy = -d2.y; F.ndz = -d2.z; F.lx = dl.x; F.ly = dl.y; F.lz = dl.z;
                dif_entry = di;
            }
            arm(F, ro, ok, vox0, cidx0);
            phase = PH_DIF;
        }
        // primary ray of the pixel (:296-315), CACHE=false only
        if (!CACHE && need_primary) {
            PixelId pix = pixel_of_local(f, lp);
            vec3 ro, rd;
            primary_ray(f, pix.px, pix.py, &ro, &rd);
            int ix, iy, iz;
            const bool ok = wrap_texel(ro, (float)R, &ix, &iy, &iz);
            set_dir(F, rd);
            dif_entry = 0xFFFFFFFFu;   // F's direction registers no longer hold a table entry
            arm(F, ro, ok, swizzled_index(ix, iy, iz, LB), coarse_index(ix, iy, iz, LOGR));
            phase = PH_PRIMARY;
            if (COUNT) c_prim++;
        }
    }
    if (COUNT) {
        DevCounters* cn = a.counters;
        const unsigned long long rays = c_prim + c_shadow + c_dif;
        wave_add(&cn->rays, rays); wave_add(&cn->rays_primary, c_prim); wave_add(&cn->rays_shadow, c_shadow);
        wave_add(&cn->rays_diffuse, c_dif); wave_add(&cn->iterations, c_iter); wave_add(&cn->minefield_fetches, rays + c_iter);
        wave_add(&cn->hits, c_hits); wave_add(&cn->material_fetches, c_hits); wave_add(&cn->sky_exits, c_sky);
        wave_add(&cn->limit_exits, c_limit); wave_add(&cn->border_fetches, c_border); wave_add(&cn->noise_fetches, c_noise);
        wave_add(&cn->pixels, c_pix);
        if (lane == 0) {
            atomicAdd(&cn->dbg_loop_iters, d_iters); atomicAdd(&cn->dbg_s_execs, d_sx); atomicAdd(&cn->dbg_f_execs, d_fx);
            atomicAdd(&cn->dbg_s_lanes, d_sl); atomicAdd(&cn->dbg_f_lanes, d_fl); atomicAdd(&cn->dbg_passes, d_pass);
            atomicAdd(&cn->dbg_pass_lanes, d_pl); atomicAdd(&cn->dbg_sky_lanes, d_sky);
        }
    }
}

// acc[pixel] (+)= the batch's samples of that pixel, in sample order (deterministic fp32 sum; raytrace.comp has one
// sample per frame, the sum over frames is the build's spp extension).  The last batch of a frame writes the pixel's
// lighting planes itself (sum / spp / 16, raytrace.comp:352-356) — the prepass has done that for the pixels it finished —
// so no separate resolve launch is needed.
template <bool CACHE, bool STREAM>
__global__ __launch_bounds__(256) void k_accumulate_paths(Frame f, Planes planes, const PathLight* __restrict__ pl,
                                                          const uint32_t* __restrict__ worklist,
                                                          const uint32_t* __restrict__ wl_count, uint32_t npix_pad,
                                                          uint32_t nsamples, int first_batch, int last_batch, float4* __restrict__ acc) {
    const uint32_t w = blockIdx.x * 256u + threadIdx.x;
    const uint32_t nwork = CACHE ? *wl_count : npix_pad;
    if (w >= nwork) return;
    const uint32_t lp = CACHE ? worklist[w] : w;
    float4 v = first_batch ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : acc[lp];
    for (uint32_t b = 0; b < nsamples; b++) {
        // STREAM: read once and too many to have stayed in the caches — streaming loads, like k_paths' stores of them (headline frame
        // 4.29 -> 4.25 ms with both; a launch whose records fit the Infinity Cache is better off with plain loads)
        const float* src = &pl[(size_t)b * nwork + w].x;
        const PathLight l = STREAM ? PathLight{__builtin_nontemporal_load(src), __builtin_nontemporal_load(src + 1), __builtin_nontemporal_load(src + 2)}
                                   : pl[(size_t)b * nwork + w];
        v.x = v.x + l.x; v.y = v.y + l.y; v.z = v.z + l.z;
    }
    if (last_batch) {
        const PixelId pix = pixel_of_local(f, lp);
        if (pix.inside) store_lighting(planes, pix.out_index, v3(v.x, v.y, v.z), f.spp);
    } else {
        acc[lp] = v;
    }
}

hipError_t launch_accumulate_paths(const Frame& f, const Planes& planes, const PathLight* pl, const uint32_t* worklist,
                                   const uint32_t* wl_count, uint32_t npix_pad, uint32_t nsamples, bool first_batch, bool last_batch,
                                   bool cache, bool stream, float4* acc, hipStream_t st) {
    if (npix_pad == 0) return hipSuccess;
    dim3 grid((npix_pad + 255u) / 256u), block(256);
#define RT_LAUNCH_ACC(C, S) hipLaunchKernelGGL((k_accumulate_paths<C, S>), grid, block, 0, st, f, planes, pl, worklist, wl_count, npix_pad, nsamples, first_batch ? 1 : 0, last_batch ? 1 : 0, acc)
    if (cache) { if (stream) RT_LAUNCH_ACC(true, true); else RT_LAUNCH_ACC(true, false); }
    else { if (stream) RT_LAUNCH_ACC(false, true); else RT_LAUNCH_ACC(false, false); }
#undef RT_LAUNCH_ACC
    return hipGetLastError();
}

hipError_t launch_sun_lut(const Frame& f, float4* lut, hipStream_t st) {
    hipLaunchKernelGGL(k_build_sun_lut, dim3(65536 / 256), dim3(256), 0, st, f, lut);
    return hipGetLastError();
}

hipError_t launch_sky_lut(const Frame& f, float4* dif_lut, hipStream_t st) {
    hipLaunchKernelGGL(k_build_sky_lut, dim3(6 * 65536 / 256), dim3(256), 0, st, f, dif_lut);
    return hipGetLastError();
}

hipError_t launch_sphere_lut(float4* lut, hipStream_t st) {
    hipLaunchKernelGGL(k_build_sphere_lut, dim3(65536 / 256), dim3(256), 0, st, lut);
    return hipGetLastError();
}

hipError_t launch_dif_lut(const float4* sphere, float4* lut, hipStream_t st) {
    hipLaunchKernelGGL(k_build_dif_lut, dim3(6 * 65536 / 256), dim3(256), 0, st, sphere, lut);
    return hipGetLastError();
}

template <int LOGR>
static void launch_primary2_logr(const Scene& sc, const Frame& f, const Planes& pl, const PrimaryArgs& a, bool count, dim3 grid,
                                 hipStream_t st) {
    const dim3 block(1024);
    if (f.lr_zero != 0) {
        if (count) hipLaunchKernelGGL((k_primary2<LOGR, true, true>), grid, block, 0, st, sc, f, pl, a);
        else hipLaunchKernelGGL((k_primary2<LOGR, true, false>), grid, block, 0, st, sc, f, pl, a);
    } else {
        if (count) hipLaunchKernelGGL((k_primary2<LOGR, false, true>), grid, block, 0, st, sc, f, pl, a);
        else hipLaunchKernelGGL((k_primary2<LOGR, false, false>), grid, block, 0, st, sc, f, pl, a);
    }
}

hipError_t launch_primary(const Scene& sc, const Frame& f, const Planes& pl, const PrimaryArgs& a, bool count, int version,
                          int nworkgroups, hipStream_t st) {
    const uint32_t npix_pad = (uint32_t)f.ntiles_local * 64u;
    if (npix_pad == 0) return hipSuccess;
    if (version == 2) {
        // one workgroup per CU, but no more than there are 16-tile shares of work
        const int want = (f.ntiles_local + 15) / 16;
        const dim3 grid(want < nworkgroups ? want : nworkgroups);
        if (f.logr == 8) launch_primary2_logr<8>(sc, f, pl, a, count, grid, st);
        else if (f.logr == 9) launch_primary2_logr<9>(sc, f, pl, a, count, grid, st);
        else if (f.logr == 10) launch_primary2_logr<10>(sc, f, pl, a, count, grid, st);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    dim3 grid((npix_pad + 255u) / 256u), block(256);
    if (count) hipLaunchKernelGGL(k_primary<true>, grid, block, 0, st, sc, f, pl, a);
    else hipLaunchKernelGGL(k_primary<false>, grid, block, 0, st, sc, f, pl, a);
    return hipGetLastError();
}

template <int LOGR>
static void launch_persist_logr(const Scene& sc, const Frame& f, const Planes& pl, const PersistArgs& a, bool count, bool cache,
                                int version, dim3 grid, dim3 block, hipStream_t st) {
    const bool lrz = f.lr_zero != 0;
#define RT_LAUNCH_PERSIST(L, C, K)                                                                        \
    do {                                                                                                  \
        hipLaunchKernelGGL((k_persist<LOGR, L, C, K>), grid, block, 0, st, sc, f, pl, a);                \
    } while (0)
    if (lrz) {
        if (count) { if (cache) RT_LAUNCH_PERSIST(true, true, true); else RT_LAUNCH_PERSIST(true, true, false); }
        else       { if (cache) RT_LAUNCH_PERSIST(true, false, true); else RT_LAUNCH_PERSIST(true, false, false); }
    } else {
        if (count) { if (cache) RT_LAUNCH_PERSIST(false, true, true); else RT_LAUNCH_PERSIST(false, true, false); }
        else       { if (cache) RT_LAUNCH_PERSIST(false, false, true); else RT_LAUNCH_PERSIST(false, false, false); }
    }
#undef RT_LAUNCH_PERSIST
}

hipError_t launch_persist(const Scene& sc, const Frame& f, const Planes& pl, const PersistArgs& a, bool count, bool cache,
                          int version, int nworkgroups, hipStream_t st) {
    dim3 grid(nworkgroups), block(1024);
    if (f.logr == 8) launch_persist_logr<8>(sc, f, pl, a, count, cache, version, grid, block, st);
    else if (f.logr == 9) launch_persist_logr<9>(sc, f, pl, a, count, cache, version, grid, block, st);
    else if (f.logr == 10) launch_persist_logr<10>(sc, f, pl, a, count, cache, version, grid, block, st);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace rtd
