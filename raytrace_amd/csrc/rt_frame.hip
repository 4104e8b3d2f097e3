// rt_frame.hip — k_frame: a whole frame in ONE launch, for frames with little work (round 4).
//
// The reference only ever renders 1024 x 1024 pixels with one sample and two levels (src/render/constants.rs:9-10, dispatch
// pipeline.rs:44-45,86-90): two million rays.  The persistent kernels are built for hundreds of millions — one 1024-thread
// workgroup per CU that first copies a 128 KiB nibble map into LDS, a prepass that writes a worklist, a path kernel that deals
// paths out of global cursors 128 at a time, an accumulate launch — and a launch of them costs ~0.14 ms however little it has to
// do.  k_frame is the opposite trade:
//
//   * 256-thread workgroups, ALL resident at once (about four waves per SIMD): nothing persistent, no global cursor, no global
//     worklist, one barrier; a wave walks 1..4 tiles of 8x8 pixels, chosen so that every workgroup gets the same mix of sky and
//     terrain (tile i of workgroup g is tile i * ngroups + g);
//   * no LDS copy of the scene: a step reads its minefield byte straight from the brick-swizzled array (one 64-byte line per 4^3
//     brick; the wave's rays start next to each other, so L1/L2 serve them) — ONE memory round trip per step where the nibble map
//     costs two on the mixed bricks near a surface, which is where these short rays live;
//   * phase A: a tile's 64 primary rays in lockstep (they are coherent), then the five primary-only planes — and the lighting of
//     sky pixels — for the whole wave at once; the pixels that have paths to walk go, with their primary hit, to a queue in the
//     workgroup's LDS (ballot-ranked, one LDS atomic per wave and tile);
//   * phase B: a lane takes a work item — sample s of queue entry e — and walks that path, a level's shadow and diffuse ray together
//     in two slots (k_persist's machinery: rt_dda.hpp); it parks when a level's rays have ended, and when `threshold` lanes are
//     parked the wave runs one transition pass for all of them, in which lanes whose path is finished take the next items.  (One
//     tile per wave and no queue — a wave's lanes idle until the tile's longest path has ended — had 20 % of the slot-lanes of a
//     step in flight and 410 K wave-steps on the reference's frame; the queue: 324 K.)  One sample per pixel — the reference's
//     frames — and the lane stores the pixel's lighting itself; more, and the paths' lights go to the light-record array and the
//     workgroup adds a pixel's samples in sample order once all its paths have ended (0 + l1 + l2 + ...: k_accumulate_paths'
//     arithmetic): no accumulate launch either way.  With many samples per pixel a workgroup gets fewer tiles than it has waves.
//
// Values are those of raytrace.comp under the rt_math.h contract; planes and exact counters equal the oracle's (cached
// primaries: the primary ray reads no noise, :306-320, and is traced once per pixel).
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "rt_dda.hpp"
#include "rt_device.hpp"
#include "rt_kernels.hpp"

namespace rtd {

constexpr uint32_t kFrameLdsStack = 7;   // albedo-stack levels per lane in LDS: frames up to depth 8
constexpr uint32_t kFrameMaxTilesPerWave = 4;   // sizes the workgroup's pixel queue in LDS (16 bytes per pixel)

#ifndef RT_FRAME_WG_WAVES
#define RT_FRAME_WG_WAVES 4      // waves per workgroup: they share the pixel queue
#endif
constexpr uint32_t kFrameWgWaves = RT_FRAME_WG_WAVES, kFrameWg = 64u * kFrameWgWaves;
#ifndef RT_FRAME_SWZ
#define RT_FRAME_SWZ 1           // 1: the swizzled index of a ray's next texel from three LDS table reads (region 256, lr = 0); 0: from shifts and masks
#endif
#ifndef RT_FRAME_WAVES_PER_SIMD
#define RT_FRAME_WAVES_PER_SIMD 5
#endif
template <int LOGR, bool LRZ, bool COUNT>
__global__ __launch_bounds__(kFrameWg, RT_FRAME_WAVES_PER_SIMD) void k_frame(Scene sc, Frame f, Planes pl, FrameArgs a) {
    __shared__ float s_albedo[128];            // (packed >> k & 0x7F) / 127.0 (raytrace.comp:156-158), exact quotients
    constexpr bool SWZ = RT_FRAME_SWZ != 0 && dda_uses_swz<LOGR, LRZ>();
    __shared__ uint32_t s_swz[SWZ ? 3 * kSwzStride : 1];   // swizzle tables (rt_dda.hpp)
    __shared__ uint32_t s_stack[kFrameLdsStack][kFrameWg];   // packed material of surface j+1 at level j, per lane
    // the workgroup's non-sky pixels: primary hit and — as bits of .w — face id << 28 | gl_WorkGroupID.y * 8 << 14 | gl_WorkGroupID.x * 8 (the
    // noise_offset terms, raytrace.comp:304: the prepass' record format); and where the pixel's planes go (a path never needs its pixel's
    // coordinates again: no division by the tile count in the pass)
    __shared__ float4 s_queue[kFrameMaxTilesPerWave * kFrameWg];
    __shared__ uint32_t s_qout[kFrameMaxTilesPerWave * kFrameWg];
    __shared__ uint32_t s_qtail, s_qhead;
    if (threadIdx.x < 128u) s_albedo[threadIdx.x] = (float)threadIdx.x / 127.0f;
    if (threadIdx.x == 0u) { s_qtail = 0u; s_qhead = 0u; }
    if (SWZ) dda_fill_swz(s_swz, threadIdx.x, kFrameWg);
    __syncthreads();

    constexpr int R = 1 << LOGR, LB = LOGR - 2;
    const float half = (float)R / 2;
    const uint32_t lane = threadIdx.x & 63u, wiw = threadIdx.x >> 6;
    const vec3 sunlight = ld3(f.sunlight);
    const uint32_t D = (uint32_t)f.depth;

    unsigned long long c_prim = 0, c_shadow = 0, c_dif = 0, c_border = 0, c_noise = 0, c_pix = 0;
    unsigned long long d_a = 0, d_b = 0, d_pass = 0, d_pl = 0, d_sl = 0, d_fl = 0;   // wave-uniform structure statistics (counting build, RT_DEBUG_STATS)
#ifdef RT_DIAG_FRAME_TIMES   // diagnostic build (tools/variant.sh): per-wave start / end on the 100 MHz clock, shipped arithmetic otherwise
    constexpr bool kTimes = true;
#else
    constexpr bool kTimes = COUNT;
#endif
    const unsigned long long t_start = kTimes ? wall_clock64() : 0ull;
    unsigned long long t_mid = 0;
    RayTally tl;
    auto advance = [&](RaySlot2& r, uint32_t step) { dda_advance<LOGR, LRZ, COUNT, false, SWZ>(r, step, f, half, c_border, s_swz); };

    RaySlot2 S, F;
    S.px = S.py = S.pz = S.ndx = S.ndy = S.lx = S.ly = S.lz = S.ux = S.uy = S.uz = 0.0f; S.ndz = -1.0f;
    S.vox = S.cidx = 0u; S.nk = PX_HIT << 16; S.axis = 2u;
    S.tracing = false; S.valid = true; S.fresh_invalid = false;
    F = S;

    // ---- phase A: the primary rays (:296-320) of this wave's tiles, in lockstep ------------------------------------------------
    // Tile i of workgroup g is local tile i * ngroups + g: a workgroup's tiles are spread evenly over the image (every workgroup
    // gets the same mix of sky and terrain — all workgroups are resident at once, nobody can take work from a neighbour), and
    // neighbouring workgroups walk neighbouring tiles at the same time.
    // Two tiles at a time, one in each ray slot: a lone primary ray leaves the wave waiting for one byte per step, two independent
    // fetch chains halve the steps of the phase (RT_FRAME_PAIR_A=0: one tile at a time).
    auto arm_primary = [&](RaySlot2& r, const PixelId& pix, vec3& pdir) {
        pdir = v3(0, 0, 1);
        r.nk = PX_HIT << 16; r.axis = 2u;
        if (pix.inside) {
            vec3 start;
            primary_ray(f, pix.px, pix.py, &start, &pdir);
            const vec3 d = vnormalize(pdir);                                                          // raytrace.comp:83
            r.lx = 1.0f / rtm_abs(d.x); r.ly = 1.0f / rtm_abs(d.y); r.lz = 1.0f / rtm_abs(d.z);       // :88
            int ix, iy, iz;
            const bool ok = wrap_texel(start, (float)R, &ix, &iy, &iz);
            dda_arm<LOGR, LRZ, COUNT, true, SWZ>(r, d.x, d.y, d.z, start.x, start.y, start.z, ok, swizzled_index(ix, iy, iz, LB), 0u, f, half,
                                            nullptr, sc, c_border, s_swz);
        }
    };
    // the tile's planes, the lighting of its sky pixels, and its other pixels into the workgroup's queue
    auto finish_primary = [&](const RaySlot2& r, const PixelId& pix, const vec3& pdir) {
        bool queue = false;
        float hx = 0, hy = 0, hz = 0;
        uint32_t nrm = 0;
        if (pix.inside) {
            const uint32_t kind = r2_kind(r);
            const bool air = kind == PX_AIR;
            nrm = r.axis == 0 ? (r.ndx < 0.0f ? 1u : 0u) : (r.axis == 1 ? (r.ndy < 0.0f ? 3u : 2u) : (r.ndz < 0.0f ? 5u : 4u));
            uint32_t material = 0;
            if (kind == PX_HIT && (LRZ || r.valid)) material = sc.mat[r.vox];   // the hit texel is the texel of the last fetch (:150-154)
            hx = r.px; hy = r.py; hz = r.pz;
            if (kind == PX_SPECIAL) { hx = hy = hz = __builtin_nanf(""); }
            const float off = 0.001f;                                           // :166-180
            if (nrm == 0) hx += off; else if (nrm == 1) hx -= off;
            else if (nrm == 2) hy += off; else if (nrm == 3) hy -= off;
            else if (nrm == 4) hz += off; else hz -= off;
            if (COUNT) { c_prim++; c_pix++; dda_tally<LOGR>(r, tl); }
            store_primary_planes(pl, pix.out_index, f, pdir, air, nrm, material, v3(hx, hy, hz));
            if (air || D < 1u) {
                // every sample of this pixel has the same light (no noise is read): summed spp times like the shader's spp frames would
                vec3 light = v3(0.0f, 0.0f, 0.0f);
                if (air) light = sample_sky(pdir, ld3(f.sunangle), sunlight, true);                    // raytrace.comp:321-322
                vec3 sum = v3(0.0f, 0.0f, 0.0f);
                for (int s = 0; s < f.spp; s++) sum = vadd(sum, light);
                store_lighting(pl, pix.out_index, sum, f.spp);
            } else {
                queue = true;
            }
        }
        // ballot-ranked slots, one LDS atomic per wave and tile
        const uint64_t m = __ballot(queue);
        if (m) {
            uint32_t base = 0;
            if (lane == 0u) base = atomicAdd(&s_qtail, (uint32_t)__popcll(m));
            base = __builtin_amdgcn_readfirstlane(base);
            if (queue) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                const uint32_t info = (nrm << 28) | (owning_workgroup((uint32_t)pix.py) * RT_SHADER_GROUP_SIZE) << 14 |
                                      (owning_workgroup((uint32_t)pix.px) * RT_SHADER_GROUP_SIZE);
                s_queue[base + rank] = make_float4(hx, hy, hz, __uint_as_float(info));
                s_qout[base + rank] = pix.out_index;
            }
        }
    };
    // the workgroup's tiles_per_group tiles are dealt to its waves in turn (wave w: tiles w, w + 4, ...); with many samples per pixel a
    // group has fewer tiles than waves — the other waves go straight to the barrier and share the paths
    const uint32_t istep = (a.pair_a ? 2u : 1u) * kFrameWgWaves;
    for (uint32_t i0 = wiw; i0 < a.tiles_per_group; i0 += istep) {
        const uint32_t i1 = i0 + kFrameWgWaves;
        const uint32_t tile0 = i0 * gridDim.x + blockIdx.x, tile1 = i1 * gridDim.x + blockIdx.x;
        if (tile0 >= (uint32_t)f.ntiles_local) break;   // wave-uniform (the tile index grows with i0)
        const bool two = a.pair_a && i1 < a.tiles_per_group && tile1 < (uint32_t)f.ntiles_local;
        const uint32_t lp0 = tile0 * 64u + lane, lp1 = tile1 * 64u + lane;
        const PixelId pix0 = pixel_of_local(f, lp0);
        PixelId pix1 = pix0;
        vec3 pdir0, pdir1 = v3(0, 0, 1);
        arm_primary(F, pix0, pdir0);
        if (two) { pix1 = pixel_of_local(f, lp1); arm_primary(S, pix1, pdir1); }
        while (__ballot(F.tracing || S.tracing)) {
            if (COUNT) { d_a++; d_fl += (uint32_t)__popcll(__ballot(F.tracing)) + (uint32_t)__popcll(__ballot(S.tracing)); }
            uint32_t stS = 0, stF = 0;
            if (F.tracing) stF = sc.mine[F.vox];
            if (S.tracing) stS = sc.mine[S.vox];
            if (F.tracing) advance(F, stF);
            if (S.tracing) advance(S, stS);
        }
        finish_primary(F, pix0, pdir0);
        if (two) finish_primary(S, pix1, pdir1);
    }
    S.tracing = false;
    F.tracing = false;
    if (kTimes) t_mid = wall_clock64();
    __syncthreads();   // the queue is complete
    const uint32_t qtotal = s_qtail;

    // ---- phase B: the paths of the workgroup's non-sky pixels ----------------------------------------------------------------
    // Work item w = s * qtotal + e: sample s of queue entry e (sample-major: a wave's lanes hold neighbouring pixels of one sample).
    // A lane takes an item, walks that path — a level's shadow and diffuse ray together in two slots — and takes the next; lanes are
    // refilled in the transition pass, ballot-ranked, one LDS atomic per wave and pass.  One sample per pixel (the reference's
    // frames): the lane stores the pixel's lighting itself (0 + light: k_accumulate_paths' arithmetic).  More: the path's light goes
    // to the light-record array (pl[pixel * spp + s]) and the workgroup adds a pixel's samples IN SAMPLE ORDER when all its paths
    // have ended (below) — the deterministic fp32 sum the oracle defines, without a launch of its own.
    if (qtotal != 0u) {
        const uint32_t threshold = a.threshold;
        const uint32_t spp = (uint32_t)f.spp;
        const uint32_t nitems = qtotal * spp;   // <= 1024 * spp (the host keeps pixel-samples per frame far below 2^31)
        bool active = false, dry = false;   // active: the lane holds a path; dry (wave-uniform): every item has been handed out
        uint32_t oidx = 0;                  // where the path's pixel's planes go (PixelId::out_index; also its row in the light-record array)
        uint32_t level = 0, sunbits = 0, samp = 0, nvtex = 0;
        uint32_t dif_entry = 0xFFFFFFFFu, sun_entry = 0xFFFFFFFFu;   // table entries held by F's / S's direction registers
        for (;;) {
            const uint64_t m_busy = __ballot(S.tracing || F.tracing);
            const uint64_t m_wait = __ballot(!(S.tracing || F.tracing) && (active || !dry));
            const uint32_t n_busy = (uint32_t)__popcll(m_busy), n_wait = (uint32_t)__popcll(m_wait);
            if (n_wait < threshold && n_busy != 0u) {
                // step loop: both rays of a lane together (their fetches overlap), until enough further lanes have parked
                const uint32_t need = threshold - n_wait;
                const uint32_t target = n_busy > need ? n_busy - need : 0u;
                do {
                    if (COUNT) { d_b++; d_sl += (uint32_t)__popcll(__ballot(S.tracing)); d_fl += (uint32_t)__popcll(__ballot(F.tracing)); }
                    uint32_t stS = 0, stF = 0;
                    if (S.tracing) stS = sc.mine[S.vox];      // :106 for a fresh ray, :137 otherwise
                    if (F.tracing) stF = sc.mine[F.vox];
                    if (S.tracing) advance(S, stS);
                    if (F.tracing) advance(F, stF);
                } while ((uint32_t)__popcll(__ballot(S.tracing || F.tracing)) > target);
            } else if (n_wait == 0u) {
                break;   // nothing in flight, nothing parked, nothing left to take
            }
            // ---- transition pass: lanes whose level has ended, lanes without a path -------------------------------------------
            const bool mine = !(S.tracing || F.tracing) && active;
            if (COUNT) { d_pass++; d_pl += (uint32_t)__popcll(__ballot(mine)); }
            bool begin_level = false;
            float sfx = 0, sfy = 0, sfz = 0;      // surface the next level stands on
            uint32_t snormal = 0;
            if (mine) {
                // diffuse result: the hit texel is the texel of the last fetch, so the material is mat[vox] (:150-154); the position
                // gets the 0.001 face offset (:166-180)
                const uint32_t fkind = r2_kind(F);
                const bool air = fkind == PX_AIR;
                const uint32_t nrm = F.axis == 0 ? (F.ndx < 0.0f ? 1u : 0u) : (F.axis == 1 ? (F.ndy < 0.0f ? 3u : 2u) : (F.ndz < 0.0f ? 5u : 4u));
                if (COUNT) { dda_tally<LOGR>(F, tl); dda_tally<LOGR>(S, tl); }
                if (r2_kind(S) == PX_AIR) sunbits |= 1u << (level - 1);                      // :326-328 / :338-340
                if (air || level == D) {
                    vec3 sky = v3(0, 0, 0);
                    if (air) { const float4 t = a.dif_lut[dif_entry + 3u]; sky = v3(t.x, t.y, t.z); }   // :331-332 / :343-345, tabulated
                    // L_j = [sun_j] S + L_{j+1} * albedo_{j+1} + emission, innermost first (raytrace.comp:346-348)
                    vec3 L = v3(0.0f, 0.0f, 0.0f);
                    if (sunbits >> (level - 1) & 1u) L = vadd(L, sunlight);
                    if (air) L = vadd(L, sky);
                    for (uint32_t j = level - 1; j >= 1u; j--) {
                        const uint32_t pm = s_stack[j - 1][threadIdx.x];
                        vec3 light2 = vmul(L, v3(s_albedo[pm >> 14 & 0x7Fu], s_albedo[pm >> 7 & 0x7Fu], s_albedo[pm & 0x7Fu]));
                        light2 = vadd(light2, v3(0.0f, 0.0f, 0.0f));      // + dif.emission, always vec3(0) (:155)
                        vec3 acc = v3(0.0f, 0.0f, 0.0f);
                        if (sunbits >> (j - 1) & 1u) acc = vadd(acc, sunlight);
                        L = vadd(acc, light2);
                    }
                    const vec3 light = vadd(v3(0.0f, 0.0f, 0.0f), L);
                    if (spp == 1u) store_lighting(pl, oidx, v3(0.0f + light.x, 0.0f + light.y, 0.0f + light.z), f.spp);
                    else a.pl[(size_t)oidx * spp + samp] = PathLight{light.x, light.y, light.z};
                    active = false;
                } else {
                    uint32_t material = 0;
                    if (fkind == PX_HIT && (LRZ || F.valid)) material = sc.mat[F.vox];
                    s_stack[level - 1][threadIdx.x] = material;   // albedo of surface level+1
                    float hx = F.px, hy = F.py, hz = F.pz;
                    if (fkind == PX_SPECIAL) { hx = hy = hz = __builtin_nanf(""); }
                    const float off = 0.001f;
                    if (nrm == 0) hx += off; else if (nrm == 1) hx -= off;
                    else if (nrm == 2) hy += off; else if (nrm == 3) hy -= off;
                    else if (nrm == 4) hz += off; else hz -= off;
                    sfx = hx; sfy = hy; sfz = hz; snormal = nrm;
                    level++; begin_level = true;
                }
            }
            // lanes without a path take the next items
            if (!dry) {
                const uint64_t want = __ballot(!active);
                if (want) {
                    uint32_t base = 0;
                    if (lane == 0u) base = atomicAdd(&s_qhead, (uint32_t)__popcll(want));
                    base = __builtin_amdgcn_readfirstlane(base);
                    const uint32_t avail = base < nitems ? nitems - base : 0u;
                    if (avail < (uint32_t)__popcll(want)) dry = true;   // (the head only grows: nothing will be left for a later pass either)
                    if (!active) {
                        const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                        if (rank < avail) {
                            const uint32_t w = base + rank;
                            samp = spp == 1u ? 0u : w / qtotal;
                            const uint32_t qe = spp == 1u ? w : w - samp * qtotal;
                            const float4 e = s_queue[qe];
                            oidx = s_qout[qe];
                            active = true;
                            // noise_offset of this sample (:298-304) and its noise_value texel (:324, :336).  The bytes are exact
                            // integers in float and the per-level offset (level-1) * 2/512 never reaches the next texel, so one integer
                            // lookup serves every level (tests/test_math_contract.py::test_noise_value_texel_is_level_independent)
                            const uint32_t info = __float_as_uint(e.w);
                            const uint32_t wgx8 = info & 0x3FFFu, wgy8 = (info >> 14) & 0x3FFFu;
                            const uint32_t seed = (f.seed + samp) % (uint32_t)RT_NOISE_BYTES;
                            const uint32_t by = seed / RT_NOISE_SIZE;
                            const uint32_t nb = sc.noise[(by > 511u ? 511u : by) * RT_NOISE_SIZE + seed % RT_NOISE_SIZE];
                            const uint32_t tx = ((nb & 0xFFu) + wgx8) & 511u, ty = (((nb >> 8) & 0xFFu) + wgy8) & 511u;
                            nvtex = sc.noise[ty * RT_NOISE_SIZE + tx];
                            sfx = e.x; sfy = e.y; sfz = e.z; snormal = info >> 28;   // the pixel's primary hit: the surface every sample starts from
                            level = 1; sunbits = 0; begin_level = true;
                        }
                    }
                }
            }
            // both rays of a level (:324-330 / :336-342): shadow ray and diffuse ray from the tables
            if (begin_level) {
                if (COUNT) { c_noise++; c_shadow++; c_dif++; }
                int ix, iy, iz;
                const bool ok = wrap_texel(v3(sfx, sfy, sfz), (float)R, &ix, &iy, &iz);
                const uint32_t vox0 = swizzled_index(ix, iy, iz, LB);
                // stepping never touches a slot's direction registers: the shadow entry depends on the sample's noise texel only
                // (same for all its levels, Q5) and the diffuse entry repeats whenever the next surface has the same face
                const uint32_t se = nvtex & 0xFFFFu;
                if (se != sun_entry) {
                    const float4 sd = a.sun_lut[2u * se], sl = a.sun_lut[2u * se + 1u];
                    S.ndx = -sd.x; S.ndy = -sd.y; S.ndz = -sd.z; S.lx = sl.x; S.ly = sl.y; S.lz = sl.z;
                    sun_entry = se;
                }
                const uint32_t di = 4u * ((snormal << 16) | se);
                if (di != dif_entry) {
                    const float4 d2 = a.dif_lut[di + 1u], dl = a.dif_lut[di + 2u];
                    F.ndx = -d2.x; F.ndy = -d2.y; F.ndz = -d2.z; F.lx = dl.x; F.ly = dl.y; F.lz = dl.z;
                    dif_entry = di;
                }
                dda_arm<LOGR, LRZ, COUNT, true, SWZ>(S, -S.ndx, -S.ndy, -S.ndz, sfx, sfy, sfz, ok, vox0, 0u, f, half, nullptr, sc, c_border, s_swz);
                dda_arm<LOGR, LRZ, COUNT, true, SWZ>(F, -F.ndx, -F.ndy, -F.ndz, sfx, sfy, sfz, ok, vox0, 0u, f, half, nullptr, sc, c_border, s_swz);
            }
        }
        // ---- more than one sample per pixel: the workgroup's ordered sums -------------------------------------------------------
        if (spp != 1u) {
            __threadfence();
            __syncthreads();   // every path of the workgroup's pixels has ended and its record is written (qtotal and spp are workgroup-uniform)
            for (uint32_t e = threadIdx.x; e < qtotal; e += kFrameWg) {
                const uint32_t out = s_qout[e];
                const PathLight* rec = a.pl + (size_t)out * spp;
                float sx = 0.0f, sy = 0.0f, sz = 0.0f;
                for (uint32_t smp = 0; smp < spp; smp++) {
                    const PathLight l = rec[smp];
                    sx = sx + l.x; sy = sy + l.y; sz = sz + l.z;
                }
                store_lighting(pl, out, v3(sx, sy, sz), f.spp);
            }
        }
    }
    if (kTimes && a.dbg_waves && lane == 0u) {   // RT_DEBUG_WAVE_DUMP: (start, end of phase A, end, phase-A steps | phase-B steps << 16 | passes << 32) per wave
        unsigned long long* w = a.dbg_waves + 4u * (blockIdx.x * kFrameWgWaves + wiw);
        w[0] = t_start; w[1] = t_mid; w[2] = wall_clock64(); w[3] = d_a | d_b << 16 | d_pass << 32;
    }
    if (COUNT) {
        DevCounters* cn = a.counters;
        const unsigned long long rays = c_prim + c_shadow + c_dif;
        c_border += tl.border;
        wave_add(&cn->rays, rays); wave_add(&cn->rays_primary, c_prim); wave_add(&cn->rays_shadow, c_shadow);
        wave_add(&cn->rays_diffuse, c_dif); wave_add(&cn->iterations, tl.iter); wave_add(&cn->minefield_fetches, rays + tl.iter);
        wave_add(&cn->hits, tl.hits); wave_add(&cn->material_fetches, tl.hits); wave_add(&cn->sky_exits, tl.sky);
        wave_add(&cn->limit_exits, tl.limit); wave_add(&cn->border_fetches, c_border); wave_add(&cn->noise_fetches, c_noise);
        wave_add(&cn->pixels, c_pix);
        if (lane == 0) {   // dbg_*: phase-A steps in s_execs, phase-B steps in loop_iters, waves with a phase B in f_execs
            atomicAdd(&cn->dbg_s_execs, d_a); atomicAdd(&cn->dbg_loop_iters, d_b); atomicAdd(&cn->dbg_passes, d_pass);
            atomicAdd(&cn->dbg_pass_lanes, d_pl); atomicAdd(&cn->dbg_s_lanes, d_sl); atomicAdd(&cn->dbg_f_lanes, d_fl);
            if (d_pass) atomicAdd(&cn->dbg_f_execs, 1ull);
            atomicMax(&cn->dbg_sky_lanes, d_a + d_b);   // longest wave, in steps
        }
    }
}

bool launch_frame_ok(const Frame& f) { return f.depth <= (int)kFrameLdsStack + 1 && f.logr >= 8 && f.logr <= 10; }

template <int LOGR>
static void launch_frame_logr(const Scene& sc, const Frame& f, const Planes& pl, const FrameArgs& a, bool count, dim3 grid, hipStream_t st) {
    const dim3 block(kFrameWg);
    if (f.lr_zero != 0) {
        if (count) hipLaunchKernelGGL((k_frame<LOGR, true, true>), grid, block, 0, st, sc, f, pl, a);
        else hipLaunchKernelGGL((k_frame<LOGR, true, false>), grid, block, 0, st, sc, f, pl, a);
    } else {
        if (count) hipLaunchKernelGGL((k_frame<LOGR, false, true>), grid, block, 0, st, sc, f, pl, a);
        else hipLaunchKernelGGL((k_frame<LOGR, false, false>), grid, block, 0, st, sc, f, pl, a);
    }
}

hipError_t launch_frame(const Scene& sc, const Frame& f, const Planes& pl, FrameArgs a, bool count, int num_cus, hipStream_t st) {
    if (f.ntiles_local <= 0) return hipSuccess;
    if (!launch_frame_ok(f)) return hipErrorInvalidValue;
    // Tiles per workgroup (four waves, which share the queue of the tiles' pixels): as many as it takes for ALL workgroups to be
    // resident at once — five waves per SIMD at this kernel's register count — so that no workgroup starts late; a wave then refills
    // its lanes from the workgroup's queue instead of leaving them idle while a tile's longest path finishes.  One sample per pixel:
    // whole tiles per wave, 1 ... kFrameMaxTilesPerWave (frames of up to 20 x CUs tiles, 512 x 512 pixels on 256 CUs, have one;
    // measured, ms per frame with 2 / 3 / 4 tiles per wave: 1280 x 720 0.182 / 0.144 / 0.155, 1920 x 1080 0.323 / 0.293 / 0.271).
    // More samples: a tile is spp times the paths, so small frames get FEWER tiles per workgroup than it has waves (down to one: the
    // other three waves skip phase A and share the paths) until the waves fill the GPU.
    const uint32_t ntiles = (uint32_t)f.ntiles_local, resident = 20u * (uint32_t)(num_cus > 0 ? num_cus : 256);
    uint32_t tpg = a.tiles_per_group;
    if (tpg == 0u) {
        if (f.spp == 1) tpg = kFrameWgWaves * ((ntiles + resident - 1u) / resident);
        else tpg = (kFrameWgWaves * ntiles + resident - 1u) / resident;
    }
    if (tpg < 1u) tpg = 1u;
    if (tpg > kFrameWgWaves * kFrameMaxTilesPerWave) tpg = kFrameWgWaves * kFrameMaxTilesPerWave;
    a.tiles_per_group = tpg;
    static const bool pair_a = getenv("RT_FRAME_PAIR_A") == nullptr || atoi(getenv("RT_FRAME_PAIR_A")) != 0;
    a.pair_a = pair_a ? 1u : 0u;
    const dim3 grid((ntiles + tpg - 1u) / tpg);
    if (a.threshold < 1u || a.threshold > 64u) a.threshold = 44u;   // parked lanes per pass: 20 0.167 ms on the reference frame, 28 0.166, 40 0.162, 48 0.161, 56 0.166
    if (f.logr == 8) launch_frame_logr<8>(sc, f, pl, a, count, grid, st);
    else if (f.logr == 9) launch_frame_logr<9>(sc, f, pl, a, count, grid, st);
    else launch_frame_logr<10>(sc, f, pl, a, count, grid, st);
    return hipGetLastError();
}

}  // namespace rtd
