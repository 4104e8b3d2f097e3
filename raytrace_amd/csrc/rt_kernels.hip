// rt_kernels.hip — gfx950 kernels of the ray-trace path.
//
//   flatten      : linear 256^3 arrays -> 4^3-brick-swizzled arrays + per-brick nibble map (src/world -> GPU layout)
//   k_mega       : one thread per pixel, every ray inline (straight restatement; generic in lr; baseline + fallback)
//   k_trace      : persistent wave64 traversal: __ballot lane compaction/refill, nibble map in LDS, SoA ray queue
//   k_shade0/N   : per-path shading between traversal waves: sky, shadow/diffuse ray spawn, light unwinding
//   k_accumulate : ordered per-pixel sum of the batch's samples
//   k_resolve    : lighting plane (fp32 + RGBA16)
//   k_untile     : scatter gathered tile-major planes into a row-major frame (multi-GPU)
//
// No MFMA anywhere: the path is dependent 1-byte gathers + fp32 VALU.
#include <hip/hip_runtime.h>

#include "rt_device.hpp"
#include "rt_kernels.hpp"

namespace rtd {

// =====================================================================================================
// Scene flattening
// =====================================================================================================
// dst index i (swizzled) <- src linear index (x fastest, util.rs:104-106).  Writes are fully coalesced; reads come
// in 4-voxel runs.  Flags minefield values above kMaxStepValue (the reference writes 0..6, chunk.rs:163-183).
__global__ __launch_bounds__(256) void k_flatten_voxels(const uint8_t* __restrict__ mine_lin,
                                                        const uint32_t* __restrict__ mat_lin,
                                                        uint8_t* __restrict__ mine_sw, uint32_t* __restrict__ mat_sw,
                                                        uint32_t* __restrict__ bad_value_flag, int logr) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;   // grid covers exactly R^3 (< 2^31)
    const int lb = logr - 2;
    const uint32_t bmask = (1u << lb) - 1u;
    uint32_t brick = i >> 6, l = i & 63u;
    uint32_t ix = ((brick & bmask) << 2) | (l & 3u);
    uint32_t iy = (((brick >> lb) & bmask) << 2) | ((l >> 2) & 3u);
    uint32_t iz = ((brick >> (2 * lb)) << 2) | (l >> 4);
    size_t src = (((((size_t)iz << logr) + iy) << logr)) + ix;
    uint8_t v = mine_lin[src];
    if (v > kMaxStepValue) atomicOr(bad_value_flag, 1u);
    mine_sw[i] = v;
    mat_sw[i] = mat_lin[src];
}

// rt_upload_slice: the same re-tiling for ONE 16-thick slab (TerrainUploadManager::upload_slice, terrain_upload.rs:84-275 ->
// vkCmdCopyBufferToImage with an offset).  The slab arrives as a dense box of extent 16 along `axis` and R along the other
// two (x fastest); thread i handles swizzled voxel i of the slab's bricks — 4 brick layers along `axis`, whole bricks, so every
// thread writes inside one 64-byte line run.  (The slab's values were checked on the host before it got here: rt_upload_slice.)
__global__ __launch_bounds__(256) void k_flatten_slab(const uint8_t* __restrict__ mine_slab, const uint32_t* __restrict__ mat_slab,
                                                      uint8_t* __restrict__ mine_sw, uint32_t* __restrict__ mat_sw,
                                                      int logr, int axis, int offset) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;   // grid covers exactly 16 * R^2
    const int lb = logr - 2;
    const uint32_t bmask = (1u << lb) - 1u;
    const uint32_t l = i & 63u, sb = i >> 6;              // sb: brick within the slab, 4 layers along `axis`
    // brick coordinates: the two full axes take lb bits each, the slab axis 2 bits (layer) on top of offset/4
    uint32_t bc[3];
    uint32_t rest = sb;
    for (int a = 0; a < 3; a++) {
        if (a == axis) { bc[a] = (uint32_t)(offset >> 2) + (rest & 3u); rest >>= 2; }
        else { bc[a] = rest & bmask; rest >>= lb; }
    }
    const uint32_t ix = (bc[0] << 2) | (l & 3u), iy = (bc[1] << 2) | ((l >> 2) & 3u), iz = (bc[2] << 2) | (l >> 4);
    const uint32_t R = 1u << logr;
    const uint32_t sx = axis == 0 ? ix - (uint32_t)offset : ix, sy = axis == 1 ? iy - (uint32_t)offset : iy,
                   sz = axis == 2 ? iz - (uint32_t)offset : iz;
    const uint32_t ex = axis == 0 ? (uint32_t)RT_SLICE_SIZE : R, ey = axis == 1 ? (uint32_t)RT_SLICE_SIZE : R;
    const size_t src = ((size_t)sz * ey + sy) * ex + sx;
    const size_t dst = ((((size_t)bc[2] << lb) + bc[1]) << lb) + bc[0];
    mine_sw[(dst << 6) | l] = mine_slab[src];
    mat_sw[(dst << 6) | l] = mat_slab[src];
}

// One thread per nibble-map word = 8 consecutive coarse cubes (x-adjacent).  A coarse cube has edge R/64 and is made of
// (R/256)^3 4^3-bricks of 64 contiguous bytes each in the swizzled minefield.
// (word0, nwords[3]): the box of words to rebuild — the whole map for an upload, the layers a slab touches for rt_upload_slice.
__global__ __launch_bounds__(256) void k_build_coarse(const uint8_t* __restrict__ mine_sw, uint32_t* __restrict__ coarse, int logr,
                                                      uint3 word0, uint3 nwords) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nwords.x * nwords.y * nwords.z) return;
    // word (wx, cy, cz): wx in [0, 8) covers coarse cubes cx = 8 wx .. 8 wx + 7
    const uint32_t wx = word0.x + t % nwords.x, cy_ = word0.y + (t / nwords.x) % nwords.y, cz_ = word0.z + t / (nwords.x * nwords.y);
    const uint32_t w = (cz_ << 9) | (cy_ << 3) | wx;
    const int lb = logr - 2, sub = logr - 8;               // sub: log2(4^3-bricks per coarse cube edge)
    const uint32_t nsub = 1u << sub;
    uint32_t word = 0;
    for (uint32_t b = 0; b < 8u; b++) {
        const uint32_t c = w * 8u + b;                     // coarse cube (cz, cy, cx), 6 bits each
        const uint32_t cx = c & 63u, cy = (c >> 6) & 63u, cz = c >> 12;
        uint32_t first = 0, diff = 0;
        for (uint32_t bz = 0; bz < nsub; bz++)
            for (uint32_t by = 0; by < nsub; by++)
                for (uint32_t bx = 0; bx < nsub; bx++) {
                    const uint32_t brick = (((((cz << sub) + bz) << lb) + ((cy << sub) + by)) << lb) + ((cx << sub) + bx);
                    const uint4* src = reinterpret_cast<const uint4*>(mine_sw + ((size_t)brick << 6));
                    const uint4 a = src[0], q = src[1], d = src[2], e = src[3];
                    if ((bz | by | bx) == 0u) first = a.x & 0xFFu;
                    const uint32_t splat = first * 0x01010101u;
                    diff |= (a.x ^ splat) | (a.y ^ splat) | (a.z ^ splat) | (a.w ^ splat) | (q.x ^ splat) | (q.y ^ splat) |
                            (q.z ^ splat) | (q.w ^ splat) | (d.x ^ splat) | (d.y ^ splat) | (d.z ^ splat) | (d.w ^ splat) |
                            (e.x ^ splat) | (e.y ^ splat) | (e.z ^ splat) | (e.w ^ splat);
                }
        const uint32_t nib = (diff == 0u && first < kNibMixed) ? first : kNibMixed;
        word |= nib << (4 * b);
    }
    coarse[w] = word;
}

// Regions above 256 (round 4): the per-BRICK nibble map behind the coarse one (Scene::brick).  One thread per word = 8 x-adjacent
// bricks; brick (bz, by, bx) is 64 consecutive bytes of the swizzled minefield at index ((bz << lb | by) << lb | bx) << 6.
// (word0, nwords): the box of words to rebuild, in (x word, brick y, brick z) — everything for an upload, a slab's for rt_upload_slice.
__global__ __launch_bounds__(256) void k_build_brick(const uint8_t* __restrict__ mine_sw, uint32_t* __restrict__ brick_words, int lb,
                                                     uint3 word0, uint3 nwords) {
    const uint32_t t = blockIdx.x * 256u + threadIdx.x;
    if (t >= nwords.x * nwords.y * nwords.z) return;
    const uint32_t wx = word0.x + t % nwords.x, by = word0.y + (t / nwords.x) % nwords.y, bz = word0.z + t / (nwords.x * nwords.y);
    const uint32_t w = (((bz << lb) + by) << (lb - 3)) + wx;      // word index: 8 bricks per word along x
    uint32_t word = 0;
    for (uint32_t b = 0; b < 8u; b++) {
        const uint4* src = reinterpret_cast<const uint4*>(mine_sw + ((size_t)(w * 8u + b) << 6));
        const uint4 a = src[0], q = src[1], d = src[2], e = src[3];
        const uint32_t first = a.x & 0xFFu, splat = first * 0x01010101u;
        const uint32_t diff = (a.x ^ splat) | (a.y ^ splat) | (a.z ^ splat) | (a.w ^ splat) | (q.x ^ splat) | (q.y ^ splat) |
                              (q.z ^ splat) | (q.w ^ splat) | (d.x ^ splat) | (d.y ^ splat) | (d.z ^ splat) | (d.w ^ splat) |
                              (e.x ^ splat) | (e.y ^ splat) | (e.z ^ splat) | (e.w ^ splat);
        word |= ((diff == 0u && first < kNibMixed) ? first : kNibMixed) << (4 * b);
    }
    brick_words[w] = word;
}

// =====================================================================================================
// k_mega — one thread per pixel
// =====================================================================================================
template <bool COUNT>
__global__ __launch_bounds__(256) void k_mega(Scene sc, Frame f, Planes pl, DevCounters* cn) {
    uint32_t lp = blockIdx.x * 256u + threadIdx.x;
    PixelId pix = pixel_of_local(f, lp);
    unsigned long long c_rays = 0, c_prim = 0, c_shadow = 0, c_dif = 0, c_iter = 0, c_mat = 0, c_noise = 0, c_hits = 0,
                       c_sky = 0, c_limit = 0, c_border = 0, c_pix = 0;
    if (pix.inside) {
        const vec3 sunangle = ld3(f.sunangle), sunlight = ld3(f.sunlight);
        vec3 start, dir;
        primary_ray(f, pix.px, pix.py, &start, &dir);
        vec3 sum = v3(0.0f, 0.0f, 0.0f);
        for (int s = 0; s < f.spp; s++) {
            uint32_t seed = (f.seed + (uint32_t)s) % (uint32_t)RT_NOISE_BYTES;
            auto tally = [&](const Hit& h) {
                if (COUNT) {
                    c_rays++; c_iter += h.iterations; c_border += h.border; c_limit += h.limit_exit;
                    if (h.air) c_sky++; else if (!h.limit_exit) { c_hits++; c_mat++; }
                }
            };
            Hit primary = trace_ray_generic(sc, f, start, dir);
            tally(primary); if (COUNT) c_prim++;
            if (s == 0) {
                store_primary_planes(pl, pix.out_index, f, dir, primary.air, primary.normal, primary.material, primary.position);
                if (COUNT) c_pix++;
            }
            vec3 light = v3(0.0f, 0.0f, 0.0f);
            if (primary.air) {
                light = sample_sky(dir, sunangle, sunlight, true);
            } else if (f.depth >= 1) {
                NoiseOffset no = noise_offset_of(sc, seed, pix.px, pix.py);
                uint32_t stack[RT_MAX_DEPTH];
                uint32_t sunbits = 0;
                Hit surface = primary;
                int K = 0; bool terminal_sky = false; vec3 sky = v3(0.0f, 0.0f, 0.0f);
                for (int level = 1; level <= f.depth; level++) {
                    uint32_t nv = noise_value_texel(sc, no, level);
                    if (COUNT) c_noise++;
                    float nr = unorm8(nv, 0), ng = unorm8(nv, 1);
                    Hit sun = trace_ray_generic(sc, f, surface.position, sun_ray_direction(sunangle, nr, ng));
                    tally(sun); if (COUNT) c_shadow++;
                    if (sun.air) sunbits |= 1u << (level - 1);
                    vec3 ddir = diffuse_direction(surface.normal, nr, ng);
                    Hit dif = trace_ray_generic(sc, f, surface.position, ddir);
                    tally(dif); if (COUNT) c_dif++;
                    K = level;
                    if (dif.air) { terminal_sky = true; sky = sample_sky(ddir, sunangle, sunlight, true); break; }
                    if (level == f.depth) break;
                    stack[level] = dif.material;     // albedo of surface level+1
                    surface = dif;
                }
                vec3 L1 = unwind_light(K, sunbits, terminal_sky, sky, sunlight, [&](int j) { return stack[j]; });
                light = vadd(light, L1);
            }
            sum = vadd(sum, light);
        }
        store_lighting(pl, pix.out_index, sum, f.spp);
    }
    if (COUNT) {
        wave_add(&cn->rays, c_rays); wave_add(&cn->rays_primary, c_prim); wave_add(&cn->rays_shadow, c_shadow);
        wave_add(&cn->rays_diffuse, c_dif); wave_add(&cn->iterations, c_iter);
        wave_add(&cn->minefield_fetches, c_rays + c_iter); wave_add(&cn->material_fetches, c_mat);
        wave_add(&cn->noise_fetches, c_noise); wave_add(&cn->hits, c_hits); wave_add(&cn->sky_exits, c_sky);
        wave_add(&cn->limit_exits, c_limit); wave_add(&cn->border_fetches, c_border); wave_add(&cn->pixels, c_pix);
    }
}

// =====================================================================================================
// k_trace — persistent wavefront traversal
// =====================================================================================================
// Grid = one 1024-thread workgroup per CU (16 waves share one 128 KiB nibble map in LDS).  Each lane walks one ray;
// finished lanes park until at least `refill_threshold` lanes of the wave are idle, then the wave retires their
// results and pulls new rays from the global cursor in one atomic: the k-th idle lane (by __ballot/mbcnt rank) takes
// ray base+k, so consecutive rays stay in neighbouring lanes and queue reads coalesce.
//
// Loop restructuring relative to raytrace.comp:106-162 (same values, different grouping): the fetch at the top of
// an iteration belongs to the previous advance (or is the initial fetch for a fresh ray), and the sky test runs
// before it — with lr == 0 a position that passes the sky test has p+128 in [0,256], so mod(p+128,256) is
// floor-identical to (int)(p+128) & 255 and the border case cannot occur; the fetch the shader performs for a
// position that then turns out to be sky is skipped (its value is never used).
enum : uint32_t { EXIT_AIR = 0, EXIT_HIT = 1, EXIT_LIMIT = 2, EXIT_SPECIAL = 3 };

template <int MODE /*0: primary rays from pixel ids, 1: ray queue*/, bool LRZ, bool COUNT>
__global__ __launch_bounds__(1024, 4) void k_trace(Scene sc, Frame f, TraceArgs a) {
    __shared__ uint32_t s_coarse[kCoarseWords];
    {
        const uint4* src = reinterpret_cast<const uint4*>(sc.coarse);
        uint4* dst = reinterpret_cast<uint4*>(s_coarse);
        for (uint32_t i = threadIdx.x; i < kCoarseWords / 4; i += 1024u) dst[i] = src[i];
    }
    __syncthreads();

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t nslots = MODE == 0 ? 0u : *a.qcount;
    const uint32_t total = MODE == 0 ? a.nprimary : 2u * nslots;
    const uint32_t threshold = a.refill_threshold;
    const float half = f.region / 2;   // this kernel is built for the reference region size only (logr == 8)
    const int lb = 6;

    bool active = false, pending = false, exhausted = false;
    // ray state
    float px = 0, py = 0, pz = 0, dx = 0, dy = 0, dz = 0, lx = 0, ly = 0, lz = 0, ux = 0, uy = 0, uz = 0;
    uint32_t sgnx = 0, sgny = 0, sgnz = 0, vox = 0, n = 0, ray = 0, axis = 2, kind = 0;
    bool valid = true, fresh = false, fresh_invalid = false;
    unsigned long long c_rays = 0, c_iter = 0, c_hits = 0, c_sky = 0, c_limit = 0, c_border = 0;

    for (;;) {
        const uint64_t idle = __ballot(!active);
        const uint32_t nidle = (uint32_t)__popcll(idle);
        if (nidle >= (exhausted ? 64u : threshold)) {
            // ---- retire finished rays ----------------------------------------------------------------
            if (!active && pending) {
                pending = false;
                const bool is_sun = MODE == 1 && ray < nslots;
                const uint32_t slot = MODE == 0 ? ray : (is_sun ? ray : ray - nslots);
                const uint32_t path = MODE == 0 ? ray : a.qid[slot];
                if (COUNT) {
                    c_rays++; c_iter += n;
                    if (kind == EXIT_AIR) {
                        c_sky++;
                        int tx, ty, tz;   // the fetch the shader makes before its sky test may hit the border
                        if (!wrap_texel(v3(px, py, pz), f.region, &tx, &ty, &tz)) c_border++;
                    } else if (kind == EXIT_LIMIT) c_limit++;
                    else c_hits++;
                    if (kind == EXIT_SPECIAL) c_border += 1u + (fresh_invalid ? 1u : 0u);
                    else if (fresh_invalid) c_border++;
                }
                if (is_sun) {
                    a.sunres[path] = kind == EXIT_AIR ? 1 : 0;
                } else {
                    uint32_t nrm = axis == 0 ? (dx > 0.0f ? 1u : 0u) : (axis == 1 ? (dy > 0.0f ? 3u : 2u) : (dz > 0.0f ? 5u : 4u));
                    uint32_t material = 0;
                    if (kind == EXIT_HIT) material = fetch_material(sc, v3(px, py, pz), f.region, lb);         // raytrace.comp:150-154
                    if (kind == EXIT_SPECIAL) { px = py = pz = __builtin_nanf(""); }
                    const float off = 0.001f;                                                  // :166-180
                    if (nrm == 0) px += off; else if (nrm == 1) px -= off;
                    else if (nrm == 2) py += off; else if (nrm == 3) py -= off;
                    else if (nrm == 4) pz += off; else pz -= off;
                    a.hx[path] = px; a.hy[path] = py; a.hz[path] = pz;
                    a.hinfo[path] = material | (nrm << 24) | (kind == EXIT_AIR ? 0x80000000u : 0u) |
                                    (kind == EXIT_LIMIT ? 0x40000000u : 0u);
                }
            }
            if (exhausted) break;   // all 64 lanes idle and retired
            // ---- refill -----------------------------------------------------------------------------
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(a.cursor, nidle);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base + nidle >= total) exhausted = true;
            if (!active) {
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
                const uint32_t r = base + rank;
                bool take = r < total;
                vec3 o = v3(0, 0, 0), d = v3(0, 0, 1);
                if (take) {
                    ray = r;
                    if (MODE == 0) {
                        uint32_t lp = r % a.npix_pad;
                        PixelId pix = pixel_of_local(f, lp);
                        take = pix.inside;          // padding pixels of partial tiles carry no ray
                        primary_ray(f, pix.px, pix.py, &o, &d);
                    } else {
                        const bool is_sun = r < nslots;
                        const uint32_t slot = is_sun ? r : r - nslots;
                        o = v3(a.qox[slot], a.qoy[slot], a.qoz[slot]);
                        const uint32_t di = is_sun ? slot : a.qcap + slot;
                        d = v3(a.qdx[di], a.qdy[di], a.qdz[di]);
                    }
                }
                if (take) {
                    d = vnormalize(d);                                                         // raytrace.comp:83
                    px = o.x; py = o.y; pz = o.z; dx = d.x; dy = d.y; dz = d.z;
                    lx = 1.0f / rtm_abs(dx); ly = 1.0f / rtm_abs(dy); lz = 1.0f / rtm_abs(dz);   // :88
                    sgnx = dx > 0.0f ? 0x80000000u : 0u; sgny = dy > 0.0f ? 0x80000000u : 0u;   // :94-98 (muls = -1 / +1)
                    sgnz = dz > 0.0f ? 0x80000000u : 0u;
                    ux = px + half; uy = py + half; uz = pz + half;
                    int ix, iy, iz;
                    valid = wrap_texel(o, f.region, &ix, &iy, &iz);                               // :106 (Q6: no bounds test)
                    fresh_invalid = !valid;
                    vox = swizzled_index(ix, iy, iz, lb);
                    n = 0; axis = 2; fresh = true; kind = EXIT_HIT;
                    if (dx != dx || dy != dy || dz != dz) { kind = EXIT_SPECIAL; n = 1; pending = true; }   // NaN direction
                    else active = true;
                }
            }
            if (__ballot(active) == 0ull) continue;   // nothing to step (e.g. only NaN rays were pulled)
        }
        if (active) {
            // ---- fetch (raytrace.comp:106 for a fresh ray, :137 otherwise) ---------------------------
            const uint32_t brick = vox >> 6;
            const uint32_t word = s_coarse[brick >> 3];
            uint32_t step = (word >> ((brick & 7u) << 2)) & 15u;
            if (step == kNibMixed) step = sc.mine[vox];
            if (!valid) step = 0u;
            if (step == 0u) {
                // hit (:146); a fresh ray that starts on a 0 has step_size 0 => mod(x,0) = NaN => defined outcome
                kind = fresh ? EXIT_SPECIAL : EXIT_HIT;
                if (fresh) n = 1;
                active = false; pending = true;
            } else if (n == (uint32_t)RT_TRACE_LIMIT) {
                kind = EXIT_LIMIT; active = false; pending = true;                             // :109 (Q8)
            } else {
                fresh = false;
                // ---- advance (:119-136) --------------------------------------------------------------
                const uint32_t sb = (step << 23) + (126u << 23);          // float(2^(step-1)) = (1<<step)/2
                const float s = __builtin_bit_cast(float, sb);
                const float is = __builtin_bit_cast(float, 0x7F000000u - sb);   // exactly 1/s
                const float qx = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, ux) ^ sgnx);
                const float qy = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, uy) ^ sgny);
                const float qz = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, uz) ^ sgnz);
                // mod(q, s) = q - s*floor(q/s); q/s == q*is and s*floor(.) are exact for a power-of-two s, so the
                // fused form below rounds once exactly like the two-step form.
                const float mx = __builtin_fmaf(-s, rtm_floor(qx * is), qx);
                const float my = __builtin_fmaf(-s, rtm_floor(qy * is), qy);
                const float mz = __builtin_fmaf(-s, rtm_floor(qz * is), qz);
                const float tx = (0.0001f + mx) * lx, ty = (0.0001f + my) * ly, tz = (0.0001f + mz) * lz;
                const bool xy = tx < ty;
                const float m1 = xy ? tx : ty;
                const bool useZ = !(m1 < tz);
                const float t = useZ ? tz : m1;
                axis = useZ ? 2u : (xy ? 0u : 1u);
                px = __builtin_fmaf(dx, t, px); py = __builtin_fmaf(dy, t, py); pz = __builtin_fmaf(dz, t, pz);   // fused (rt_math.h contract)
                n++;
                ux = px + half; uy = py + half; uz = pz + half;
                // ---- sky test (:138-145), then address of the next fetch ------------------------------
                const bool sky = rtm_abs(px - f.lr[0]) >= half || rtm_abs(py - f.lr[1]) >= half || rtm_abs(pz - f.lr[2]) >= half;
                if (sky) {
                    kind = EXIT_AIR; active = false; pending = true;
                } else if (LRZ) {
                    const int ix = (int)ux & 255, iy = (int)uy & 255, iz = (int)uz & 255;
                    vox = swizzled_index(ix, iy, iz, lb);
                } else {
                    int ix, iy, iz;
                    valid = wrap_texel(v3(px, py, pz), f.region, &ix, &iy, &iz);
                    if (COUNT && !valid) c_border++;
                    vox = swizzled_index(ix, iy, iz, lb);
                }
            }
        }
    }
    if (COUNT) {
        DevCounters* cn = a.counters;
        wave_add(&cn->rays, c_rays); wave_add(&cn->iterations, c_iter); wave_add(&cn->minefield_fetches, c_rays + c_iter);
        wave_add(&cn->hits, c_hits); wave_add(&cn->material_fetches, c_hits); wave_add(&cn->sky_exits, c_sky);
        wave_add(&cn->limit_exits, c_limit); wave_add(&cn->border_fetches, c_border);
    }
}

// =====================================================================================================
// Shading stages
// =====================================================================================================
// Append one (shadow, diffuse) ray pair per spawning lane: one atomicAdd per wave, ballot-rank slots.
__device__ __forceinline__ void spawn_pair(const ShadeArgs& a, bool spawn, uint32_t path, vec3 origin, vec3 sun_dir, vec3 dif_dir) {
    const uint64_t m = __ballot(spawn);
    if (m == 0ull) return;
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t base = 0;
    if (lane == (uint32_t)__builtin_ctzll(m)) base = atomicAdd(a.qcount_next, (uint32_t)__popcll(m));
    base = __shfl(base, __builtin_ctzll(m), 64);
    if (spawn) {
        const uint32_t slot = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        a.qox[slot] = origin.x; a.qoy[slot] = origin.y; a.qoz[slot] = origin.z;
        a.qdx[slot] = sun_dir.x; a.qdy[slot] = sun_dir.y; a.qdz[slot] = sun_dir.z;
        a.qdx[a.qcap + slot] = dif_dir.x; a.qdy[a.qcap + slot] = dif_dir.y; a.qdz[a.qcap + slot] = dif_dir.z;
        a.qid[slot] = path;
    }
}

// Level-`level` rays of a path standing on `surface` (position, face id): raytrace.comp:324-330 / :336-342.
__device__ __forceinline__ void make_level_rays(const Scene& sc, const Frame& f, NoiseOffset no, int level, uint32_t normal,
                                                vec3* sun_dir, vec3* dif_dir) {
    uint32_t nv = noise_value_texel(sc, no, level);
    float nr = unorm8(nv, 0), ng = unorm8(nv, 1);
    *sun_dir = sun_ray_direction(ld3(f.sunangle), nr, ng);
    *dif_dir = diffuse_direction(normal, nr, ng);
}

// After the primary wave: G-buffer planes, sky for air pixels, first shadow/diffuse pair otherwise.
template <bool COUNT>
__global__ __launch_bounds__(256) void k_shade0(Scene sc, Frame f, ShadeArgs a, Planes pl) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    bool spawn = false;
    vec3 origin = v3(0, 0, 0), sun_dir = v3(0, 0, 0), dif_dir = v3(0, 0, 0);
    unsigned long long c_noise = 0, c_pix = 0, c_prim = 0;
    if (p < a.npaths) {
        const uint32_t b = p / a.npix_pad, lp = p % a.npix_pad;
        PixelId pix = pixel_of_local(f, lp);
        uint8_t state = 0;
        if (pix.inside) {
            if (COUNT) c_prim++;
            const uint32_t info = a.hinfo[p];
            const bool air = (info & 0x80000000u) != 0u;
            const uint32_t normal = (info >> 24) & 7u, material = info & 0x1FFFFFu;
            const vec3 pos = v3(a.hx[p], a.hy[p], a.hz[p]);
            vec3 start, dir;
            primary_ray(f, pix.px, pix.py, &start, &dir);
            if (a.sample0 + b == 0u) {
                store_primary_planes(pl, pix.out_index, f, dir, air, normal, material, pos);
                if (COUNT) c_pix++;
            }
            if (air) {
                vec3 light = sample_sky(dir, ld3(f.sunangle), ld3(f.sunlight), true);            // raytrace.comp:321-322
                a.plx[p] = light.x; a.ply[p] = light.y; a.plz[p] = light.z;
            } else if (f.depth < 1) {
                a.plx[p] = 0.0f; a.ply[p] = 0.0f; a.plz[p] = 0.0f;
            } else {
                const uint32_t seed = (f.seed + a.sample0 + b) % (uint32_t)RT_NOISE_BYTES;
                NoiseOffset no = noise_offset_of(sc, seed, pix.px, pix.py);
                make_level_rays(sc, f, no, 1, normal, &sun_dir, &dif_dir);
                if (COUNT) c_noise++;
                origin = pos;
                a.pdx[p] = dif_dir.x; a.pdy[p] = dif_dir.y; a.pdz[p] = dif_dir.z;
                a.pnormal[p] = (uint8_t)normal;
                a.sunbits[p] = 0u;
                spawn = true;
                state = 1;
            }
        }
        a.pstate[p] = state;
    }
    spawn_pair(a, spawn, p, origin, sun_dir, dif_dir);
    if (COUNT) {
        wave_add(&a.counters->noise_fetches, c_noise); wave_add(&a.counters->pixels, c_pix);
        wave_add(&a.counters->rays_primary, c_prim);
    }
}

// After the level-`level` wave (level >= 1): fold in the shadow result, terminate on sky / depth, or descend.
template <bool COUNT>
__global__ __launch_bounds__(256) void k_shadeN(Scene sc, Frame f, ShadeArgs a, int level) {
    const uint32_t p = blockIdx.x * 256u + threadIdx.x;
    bool spawn = false;
    vec3 origin = v3(0, 0, 0), sun_dir = v3(0, 0, 0), dif_dir = v3(0, 0, 0);
    unsigned long long c_noise = 0, c_pairs = 0;
    if (p < a.npaths && a.pstate[p] != 0) {
        if (COUNT) c_pairs++;
        uint32_t sunbits = a.sunbits[p] | ((uint32_t)a.sunres[p] << (level - 1));
        const uint32_t info = a.hinfo[p];
        const bool air = (info & 0x80000000u) != 0u;
        if (air || level == f.depth) {
            vec3 sky = v3(0, 0, 0);
            if (air) sky = sample_sky(v3(a.pdx[p], a.pdy[p], a.pdz[p]), ld3(f.sunangle), ld3(f.sunlight), true);   // :331-332 / :343-345
            vec3 L1 = unwind_light(level, sunbits, air, sky, ld3(f.sunlight),
                                   [&](int j) { return a.stack[(size_t)(j - 1) * a.npaths_cap + p]; });
            vec3 light = vadd(v3(0.0f, 0.0f, 0.0f), L1);
            a.plx[p] = light.x; a.ply[p] = light.y; a.plz[p] = light.z;
            a.pstate[p] = 0;
        } else {
            const uint32_t normal = (info >> 24) & 7u;
            a.stack[(size_t)(level - 1) * a.npaths_cap + p] = info & 0x1FFFFFu;   // albedo of surface level+1
            a.sunbits[p] = sunbits;
            const uint32_t b = p / a.npix_pad, lp = p % a.npix_pad;
            PixelId pix = pixel_of_local(f, lp);
            const uint32_t seed = (f.seed + a.sample0 + b) % (uint32_t)RT_NOISE_BYTES;
            NoiseOffset no = noise_offset_of(sc, seed, pix.px, pix.py);
            make_level_rays(sc, f, no, level + 1, normal, &sun_dir, &dif_dir);
            if (COUNT) c_noise++;
            origin = v3(a.hx[p], a.hy[p], a.hz[p]);
            a.pdx[p] = dif_dir.x; a.pdy[p] = dif_dir.y; a.pdz[p] = dif_dir.z;
            a.pnormal[p] = (uint8_t)normal;
            spawn = true;
        }
    }
    spawn_pair(a, spawn, p, origin, sun_dir, dif_dir);
    if (COUNT) {
        wave_add(&a.counters->noise_fetches, c_noise);
        wave_add(&a.counters->rays_shadow, c_pairs); wave_add(&a.counters->rays_diffuse, c_pairs);
    }
}

// acc[pixel] += light of each of the batch's samples, in sample order (deterministic fp32 sum).
__global__ __launch_bounds__(256) void k_accumulate(const float* __restrict__ plx, const float* __restrict__ ply,
                                                    const float* __restrict__ plz, float4* __restrict__ acc,
                                                    uint32_t npix_pad, uint32_t nsamples, int first_batch) {
    const uint32_t lp = blockIdx.x * 256u + threadIdx.x;
    if (lp >= npix_pad) return;
    float4 v = first_batch ? make_float4(0.0f, 0.0f, 0.0f, 0.0f) : acc[lp];
    for (uint32_t b = 0; b < nsamples; b++) {
        const size_t p = (size_t)b * npix_pad + lp;
        v.x = v.x + plx[p]; v.y = v.y + ply[p]; v.z = v.z + plz[p];
    }
    acc[lp] = v;
}

__global__ __launch_bounds__(256) void k_resolve(Frame f, const float4* __restrict__ acc, Planes pl, uint32_t npix_pad) {
    const uint32_t lp = blockIdx.x * 256u + threadIdx.x;
    if (lp >= npix_pad) return;
    PixelId pix = pixel_of_local(f, lp);
    if (!pix.inside) return;
    const float4 v = acc[lp];
    store_lighting(pl, pix.out_index, v3(v.x, v.y, v.z), f.spp);
}

// Gathered planes are rank-major: rank r's plane [capacity tiles][64 px][bpp bytes] starts at r * rank_stride.
// Tile j of rank r is global tile r + j*world.
__global__ __launch_bounds__(256) void k_untile(const uint8_t* __restrict__ gathered, size_t rank_stride, uint8_t* __restrict__ frame,
                                                int world, int capacity, int tiles_x, int tiles_y, int width, int height, int bpp) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;   // one thread per gathered pixel
    const uint32_t total = (uint32_t)world * (uint32_t)capacity * 64u;
    if (i >= total) return;
    const uint32_t l = i & 63u, j = (i >> 6) % (uint32_t)capacity, r = (i >> 6) / (uint32_t)capacity;
    const uint32_t t = r + j * (uint32_t)world;
    if (t >= (uint32_t)(tiles_x * tiles_y)) return;
    const int px = (int)(t % (uint32_t)tiles_x) * 8 + (int)(l & 7u), py = (int)(t / (uint32_t)tiles_x) * 8 + (int)(l >> 3);
    if (px >= width || py >= height) return;
    const uint8_t* s = gathered + (size_t)r * rank_stride + ((size_t)j * 64u + l) * bpp;
    uint8_t* d = frame + ((size_t)py * width + px) * bpp;
    for (int k = 0; k < bpp; k++) d[k] = s[k];
}

// =====================================================================================================
// Host-callable launchers (declared in rt_kernels.hpp)
// =====================================================================================================
hipError_t launch_flatten(const uint8_t* mine_lin, const uint32_t* mat_lin, uint8_t* mine_sw, uint32_t* mat_sw,
                          uint32_t* coarse, uint32_t* brick, uint32_t* bad_flag, int logr, hipStream_t st) {
    hipLaunchKernelGGL(k_flatten_voxels, dim3((1u << (3 * logr)) / 256u), dim3(256), 0, st, mine_lin, mat_lin, mine_sw, mat_sw,
                       bad_flag, logr);
    hipLaunchKernelGGL(k_build_coarse, dim3(kCoarseWords / 256), dim3(256), 0, st, mine_sw, coarse, logr, make_uint3(0, 0, 0),
                       make_uint3(8, 64, 64));
    if (logr > 8 && brick != nullptr) {
        const int lb = logr - 2;
        const uint32_t nb = 1u << lb, n = (nb / 8u) * nb * nb;
        hipLaunchKernelGGL(k_build_brick, dim3((n + 255u) / 256u), dim3(256), 0, st, mine_sw, brick, lb, make_uint3(0, 0, 0), make_uint3(nb / 8u, nb, nb));
    }
    return hipGetLastError();
}

hipError_t launch_flatten_slab(const uint8_t* mine_slab, const uint32_t* mat_slab, uint8_t* mine_sw, uint32_t* mat_sw, uint32_t* coarse,
                               uint32_t* brick, int logr, int axis, int offset, hipStream_t st) {
    const uint32_t R = 1u << logr;
    hipLaunchKernelGGL(k_flatten_slab, dim3(RT_SLICE_SIZE * R * R / 256u), dim3(256), 0, st, mine_slab, mat_slab, mine_sw, mat_sw,
                       logr, axis, offset);
    // nibble-map entries the slab touches: coarse cubes have edge R/64, so 16 voxels are 1024/R layers (4, 2, 1) — rounded out
    // to whole words along x (a word holds 8 x-adjacent cubes, all recomputed from the re-tiled bytes)
    const uint32_t e = R / 64u;
    const uint32_t c0 = (uint32_t)offset / e, c1 = ((uint32_t)offset + RT_SLICE_SIZE - 1u) / e;   // inclusive cube range along `axis`
    uint3 w0 = make_uint3(0, 0, 0), nw = make_uint3(8, 64, 64);
    if (axis == 0) { w0.x = c0 / 8u; nw.x = c1 / 8u - w0.x + 1u; }
    else if (axis == 1) { w0.y = c0; nw.y = c1 - c0 + 1u; }
    else { w0.z = c0; nw.z = c1 - c0 + 1u; }
    const uint32_t n = nw.x * nw.y * nw.z;
    hipLaunchKernelGGL(k_build_coarse, dim3((n + 255u) / 256u), dim3(256), 0, st, mine_sw, coarse, logr, w0, nw);
    if (logr > 8 && brick != nullptr) {   // the slab's four brick layers of the per-brick map (whole words along x)
        const int lb = logr - 2;
        const uint32_t nb = 1u << lb, b0 = (uint32_t)offset / 4u;
        uint3 bw0 = make_uint3(0, 0, 0), bnw = make_uint3(nb / 8u, nb, nb);
        if (axis == 0) { bw0.x = b0 / 8u; bnw.x = (b0 + 3u) / 8u - bw0.x + 1u; }
        else if (axis == 1) { bw0.y = b0; bnw.y = 4u; }
        else { bw0.z = b0; bnw.z = 4u; }
        const uint32_t bn = bnw.x * bnw.y * bnw.z;
        hipLaunchKernelGGL(k_build_brick, dim3((bn + 255u) / 256u), dim3(256), 0, st, mine_sw, brick, lb, bw0, bnw);
    }
    return hipGetLastError();
}

hipError_t launch_mega(const Scene& sc, const Frame& f, const Planes& pl, DevCounters* cn, bool count, hipStream_t st) {
    const uint32_t npix_pad = (uint32_t)f.ntiles_local * 64u;
    if (npix_pad == 0) return hipSuccess;
    dim3 grid((npix_pad + 255u) / 256u), block(256);
    if (count) hipLaunchKernelGGL(k_mega<true>, grid, block, 0, st, sc, f, pl, cn);
    else hipLaunchKernelGGL(k_mega<false>, grid, block, 0, st, sc, f, pl, cn);
    return hipGetLastError();
}

hipError_t launch_trace(const Scene& sc, const Frame& f, const TraceArgs& a, bool primary, bool count, int nworkgroups, hipStream_t st) {
    dim3 grid(nworkgroups), block(1024);
    const bool lrz = f.lr_zero != 0;
#define RT_LAUNCH_TRACE(M, L, C) hipLaunchKernelGGL((k_trace<M, L, C>), grid, block, 0, st, sc, f, a)
    if (primary) {
        if (lrz) { if (count) RT_LAUNCH_TRACE(0, true, true); else RT_LAUNCH_TRACE(0, true, false); }
        else     { if (count) RT_LAUNCH_TRACE(0, false, true); else RT_LAUNCH_TRACE(0, false, false); }
    } else {
        if (lrz) { if (count) RT_LAUNCH_TRACE(1, true, true); else RT_LAUNCH_TRACE(1, true, false); }
        else     { if (count) RT_LAUNCH_TRACE(1, false, true); else RT_LAUNCH_TRACE(1, false, false); }
    }
#undef RT_LAUNCH_TRACE
    return hipGetLastError();
}

hipError_t launch_shade0(const Scene& sc, const Frame& f, const ShadeArgs& a, const Planes& pl, bool count, hipStream_t st) {
    if (a.npaths == 0) return hipSuccess;
    dim3 grid((a.npaths + 255u) / 256u), block(256);
    if (count) hipLaunchKernelGGL(k_shade0<true>, grid, block, 0, st, sc, f, a, pl);
    else hipLaunchKernelGGL(k_shade0<false>, grid, block, 0, st, sc, f, a, pl);
    return hipGetLastError();
}

hipError_t launch_shadeN(const Scene& sc, const Frame& f, const ShadeArgs& a, int level, bool count, hipStream_t st) {
    if (a.npaths == 0) return hipSuccess;
    dim3 grid((a.npaths + 255u) / 256u), block(256);
    if (count) hipLaunchKernelGGL(k_shadeN<true>, grid, block, 0, st, sc, f, a, level);
    else hipLaunchKernelGGL(k_shadeN<false>, grid, block, 0, st, sc, f, a, level);
    return hipGetLastError();
}

hipError_t launch_accumulate(const float* plx, const float* ply, const float* plz, float4* acc, uint32_t npix_pad,
                             uint32_t nsamples, bool first_batch, hipStream_t st) {
    if (npix_pad == 0) return hipSuccess;
    hipLaunchKernelGGL(k_accumulate, dim3((npix_pad + 255u) / 256u), dim3(256), 0, st, plx, ply, plz, acc, npix_pad, nsamples,
                       first_batch ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_resolve(const Frame& f, const float4* acc, const Planes& pl, uint32_t npix_pad, hipStream_t st) {
    if (npix_pad == 0) return hipSuccess;
    hipLaunchKernelGGL(k_resolve, dim3((npix_pad + 255u) / 256u), dim3(256), 0, st, f, acc, pl, npix_pad);
    return hipGetLastError();
}

hipError_t launch_untile_strided(const void* gathered, size_t rank_stride, void* frame, int world, int capacity, int tiles_x,
                                 int tiles_y, int width, int height, int bpp, hipStream_t st) {
    const uint32_t total = (uint32_t)world * (uint32_t)capacity * 64u;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(k_untile, dim3((total + 255u) / 256u), dim3(256), 0, st, (const uint8_t*)gathered, rank_stride, (uint8_t*)frame,
                       world, capacity, tiles_x, tiles_y, width, height, bpp);
    return hipGetLastError();
}

hipError_t launch_untile(const void* gathered, void* frame, int world, int capacity, int tiles_x, int tiles_y, int width,
                         int height, int bpp, hipStream_t st) {
    return launch_untile_strided(gathered, (size_t)capacity * 64u * (size_t)bpp, frame, world, capacity, tiles_x, tiles_y, width,
                                 height, bpp, st);
}

}  // namespace rtd
