// rt_seq.hip — k_seq: the path kernel with SEQUENTIAL ray slots.
//
// k_paths (rt_paths.hip) gives every path two ray slots that step together; the shadow ray of a level is short (4.5 steps on
// the benchmark scene against 7.9 for the diffuse ray), so the shadow slot is idle for half of the level and the branch-free
// step loop — which executes every slot for every lane — spends its VALU time on it all the same (slot-lanes in flight:
// 47 %).  Here a lane carries NC paths with ONE slot each: the slot walks the level's shadow ray and then its diffuse ray from
// the same surface point (the diffuse ray's direction, origin and first texel wait in the context's q*/o* registers), so a
// slot is busy for the whole level.  A context whose shadow ray has ended is re-armed by a short block inside the step loop
// (run when `rmin` contexts of the wave wait — register moves only, no memory); a context whose diffuse ray has ended parks
// for the transition pass as in k_paths.  Everything else — the branch-free step (rt_pslot.hpp), nibble map, tables, cursors,
// values — is k_paths'.
//
// Measured on the headline frame (profiles/README.md): the slots are fuller as intended — with three paths per lane 35.0 M
// slot-steps at 61 % against k_paths' 45.2 M at 47 %, 2.76 M passes against 3.34 M — but every slot now tracks the step axis, the
// re-arm block runs almost every iteration at small `rmin` (or leaves slots waiting at a large one), and the loop's scalar
// bookkeeping doubles: 3.44 G VALU + 1.28 G SALU instructions per launch against 3.58 G + 0.67 G, 5.42 ms against 5.23 ms
// (one look at the contexts per step).  With three steps per look: 5.13 ms with three paths per lane (128 VGPRs: the
// scheduler has no room left), **5.01 ms with two** (106 VGPRs) — level with k_paths' 5.01 ms at the time, not ahead; k_paths has
// since reworked its transition pass around memory latency (4.39 ms, rt_paths.hip), which this kernel has not followed.  k_paths is
// RT_KERNEL_DEFAULT's kernel; this one is selectable (RT_KERNEL_SEQ, RT_SEQ_NC=2|3, default 2) and runs the same parity tests.
//
// Restrictions as k_paths: RT_FLAG_CACHE_PRIMARY, lr = (0,0,0), region 256.
#include <hip/hip_runtime.h>

#include "rt_device.hpp"
#include "rt_kernels.hpp"
#include "rt_pslot.hpp"

#ifndef RT_SEQ_STEPS_PER_CHECK
#define RT_SEQ_STEPS_PER_CHECK 3   // step iterations between two looks at the waiting / parked contexts (4: 5.04 ms, 6: 5.24 ms)
#endif

namespace rtd {
using namespace pslot;

namespace {
constexpr uint32_t PP_SHADOW = 1u << 18;      // PPath::st: the slot walks the level's shadow ray (the diffuse ray waits in q*/o*)
// flags above the voxel index in SeqCtx::ow
constexpr uint32_t OW_BAD = 1u << 24;         // the waiting diffuse ray ends at once (NaN direction or first texel outside the texture)
constexpr uint32_t OW_OUTSIDE = 1u << 25;     // the level's origin lies outside the region: first step with the generic q (p_advance)
constexpr uint32_t OW_FRESHINV = 1u << 26;    // counting builds: first texel outside the texture

struct SeqCtx {
    PSlot r;
    float qx, qy, qz, qlx, qly, qlz;   // the level's diffuse ray: negated direction, 1/|direction| (the table entry of P.ent)
    float ox, oy, oz;                  // the level's surface point (origin of both rays)
    uint32_t ow;                       // its first texel (swizzled index) | OW_* flags
    PPath p;
};
}  // namespace

template <int NC, bool COUNT, int STK>
__global__ __launch_bounds__(1024) void k_seq(Scene sc, Frame f, Planes pl, PersistArgs a) {
    __shared__ uint32_t s_coarse[kCoarseWords];
    __shared__ __attribute__((aligned(2048))) uint32_t s_swz[3 * 512];   // swizzle tables (see p_advance)
    __shared__ float s_albedo[128];            // (packed >> k & 0x7F) / 127.0 (raytrace.comp:156-158), exact quotients
    // albedo stack for depth <= 4: three levels x 21 albedo bits in one 64-bit word per context and thread
    __shared__ uint2 s_stack[STK == 0 ? NC : 1][STK == 0 ? 1024 : 1];
    const uint32_t nwork = *a.wl_count;
    const uint32_t nitems = nwork * a.nsamples;
    if (nitems == 0u) return;
    {
        const uint4* src = reinterpret_cast<const uint4*>(sc.coarse);
        uint4* dst = reinterpret_cast<uint4*>(s_coarse);
        for (uint32_t i = threadIdx.x; i < kCoarseWords / 4; i += 1024u) dst[i] = src[i];
        if (threadIdx.x < 128u) s_albedo[threadIdx.x] = (float)threadIdx.x / 127.0f;
        for (uint32_t i = threadIdx.x; i < 3u * 512u; i += 1024u) {   // entry 256 = the wrap to texel 0; 257.. are never used
            const uint32_t ax = i >> 9, v = i & 255u;
            s_swz[i] = ((v & 3u) << (2u * ax)) | ((v >> 2) << (6u + 6u * ax));
        }
    }
    __syncthreads();

    constexpr float half = 128.0f;
    constexpr int R = 256;
    const uint8_t* s_nib = reinterpret_cast<const uint8_t*>(s_coarse);
    const uint32_t swz = (uint32_t)(uintptr_t)(lds_u32*)s_swz;   // LDS byte address of the tables
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gtid = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t threshold = a.threshold, rmin = a.rmin;
    const vec3 sunlight = ld3(f.sunlight);
    const uint32_t D = (uint32_t)f.depth;
    const uint32_t stack_levels = D > 1u ? D - 1u : 1u;
    const auto mine_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sc.mine), (short)0, R * R * R, 0x00020000);

    SeqCtx C[NC];
#pragma unroll
    for (int c = 0; c < NC; c++) {
        PSlot& r = C[c].r;
        r.px = r.py = r.pz = r.ndx = r.ndy = r.lx = r.ly = r.lz = 0.0f; r.ndz = -1.0f;
        r.sx = r.sy = r.sz = 0u; r.nk = K_DEAD | K_END; r.axis = 0u;
        C[c].qx = C[c].qy = 0.0f; C[c].qz = -1.0f; C[c].qlx = C[c].qly = C[c].qlz = 0.0f;
        C[c].ox = C[c].oy = C[c].oz = 0.0f; C[c].ow = 0u;
        C[c].p.st = PP_FINAL; C[c].p.item = 0u; C[c].p.ent = 7u << 16;
    }

    bool exhausted = false;
    const uint32_t kChunk = a.chunk ? a.chunk : 128u;   // paths per cursor atomic (see k_persist)
    uint32_t chunk_next = 0, chunk_end = 0;   // wave-uniform: the wave's current chunk of an XCD group's share of the paths
    uint32_t chunk_sb = 0, chunk_w = 0;       // (sample-in-batch, slot within the share) of path chunk_next
    uint32_t chunk_w0 = 0, chunk_nw = 1;      // the share's slot range
    const uint32_t home_grp = blockIdx.x & 7u;   // workgroups b and b + 8 share an XCD (round-robin dispatch; speed only)
    uint32_t grp_tries = 0;

    unsigned long long c_shadow = 0, c_dif = 0, c_iter = 0, c_hits = 0, c_sky = 0, c_limit = 0, c_border = 0, c_noise = 0;
    unsigned long long d_iters = 0, d_pass = 0, d_pl = 0, d_live = 0, d_rearm = 0, d_rl = 0;   // wave-uniform structure statistics

    auto lookup = [&](const PSlot& r) -> uint32_t {
        const uint32_t vox = ps_vox(r);
        uint32_t st = (s_nib[vox >> 7] >> ((vox >> 4) & 4u)) & 15u;
        if (st == kNibMixed) st = sc.mine[vox];
        return st;
    };
    // how an ENDED ray that did not reach the sky stopped: 0 = hit, 1 = loop limit, 2 = special (see k_paths)
    auto stop_kind = [&](const PSlot& r) -> uint32_t {
        const uint32_t left = r.nk & 0xFFFFu;
        uint32_t kind = ((r.nk & K_DEAD) != 0u || left == (uint32_t)RT_TRACE_LIMIT) ? 2u : 0u;
        const bool at_limit = kind == 0u && left == 0u;
        if (__builtin_expect(__ballot(at_limit) != 0ull, 0)) {
            if (at_limit && lookup(r) != 0u) kind = 1u;
        }
        return kind;
    };
    auto tally = [&](const PSlot& r) {   // exact counters of one finished ray
        const uint32_t left = r.nk & 0xFFFFu;
        if (r.nk & K_AIR) {
            c_iter += (uint32_t)RT_TRACE_LIMIT - left;
            c_sky++;
            int tx, ty, tz;   // the fetch the shader makes before its sky test may hit the border
            if (!wrap_texel(v3(r.px, r.py, r.pz), (float)R, &tx, &ty, &tz)) c_border++;
        } else {
            const uint32_t kind = stop_kind(r);
            c_iter += kind == 2u ? 1u : (uint32_t)RT_TRACE_LIMIT - left;
            if (kind == 1u) c_limit++; else c_hits++;
            if (kind == 2u) c_border += 1u + ((r.nk & kFreshInvalid) ? 1u : 0u);
        }
    };
    // the first step of a ray whose origin lies outside the region (rare): q for u of either sign
    auto first_step_outside = [&](PSlot& r, bool outside) {
        if (__builtin_expect(__ballot(outside) != 0ull, 0)) {
            uint32_t st = 0;
            if (outside) st = lookup(r);
            p_advance<true, 2>(r, st, outside, swz);
        }
    };

    auto stack_at = [&](uint32_t c, uint32_t j) -> uint32_t {
        if constexpr (STK == 0) {
            const uint2 w = s_stack[c][threadIdx.x];
            const uint64_t v = (uint64_t)w.y << 32 | w.x;
            return (uint32_t)(v >> (21u * j)) & 0x1FFFFFu;
        } else {
            return a.stack[((size_t)c * stack_levels + j) * a.nthreads + gtid];
        }
    };
    auto stack_put = [&](uint32_t c, uint32_t j, uint32_t m) {
        if constexpr (STK == 0) {
            const uint2 w = s_stack[c][threadIdx.x];
            uint64_t v = (uint64_t)w.y << 32 | w.x;
            const uint64_t m21 = m & 0x1FFFFFu;      // albedo bits of the packed material (raytrace.comp:156-158 read nothing else)
            v = j == 0u ? m21 : (j == 1u ? ((v & 0x1FFFFFull) | m21 << 21) : ((v & 0x3FFFFFFFFFFull) | m21 << 42));
            s_stack[c][threadIdx.x] = make_uint2((uint32_t)v, (uint32_t)(v >> 32));
        } else {
            a.stack[((size_t)c * stack_levels + j) * a.nthreads + gtid] = m;
        }
    };

    // ---- the shadow ray of a context ended: note its result (:326-328 / :338-340), start the level's diffuse ray (:330 / :342)
    auto rearm = [&](SeqCtx& x) {
        PSlot& r = x.r;
        const bool me = (x.p.st & PP_SHADOW) != 0u && r.nk >= K_END;
        if (COUNT) { d_rearm++; d_rl += (uint32_t)__popcll(__ballot(me)); }
        if (me) {
            if (COUNT) tally(r);
            const uint32_t level = x.p.st >> 20;
            uint32_t st = x.p.st & ~PP_SHADOW;
            if (r.nk & K_AIR) st |= 1u << (level - 1u);
            x.p.st = st;
            r.px = x.ox; r.py = x.oy; r.pz = x.oz;
            r.ndx = x.qx; r.ndy = x.qy; r.ndz = x.qz; r.lx = x.qlx; r.ly = x.qly; r.lz = x.qlz;
            r.sx = x.ow & 0xFFFFFFu; r.sy = 0u; r.sz = 0u;
            r.axis = 0u;              // code of "z": a ray that ends before its first step reports the z face (:90)
            r.nk = (x.ow & OW_BAD) ? (K_DEAD | K_END | ((COUNT && (x.ow & OW_FRESHINV)) ? kFreshInvalid : 0u)) : (uint32_t)RT_TRACE_LIMIT;
        }
        first_step_outside(r, me && (x.ow & (OW_OUTSIDE | OW_BAD)) == OW_OUTSIDE);
    };

    // ---- both rays of a level (:324-330 / :336-342) from the surface point (sfx, sfy, sfz) with face id snormal: the slot starts
    // on the shadow ray (its direction comes from the per-frame table: the slot's registers held the diffuse ray), the diffuse
    // ray's table entry, origin and first texel wait in the context
    auto begin_level = [&](SeqCtx& x, float sfx, float sfy, float sfz, uint32_t snormal) {
        if (COUNT) { c_noise++; c_shadow++; c_dif++; }
        PSlot& r = x.r;
        int ix, iy, iz;
        const bool ok = wrap_texel(v3(sfx, sfy, sfz), (float)R, &ix, &iy, &iz);
        const uint32_t tx = s_swz[ix], ty = s_swz[512 + iy], tz = s_swz[1024 + iz];
        const uint32_t se = x.p.ent & 0xFFFFu;
        const float4 sd = a.sun_lut[2u * se], sl = a.sun_lut[2u * se + 1u];
        if (snormal != x.p.ent >> 16) {   // q* still hold the entry of the path's previous level when the face repeats
            const uint32_t di = 4u * ((snormal << 16) | se);
            const float4 d2 = a.dif_lut[di + 1u], dl = a.dif_lut[di + 2u];
            x.qx = -d2.x; x.qy = -d2.y; x.qz = -d2.z; x.qlx = dl.x; x.qly = dl.y; x.qlz = dl.z;
            x.p.ent = se | snormal << 16;
        }
        r.ndx = -sd.x; r.ndy = -sd.y; r.ndz = -sd.z; r.lx = sl.x; r.ly = sl.y; r.lz = sl.z;
        x.ox = sfx; x.oy = sfy; x.oz = sfz;
        const bool bad_q = x.qx != x.qx || x.qy != x.qy || x.qz != x.qz || !ok;
        const bool bad_s = r.ndx != r.ndx || r.ndy != r.ndy || r.ndz != r.ndz || !ok;
        const bool outside = ok && (sfx + half < 0.0f || sfy + half < 0.0f || sfz + half < 0.0f);
        x.ow = (tx | ty | tz) | (bad_q ? OW_BAD : 0u) | (outside ? OW_OUTSIDE : 0u) | ((COUNT && !ok) ? OW_FRESHINV : 0u);
        r.px = sfx; r.py = sfy; r.pz = sfz;
        r.sx = tx; r.sy = ty; r.sz = tz;
        r.axis = 0u;
        r.nk = bad_s ? (K_DEAD | K_END | ((COUNT && !ok) ? kFreshInvalid : 0u)) : (uint32_t)RT_TRACE_LIMIT;
        x.p.st |= PP_SHADOW;
        first_step_outside(r, outside && !bad_s);
    };

    // =========================== transition pass of one context ============================================
    auto pass = [&](SeqCtx& x, const uint32_t c) {
        PSlot& F = x.r;
        PPath& P = x.p;
        const bool ended = F.nk >= K_END && (P.st & PP_SHADOW) == 0u;
        const uint32_t level = P.st >> 20;
        const bool mine = ended && level != 0u;
        if (COUNT) { d_pass++; d_pl += (uint32_t)__popcll(__ballot(mine)); }
        bool start = false;
        float sfx = 0, sfy = 0, sfz = 0;      // surface the next level stands on
        uint32_t snormal = 0;
        if (mine) {
            if (COUNT) tally(F);
            const uint32_t sunbits = P.st & 0xFFFFu;     // this level's shadow result is in already (rearm)
            const bool air = (F.nk & K_AIR) != 0u;
            if (((F.nk | P.st) & K_AIR) != 0u) {   // sky exit or last level: the path ends
                vec3 sky = v3(0, 0, 0);
                if (air) {   // :331-332 / :343-345, tabulated per frame
                    const float4 t = a.dif_lut[4u * P.ent + 3u];   // P.ent = (face << 16 | noise bytes) = the entry F walked
                    sky = v3(t.x, t.y, t.z);
                }
                // L_j = [sun_j] S + L_{j+1} * albedo_{j+1} + emission, innermost first (raytrace.comp:346-348)
                vec3 L = v3(0.0f, 0.0f, 0.0f);
                if (sunbits >> (level - 1u) & 1u) L = vadd(L, sunlight);
                if (air) L = vadd(L, sky);
                for (uint32_t j = level - 1u; j >= 1u; j--) {
                    const uint32_t pm = stack_at(c, j - 1u);
                    vec3 light2 = vmul(L, v3(s_albedo[pm >> 14 & 0x7Fu], s_albedo[pm >> 7 & 0x7Fu], s_albedo[pm & 0x7Fu]));
                    light2 = vadd(light2, v3(0.0f, 0.0f, 0.0f));      // + dif.emission, always vec3(0) (:155)
                    vec3 acc = v3(0.0f, 0.0f, 0.0f);
                    if (sunbits >> (j - 1u) & 1u) acc = vadd(acc, sunlight);
                    L = vadd(acc, light2);
                }
                const vec3 light = vadd(v3(0.0f, 0.0f, 0.0f), L);
                a.pl[P.item] = PathLight{light.x, light.y, light.z};   // k_accumulate_paths adds a pixel's samples in order
                P.st = PP_FINAL;
            } else {
                // Diffuse result: material of the hit texel (:150-154), position with the 0.001 face offset (:166-180)
                const uint32_t kind = stop_kind(F);
                const uint32_t axis = ps_axis_of_code(F.axis);
                const uint32_t nrm = axis == 0u ? (F.ndx < 0.0f ? 1u : 0u) : (axis == 1u ? (F.ndy < 0.0f ? 3u : 2u) : (F.ndz < 0.0f ? 5u : 4u));
                uint32_t material = 0;
                if (kind == 0u) material = sc.mat[ps_vox(F)];
                float hx = F.px, hy = F.py, hz = F.pz;
                if (kind == 2u) { hx = hy = hz = __builtin_nanf(""); }
                const float offv = (nrm & 1u) ? -0.001f : 0.001f;
                hx = axis == 0u ? hx + offv : hx; hy = axis == 1u ? hy + offv : hy; hz = axis == 2u ? hz + offv : hz;
                stack_put(c, level - 1u, material);   // albedo of surface level+1
                P.st = sunbits | (level + 1u) << 20 | (level + 1u == D ? PP_FINAL : 0u);
                sfx = hx; sfy = hy; sfz = hz; snormal = nrm;
                start = true;
            }
        }
        // contexts without a path pull the next ones (chunked per-XCD cursors, see k_persist)
        if (!exhausted) {
            const bool wantme = ended && (P.st >> 20) == 0u;
            const uint64_t want = __ballot(wantme);
            const uint32_t nwant = (uint32_t)__popcll(want);
            if (nwant) {
                if (chunk_next >= chunk_end) {
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
                if (wantme) {
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                    if (rank < take) {
                        uint32_t sb = chunk_sb, w = chunk_w + rank;
                        while (w >= chunk_nw) { w -= chunk_nw; sb++; }
                        w += chunk_w0;   // worklist slot
                        const float4 ph = a.phit[w];
                        const uint32_t info = __float_as_uint(ph.w);
                        sfx = ph.x; sfy = ph.y; sfz = ph.z;
                        snormal = info >> 28;
                        const uint32_t wgx8 = info & 0x3FFFu, wgy8 = (info >> 14) & 0x3FFFu;
                        P.item = sb * nwork + w;
                        // noise_offset of this path (:298-304) and its noise_value texel (:324, :336; one lookup serves every level, Q5)
                        const uint32_t seed = (f.seed + a.sample0 + sb) % (uint32_t)RT_NOISE_BYTES;
                        const uint32_t by = seed / RT_NOISE_SIZE;
                        const uint32_t nb = sc.noise[(by > 511u ? 511u : by) * RT_NOISE_SIZE + seed % RT_NOISE_SIZE];
                        const uint32_t tx = ((nb & 0xFFu) + wgx8) & 511u, ty = (((nb >> 8) & 0xFFu) + wgy8) & 511u;
                        P.ent = (sc.noise[ty * RT_NOISE_SIZE + tx] & 0xFFFFu) | 7u << 16;
                        P.st = 1u << 20 | (D == 1u ? PP_FINAL : 0u);
                        start = true;
                    }
                }
                chunk_w += take;
                while (chunk_w >= chunk_nw) { chunk_w -= chunk_nw; chunk_sb++; }
            }
        }
        if (start) begin_level(x, sfx, sfy, sfz, snormal);
    };

    uint64_t idle[NC], sh[NC];   // lanes whose context is empty for good; lanes whose context is on its shadow ray
#pragma unroll
    for (int c = 0; c < NC; c++) { idle[c] = 0ull; sh[c] = 0ull; }
    for (;;) {
        uint64_t park[NC];
        bool leave = false;
        for (;;) {
            // a context whose shadow ray ended waits for the re-arm block; one whose diffuse ray ended (or that has no path)
            // parks for its transition pass
            uint64_t need[NC];
            uint64_t live = 0ull;
            uint32_t nre = 0, best = 0, nlive = 0;
#pragma unroll
            for (int c = 0; c < NC; c++) {
                const uint64_t e = __ballot(C[c].r.nk >= K_END);
                if (COUNT) nlive += (uint32_t)__popcll(~e);
                need[c] = e & sh[c];
                park[c] = e & ~sh[c] & ~idle[c];
                live |= ~e;
                nre += (uint32_t)__popcll(need[c]);
                best = max(best, (uint32_t)__popcll(park[c]));
            }
            if (best >= threshold) break;
            if (nre >= rmin || (live == 0ull && nre != 0u)) {
#pragma unroll
                for (int c = 0; c < NC; c++)
                    if (need[c]) { rearm(C[c]); sh[c] &= ~need[c]; }
                continue;
            }
            if (live == 0ull) { leave = best == 0u; break; }
            if (COUNT) { d_iters++; d_live += nlive; }
            // ---- one step of all slots: nibble reads, then byte loads, then the arithmetic ----
#pragma unroll
            for (int rep = 0; rep < RT_SEQ_STEPS_PER_CHECK; rep++) {
            uint32_t v[NC], w[NC], t[NC], b[NC];
            bool g[NC];
#pragma unroll
            for (int c = 0; c < NC; c++) { v[c] = ps_vox(C[c].r); w[c] = s_nib[v[c] >> 7]; }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                t[c] = __builtin_amdgcn_ubfe(w[c], (v[c] >> 4) & 4u, 4u);
                g[c] = ps_running(C[c].r.nk) && t[c] == kNibMixed;
                b[c] = __builtin_amdgcn_raw_buffer_load_b8(mine_rsrc, g[c] ? v[c] : 0xFFFFFFFFu, 0, 0);
            }
#pragma unroll
            for (int c = 0; c < NC; c++) {
                t[c] = g[c] ? b[c] : t[c];
                p_advance<false, 2>(C[c].r, t[c], true, swz);
            }
            }
        }
        if (leave) break;   // nothing in flight, nothing waiting, nothing parked, no paths left
        // serve the context kind with the most parked lanes
        uint32_t pick = 0, bestn = (uint32_t)__popcll(park[0]);
#pragma unroll
        for (int c = 1; c < NC; c++) { const uint32_t n = (uint32_t)__popcll(park[c]); if (n > bestn) { bestn = n; pick = (uint32_t)c; } }
#pragma unroll
        for (int c = 0; c < NC; c++)
            if (pick == (uint32_t)c) { pass(C[c], (uint32_t)c); sh[c] = __ballot((C[c].p.st & PP_SHADOW) != 0u); }
        if (exhausted) {
#pragma unroll
            for (int c = 0; c < NC; c++) idle[c] = __ballot((C[c].p.st >> 20) == 0u);
        }
    }
    if (COUNT) {
        DevCounters* cn = a.counters;
        const unsigned long long rays = c_shadow + c_dif;
        wave_add(&cn->rays, rays); wave_add(&cn->rays_shadow, c_shadow);
        wave_add(&cn->rays_diffuse, c_dif); wave_add(&cn->iterations, c_iter); wave_add(&cn->minefield_fetches, rays + c_iter);
        wave_add(&cn->hits, c_hits); wave_add(&cn->material_fetches, c_hits); wave_add(&cn->sky_exits, c_sky);
        wave_add(&cn->limit_exits, c_limit); wave_add(&cn->border_fetches, c_border); wave_add(&cn->noise_fetches, c_noise);
        if (lane == 0) {
            atomicAdd(&cn->dbg_loop_iters, d_iters); atomicAdd(&cn->dbg_f_lanes, d_live);
            atomicAdd(&cn->dbg_passes, d_pass); atomicAdd(&cn->dbg_pass_lanes, d_pl);
            atomicAdd(&cn->dbg_s_execs, d_rearm); atomicAdd(&cn->dbg_s_lanes, d_rl);
        }
    }
}

hipError_t launch_seq(const Scene& sc, const Frame& f, const Planes& pl, const PersistArgs& a, bool count, int nc, int nworkgroups,
                      hipStream_t st) {
    if (f.logr != 8 || f.lr_zero == 0 || (nc != 2 && nc != 3)) return hipErrorInvalidValue;
    const dim3 grid(nworkgroups), block(1024);
    const bool lds_stack = f.depth <= 4;
#define RT_LAUNCH_SEQ(N, C, S) hipLaunchKernelGGL((k_seq<N, C, S>), grid, block, 0, st, sc, f, pl, a)
#define RT_LAUNCH_SEQ_N(N) do { if (lds_stack) { if (count) RT_LAUNCH_SEQ(N, true, 0); else RT_LAUNCH_SEQ(N, false, 0); } \
                                else { if (count) RT_LAUNCH_SEQ(N, true, 1); else RT_LAUNCH_SEQ(N, false, 1); } } while (0)
    if (nc == 2) RT_LAUNCH_SEQ_N(2); else RT_LAUNCH_SEQ_N(3);
#undef RT_LAUNCH_SEQ_N
#undef RT_LAUNCH_SEQ
    return hipGetLastError();
}

}  // namespace rtd
