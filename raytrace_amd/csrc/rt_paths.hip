// rt_paths.hip — k_paths, the path kernel RT_KERNEL_DEFAULT runs for cached-primary frames (launches of 3 M pixel-samples and more).
//
// Same work, same values as k_persist (rt_persist.hip); what differs is how much of it a wave keeps in flight and how the
// instructions are spent (measured on gfx950, DESIGN.md 5: VALU issue — tools/ubench/valu_rate.hip —, the vector-memory path's
// address and tag work, and the latency chains of the transition passes share the bound):
//   * a lane carries TWO paths (contexts A and B), each with the level's shadow ray and diffuse ray in their own ray slots:
//     four independent fetch chains per lane instead of two.  k_persist ran at an LDS-pinned four waves per SIMD with 75 of
//     its 128 VGPRs and its waves parked at s_waitcnt half of their life (round-1 counters); the idle registers now hold
//     the second path.
//   * the step loop is ONE basic block: every slot's step is predicated with selects instead of exec-mask branches and the
//     byte behind a "mixed" nibble-map entry is a buffer load whose offset is out of range for lanes that do not need it
//     (no memory access, returns 0).  The compiler issues the four nibble reads, then the four byte loads, and overlaps each
//     slot's arithmetic with the others' latency; the ~65 scalar/branch instructions k_persist spent per slot and step
//     are gone.
//   * the swizzle-table words of a slot's next texel are consumed at the top of the NEXT iteration (the slot keeps the
//     three words, not their OR), so that LDS latency is off the critical path too.
//   * (round 3) what the loop needs to know about a ray besides its numbers — in flight?  last step along z?  tx < ty? — are wave
//     masks in SGPR pairs merged with scalar instructions (rt_pslot.hpp, p_step), not flag bits that cost every slot and step a
//     subtract, two compares, an OR and three selects; the position update and the iteration count run under the movers' EXEC
//     mask; the counter only counts — the loop limit is tested in a separate small loop the wave enters when a ray in flight
//     comes within reach of it (see near_limit); whether an ended ray reached the sky, hit, met the limit or never moved is read
//     off its position and counter when the transition pass consumes it.  Step group 535 -> 468 VALU, launch 4.31 -> 4.06 ms.
//   * the shadow slots sit out one repetition in three of the step group (a level's shadow ray is the shorter one) — see
//     RT_PATHS_SHADOW_REPS.
// Transitions work as in k_persist (parked lanes, __ballot threshold, per-XCD chunked cursors, direction tables), but a wave in
// its pass steps none of its rays, so the pass is built around its memory round trips rather than its instructions: one pass
// serves every lane that has a parked context (A or B), and its loads go out in two batches whatever the lanes' transitions
// are (see `pass`).  What was measured on the way, same box each time (tools/abn.sh): per-context passes with the loads inside
// their divergent branches 4.92 ms per launch, loads batched 4.64, one pass for both contexts 4.43, three steps per look 4.39.
// Built and measured without gain: passes specialised by what the parked level needs (6.08 against 5.3 ms); reserving the next
// chunk of paths one pass ahead (4.83 against 4.64: the atomic's return sits in front of the pass's loads); the step group
// software-pipelined by context, one context's arithmetic under the other's byte loads (4.98 against 4.65 — the loop wants its
// four slots' accesses batched); nibble map and byte array biased by one so that "mixed or not" needs
// no compare (2.5 % fewer loop instructions, 1 % slower: ended slots then keep fetching their byte); positions scaled by four
// to share the table-index multiply (6 % fewer instructions, 1.7 % slower: more packed-math issue slots).
// Round 3, same way (profiles/r3_kpaths_variants_*.txt): selects forced into SGPR-pair form, selects instead of the EXEC-masked
// update, byte loads skipped by a branch when no lane needs one (+5 %), 1 / |direction| computed instead of loaded (+3 %), one byte
// load per pair of rays (+9 %), a wave's new paths from one pixel (+5.5 %) — all lost; per-lane `bool`s carried round the loop
// (the compiler makes 0/1 registers of them: 550 VALU per group) lost to explicit 64-bit masks.
//
// Restrictions (rt_api.hip dispatches everything else to k_persist): RT_FLAG_CACHE_PRIMARY.  Scrolled regions (lr != 0, terrain
// streaming) run the LRZ = false instantiations: generic q, lr in the sky test, the shader's own mod for the texel and its border
// case (rt_pslot.hpp) — 17 more VALU per slot and step.  Regions 512 and
// 1024 (LOGR 9, 10) use larger swizzle tables (2 R entries per axis) and derive the nibble-map entry from the brick coordinates;
// their tables leave no LDS for the albedo stack, which then lives in global memory (STK = 1).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "rt_device.hpp"
#include "rt_kernels.hpp"
#include "rt_pslot.hpp"

// Step iterations between two looks at the parked contexts (the look costs two v_min + two v_cmp + a dozen scalar instructions),
// and — bit r — whether the shadow slots step in repetition r of the group.  A level's shadow ray is short (4.5 steps on the
// benchmark scene against the diffuse ray's 7.9) and its slot then idles until the diffuse ray ends; stepping it in two of three
// repetitions removes a sixth of the loop's instructions and rarely lengthens a level.  Same-box timings (tools/abn.sh) with
// the merged pass: 3 steps / 0x3 4.41 ms per launch, 3 / 0x7 4.53, 4 / 0x7 4.47, 4 / 0xF 4.56, 5 / 0x15 4.46, 5 / 0x0F 4.54,
// 6 / 0x1B 4.88.  Round 3 (wave masks, cheaper looks), same box: 3 / 0x3 4.06-4.08 ms, 4 / 0x7 4.09-4.14, 4 / 0xB 4.08-4.11, 5 / 0x17 4.15-4.18,
// 5 / 0xF 4.16-4.19, 2 / 0x1 4.30, 3 / 0x7 4.25.
#ifndef RT_PATHS_STEPS_PER_CHECK
#define RT_PATHS_STEPS_PER_CHECK 3
#endif
#ifndef RT_PATHS_DRAIN_PARK
#define RT_PATHS_DRAIN_PARK 8   // parked lanes that run the pass of a wave whose paths have run out (1, 2, 4, 8, 16 measured: r3_drain_park_threshold.txt)
#endif
#ifndef RT_PATHS_SHADOW_REPS
#define RT_PATHS_SHADOW_REPS 0x3
#endif
#ifndef RT_PATHS_BRICK_MAP
#define RT_PATHS_BRICK_MAP 0    // regions above 256: 1 = consult a global per-brick nibble map (Scene::brick; RT_BRICK_MAP=1 makes the host build it) before
                                // the byte array.  Measured and NOT kept: it answers 11 % of the loop's fetches at R = 1024 and costs every step on a
                                // "mixed" cube one more dependent round trip: C5 128 -> 151 ms per frame (profiles/r4_c5_brick_map.txt)
#endif

namespace rtd {
using namespace pslot;

// STK: where the albedo stack of the two paths lives — 0 = LDS (depth <= 4: three levels per path: the 24 KiB the nibble map and
// the tables leave hold exactly 2 x 3 rows); 1 = global memory; 2 (depth 5..8 at region 256) = levels 1-3 in those LDS rows, levels
// 4-7 in global memory and fetched with the pass's first batch of loads, so that the sum over a path's levels makes no round
// trips of its own (the 4K spp-256 depth-8 frame: 6.75 ms per launch against 7.19 with the whole stack in global memory; LDS rows
// alone: 7.14; the whole seven-level column fetched ahead: 8.40 — its registers spill).
template <bool COUNT, int STK, int LOGR, bool LRZ>
__global__ __launch_bounds__(1024) void k_paths(Scene sc, Frame f, Planes pl, PersistArgs a) {
    static_assert(LOGR == 8 || STK == 1, "the larger regions' tables leave no LDS for the albedo stack");
    constexpr bool LDS_ROWS = STK == 0 || STK == 2;   // STK 2 (depth 5..8 at region 256): levels 1-3 in LDS, the rest in global memory
    constexpr int R = 1 << LOGR, LB = LOGR - 2;
    constexpr uint32_t kTabWords = swz_bytes<LOGR>() / 4u;   // 2 R entries per axis
    __shared__ uint32_t s_coarse[kCoarseWords];
    __shared__ __attribute__((aligned(swz_bytes<LOGR>()))) uint32_t s_swz[3 * kTabWords];   // swizzle tables (see p_advance)
    __shared__ float s_albedo[128];            // (packed >> k & 0x7F) / 127.0 (raytrace.comp:156-158), exact quotients
    __shared__ uint32_t s_stack[LDS_ROWS ? 2 : 1][LDS_ROWS ? 3 : 1][LDS_ROWS ? 1024 : 1];
    const uint32_t nwork = *a.wl_count;
    const uint32_t nitems = nwork * a.nsamples;
    if (nitems == 0u) return;
    {
        const uint4* src = reinterpret_cast<const uint4*>(sc.coarse);
        uint4* dst = reinterpret_cast<uint4*>(s_coarse);
        for (uint32_t i = threadIdx.x; i < kCoarseWords / 4; i += 1024u) dst[i] = src[i];
        if (threadIdx.x < 128u) s_albedo[threadIdx.x] = (float)threadIdx.x / 127.0f;
        for (uint32_t i = threadIdx.x; i < 3u * kTabWords; i += 1024u) {   // entry R = the wrap to texel 0; R+1.. are never used
            const uint32_t ax = i / kTabWords, e = i % kTabWords, v = e & (uint32_t)(R - 1);
            const uint32_t word = ((v & 3u) << (2u * ax)) | ((v >> 2) << (6u + (uint32_t)LB * ax));
            s_swz[i] = (!LRZ && e == (uint32_t)R) ? kSwzBorder : word;   // scrolled region: entry R = the border texel (p_advance)
        }
    }
    __syncthreads();

    constexpr float half = (float)(R / 2);
    const uint8_t* s_nib = reinterpret_cast<const uint8_t*>(s_coarse);
    const uint32_t swz = (uint32_t)(uintptr_t)(lds_u32*)s_swz;   // LDS byte address of the tables
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t gtid = blockIdx.x * 1024u + threadIdx.x;
    const uint32_t threshold = a.threshold;   // parked lanes of one context kind that trigger its transition pass
    const vec3 sunlight = ld3(f.sunlight);
    const uint32_t D = (uint32_t)f.depth;
    const uint32_t stack_levels = D > 1u ? D - 1u : 1u;
    // minefield bytes as a buffer: a lane that needs no byte passes an out-of-range offset (no access, returns 0)
    const auto mine_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sc.mine), (short)0, 1 << (3 * LOGR), 0x00020000);
    // regions above 256: the per-brick nibble map between the LDS map (one entry per (R/64)^3 cube) and the bytes — same trick, a
    // lane whose cube is not "mixed" passes an offset beyond the map
    const auto brick_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(sc.brick), (short)0, 1 << (3 * LOGR - 7), 0x00020000);
    unsigned long long d_fetch = 0, d_fetch_cube = 0, d_fetch_brick = 0;   // counting builds: fetches of rays in flight / not answered by the LDS map / nor by the brick map

#ifdef RT_DIAG_WAVE_TIMES
    const unsigned long long t_wave0 = wall_clock64();
    unsigned long long t_exh = 0;
#endif
    // the four rays of the lane (shadow / diffuse of context A and B) and their per-lane booleans (rt_pslot.hpp: wave masks in SGPRs)
    PRay SA, FA, SB, FB;
    SA.px = SA.py = SA.pz = SA.ndx = SA.ndy = SA.lx = SA.ly = SA.lz = 0.0f; SA.ndz = -1.0f;
    SA.sx = SA.sy = SA.sz = 0u; SA.nk = (uint32_t)RT_TRACE_LIMIT;
    FA = SA; SB = SA; FB = SA;
    lanemask rSA = 0, rFA = 0, rSB = 0, rFB = 0;                 // in flight
    lanemask zFA = ~0ull, xyFA = 0, zFB = ~0ull, xyFB = 0;       // the diffuse ray's last step: along z / tx < ty (z: a ray that ends before its first step reports the z face, :90)
    lanemask nul = 0, nul2 = 0;                                  // the shadow rays' unused axis masks
    uint32_t looks = 0;
    PPath PA, PB;
    PA.st = PP_FINAL; PA.item = 0u; PA.ent = 7u << 16;
    PB = PA;

    bool exhausted = false;
    const uint32_t kChunk = a.chunk ? a.chunk : 128u;   // paths per cursor atomic (see k_persist)
    uint32_t chunk_next = 0, chunk_end = 0;   // wave-uniform: the wave's current chunk of an XCD group's share of the paths
    uint32_t chunk_sb = 0, chunk_w = 0;       // (sample-in-batch, slot within the share) of path chunk_next
    uint32_t chunk_w0 = 0, chunk_nw = 1;      // the share's slot range
    const uint32_t home_grp = blockIdx.x & 7u;   // workgroups b and b + 8 share an XCD (round-robin dispatch; speed only)
    uint32_t grp_tries = 0;

    unsigned long long c_shadow = 0, c_dif = 0, c_iter = 0, c_hits = 0, c_sky = 0, c_limit = 0, c_border = 0, c_noise = 0;
    unsigned long long d_iters = 0, d_passf = 0, d_plf = 0, d_live = 0;   // wave-uniform structure statistics (counting builds)

    // minefield value of a texel (nibble map, byte array behind it), outside the step loop
    auto lookup = [&](uint32_t vox) -> uint32_t {
        if (!LRZ && ps_border(vox)) return 0u;
        uint32_t nb, nsh;
        ps_nibble_of<LOGR, LRZ>(vox, &nb, &nsh);
        uint32_t st = (s_nib[nb] >> nsh) & 15u;
        if (LOGR > 8 && RT_PATHS_BRICK_MAP != 0 && st == kNibMixed) st = (sc.brick[vox >> 7] >> ((vox >> 4) & 4u)) & 15u;
        if (st == kNibMixed) st = sc.mine[vox];
        return st;
    };
    // How an ended ray that did not reach the sky stopped: 0 = hit (:146-160), 1 = loop limit (Q8: a non-air hit with
    // material 0), 2 = special (Q12: it never moved — fresh ray on a 0, NaN direction, first texel outside the texture: NaN
    // position, material 0).  The limit case needs the value of the ray's texel once more — practically never taken.
    auto stop_kind = [&](uint32_t nk, uint32_t vox, bool consider) -> uint32_t {
        const uint32_t left = nk & 0xFFFFu;
        uint32_t kind = left == (uint32_t)RT_TRACE_LIMIT ? 2u : 0u;
        const bool at_limit = consider && left == 0u;
        if (__builtin_expect(__ballot(at_limit) != 0ull, 0)) {
            if (at_limit && lookup(vox) != 0u) kind = 1u;
        }
        return kind;
    };
    auto tally = [&](const PRay& r) {   // exact counters of one finished ray (counting builds)
        const uint32_t left = r.nk & 0xFFFFu;
        if (pr_outside<LOGR, LRZ>(r.px, r.py, r.pz, f.lr[0], f.lr[1], f.lr[2])) {
            c_iter += (uint32_t)RT_TRACE_LIMIT - left;
            c_sky++;
            int tx, ty, tz;   // the fetch the shader makes before its sky test may hit the border
            if (!wrap_texel(v3(r.px, r.py, r.pz), (float)R, &tx, &ty, &tz)) c_border++;
        } else {
            const uint32_t kind = stop_kind(r.nk, pr_vox(r), true);
            c_iter += kind == 2u ? 1u : (uint32_t)RT_TRACE_LIMIT - left;   // a ray that is special ends inside its first iteration
            if (kind == 1u) c_limit++; else c_hits++;
            if (kind == 2u) c_border += 1u + ((r.nk & kFreshInvalid) ? 1u : 0u);
            else if (!LRZ && ps_border(pr_vox(r))) c_border++;   // the fetch that ended it went to the border texel
        }
    };
    // the first step of a fresh ray whose origin lies outside the region (rare; the generic q of p_advance, flag-word bookkeeping)
    auto first_step = [&](PSlot& r, bool en) {
        en = en && ps_running(r.nk);
        uint32_t st = 0;
        if (en) st = lookup(ps_vox(r));
        p_advance<true, 1, LOGR, LRZ>(r, st, en, swz, f.lr[0], f.lr[1], f.lr[2]);
    };

    // albedo stack of context c: packed material of surface j+2 at slot j
    auto stack_at = [&](uint32_t c, uint32_t j) -> uint32_t {
        if constexpr (STK == 0) {
            return s_stack[c][j][threadIdx.x];
        } else if constexpr (STK == 2) {
            if (j < 3u) return s_stack[c][j][threadIdx.x];
            return a.stack[((size_t)c * stack_levels + j) * a.nthreads + gtid];
        } else {
            return a.stack[((size_t)c * stack_levels + j) * a.nthreads + gtid];
        }
    };
    auto stack_put = [&](uint32_t c, uint32_t j, uint32_t m) {
        if (LDS_ROWS && (STK == 0 || j < 3u)) s_stack[c][j][threadIdx.x] = m;
        else a.stack[((size_t)c * stack_levels + j) * a.nthreads + gtid] = m;
    };

    // =========================== transition pass ============================================
    // A wave inside its pass steps none of its rays, so the pass is written for latency.  (i) ONE pass serves every lane that has
    // a parked context, whichever of its two it is (context A if both are: B follows in the next pass): the lane's context is
    // picked into working values on entry and written back under lane masks on exit, so the wave makes one trip through the
    // pass's memory round trips where a pass per context kind made two.  (ii) Every lane's first loads — the sky entry of a path
    // that ends in the sky, the material of a hit, worklist entry and noise bytes of a new path — go out together, whatever
    // kind of transition the lane makes, then the direction-table entries of everyone who starts a level; each batch is
    // waited for once.  (Loads inside the three divergent branches, as the pass was first written, made a chain of five
    // dependent round trips.)  A lane that has nothing to fetch reads element 0.
    constexpr uint32_t PP_SNAN = 1u << 19, PP_FNAN = 1u << 18;   // PPath::st: the shadow / diffuse direction in the slot registers has a NaN
    auto pass = [&]() {
        const bool endA = !lm_lane(rSA | rFA), endB = !lm_lane(rSB | rFB);
        const bool workA = endA && !(exhausted && (PA.st >> 20) == 0u), workB = endB && !(exhausted && (PB.st >> 20) == 0u);
        const bool act = workA || workB, useB = !workA && workB;
        const uint32_t c = useB ? 1u : 0u;
        // the lane's context
        const uint32_t Fnk = useB ? FB.nk : FA.nk;
        const float Fpx = useB ? FB.px : FA.px, Fpy = useB ? FB.py : FA.py, Fpz = useB ? FB.pz : FA.pz;
        const float Fdx = useB ? FB.ndx : FA.ndx, Fdy = useB ? FB.ndy : FA.ndy, Fdz = useB ? FB.ndz : FA.ndz;
        const uint32_t Fvox = useB ? pr_vox(FB) : pr_vox(FA);
        const lanemask useBm = __ballot(useB);
        const uint32_t Faxis = lm_lane((zFB & useBm) | (zFA & ~useBm)) ? 2u : (lm_lane((xyFB & useBm) | (xyFA & ~useBm)) ? 0u : 1u);
        // whether an ended ray reached the sky is read off its position (:138-145)
        const bool Sair = useB ? pr_outside<LOGR, LRZ>(SB.px, SB.py, SB.pz, f.lr[0], f.lr[1], f.lr[2])
                               : pr_outside<LOGR, LRZ>(SA.px, SA.py, SA.pz, f.lr[0], f.lr[1], f.lr[2]);
        const bool Fair = pr_outside<LOGR, LRZ>(Fpx, Fpy, Fpz, f.lr[0], f.lr[1], f.lr[2]);
        uint32_t Pst = useB ? PB.st : PA.st, Pitem = useB ? PB.item : PA.item, Pent = useB ? PB.ent : PA.ent;

        const uint32_t level = Pst >> 20;
        const bool mine = act && level != 0u;
        const bool fin = mine && (Fair || (Pst & PP_FINAL) != 0u);   // sky exit or last level: the path ends
        const bool cont = mine && !fin;
        if (COUNT) {
            d_passf++; d_plf += (uint32_t)__popcll(__ballot(mine));
            if (mine) { if (useB) { tally(FB); tally(SB); } else { tally(FA); tally(SA); } }
        }
        // contexts without a path (or whose path ends here) pull the next ones: the wave owns a chunk of kChunk consecutive paths
        // of its XCD group's share (one atomicAdd per chunk) and deals them out ballot-ranked (see k_persist)
        bool getnew = false;
        uint32_t nsb = 0, nw = 0;   // (sample-in-batch, worklist slot) of the new path
        if (!exhausted) {
            const bool wantme = act && (level == 0u || fin);
            const uint64_t want = __ballot(wantme);
            const uint32_t nwant = (uint32_t)__popcll(want);
            if (nwant) {
                if (chunk_next >= chunk_end) {
                    for (;;) {
                        if (grp_tries == 8u) { exhausted = true; chunk_next = chunk_end = 0u; break; }
                        const uint32_t g = (home_grp + grp_tries) & 7u;
                        const uint32_t w0 = (uint32_t)((uint64_t)nwork * g >> 3), gw = (uint32_t)((uint64_t)nwork * (g + 1u) >> 3) - w0;
                        const uint32_t ng = gw * a.nsamples;
                        uint32_t base = 0;
                        if (lane == 0) base = atomicAdd(a.cursor + 32u * g, kChunk);
                        base = __builtin_amdgcn_readfirstlane(base);
                        if (base < ng) {
                            chunk_next = base; chunk_end = base + kChunk < ng ? base + kChunk : ng;
                            chunk_w0 = w0; chunk_nw = gw;
                            chunk_sb = base / gw; chunk_w = base - chunk_sb * gw;   // once per chunk
                            break;
                        }
                        grp_tries++;   // that group's share is handed out for good (its cursor only grows)
                    }
                }
                const uint32_t take = min(nwant, chunk_end - chunk_next);
                chunk_next += take;
                const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(want >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)want, 0u));
                if (wantme && rank < take) {
                    // (sample-in-batch, slot) of path first+rank, stepped from the chunk's running position (no division)
                    uint32_t sb = chunk_sb, w = chunk_w + rank;
                    while (w >= chunk_nw) { w -= chunk_nw; sb++; }
                    nsb = sb; nw = w + chunk_w0;
                    getnew = true;
                }
                chunk_w += take;
                while (chunk_w >= chunk_nw) { chunk_w -= chunk_nw; chunk_sb++; }
            }
        }

        // ---- first batch of loads
        const bool air = fin && Fair;
        const uint32_t kind = stop_kind(Fnk, Fvox, cont);   // 0 = hit, 1 = loop limit, 2 = special
        const bool hit = cont && kind == 0u;
        const float4 skyv = a.dif_lut[air ? 4u * Pent + 3u : 0u];   // :331-332 / :343-345, tabulated per frame; Pent = the entry F walked
        const uint32_t matv = sc.mat[(hit && !(!LRZ && ps_border(Fvox))) ? Fvox : 0u];   // the hit texel is the texel of the last fetch (:150-154); border: 0
        // one sample per pixel (a.direct): the ending path's pixel, to store its lighting here (Pitem is its worklist slot then)
        // (depth <= 4 builds only — the reference's frames are depth 2; launch_paths_direct_ok tells the host)
        uint32_t lp_done = 0;
        if (STK == 0 && a.direct) lp_done = a.worklist[fin ? Pitem : 0u];
        const float4 ph = a.phit[nw];       // the prepass' record of the new path's pixel: one 16-byte load
        const uint32_t info = u_bits(ph.w);
        const float ox = ph.x, oy = ph.y, oz = ph.z;
        // noise_offset of the new path (:298-304) and its noise_value texel (:324, :336); one integer lookup serves every level
        // (Q5; tests/test_math_contract.py::test_noise_value_texel_is_level_independent)
        const uint32_t seed = (f.seed + a.sample0 + nsb) % (uint32_t)RT_NOISE_BYTES;
        const uint32_t by = seed / RT_NOISE_SIZE;
        const uint32_t nb = sc.noise[(by > 511u ? 511u : by) * RT_NOISE_SIZE + seed % RT_NOISE_SIZE];
        const uint32_t wgx8 = info & 0x3FFFu, wgy8 = (info >> 14) & 0x3FFFu;
        const uint32_t ntx = ((nb & 0xFFu) + wgx8) & 511u, nty = (((nb >> 8) & 0xFFu) + wgy8) & 511u;
        const uint32_t nse = sc.noise[nty * RT_NOISE_SIZE + ntx] & 0xFFFFu;   // second round trip, new paths only
        // depth 5..8 at region 256: the stack levels that live in global memory (4..7) ride in the first batch — the sum below
        // then makes no round trips of its own
        uint32_t pf0 = 0, pf1 = 0, pf2 = 0, pf3 = 0;
        if constexpr (STK == 2) {
            const size_t col = (size_t)c * stack_levels * a.nthreads + gtid;
            pf0 = a.stack[col + (size_t)min(3u, stack_levels - 1u) * a.nthreads]; pf1 = a.stack[col + (size_t)min(4u, stack_levels - 1u) * a.nthreads];
            pf2 = a.stack[col + (size_t)min(5u, stack_levels - 1u) * a.nthreads]; pf3 = a.stack[col + (size_t)min(6u, stack_levels - 1u) * a.nthreads];
        }

        // ---- a path ends: L_j = [sun_j] S + L_{j+1} * albedo_{j+1} + emission, innermost first (raytrace.comp:346-348)
        uint32_t sunbits = Pst & 0xFFFFu;
        if (mine && Sair) sunbits |= 1u << (level - 1u);                          // :326-328 / :338-340
        if (fin) {
            vec3 L = v3(0.0f, 0.0f, 0.0f);
            if (sunbits >> (level - 1u) & 1u) L = vadd(L, sunlight);
            if (air) L = vadd(L, v3(skyv.x, skyv.y, skyv.z));
            for (uint32_t j = level - 1u; j >= 1u; j--) {
                uint32_t pm;
                if constexpr (STK == 2) {
                    const uint32_t k = j - 1u;
                    pm = k < 3u ? s_stack[c][k < 3u ? k : 0u][threadIdx.x] : (k == 3u ? pf0 : (k == 4u ? pf1 : (k == 5u ? pf2 : pf3)));
                } else {
                    pm = stack_at(c, j - 1u);
                }
                vec3 light2 = vmul(L, v3(s_albedo[pm >> 14 & 0x7Fu], s_albedo[pm >> 7 & 0x7Fu], s_albedo[pm & 0x7Fu]));
                light2 = vadd(light2, v3(0.0f, 0.0f, 0.0f));      // + dif.emission, always vec3(0) (:155)
                vec3 acc = v3(0.0f, 0.0f, 0.0f);
                if (sunbits >> (j - 1u) & 1u) acc = vadd(acc, sunlight);
                L = vadd(acc, light2);
            }
            const vec3 light = vadd(v3(0.0f, 0.0f, 0.0f), L);
            if (STK == 0 && a.direct) {   // the sum over the pixel's one sample is 0 + light, and the pixel is finished (k_accumulate_paths' arithmetic)
                const PixelId dp = pixel_of_local(f, lp_done);
                if (dp.inside) store_lighting(pl, dp.out_index, v3(0.0f + light.x, 0.0f + light.y, 0.0f + light.z), f.spp);
            } else
#ifndef RT_DIAG_NO_PL_STORE   // diagnostic build (tools/variant.sh nopl rt_paths.hip -DRT_DIAG_NO_PL_STORE): wrong frames, timing only
            {   // k_accumulate_paths adds a pixel's samples in order.  The record is read once, after the launch: when a launch
                // writes more of them than the caches would keep (PersistArgs::pl_stream; the headline launch writes 1.6 GB) they go
                // out as streaming stores (`nt`) and do not push the scene out of L2 — headline launch 4.06 -> 4.03 ms, C4 50.0 -> 48.9,
                // C5 20.85 -> 20.73 (profiles/r3_pl_nontemporal.txt)
                typedef float f3v __attribute__((ext_vector_type(3)));
                const f3v val = {light.x, light.y, light.z};
                PathLight* dst = a.pl + Pitem;
                if (a.pl_stream) asm volatile("global_store_dwordx3 %0, %1, off nt" :: "v"(dst), "v"(val) : "memory");
                else asm volatile("global_store_dwordx3 %0, %1, off" :: "v"(dst), "v"(val) : "memory");
            }
#else
            if (light.x == 12345.678f) a.pl[Pitem] = PathLight{light.x, light.y, light.z};   // keeps the unwinding alive, never stores
#endif
            Pst = PP_FINAL;
        }
        // ---- a diffuse ray hit: the next level stands on the hit point with the 0.001 face offset (:166-180)
        bool start = false;
        float sfx = 0, sfy = 0, sfz = 0;
        uint32_t snormal = 0;
        if (cont) {
            const uint32_t nrm = Faxis == 0u ? (Fdx < 0.0f ? 1u : 0u) : (Faxis == 1u ? (Fdy < 0.0f ? 3u : 2u) : (Fdz < 0.0f ? 5u : 4u));
            float hx = Fpx, hy = Fpy, hz = Fpz;
            if (kind == 2u) { hx = hy = hz = __builtin_nanf(""); }
            const float offv = (nrm & 1u) ? -0.001f : 0.001f;
            hx = Faxis == 0u ? hx + offv : hx; hy = Faxis == 1u ? hy + offv : hy; hz = Faxis == 2u ? hz + offv : hz;
            stack_put(c, level - 1u, (hit && !(!LRZ && ps_border(Fvox))) ? matv : 0u);   // albedo of surface level+1
            Pst = sunbits | (Pst & (PP_SNAN | PP_FNAN)) | (level + 1u) << 20 | (level + 1u == D ? PP_FINAL : 0u);
            sfx = hx; sfy = hy; sfz = hz; snormal = nrm;
            start = true;
        }
        if (getnew) {
            sfx = ox; sfy = oy; sfz = oz;
            snormal = info >> 28;
            Pitem = nsb * nwork + nw;
            Pent = nse | 7u << 16;
            Pst = 1u << 20 | (D == 1u ? PP_FINAL : 0u);
            start = true;
        }

        // ---- second batch: the direction tables.  The shadow ray's direction depends on the path's noise bytes only (one read per
        // path); F's direction registers still hold the entry of the path's previous level, which repeats whenever the next
        // surface has the same face.
        const uint32_t se = Pent & 0xFFFFu;
        const bool newface = start && snormal != Pent >> 16;
        const uint32_t di = newface ? 4u * ((snormal << 16) | se) : 0u;
        const uint32_t si = getnew ? 2u * se : 0u;
        // (1 / |direction| comes from the tables too: computing the six IEEE divisions here instead of the two extra 16-byte loads
        // was measured at +3.0 % on the headline launch, +2.5 % on C4, +1.7 % on C5 — profiles/r3_kpaths_variants_c.txt)
        const float4 d2 = a.dif_lut[di + 1u], dl = a.dif_lut[di + 2u];
        const float4 sd = a.sun_lut[si], sl = a.sun_lut[si + 1u];
        // both rays of the level (:324-330 / :336-342) from the surface point (sfx, sfy, sfz) with face id snormal: head of
        // trace_ray (:83-107)
        if (COUNT && start) { c_noise++; c_shadow++; c_dif++; }
        int ix, iy, iz;
        const bool ok = wrap_texel(v3(sfx, sfy, sfz), (float)R, &ix, &iy, &iz);
        const uint32_t tx = s_swz[ix], ty = s_swz[kTabWords + iy], tz = s_swz[2u * kTabWords + iz];
        if (newface) {
            Pent = se | snormal << 16;
            Pst = (Pst & ~PP_FNAN) | ((d2.x != d2.x || d2.y != d2.y || d2.z != d2.z) ? PP_FNAN : 0u);
        }
        if (getnew) Pst |= (sd.x != sd.x || sd.y != sd.y || sd.z != sd.z) ? PP_SNAN : 0u;
        // NaN direction, or a first texel outside the texture (border value 0: step_size 0 on a fresh ray): the ray never runs
        bool runS = start && !((Pst & PP_SNAN) || !ok), runF = start && !((Pst & PP_FNAN) || !ok);
        const uint32_t nk0 = (uint32_t)RT_TRACE_LIMIT | ((COUNT && !ok) ? kFreshInvalid : 0u);
        // a level whose origin lies outside the region (rare: a camera outside, a hit on the region's face).  lr = 0 builds step with
        // the q of positions at or above -R/2, so an origin below takes its first step apart; and a ray that will never move from out
        // there must not be taken for one that left the region (the pass reads "reached the sky" off the position): see below
        const bool outside = start && ok && pr_outside<LOGR, LRZ>(sfx, sfy, sfz, f.lr[0], f.lr[1], f.lr[2]);
        const bool below = LRZ && outside && (sfx + half < 0.0f || sfy + half < 0.0f || sfz + half < 0.0f);

        // the two fresh rays as they enter the step loop
        float Spx = sfx, Spy = sfy, Spz = sfz, Tpx = sfx, Tpy = sfy, Tpz = sfz;
        uint32_t Ssx = tx, Ssy = ty, Ssz = tz, Tsx = tx, Tsy = ty, Tsz = tz, nkS = nk0, nkF = nk0;
        bool Tz = true, Txy = false;     // a ray that ends before its first step reports the z face (:90)
        if (__builtin_expect(__ballot(below) != 0ull, 0)) {
            PSlot TS, TF;
            TS.px = sfx; TS.py = sfy; TS.pz = sfz; TS.sx = tx; TS.sy = ty; TS.sz = tz; TS.axis = 2u;
            TS.nk = runS ? (uint32_t)RT_TRACE_LIMIT : (K_DEAD | K_END);
            TF = TS; TF.nk = runF ? (uint32_t)RT_TRACE_LIMIT : (K_DEAD | K_END);
            TS.ndx = getnew ? -sd.x : (useB ? SB.ndx : SA.ndx); TS.ndy = getnew ? -sd.y : (useB ? SB.ndy : SA.ndy);
            TS.ndz = getnew ? -sd.z : (useB ? SB.ndz : SA.ndz);
            TS.lx = getnew ? sl.x : (useB ? SB.lx : SA.lx); TS.ly = getnew ? sl.y : (useB ? SB.ly : SA.ly); TS.lz = getnew ? sl.z : (useB ? SB.lz : SA.lz);
            TF.ndx = newface ? -d2.x : Fdx; TF.ndy = newface ? -d2.y : Fdy; TF.ndz = newface ? -d2.z : Fdz;
            TF.lx = newface ? dl.x : (useB ? FB.lx : FA.lx); TF.ly = newface ? dl.y : (useB ? FB.ly : FA.ly); TF.lz = newface ? dl.z : (useB ? FB.lz : FA.lz);
            first_step(TS, below); first_step(TF, below);
            if (below) {   // back from the flag word to the booleans
                Spx = TS.px; Spy = TS.py; Spz = TS.pz; Ssx = TS.sx; Ssy = TS.sy; Ssz = TS.sz;
                Tpx = TF.px; Tpy = TF.py; Tpz = TF.pz; Tsx = TF.sx; Tsy = TF.sy; Tsz = TF.sz;
                if (runS) { runS = ps_running(TS.nk); nkS = TS.nk & 0xFFFFu; }
                if (runF) { runF = ps_running(TF.nk); nkF = TF.nk & 0xFFFFu; Tz = TF.axis == 2u; Txy = TF.axis == 0u; }
            }
        }
        if (__builtin_expect(__ballot(outside) != 0ull, 0)) {
            // a ray that has not moved from an origin outside the region and will not — it is dead, or it stands on a 0 — is "special"
            // (Q12: NaN position, material 0), not a sky exit: give it its NaN position now
            uint32_t st0 = 1u;
            if (outside) st0 = lookup(tx | ty | tz);
            const float qnan = __builtin_nanf("");
            if (outside && (nkS & 0xFFFFu) == (uint32_t)RT_TRACE_LIMIT && !(runS && st0 != 0u)) { Spx = Spy = Spz = qnan; runS = false; }
            if (outside && (nkF & 0xFFFFu) == (uint32_t)RT_TRACE_LIMIT && !(runF && st0 != 0u)) { Tpx = Tpy = Tpz = qnan; runF = false; }
        }

        // ---- write the lane's context back
        if (useB) { PB.st = Pst; PB.item = Pitem; PB.ent = Pent; } else { PA.st = Pst; PA.item = Pitem; PA.ent = Pent; }
        const bool startA = start && !useB, startB = start && useB;
        if (startA) {
            SA.px = Spx; SA.py = Spy; SA.pz = Spz; SA.sx = Ssx; SA.sy = Ssy; SA.sz = Ssz; SA.nk = nkS;
            FA.px = Tpx; FA.py = Tpy; FA.pz = Tpz; FA.sx = Tsx; FA.sy = Tsy; FA.sz = Tsz; FA.nk = nkF;
        }
        if (startB) {
            SB.px = Spx; SB.py = Spy; SB.pz = Spz; SB.sx = Ssx; SB.sy = Ssy; SB.sz = Ssz; SB.nk = nkS;
            FB.px = Tpx; FB.py = Tpy; FB.pz = Tpz; FB.sx = Tsx; FB.sy = Tsy; FB.sz = Tsz; FB.nk = nkF;
        }
        {
            const lanemask sA = __ballot(startA), sB = __ballot(startB), mS = __ballot(runS), mF = __ballot(runF), mz = __ballot(Tz), mxy = __ballot(Txy);
            rSA = (rSA & ~sA) | (mS & sA); rFA = (rFA & ~sA) | (mF & sA); zFA = (zFA & ~sA) | (mz & sA); xyFA = (xyFA & ~sA) | (mxy & sA);
            rSB = (rSB & ~sB) | (mS & sB); rFB = (rFB & ~sB) | (mF & sB); zFB = (zFB & ~sB) | (mz & sB); xyFB = (xyFB & ~sB) | (mxy & sB);
        }
        if (newface && !useB) { FA.ndx = -d2.x; FA.ndy = -d2.y; FA.ndz = -d2.z; FA.lx = dl.x; FA.ly = dl.y; FA.lz = dl.z; }
        if (newface && useB) { FB.ndx = -d2.x; FB.ndy = -d2.y; FB.ndz = -d2.z; FB.lx = dl.x; FB.ly = dl.y; FB.lz = dl.z; }
        if (getnew && !useB) { SA.ndx = -sd.x; SA.ndy = -sd.y; SA.ndz = -sd.z; SA.lx = sl.x; SA.ly = sl.y; SA.lz = sl.z; }
        if (getnew && useB) { SB.ndx = -sd.x; SB.ndy = -sd.y; SB.ndz = -sd.z; SB.lx = sl.x; SB.ly = sl.y; SB.lz = sl.z; }
    };

    // ---- one step of all four rays: nibble reads, then byte loads, then the arithmetic ----
    // SHADOW: the shadow rays step too.  CAREFUL: the steps test the iteration counter (p_step).
#ifdef RT_DIAG_STEP_TIMES   // where a step's time goes once the paths have run out (shader-clock ticks; tools/lab/r4/step_times.py)
    unsigned long long t_lds = 0, t_mem = 0, t_alu = 0, t_steps = 0, t_bytes = 0;
#endif
    auto step_all = [&](auto shadow_tag, auto careful_tag) {
        constexpr bool SHADOW = decltype(shadow_tag)::value, CAREFUL = decltype(careful_tag)::value;
#ifdef RT_DIAG_STEP_TIMES
        const bool timed = exhausted;
        unsigned long long tq0 = 0, tq1 = 0, tq2 = 0;
        if (timed) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); tq0 = clock64(); }
#endif
        const uint32_t v1 = pr_vox(FA), v3_ = pr_vox(FB);
        uint32_t n1, n3, h1, h3;     // nibble-map byte and nibble offset of each ray's texel
        ps_nibble_of<LOGR, LRZ>(v1, &n1, &h1); ps_nibble_of<LOGR, LRZ>(v3_, &n3, &h3);
        uint32_t v0 = 0, v2 = 0, n0 = 0, n2 = 0, h0 = 0, h2 = 0, w0 = 0, w2 = 0;
        if (SHADOW) {
            v0 = pr_vox(SA); v2 = pr_vox(SB);
            ps_nibble_of<LOGR, LRZ>(v0, &n0, &h0); ps_nibble_of<LOGR, LRZ>(v2, &n2, &h2);
            w0 = s_nib[n0]; w2 = s_nib[n2];
        }
        const uint32_t w1 = s_nib[n1], w3 = s_nib[n3];
        uint32_t t1 = __builtin_amdgcn_ubfe(w1, h1, 4u), t3 = __builtin_amdgcn_ubfe(w3, h3, 4u);
        if (!LRZ) {   // the border texel reads as "mixed", and its byte offset is out of the buffer's range: value 0
            t1 |= (uint32_t)((int32_t)v1 >> 31) & 15u; t3 |= (uint32_t)((int32_t)v3_ >> 31) & 15u;
        }
#ifdef RT_DIAG_STEP_TIMES
        if (timed) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq1 = clock64(); }
#endif
        const lanemask m1 = rFA & __ballot(t1 == kNibMixed), m3 = rFB & __ballot(t3 == kNibMixed);   // only a ray in flight on a mixed cube fetches
        const bool g1 = lm_lane(m1), g3 = lm_lane(m3);
        uint32_t t0 = 0, t2 = 0, b0 = 0, b2 = 0;
        bool g0 = false, g2 = false;
        lanemask m0 = 0, m2 = 0;
        if (SHADOW) {
            t0 = __builtin_amdgcn_ubfe(w0, h0, 4u); t2 = __builtin_amdgcn_ubfe(w2, h2, 4u);
            if (!LRZ) { t0 |= (uint32_t)((int32_t)v0 >> 31) & 15u; t2 |= (uint32_t)((int32_t)v2 >> 31) & 15u; }
            m0 = rSA & __ballot(t0 == kNibMixed); m2 = rSB & __ballot(t2 == kNibMixed);
            g0 = lm_lane(m0); g2 = lm_lane(m2);
        }
        if (COUNT) {
            d_fetch += (uint32_t)__popcll(rFA) + (uint32_t)__popcll(rFB) + (SHADOW ? (uint32_t)__popcll(rSA) + (uint32_t)__popcll(rSB) : 0u);
            d_fetch_cube += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m3) + (uint32_t)__popcll(m0) + (uint32_t)__popcll(m2);
        }
        if constexpr (LOGR > 8 && RT_PATHS_BRICK_MAP != 0) {
            // second level (round 4): the brick's own nibble from the global per-brick map (L2-resident: 1 MiB / 8 MiB); the byte
            // array is read only by lanes whose BRICK is mixed.  At R = 1024 a cube of the LDS map is 16^3 voxels and most cubes
            // near the terrain read "mixed" although 60-80 % of their 4^3 bricks are uniform: this turns those fetches from one
            // 64-byte line of a 1 GiB array each into hits in an 8 MiB map.
            uint32_t u0 = 0, u2 = 0;
            if (SHADOW) u0 = __builtin_amdgcn_raw_buffer_load_b8(brick_rsrc, g0 ? v0 >> 7 : 0xFFFFFFFFu, 0, 0);
            uint32_t u1 = __builtin_amdgcn_raw_buffer_load_b8(brick_rsrc, g1 ? v1 >> 7 : 0xFFFFFFFFu, 0, 0);
            if (SHADOW) u2 = __builtin_amdgcn_raw_buffer_load_b8(brick_rsrc, g2 ? v2 >> 7 : 0xFFFFFFFFu, 0, 0);
            uint32_t u3 = __builtin_amdgcn_raw_buffer_load_b8(brick_rsrc, g3 ? v3_ >> 7 : 0xFFFFFFFFu, 0, 0);
            u1 = __builtin_amdgcn_ubfe(u1, (v1 >> 4) & 4u, 4u); u3 = __builtin_amdgcn_ubfe(u3, (v3_ >> 4) & 4u, 4u);
            const lanemask n1 = m1 & __ballot(u1 == kNibMixed), n3 = m3 & __ballot(u3 == kNibMixed);
            const bool k1 = lm_lane(n1), k3 = lm_lane(n3);
            bool k0 = false, k2 = false;
            lanemask n0 = 0, n2 = 0;
            if (SHADOW) {
                u0 = __builtin_amdgcn_ubfe(u0, (v0 >> 4) & 4u, 4u); u2 = __builtin_amdgcn_ubfe(u2, (v2 >> 4) & 4u, 4u);
                n0 = m0 & __ballot(u0 == kNibMixed); n2 = m2 & __ballot(u2 == kNibMixed);
                k0 = lm_lane(n0); k2 = lm_lane(n2);
                b0 = __builtin_amdgcn_raw_buffer_load_b8(mine_rsrc, k0 ? v0 : 0xFFFFFFFFu, 0, 0);
            }
            if (COUNT) d_fetch_brick += (uint32_t)__popcll(n1) + (uint32_t)__popcll(n3) + (uint32_t)__popcll(n0) + (uint32_t)__popcll(n2);
            const uint32_t b1 = __builtin_amdgcn_raw_buffer_load_b8(mine_rsrc, k1 ? v1 : 0xFFFFFFFFu, 0, 0);
            if (SHADOW) b2 = __builtin_amdgcn_raw_buffer_load_b8(mine_rsrc, k2 ? v2 : 0xFFFFFFFFu, 0, 0);
            const uint32_t b3 = __builtin_amdgcn_raw_buffer_load_b8(mine_rsrc, k3 ? v3_ : 0xFFFFFFFFu, 0, 0);
            t1 = g1 ? (k1 ? b1 : u1) : t1; t3 = g3 ? (k3 ? b3 : u3) : t3;
            if (SHADOW) { t0 = g0 ? (k0 ? b0 : u0) : t0; t2 = g2 ? (k2 ? b2 : u2) : t2; }
        } else {
            if (COUNT) d_fetch_brick += (uint32_t)__popcll(m1) + (uint32_t)__popcll(m3) + (uint32_t)__popcll(m0) + (uint32_t)__popcll(m2);
            if (SHADOW) b0 = __builtin_amdgcn_raw_buffer_load_b8(mine_rsrc, g0 ? v0 : 0xFFFFFFFFu, 0, 0);
            const uint32_t b1 = __builtin_amdgcn_raw_buffer_load_b8(mine_rsrc, g1 ? v1 : 0xFFFFFFFFu, 0, 0);
            if (SHADOW) b2 = __builtin_amdgcn_raw_buffer_load_b8(mine_rsrc, g2 ? v2 : 0xFFFFFFFFu, 0, 0);
            const uint32_t b3 = __builtin_amdgcn_raw_buffer_load_b8(mine_rsrc, g3 ? v3_ : 0xFFFFFFFFu, 0, 0);
            t1 = g1 ? b1 : t1; t3 = g3 ? b3 : t3;
            if (SHADOW) { t0 = g0 ? b0 : t0; t2 = g2 ? b2 : t2; }
        }
#ifdef RT_DIAG_STEP_TIMES
        if (timed) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tq2 = clock64();
            t_bytes += (uint32_t)__popcll(__ballot(g1)) + (uint32_t)__popcll(__ballot(g3)) + (uint32_t)__popcll(__ballot(g0)) + (uint32_t)__popcll(__ballot(g2));
        }
#endif
        if (SHADOW) p_step<false, LOGR, LRZ, CAREFUL>(SA, rSA, nul, nul2, t0, swz, f.lr[0], f.lr[1], f.lr[2]);
        p_step<true, LOGR, LRZ, CAREFUL>(FA, rFA, zFA, xyFA, t1, swz, f.lr[0], f.lr[1], f.lr[2]);
        if (SHADOW) p_step<false, LOGR, LRZ, CAREFUL>(SB, rSB, nul, nul2, t2, swz, f.lr[0], f.lr[1], f.lr[2]);
        p_step<true, LOGR, LRZ, CAREFUL>(FB, rFB, zFB, xyFB, t3, swz, f.lr[0], f.lr[1], f.lr[2]);
#ifdef RT_DIAG_STEP_TIMES
        if (timed) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long tq3 = clock64();
            t_lds += tq1 - tq0; t_mem += tq2 - tq1; t_alu += tq3 - tq2; t_steps++;
        }
#endif
    };
    // The loop limit (raytrace.comp:109).  The steps of the main loop only count a ray's iterations; every 16th look the wave
    // asks whether a ray in flight has come within reach of the limit (16 looks = 48 iterations) and, if so, steps with the
    // counter tested in every iteration until no such ray is left — at most kLimitMargin iterations, no pass in between.  On
    // terrain this never runs (a ray crosses a unit plane per iteration: <= 969 at R = 256); a 1024^3 region full of value 1
    // gets there, and so does a ray stalled at a cell boundary far from the origin, where 1e-4 is below the positions' spacing.
    constexpr uint32_t kLimitMargin = 16u * RT_PATHS_STEPS_PER_CHECK + 4u;
    auto near_limit = [&]() -> lanemask {
        return (rSA & __ballot((SA.nk & 0xFFFFu) < kLimitMargin)) | (rFA & __ballot((FA.nk & 0xFFFFu) < kLimitMargin)) |
               (rSB & __ballot((SB.nk & 0xFFFFu) < kLimitMargin)) | (rFB & __ballot((FB.nk & 0xFFFFu) < kLimitMargin));
    };

    for (;;) {
        uint64_t park;
        for (;;) {
            // a context parks when both its rays have ended; when `threshold` lanes have a parked context the pass runs
            const uint64_t eA = ~(rSA | rFA), eB = ~(rSB | rFB);     // (every lane of the workgroup's waves is active: 1024 threads)
            uint64_t idleA = 0ull, idleB = 0ull;   // lanes whose context is empty for good (no paths left)
            if (exhausted) { idleA = __ballot((PA.st >> 20) == 0u); idleB = __ballot((PB.st >> 20) == 0u); }
            park = (eA & ~idleA) | (eB & ~idleB);
            // (once the paths have run out nobody refills the wave, and fewer parked lanes are enough.  From there a wave makes 24 more
            // looks and 22 passes on the headline frame — 0.21 ms, whatever the launch's size; the slowest waves twice that:
            // tools/drain_times.py, profiles/r3_drain_*.txt, DESIGN.md 5 "The drain")
            if ((uint32_t)__popcll(park) >= (exhausted ? (uint32_t)RT_PATHS_DRAIN_PARK : threshold) || (eA & eB) == ~0ull) break;
            if (COUNT) { d_iters++; d_live += (uint32_t)__popcll(~eA) + (uint32_t)__popcll(~eB); }
            if (__builtin_expect((looks++ & 15u) == 0u, 0)) {
                while (__builtin_expect(near_limit() != 0ull, 0)) step_all(std::true_type{}, std::true_type{});
            }
#pragma unroll
            for (int rep = 0; rep < RT_PATHS_STEPS_PER_CHECK; rep++) {
                // compile-time (the loop is unrolled).  The larger regions' steps wait on memory, not on the VALU: every ray steps
                // in every repetition there (1024^3 region, 4K spp-1024 depth-8 frame: 182 ms against 188 ms with 0x7)
                if (LOGR != 8 || (RT_PATHS_SHADOW_REPS >> rep & 1) != 0) step_all(std::true_type{}, std::false_type{});
                else step_all(std::false_type{}, std::false_type{});
            }
        }
        if (park == 0ull) break;   // nothing in flight, nothing parked, no paths left
#ifdef RT_DIAG_WAVE_TIMES
        if (exhausted && !t_exh) t_exh = wall_clock64();
#endif
        __builtin_amdgcn_s_setprio(1);   // a wave in its pass holds 256 slot-lanes still: let it through (0.6 %)
        pass();
        __builtin_amdgcn_s_setprio(0);
    }
    if (COUNT) {
        DevCounters* cn = a.counters;
        const unsigned long long rays = c_shadow + c_dif;
        wave_add(&cn->rays, rays); wave_add(&cn->rays_shadow, c_shadow);
        wave_add(&cn->rays_diffuse, c_dif); wave_add(&cn->iterations, c_iter); wave_add(&cn->minefield_fetches, rays + c_iter);
        wave_add(&cn->hits, c_hits); wave_add(&cn->material_fetches, c_hits); wave_add(&cn->sky_exits, c_sky);
        wave_add(&cn->limit_exits, c_limit); wave_add(&cn->border_fetches, c_border); wave_add(&cn->noise_fetches, c_noise);
        if (lane == 0) {
#ifndef RT_DIAG_WAVE_TIMES
            atomicAdd(&cn->dbg_loop_iters, d_iters); atomicAdd(&cn->dbg_f_lanes, d_live);
            atomicAdd(&cn->dbg_passes, d_passf); atomicAdd(&cn->dbg_pass_lanes, d_plf);
            // where the step loop's fetches are answered (tools/fetch_levels.py): all / beyond the LDS map / beyond the brick map
#ifdef RT_DIAG_STEP_TIMES
            (void)d_fetch; (void)d_fetch_cube; (void)d_fetch_brick;
            atomicAdd(&cn->dbg_s_execs, t_lds); atomicAdd(&cn->dbg_s_lanes, t_mem); atomicAdd(&cn->dbg_sky_lanes, t_alu);
            atomicAdd(&cn->dbg_f_execs, t_steps); atomicAdd(&cn->dbg_loop_iters, t_bytes);
#else
            atomicAdd(&cn->dbg_s_execs, d_fetch); atomicAdd(&cn->dbg_s_lanes, d_fetch_cube); atomicAdd(&cn->dbg_sky_lanes, d_fetch_brick);
#endif
#else       // wave lifetimes on the 100 MHz clock (tools/drain_times.py reads them from RT_DEBUG_STATS' raw lines)
            (void)d_iters; (void)d_live; (void)d_passf; (void)d_plf; (void)d_fetch; (void)d_fetch_cube; (void)d_fetch_brick;
            const unsigned long long t1 = wall_clock64();
            atomicAdd(&cn->dbg_s_execs, t1 - t_wave0); atomicMax(&cn->dbg_f_execs, t1); atomicMax(&cn->dbg_sky_lanes, ~t_wave0);
            atomicAdd(&cn->dbg_s_lanes, t_exh ? t1 - t_exh : 0ull);
            atomicMax(&cn->dbg_loop_iters, t_exh ? ~t_exh : 0ull);    // earliest moment a wave found the paths handed out
            atomicAdd(&cn->dbg_passes, 1ull);                          // waves
            atomicAdd(&cn->dbg_pass_lanes, t1 & 0xFFFFFFFFFFull);      // sum of the end times (40 bits each)
            atomicAdd(&cn->dbg_f_lanes, (t_exh ? t_exh : t1) & 0xFFFFFFFFFFull);   // sum of the exhaustion times
            if (STK == 0) {   // per-wave record in the (otherwise unused) global stack: start, out of paths, end (RT_DEBUG_WAVE_DUMP, tools/lab/r4/wg_end_times.py)
                unsigned long long* rec = reinterpret_cast<unsigned long long*>(a.stack) + (size_t)(gtid >> 6) * 4u;
                rec[0] = t_wave0; rec[1] = t_exh; rec[2] = t1; rec[3] = blockIdx.x;
            }
#endif
        }
    }
}

// PersistArgs::direct is honoured by the depth <= 4, region-256 instantiations only (STK = 0)
bool launch_paths_direct_ok(const Frame& f) { return f.depth <= 4 && f.logr == 8; }

hipError_t launch_paths(const Scene& sc, const Frame& f, const Planes& pl, const PersistArgs& a, bool count, int nworkgroups,
                        hipStream_t st) {
    if (f.logr < 8 || f.logr > 10) return hipErrorInvalidValue;
    const dim3 grid(nworkgroups), block(1024);
    // albedo stack: see STK (the larger regions' swizzle tables take the LDS rows)
    const bool lds_stack = f.depth <= 4 && f.logr == 8;
    const bool lrz = f.lr_zero != 0;   // false: a scrolled region (rt_pslot.hpp, p_advance)
#define RT_LAUNCH_PATHS(C, S, L, Z) hipLaunchKernelGGL((k_paths<C, S, L, Z>), grid, block, 0, st, sc, f, pl, a)
#define RT_LAUNCH_PATHS_CZ(S, L) do { if (count) { if (lrz) RT_LAUNCH_PATHS(true, S, L, true); else RT_LAUNCH_PATHS(true, S, L, false); } \
                                      else { if (lrz) RT_LAUNCH_PATHS(false, S, L, true); else RT_LAUNCH_PATHS(false, S, L, false); } } while (0)
    if (f.logr == 8) {
        if (lds_stack) RT_LAUNCH_PATHS_CZ(0, 8); else if (f.depth <= 8) RT_LAUNCH_PATHS_CZ(2, 8); else RT_LAUNCH_PATHS_CZ(1, 8);
    } else if (f.logr == 9) {
        RT_LAUNCH_PATHS_CZ(1, 9);
    } else {
        RT_LAUNCH_PATHS_CZ(1, 10);
    }
#undef RT_LAUNCH_PATHS_CZ
#undef RT_LAUNCH_PATHS
    return hipGetLastError();
}

}  // namespace rtd
