// rt_kernels.hpp — argument blocks and host-callable launchers of the kernels in rt_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_device.hpp"

namespace rtd {

// SoA ray queue + per-path hit/state arrays, all in HBM (sizes in elements; cap = paths per batch).
//   queue : origin qo*[cap] shared by the pair, directions qd*[2*cap] (shadow at slot, diffuse at cap+slot), qid[cap] = path
//   hits  : hx,hy,hz,hinfo[cap] (diffuse / primary result per path), sunres[cap] (shadow ray reached the sky)
struct TraceArgs {
    const float *qox, *qoy, *qoz, *qdx, *qdy, *qdz;
    const uint32_t* qid;
    const uint32_t* qcount;     // number of queued pairs (written by the previous shade stage)
    uint32_t* cursor;           // global ray cursor, zero before launch
    uint32_t qcap;
    uint32_t nprimary;          // MODE 0: number of paths in the batch
    uint32_t npix_pad;          // MODE 0: paths per sample (local pixels padded to whole tiles)
    uint32_t refill_threshold;  // idle lanes per wave that trigger retire + refill (1..64)
    float *hx, *hy, *hz;
    uint32_t* hinfo;            // material[20:0] | face id << 24 | limit << 30 | air << 31
    uint8_t* sunres;
    DevCounters* counters;
};

struct ShadeArgs {
    float *qox, *qoy, *qoz, *qdx, *qdy, *qdz;
    uint32_t* qid;
    uint32_t* qcount_next;      // pairs appended by this stage (zero before launch)
    uint32_t qcap;
    uint32_t npaths;            // paths in this batch
    uint32_t npaths_cap;        // allocation stride of the per-level albedo stack
    uint32_t npix_pad;
    uint32_t sample0;           // index of the batch's first sample within the frame
    const float *hx, *hy, *hz;
    const uint32_t* hinfo;
    const uint8_t* sunres;
    float *pdx, *pdy, *pdz;     // diffuse direction of the level in flight (sample_sky argument on a sky exit)
    uint8_t* pnormal;
    uint8_t* pstate;            // 1 = path still has rays in flight
    uint32_t* sunbits;          // bit j-1: shadow ray of level j reached the sky
    uint32_t* stack;            // [(depth-1)][npaths_cap] packed material of surface j+1
    float *plx, *ply, *plz;     // finished light of the path
    DevCounters* counters;
};

// k_primary / k_primary2 (rt_persist.hip): primary prepass.  Arrays are indexed by worklist slot (acc: by local pixel).
struct PrimaryArgs {
    float4* phit;               // per queued pixel, ONE 16-byte record: primary hit position (with the 0.001 face offset) and, as bits of .w,
                                // face id << 28 | gl_WorkGroupID.y * 8 << 14 | gl_WorkGroupID.x * 8 (the noise_offset terms, raytrace.comp:304)
    uint32_t* worklist;         // local pixel ids that need shadow/diffuse rays
    uint32_t* wl_count;         // zero before launch
    uint32_t* zero_words;       // k_primary2: words the prepass clears for later launches instead of a memset of their own: the slot's other
    uint32_t zero_count;        // worklist counter (the slot's next frame) ...
    uint32_t* zero_words2;      // ... and the path cursors of the lane this frame's first path launch runs on; either may be null
    uint32_t zero_count2;
    float4* acc;                // (unused by the prepass since it stores the lighting of the pixels it finishes itself)
    DevCounters* counters;
};
// k_persist (rt_persist.hip): persistent path kernel.  Work item r = sample_in_batch * nwork + w.
// light of one path, summed per pixel in sample order by k_accumulate_paths (12 bytes: the records are the path kernels' whole
// write traffic)
struct PathLight { float x, y, z; };
struct PersistArgs {
    uint32_t* cursor;           // [8][32] next path of each XCD group's share, one word per 128-byte line (zero before launch)
    const uint32_t* worklist;   // CACHE: pixels queued by the prepass
    const uint32_t* wl_count;   // CACHE: number of queued pixels (nwork)
    uint32_t npix_pad;          // CACHE=false: nwork = all local pixels (padded to whole 8x8 tiles)
    uint32_t sample0, nsamples; // samples of this batch: sample0 .. sample0+nsamples-1
    uint32_t threshold;         // parked lanes per wave that trigger a transition pass (1..64)
    uint32_t rmin;              // (was k_seq's re-arm threshold; unused)
    uint32_t chunk;             // paths per cursor atomic; 0 = the default (128)
    uint32_t nthreads;          // grid size in threads (stride of the albedo stack)
    uint32_t direct;            // 1: the frame has ONE sample, so a path's light is its pixel's: the kernel stores the lighting planes
                                // itself (0 + light, / spp / 16: what k_accumulate_paths would do) and writes no light record
    uint32_t pl_stream;         // k_paths: 1 = the light records go out as streaming (`nt`) stores — set when a launch's records are too many
                                // to be worth keeping in L2 / the Infinity Cache until k_accumulate_paths reads them
    uint32_t* stack;            // [2][(depth-1)][nthreads] packed material of surface j+1 (only touched when depth >= 2; k_persist uses half)
    const float4* phit;         // CACHE: the primary prepass' record per worklist slot (PrimaryArgs::phit): one load per new path
    const float4* sun_lut;      // [2*65536] per-frame shadow-ray table: direction, 1/|direction|
    const float4* dif_lut;      // [4*6*65536] diffuse-ray table, one 64-byte line per (face, noise byte pair): dir, normalized dir, 1/|dir|, per-frame sky(dir)
    PathLight* pl;              // [nsamples * nwork] light of each path (one 12-byte store per path)
    DevCounters* counters;
};
// k_frame (rt_frame.hip): the whole frame in one launch — primary rays, every sample's paths, the planes — for frames with little work
struct FrameArgs {
    uint32_t threshold;         // parked lanes per wave that trigger a transition pass (1..64; 0 = the default)
    uint32_t tiles_per_group;   // tiles of a (four-wave) workgroup; 0 = chosen by launch_frame from the frame's size and spp (RT_FRAME_TILES overrides)
    uint32_t pair_a;            // set by launch_frame: phase A walks two tiles at a time, one per ray slot (RT_FRAME_PAIR_A=0: one)
    const float4* sun_lut;      // as PersistArgs
    const float4* dif_lut;
    PathLight* pl;              // spp > 1: [pixel's out_index * spp + sample] light of each path, summed in sample order by the pixel's workgroup
    DevCounters* counters;
    unsigned long long* dbg_waves;   // counting build, diagnostics: four words per tile (null: none)
};
bool launch_frame_ok(const Frame& f);   // does k_frame cover this frame (depth <= 8)?
hipError_t launch_frame(const Scene& sc, const Frame& f, const Planes& pl, FrameArgs a, bool count, int num_cus, hipStream_t st);
hipError_t launch_accumulate_paths(const Frame& f, const Planes& planes, const PathLight* pl, const uint32_t* worklist,
                                   const uint32_t* wl_count, uint32_t npix_pad, uint32_t nsamples, bool first_batch, bool last_batch,
                                   bool cache, bool stream, float4* acc, hipStream_t st);
hipError_t launch_sphere_lut(float4* lut, hipStream_t st);
hipError_t launch_dif_lut(const float4* sphere, float4* lut, hipStream_t st);
hipError_t launch_sun_lut(const Frame& f, float4* lut, hipStream_t st);
hipError_t launch_sky_lut(const Frame& f, float4* dif_lut, hipStream_t st);
hipError_t launch_primary(const Scene& sc, const Frame& f, const Planes& pl, const PrimaryArgs& a, bool count,
                          int version /* 1 = k_primary, 2 = k_primary2 */, int nworkgroups, hipStream_t st);
hipError_t launch_persist(const Scene& sc, const Frame& f, const Planes& pl, const PersistArgs& a, bool count, bool cache,
                          int version /* 1 = k_persist */, int nworkgroups, hipStream_t st);

// k_paths (rt_paths.hip): cached-primary frames with lr = 0 and region 256 only
bool launch_paths_direct_ok(const Frame& f);   // does k_paths honour PersistArgs::direct for this frame?
hipError_t launch_paths(const Scene& sc, const Frame& f, const Planes& pl, const PersistArgs& a, bool count, int nworkgroups,
                        hipStream_t st);

hipError_t launch_flatten(const uint8_t* mine_lin, const uint32_t* mat_lin, uint8_t* mine_sw, uint32_t* mat_sw,
                          uint32_t* coarse, uint32_t* brick /* R > 256: the per-brick map, else null */, uint32_t* bad_flag, int logr, hipStream_t st);
hipError_t launch_flatten_slab(const uint8_t* mine_slab, const uint32_t* mat_slab, uint8_t* mine_sw, uint32_t* mat_sw, uint32_t* coarse,
                               uint32_t* brick, int logr, int axis, int offset, hipStream_t st);
hipError_t launch_mega(const Scene& sc, const Frame& f, const Planes& pl, DevCounters* cn, bool count, hipStream_t st);
hipError_t launch_trace(const Scene& sc, const Frame& f, const TraceArgs& a, bool primary, bool count, int nworkgroups, hipStream_t st);
hipError_t launch_shade0(const Scene& sc, const Frame& f, const ShadeArgs& a, const Planes& pl, bool count, hipStream_t st);
hipError_t launch_shadeN(const Scene& sc, const Frame& f, const ShadeArgs& a, int level, bool count, hipStream_t st);
hipError_t launch_accumulate(const float* plx, const float* ply, const float* plz, float4* acc, uint32_t npix_pad,
                             uint32_t nsamples, bool first_batch, hipStream_t st);
hipError_t launch_resolve(const Frame& f, const float4* acc, const Planes& pl, uint32_t npix_pad, hipStream_t st);
hipError_t launch_untile_strided(const void* gathered, size_t rank_stride, void* frame, int world, int capacity, int tiles_x,
                                 int tiles_y, int width, int height, int bpp, hipStream_t st);
hipError_t launch_untile(const void* gathered, void* frame, int world, int capacity, int tiles_x, int tiles_y, int width,
                         int height, int bpp, hipStream_t st);

// rt_post.hip: the reference's post passes
hipError_t launch_denoise_prepare(const void* lighting, const void* depth, const void* normal, int W, int H, void* work,
                                  hipStream_t st);
hipError_t launch_denoise(const void* work_in, int W, int H, int size, bool swapped, bool last, void* work_out, void* lighting,
                          hipStream_t st);
hipError_t launch_selftest_dn_div(unsigned long long* mismatches_dev, hipStream_t st);
hipError_t launch_finalize(const void* albedo, const void* emission, const void* fog, const void* lighting, const void* depth,
                           const uint32_t* noise, int W, int H, void* out_bgra8, hipStream_t st);

}  // namespace rtd
