// rt_api.hip — implementation of the C ABI in include/rt_abi.h on top of the kernels in rt_kernels.hip.
//
// One context = one GPU = one HIP stream.  Nothing here runs ray-trace work on the CPU: if the HIP runtime or a
// device is missing, rt_create fails with RT_ERR_NO_DEVICE and the caller gets no context.
#include <hip/hip_runtime.h>

#include <dlfcn.h>
#include <rccl/rccl.h>   // types and prototypes only: the library is dlopen'ed on first use (rt_comm_* / rt_gather_gbuffer)

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "../../include/rt_abi.h"
#include "../../include/rt_math.h"
#include "rt_kernels.hpp"

namespace {

thread_local std::string g_create_error = "";


const size_t kBytesPerPixel[RT_BUF_COUNT] = {8, 2, 1, 4, 4, 4, 16, 16, 4, 4};

}  // namespace

constexpr size_t kCursorWords = 8 * 32;   // path cursors of the persistent kernels: one word per XCD group, each on its own 128-byte line
// RT_KERNEL_DEFAULT: launches of at least this many pixel-samples run on k_paths, smaller ones on k_persist — unless the whole
// frame is small enough for k_frame (kFrameCrossover, kFrameCrossoverMulti below) — (measured crossover,
// round 3, same box: 1080p spp 1 (2.1 M) 0.351 against 0.348 ms per frame, spp 2 (4.1 M) 0.409 against 0.442; 1024^2 spp 1 0.252 against 0.229)
constexpr uint64_t kPathsCrossover = 3ull << 20;
// ... and ONE-sample frames of fewer pixels than this run on k_frame — the whole frame in one launch, no prepass, worklist or
// accumulate (rt_frame.hip; RT_FRAME_CROSSOVER overrides).  Measured (profiles/r4_frame_kernel.txt), k_frame against prepass +
// k_persist, ms per frame: 256^2 0.084 / 0.186, 1024^2 (the reference's frame) 0.163 / 0.225, 1280x720 0.155 / 0.218, 1920x1080
// 0.275 / 0.297, 2048x1440 0.355 / 0.360 (k_paths: 0.353); with more than one sample per pixel a lane walks them one after the
// other and the persistent kernels win (256^2 spp 4: 0.200 / 0.188).
constexpr uint64_t kFrameCrossover = 5ull << 19;
// More than one sample per pixel: k_frame deals the (pixel, sample) paths of a workgroup's pixels to its lanes and adds a pixel's
// samples in order itself.  It wins while the frame is so small that the persistent kernels' ~0.14 ms per launch dominates — measured
// (k_frame / k_persist, ms per frame): 256^2 spp 2 depth 2 0.105 / 0.176, spp 4 depth 2 0.106 / 0.179, 128^2 spp 16 depth 4 0.213 / 0.206,
// 256^2 spp 16 depth 4 0.246 / 0.211, 512^2 spp 4 depth 4 0.234 / 0.217, 1024^2 spp 2 depth 4 0.352 / 0.290 — i.e. below about 1.5 M
// pixel-sample-levels (pixels x spp x depth; RT_FRAME_CROSSOVER_MULTI overrides).
constexpr uint64_t kFrameCrossoverMulti = 3ull << 19;
// light records of one launch above which k_paths streams them out and k_accumulate_paths streams them in (see rt_draw_frame)
constexpr uint64_t kStreamRecordBytes = 384ull << 20;

struct Lane {
    hipStream_t stream = nullptr;      // lane 0: the context's stream (its own or the caller's, rt_set_stream); lane 1: the library's second stream
    uint32_t* cursor = nullptr;        // kCursorWords: eight path cursors, one 128-byte line each
    bool cursor_clean = false;         // all zero (create-time memset, or the prepass in front of the launch cleared it)
    uint32_t* pstack = nullptr;        // albedo stack of the launch in flight (global part)
    rtd::PathLight* ppl = nullptr;     // light records of the launch in flight
    hipEvent_t ev_join = nullptr;      // "everything submitted to this lane so far" (join_lanes_into)
};
struct FrameSlot {
    void* planes[RT_BUF_COUNT] = {};
    void* gbuffer = nullptr;           // planes 0..5 back to back (256-byte aligned each)
    uint32_t* worklist = nullptr;
    float4* phit = nullptr;            // primary-hit records of the worklist (rtd::PrimaryArgs::phit)
    float4* pacc = nullptr;
    uint32_t* wl_count = nullptr;      // two words on lines of their own: the frame in the slot counts in [wl_parity] while its prepass clears the other
    int wl_parity = 0;
    bool wl_clean[2] = {false, false};
    void* denoise_work[2] = {nullptr, nullptr};   // ping/pong working planes of the denoise passes (16 B/pixel), allocated on first use
    hipEvent_t ev_prepass = nullptr;   // the frame's prepass and tables are done (launches on the other lane wait for it)
    hipEvent_t ev_acc = nullptr;       // the frame's most recent accumulate launch (the next one adds to the same sums, in sample order)
    hipEvent_t ev_tail = nullptr;      // everything submitted for the frame in this slot: the slot's next frame starts after it
    bool tail_recorded = false;
};

struct RtContext {
    RtConfig cfg{};
    int device = 0;
    int num_cus = 256;
    int logr = 8;                       // log2 of the region edge
    int region = RT_ROOT_BLOCK_SIZE;    // R
    size_t vox = 0;                     // R^3
    hipStream_t own_stream = nullptr, stream = nullptr;   // stream: where the frame drawn last ended (post passes, gather, readback follow it there)
    bool user_stream = false;           // rt_set_stream gave a stream: one lane, one slot
    std::string err = "";
    bool has_world = false, has_noise = false;
    bool world_resident = false;          // a full region has been uploaded once (slabs may patch it)

    // scene
    // (the caller-layout copy of the region exists only inside rt_upload_world: 5 B/voxel, 5 GiB at R = 1024)
    uint8_t* d_mine_sw = nullptr; uint32_t* d_mat_sw = nullptr;
    uint32_t* d_coarse = nullptr; uint32_t* d_noise = nullptr; uint32_t* d_flag = nullptr;
    uint32_t* d_brick = nullptr;          // R > 256: per-brick nibble map (rtd::Scene::brick), R^3 / 128 bytes
    // rt_upload_slice: a 16-thick slab travels pinned host staging -> device staging (own stream) -> re-tile (render stream).
    // TWO staging sets used in turn: slab n's transfer waits for slab n - 2's re-tile, not for slab n - 1's, which sits
    // behind the frames on the render stream — a host that uploads one slab per frame never waits for a frame
    uint8_t* d_slab_mine[2] = {nullptr, nullptr}; uint32_t* d_slab_mat[2] = {nullptr, nullptr};
    uint8_t* h_slab_mine[2] = {nullptr, nullptr}; uint32_t* h_slab_mat[2] = {nullptr, nullptr};   // hipHostMalloc
    uint64_t slabs = 0;                   // slabs submitted: the next one uses set slabs & 1
    hipStream_t upload_stream = nullptr;
    hipEvent_t ev_slab_copied[2] = {nullptr, nullptr}, ev_slab_applied[2] = {nullptr, nullptr};   // per set: host staging read / device staging consumed
    bool slab_copy_pending[2] = {false, false}, slab_apply_recorded[2] = {false, false};

    // tiling
    int tiles_x = 0, tiles_y = 0, ntiles_total = 0, ntiles_local = 0, tile_capacity = 0;
    uint32_t npix_pad = 0;     // ntiles_local * 64
    size_t plane_pixels = 0;   // pixels per output plane

    // the planes of the frame drawn last (aliases of its slot's: rt_device_ptr, rt_readback, the post passes and the gather use them)
    void* planes[RT_BUF_COUNT] = {};
    void* gbuffer = nullptr;
    size_t gbuffer_offset[6] = {};
    size_t gbuffer_bytes = 0;
    Lane lanes[2];
    int nlanes = 1;                     // 2: path launches alternate between two streams (persistent kernels on the context's own stream)
    FrameSlot slots[2];
    int nslots = 1;                     // 2 with RT_FLAG_FRAMES_IN_FLIGHT_2
    int cur_slot = 0;                   // slot of the frame drawn last
    uint64_t frames_drawn = 0, path_launches = 0;
    hipEvent_t ev_fence = nullptr;      // fence_lanes_after
    hipEvent_t ev_gather = nullptr;     // the previous rt_gather_gbuffer (gathers share staging: they run one after the other)
    bool gather_recorded = false;

    // wavefront pipeline state
    int kernel = RT_KERNEL_PERSISTENT;
    uint32_t batch_samples = 1;   // samples per batch
    uint32_t cap = 0;             // paths per batch (allocation)
    uint32_t refill_threshold = 24;
    float *qox = nullptr, *qoy = nullptr, *qoz = nullptr, *qdx = nullptr, *qdy = nullptr, *qdz = nullptr;
    uint32_t* qid = nullptr;
    float *hx = nullptr, *hy = nullptr, *hz = nullptr;
    uint32_t* hinfo = nullptr;
    uint8_t *sunres = nullptr, *pnormal = nullptr, *pstate = nullptr;
    float *pdx = nullptr, *pdy = nullptr, *pdz = nullptr, *plx = nullptr, *ply = nullptr, *plz = nullptr;
    uint32_t *sunbits = nullptr, *stack = nullptr;
    float4* acc = nullptr;
    uint32_t* ctrl = nullptr;     // per batch: [RT_MAX_DEPTH+2] pair counts, then [RT_MAX_DEPTH+2] cursors
    // persistent kernel state (per-launch and per-frame buffers live in the lanes and slots above)
    float4* sphere_lut = nullptr;
    float4* sun_lut = nullptr;
    float4* dif_lut = nullptr;
    uint32_t persist_batch = 1;
    int pl_stream_mode = -1;      // RT_PL_STREAM: -1 = by size (kStreamRecordBytes); bit 0 streaming stores, bit 1 streaming loads
    uint32_t persist_chunk = 0;   // RT_PERSIST_CHUNK: paths per cursor atomic (multiple of 64); 0 = automatic
    uint32_t persist_threshold = 0, persist_rmin = 12;   // threshold 0 = the kernel version's default
    int persist_version = 1;      // 1 = k_persist, 3 = k_paths (RT_KERNEL_PATHS)
    bool paths_by_size = false;   // RT_KERNEL_DEFAULT: k_paths for launches with enough work, k_persist for small ones
    int frame_mode = 0;           // k_frame: 0 = never, 1 = one-sample frames below frame_crossover pixels (RT_KERNEL_DEFAULT), 2 = every frame it covers (RT_KERNEL_FRAME)
    uint64_t frame_crossover = kFrameCrossover;
    uint64_t frame_crossover_multi = kFrameCrossoverMulti;   // ... and frames of more than one sample per pixel below this many pixel-sample-levels
    uint32_t frame_threshold = 0; // RT_FRAME_THRESHOLD: parked lanes per wave that trigger k_frame's pass; 0 = the default
    uint32_t frame_tiles = 0;     // RT_FRAME_TILES (per wave, 1..4) / RT_FRAME_GROUP_TILES (1..16): tiles per four-wave workgroup of k_frame; 0 = by frame size and spp
    int last_path_kernel = 0;     // RtKernel the most recent frame's path launches ran on (0 = no frame yet)
    float lut_key[6] = {0, 0, 0, 0, 0, 0};   // sun vector + colour the per-frame tables were built for
    bool lut_valid = false;
    int primary_version = 2;      // 1 = k_primary (thread per pixel), 2 = k_primary2 (nibble map in LDS); RT_PRIMARY_V
    rtd::DevCounters* d_counters = nullptr;
    unsigned long long* d_selftest = nullptr;
    uint64_t host_noise_base = 0, host_frames = 0;

    // timing
    hipEvent_t ev_frame0 = nullptr, ev_frame1 = nullptr;
    std::vector<hipEvent_t> ev_pool;
    std::vector<int> ev_kind;     // per (start,stop) pair: 0 = trace, 1 = other
    size_t ev_used = 0;
    bool frame_recorded = false;
    uint64_t rays_bound = 0;

    // multi-GPU gather (rt_gather_gbuffer): root's staging for the ranks' blocks; the overlapped mode's second stream,
    // two send-side staging copies of this rank's block and the events that order them
    void* frame_planes[6] = {};   // root, frames_dev == NULL: the library's own row-major full-frame planes (rt_frame_ptr)
    uint8_t* gathered[2] = {nullptr, nullptr};
    uint8_t* stage[2] = {nullptr, nullptr};
    hipStream_t gather_stream = nullptr;
    hipEvent_t ev_ready[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    bool ev_free_recorded[2] = {false, false};
    uint64_t gathers = 0;
    uint32_t timer_overflow = 0;   // launches that found the event pool full (RT_FLAG_TIMING without rt_get_timing)
    std::vector<hipEvent_t> gather_ev;   // RT_FLAG_TIMING: (start, stop) pairs round the rt_gather_gbuffer calls (rt_get_gather_timing)
    size_t gather_ev_used = 0;

    std::vector<void*> allocs;
    uint64_t device_bytes = 0;       // sum of the context's device allocations (rt_get_info)
    uint64_t light_budget_bytes = 0; // what the per-path light records were sized for
};

constexpr size_t kMaxTimerEvents = 2 * 4096;   // LaunchTimer pool cap: pairs beyond it are not timed (counted in timer_overflow)

namespace {

int fail(RtContext* c, int code, const std::string& msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}

#define RT_HIP(ctx, call)                                                                             \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(ctx, e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP,                     \
                        std::string(#call) + ": " + hipGetErrorString(e_));                           \
    } while (0)

template <typename T>
hipError_t dev_alloc(RtContext* c, T** p, size_t count) {
    *p = nullptr;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e == hipSuccess) { c->allocs.push_back((void*)*p); c->device_bytes += count * sizeof(T); }
    return e;
}

// RT_KERNEL_DEFAULT: is this context's frame small enough for k_frame (one launch per frame)?
bool frame_by_size(const RtContext* c) {
    if (c->cfg.spp == 1) return (uint64_t)c->npix_pad < c->frame_crossover;
    return (uint64_t)c->npix_pad * (uint64_t)c->cfg.spp * (uint64_t)(c->cfg.depth > 1 ? c->cfg.depth : 1) < c->frame_crossover_multi;
}

rtd::Scene scene_of(const RtContext* c) {
    rtd::Scene s;
    s.mine = c->d_mine_sw; s.mat = c->d_mat_sw; s.coarse = c->d_coarse; s.noise = c->d_noise;
    s.brick = c->d_brick ? reinterpret_cast<const uint8_t*>(c->d_brick) : reinterpret_cast<const uint8_t*>(c->d_coarse);
    return s;
}

rtd::Planes planes_of(void* const* planes) {
    rtd::Planes p;
    p.lighting_rgba16 = (uint16_t*)planes[RT_BUF_LIGHTING_RGBA16];
    p.depth_r16 = (uint16_t*)planes[RT_BUF_DEPTH_R16UI];
    p.normal_r8 = (uint8_t*)planes[RT_BUF_NORMAL_R8UI];
    p.albedo_rgba8 = (uint32_t*)planes[RT_BUF_ALBEDO_RGBA8];
    p.emission_rgba8 = (uint32_t*)planes[RT_BUF_EMISSION_RGBA8];
    p.fog_rgba8 = (uint32_t*)planes[RT_BUF_FOG_RGBA8];
    p.lighting_f32 = (float*)planes[RT_BUF_LIGHTING_F32];
    p.fog_f32 = (float*)planes[RT_BUF_FOG_F32];
    p.depth_f32 = (float*)planes[RT_BUF_DEPTH_F32];
    return p;
}
rtd::Planes planes_of(const RtContext* c) { return planes_of(c->planes); }

// raytrace.comp:317-318 — uniform over the frame, so evaluated once here with the same rt_math.h the kernels use.
void sun_constants(float a, float* sunangle, float* sunlight) {
    float s, c;
    rtm_sincos(a, &s, &c);
    rtm_vec3 v = rtm_normalize3({c * 0.5f + (a - 0.5f) * 0.5f, s, c});
    sunangle[0] = v.x; sunangle[1] = v.y; sunangle[2] = v.z;
    // sun_color, raytrace.comp:259-269
    float horizon = rtm_length2(v.x, v.y);
    float sun_amount = rtm_min(1.0f - horizon, 0.02f) * 50.0f;
    const float main_color[3] = {0.9647f * 2.0f, 0.7843f * 2.0f, 0.8824f * 2.0f};
    const float sunset_color[3] = {0.7412f * 2.0f, 0.2157f * 2.0f, 0.1686f * 2.0f};
    for (int k = 0; k < 3; k++)
        sunlight[k] = v.z >= 0.0f ? rtm_mix(sunset_color[k], main_color[k], sun_amount)
                                  : rtm_mix(sunset_color[k], 0.0f, sun_amount * 2);
}

rtd::Frame frame_of(const RtContext* c, const RtUniforms* u) {
    rtd::Frame f;
    for (int k = 0; k < 3; k++) {
        f.origin[k] = u->origin[k]; f.forward[k] = u->forward[k]; f.up[k] = u->up[k]; f.right[k] = u->right[k];
        f.lr[k] = (float)u->lr[k];    // vec3 current_rotation = uniform_data.lr (raytrace.comp:104)
    }
    sun_constants(u->sun_angle, f.sunangle, f.sunlight);
    f.seed = u->seed;
    f.width = c->cfg.width; f.height = c->cfg.height;
    f.tiles_x = c->tiles_x; f.tiles_y = c->tiles_y;
    f.tile_rank = c->cfg.tile_rank; f.tile_world = c->cfg.tile_world;
    f.ntiles_local = c->ntiles_local;
    f.spp = c->cfg.spp; f.depth = c->cfg.depth;
    f.lr_zero = (u->lr[0] == 0 && u->lr[1] == 0 && u->lr[2] == 0) ? 1 : 0;
    f.logr = c->logr;
    f.region = (float)c->region;
    return f;
}

// Event pair bracketing one launch (only with RT_FLAG_TIMING), recorded on the stream the launch goes to.
struct LaunchTimer {
    RtContext* c; bool on; size_t idx; hipStream_t st;
    // kind 0 = traversal kernel (RT_FLAG_TIMING), 1 = any other launch (RT_FLAG_TIMING_ALL: each pair of events costs a few us)
    LaunchTimer(RtContext* ctx, int kind, hipStream_t stream = nullptr)
        : c(ctx), on(kind == 0 ? (ctx->cfg.flags & RT_FLAG_TIMING) != 0 : (ctx->cfg.flags & RT_FLAG_TIMING_ALL) == RT_FLAG_TIMING_ALL), idx(0),
          st(stream ? stream : ctx->stream) {
        if (!on) return;
        if (c->ev_used + 2 > kMaxTimerEvents) { c->timer_overflow++; on = false; return; }
        if (c->ev_used + 2 > c->ev_pool.size()) {
            for (int k = 0; k < 2; k++) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; return; } c->ev_pool.push_back(e); }
        }
        idx = c->ev_used; c->ev_used += 2;
        if (c->ev_kind.size() < c->ev_used / 2) c->ev_kind.resize(c->ev_used / 2);
        c->ev_kind[idx / 2] = kind;
        (void)hipEventRecord(c->ev_pool[idx], st);
    }
    ~LaunchTimer() { if (on) (void)hipEventRecord(c->ev_pool[idx + 1], st); }
};

// ---- lanes (round 4): ordering between the library's streams, all on the device -------------------------------------------
// `st` waits for everything submitted so far on the other lanes (and on the stream the last frame ended on)
hipError_t join_lanes_into(RtContext* c, hipStream_t st) {
    for (int l = 0; l < c->nlanes; l++) {
        Lane& ln = c->lanes[l];
        if (ln.stream == st || !ln.stream) continue;
        hipError_t e = hipEventRecord(ln.ev_join, ln.stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(st, ln.ev_join, 0);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
// ... and the other lanes wait for what `st` holds now
hipError_t fence_lanes_after(RtContext* c, hipStream_t st) {
    if (c->nlanes < 2) return hipSuccess;
    hipError_t e = hipEventRecord(c->ev_fence, st);
    for (int l = 0; l < c->nlanes && e == hipSuccess; l++)
        if (c->lanes[l].stream != st) e = hipStreamWaitEvent(c->lanes[l].stream, c->ev_fence, 0);
    return e;
}
// host-side wait for every stream of the context that renders
hipError_t sync_lanes(RtContext* c) {
    hipError_t e = hipStreamSynchronize(c->stream);
    for (int l = 0; l < c->nlanes && e == hipSuccess; l++)
        if (c->lanes[l].stream && c->lanes[l].stream != c->stream) e = hipStreamSynchronize(c->lanes[l].stream);
    return e;
}

int reflatten(RtContext* c, const uint8_t* d_mine_lin, const uint32_t* d_mat_lin) {
    RT_HIP(c, hipMemsetAsync(c->d_flag, 0, sizeof(uint32_t), c->stream));
    RT_HIP(c, rtd::launch_flatten(d_mine_lin, d_mat_lin, c->d_mine_sw, c->d_mat_sw, c->d_coarse, c->d_brick, c->d_flag, c->logr, c->stream));
    uint32_t flag = 0;
    RT_HIP(c, hipMemcpyAsync(&flag, c->d_flag, sizeof(flag), hipMemcpyDeviceToHost, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    if (flag) return fail(c, RT_ERR_INVALID_ARG, "minefield holds a value above 30 (the reference writes 0..6, src/world/chunk.rs:163-183)");
    return RT_OK;
}

int draw_wavefront(RtContext* c, const rtd::Frame& f) {
    const bool count = (c->cfg.flags & RT_FLAG_COUNTERS) != 0;
    const rtd::Scene sc = scene_of(c);
    const rtd::Planes pl = planes_of(c);
    const int D = c->cfg.depth;
    const uint32_t spp = (uint32_t)c->cfg.spp;
    const uint32_t B = c->batch_samples;
    const uint32_t ctrl_words = 2 * (RT_MAX_DEPTH + 2);
    uint32_t batch = 0;
    for (uint32_t s0 = 0; s0 < spp; s0 += B, batch++) {
        const uint32_t ns = spp - s0 < B ? spp - s0 : B;
        const uint32_t npaths = c->npix_pad * ns;
        uint32_t* counts = c->ctrl;
        uint32_t* cursors = c->ctrl + (RT_MAX_DEPTH + 2);
        RT_HIP(c, hipMemsetAsync(c->ctrl, 0, ctrl_words * sizeof(uint32_t), c->stream));

        rtd::TraceArgs ta{};
        ta.qox = c->qox; ta.qoy = c->qoy; ta.qoz = c->qoz; ta.qdx = c->qdx; ta.qdy = c->qdy; ta.qdz = c->qdz; ta.qid = c->qid;
        ta.qcap = c->cap; ta.nprimary = npaths; ta.npix_pad = c->npix_pad; ta.refill_threshold = c->refill_threshold;
        ta.hx = c->hx; ta.hy = c->hy; ta.hz = c->hz; ta.hinfo = c->hinfo; ta.sunres = c->sunres; ta.counters = c->d_counters;

        rtd::ShadeArgs sa{};
        sa.qox = c->qox; sa.qoy = c->qoy; sa.qoz = c->qoz; sa.qdx = c->qdx; sa.qdy = c->qdy; sa.qdz = c->qdz; sa.qid = c->qid;
        sa.qcap = c->cap; sa.npaths = npaths; sa.npaths_cap = c->cap; sa.npix_pad = c->npix_pad; sa.sample0 = s0;
        sa.hx = c->hx; sa.hy = c->hy; sa.hz = c->hz; sa.hinfo = c->hinfo; sa.sunres = c->sunres;
        sa.pdx = c->pdx; sa.pdy = c->pdy; sa.pdz = c->pdz; sa.pnormal = c->pnormal; sa.pstate = c->pstate;
        sa.sunbits = c->sunbits; sa.stack = c->stack; sa.plx = c->plx; sa.ply = c->ply; sa.plz = c->plz;
        sa.counters = c->d_counters;

        // primary wave
        ta.qcount = counts + 0; ta.cursor = cursors + 0;
        { LaunchTimer t(c, 0); RT_HIP(c, rtd::launch_trace(sc, f, ta, true, count, c->num_cus, c->stream)); }
        sa.qcount_next = counts + 1;
        { LaunchTimer t(c, 1); RT_HIP(c, rtd::launch_shade0(sc, f, sa, pl, count, c->stream)); }
        for (int level = 1; level <= D; level++) {
            ta.qcount = counts + level; ta.cursor = cursors + level;
            { LaunchTimer t(c, 0); RT_HIP(c, rtd::launch_trace(sc, f, ta, false, count, c->num_cus, c->stream)); }
            sa.qcount_next = counts + level + 1;
            { LaunchTimer t(c, 1); RT_HIP(c, rtd::launch_shadeN(sc, f, sa, level, count, c->stream)); }
        }
        { LaunchTimer t(c, 1); RT_HIP(c, rtd::launch_accumulate(c->plx, c->ply, c->plz, c->acc, c->npix_pad, ns, batch == 0, c->stream)); }
    }
    { LaunchTimer t(c, 1); RT_HIP(c, rtd::launch_resolve(f, c->acc, pl, c->npix_pad, c->stream)); }
    return RT_OK;
}

}  // namespace

// Light-record budget of a context (all lanes together): RT_PERSIST_LIGHT_GIB, else min(kDefaultLightBytes, a tenth of the free
// device memory): a context is one tenant of the GPU.  Pure function of its inputs apart from hipMemGetInfo.
constexpr uint64_t kDefaultLightBytes = 16ull << 30;
uint64_t rt_light_budget_bytes(const char* env_gib) {
    uint64_t light_bytes = kDefaultLightBytes;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) { if ((uint64_t)free_b / 10u < light_bytes) light_bytes = (uint64_t)free_b / 10u; }
    else (void)hipGetLastError();
    if (env_gib) { long long v = atoll(env_gib); if (v >= 1 && v <= 128) light_bytes = (uint64_t)v << 30; }
    return light_bytes;
}
// Samples of every pixel one launch covers: what `lane_bytes` of 12-byte records hold for `npix` worklist slots, at most 2^31
// paths (path indices are 32-bit), at most spp, at least 1; RT_PERSIST_BATCH may only lower it.  (tests/test_abi.py pins it.)
extern "C" uint64_t rt_samples_per_launch(uint64_t lane_bytes, uint64_t npix, uint64_t spp, const char* env_batch) {
    if (npix == 0) npix = 1;
    uint64_t B = lane_bytes / (sizeof(rtd::PathLight) * npix);
    if (B > (1ull << 31) / npix) B = (1ull << 31) / npix;
    if (env_batch) { long long v = atoll(env_batch); if (v > 0 && (uint64_t)v < B) B = (uint64_t)v; }
    if (B < 1) B = 1;
    if (B > spp) B = spp;
    return B;
}

extern "C" {

uint32_t rt_abi_version(void) { return ((uint32_t)RT_ABI_VERSION_MAJOR << 16) | (uint32_t)RT_ABI_VERSION_MINOR; }

const char* rt_last_error(RtContext* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int rt_create(const RtConfig* cfg, RtContext** out) {
    if (out) *out = nullptr;
    if (!cfg || !out) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: null argument");
    if (cfg->struct_size != sizeof(RtConfig)) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: RtConfig.struct_size mismatch");
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->width > 16384 || cfg->height > 16384)
        return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: width/height out of range");
    if (cfg->region != 256 && cfg->region != 512 && cfg->region != 1024)
        return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: region must be 256 (the reference), 512 or 1024");
    if (cfg->region != 256 && cfg->kernel == RT_KERNEL_WAVEFRONT)
        return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: the split wavefront baseline supports region 256 only");
    if (cfg->spp < 1 || cfg->spp > RT_NOISE_BYTES) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: spp out of range");
    if (cfg->depth < 0 || cfg->depth > RT_MAX_DEPTH) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: depth out of range");
    if (cfg->tile_world < 1 || cfg->tile_rank < 0 || cfg->tile_rank >= cfg->tile_world)
        return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: bad tile_rank/tile_world");
    if (cfg->kernel < RT_KERNEL_DEFAULT || cfg->kernel > RT_KERNEL_FRAME || cfg->kernel == 4 /* RT_KERNEL_PERSISTENT2, retired in round 3 */ ||
        cfg->kernel == 6 /* RT_KERNEL_SEQ, retired in round 4 */)
        return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: unknown kernel");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, std::string("rt_create: no HIP device (") + hipGetErrorString(e) + ")");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_create: device ordinal out of range");

    RtContext* c = new (std::nothrow) RtContext();
    if (!c) return fail(nullptr, RT_ERR_OOM, "rt_create: host allocation failed");
    auto bail = [&](int code, const std::string& msg) { g_create_error = msg; rt_destroy(c); return code; };
#define RT_HIP_CREATE(call)                                                                                        \
    do { hipError_t e_ = (call); if (e_ != hipSuccess)                                                             \
        return bail(e_ == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); } while (0)

    c->cfg = *cfg;
    c->device = cfg->device;
    c->region = cfg->region;
    c->logr = cfg->region == 256 ? 8 : (cfg->region == 512 ? 9 : 10);
    c->vox = (size_t)cfg->region * cfg->region * cfg->region;
    {
        // RT_KERNEL_DEFAULT = k_paths (two paths, four ray slots per lane, branch-free step loop); the frames it does not
        // cover run on k_persist (rt_draw_frame decides per frame: lr is a per-frame uniform)
        c->paths_by_size = cfg->kernel == RT_KERNEL_DEFAULT;
        // k_frame (one launch per frame) takes the small frames of RT_KERNEL_DEFAULT and whatever it covers of RT_KERNEL_FRAME; the
        // rest of those contexts' frames go the persistent way, so they hold its buffers as well
        c->frame_mode = cfg->kernel == RT_KERNEL_FRAME ? 2 : (cfg->kernel == RT_KERNEL_DEFAULT ? 1 : 0);
        if (const char* s = getenv("RT_FRAME_CROSSOVER")) { long long v = atoll(s); if (v >= 0) c->frame_crossover = (uint64_t)v; }
        if (const char* s = getenv("RT_FRAME_CROSSOVER_MULTI")) { long long v = atoll(s); if (v >= 0) c->frame_crossover_multi = (uint64_t)v; }
        if (const char* s = getenv("RT_FRAME_THRESHOLD")) { int v = atoi(s); if (v >= 1 && v <= 64) c->frame_threshold = (uint32_t)v; }
        if (const char* s = getenv("RT_FRAME_TILES")) { int v = atoi(s); if (v >= 1 && v <= 4) c->frame_tiles = 4u * (uint32_t)v; }           // per wave
        if (const char* s = getenv("RT_FRAME_GROUP_TILES")) { int v = atoi(s); if (v >= 1 && v <= 16) c->frame_tiles = (uint32_t)v; }     // per four-wave workgroup
        c->kernel = (cfg->kernel == RT_KERNEL_DEFAULT || cfg->kernel == RT_KERNEL_FRAME) ? RT_KERNEL_PATHS : cfg->kernel;
        if (c->kernel == RT_KERNEL_PATHS) { c->kernel = RT_KERNEL_PERSISTENT; c->persist_version = 3; }
    }
    RT_HIP_CREATE(hipSetDevice(c->device));
    hipDeviceProp_t prop;
    RT_HIP_CREATE(hipGetDeviceProperties(&prop, c->device));
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // RT_RESERVE_CUS=n: the persistent kernels (one workgroup per CU) leave n CUs free, e.g. for an RCCL kernel running
    // beside them (bench.py's overlapped gather)
    if (const char* s = getenv("RT_RESERVE_CUS")) { int v = atoi(s); if (v > 0 && v < c->num_cus) c->num_cus -= v; }
    RT_HIP_CREATE(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    RT_HIP_CREATE(hipEventCreate(&c->ev_frame0));
    RT_HIP_CREATE(hipEventCreate(&c->ev_frame1));

    // tiling: 8x8-pixel tiles dealt round-robin over tile_world contexts
    c->tiles_x = (cfg->width + 7) / 8; c->tiles_y = (cfg->height + 7) / 8;
    c->ntiles_total = c->tiles_x * c->tiles_y;
    c->tile_capacity = (c->ntiles_total + cfg->tile_world - 1) / cfg->tile_world;
    c->ntiles_local = (c->ntiles_total - cfg->tile_rank + cfg->tile_world - 1) / cfg->tile_world;
    if (c->ntiles_local < 0) c->ntiles_local = 0;
    c->npix_pad = (uint32_t)c->ntiles_local * 64u;
    c->plane_pixels = cfg->tile_world == 1 ? (size_t)cfg->width * cfg->height : (size_t)c->tile_capacity * 64;

    // scene
    RT_HIP_CREATE(dev_alloc(c, &c->d_mine_sw, c->vox)); RT_HIP_CREATE(dev_alloc(c, &c->d_mat_sw, c->vox));
    RT_HIP_CREATE(dev_alloc(c, &c->d_coarse, (size_t)rtd::kCoarseWords));
    // (the per-brick map of a -DRT_PATHS_BRICK_MAP=1 build of k_paths: an experiment that lost, profiles/r4_c5_brick_map.txt)
    if (c->logr > 8 && getenv("RT_BRICK_MAP")) RT_HIP_CREATE(dev_alloc(c, &c->d_brick, c->vox / 128u / 4u));
    RT_HIP_CREATE(dev_alloc(c, &c->d_noise, (size_t)RT_NOISE_SIZE * RT_NOISE_SIZE));
    RT_HIP_CREATE(dev_alloc(c, &c->d_flag, 4));
    RT_HIP_CREATE(dev_alloc(c, &c->d_counters, 1));
    RT_HIP_CREATE(hipMemset(c->d_counters, 0, sizeof(rtd::DevCounters)));

    // lanes and frame slots (see Lane / FrameSlot).  Two lanes for the persistent kernels; two slots when the host asks for two
    // frames in flight.  RT_LANES=1 / RT_FRAMES_IN_FLIGHT=1|2 override (experiments, A/B timing).
    const bool persistent = cfg->kernel == RT_KERNEL_DEFAULT || cfg->kernel == RT_KERNEL_PERSISTENT || cfg->kernel == RT_KERNEL_PATHS || cfg->kernel == RT_KERNEL_FRAME;
    c->nlanes = persistent ? 2 : 1;
    c->nslots = (persistent && (cfg->flags & RT_FLAG_FRAMES_IN_FLIGHT_2)) ? 2 : 1;
    if (const char* s = getenv("RT_LANES")) { int v = atoi(s); if (v == 1 || (v == 2 && persistent)) c->nlanes = v; }
    if (const char* s = getenv("RT_FRAMES_IN_FLIGHT")) { int v = atoi(s); if (v == 1 || (v == 2 && persistent)) c->nslots = v; }
    if (c->nlanes < 2) c->nslots = 1;
    c->lanes[0].stream = c->own_stream;
    if (c->nlanes == 2) RT_HIP_CREATE(hipStreamCreateWithFlags(&c->lanes[1].stream, hipStreamNonBlocking));
    for (int l = 0; l < c->nlanes; l++) RT_HIP_CREATE(hipEventCreateWithFlags(&c->lanes[l].ev_join, hipEventDisableTiming));
    RT_HIP_CREATE(hipEventCreateWithFlags(&c->ev_fence, hipEventDisableTiming));
    RT_HIP_CREATE(hipEventCreateWithFlags(&c->ev_gather, hipEventDisableTiming));

    {   // the six reference-format planes of a slot live in ONE block (each padded to 256 B) so a multi-GPU host can gather them
        // with a single collective; the other planes are separate allocations
        size_t off = 0;
        for (int b = 0; b <= RT_BUF_FOG_RGBA8; b++) {
            c->gbuffer_offset[b] = off;
            off += (c->plane_pixels * kBytesPerPixel[b] + 255) / 256 * 256;
        }
        c->gbuffer_bytes = off;
        for (int sl = 0; sl < c->nslots; sl++) {
            FrameSlot& fs = c->slots[sl];
            uint8_t* block = nullptr;
            RT_HIP_CREATE(dev_alloc(c, &block, off));
            RT_HIP_CREATE(hipMemset(block, 0, off));
            fs.gbuffer = block;
            for (int b = 0; b <= RT_BUF_FOG_RGBA8; b++) fs.planes[b] = block + c->gbuffer_offset[b];
            for (int b = RT_BUF_FOG_RGBA8 + 1; b < RT_BUF_COUNT; b++) {
                uint8_t* p = nullptr;
                RT_HIP_CREATE(dev_alloc(c, &p, c->plane_pixels * kBytesPerPixel[b]));
                RT_HIP_CREATE(hipMemset(p, 0, c->plane_pixels * kBytesPerPixel[b]));
                fs.planes[b] = p;
            }
            RT_HIP_CREATE(hipEventCreateWithFlags(&fs.ev_prepass, hipEventDisableTiming));
            RT_HIP_CREATE(hipEventCreateWithFlags(&fs.ev_acc, hipEventDisableTiming));
            RT_HIP_CREATE(hipEventCreateWithFlags(&fs.ev_tail, hipEventDisableTiming));
        }
        c->cur_slot = 0;
        c->gbuffer = c->slots[0].gbuffer;
        for (int b = 0; b < RT_BUF_COUNT; b++) c->planes[b] = c->slots[0].planes[b];
    }

    if (const char* s = getenv("RT_REFILL_THRESHOLD")) { int v = atoi(s); if (v >= 1 && v <= 64) c->refill_threshold = (uint32_t)v; }
    if (const char* s = getenv("RT_PRIMARY_V")) { int v = atoi(s); if (v == 1 || v == 2) c->primary_version = v; }
    if (const char* s = getenv("RT_PERSIST_THRESHOLD")) { int v = atoi(s); if (v >= 1 && v <= 64) c->persist_threshold = (uint32_t)v; }
    if (const char* s = getenv("RT_PL_STREAM")) { int v = atoi(s); if (v >= 0 && v <= 3) c->pl_stream_mode = v; }
    if (const char* s = getenv("RT_PERSIST_CHUNK")) { int v = atoi(s); if (v >= 64 && v <= 4096) c->persist_chunk = (uint32_t)v & ~63u; }
    if (const char* s = getenv("RT_PERSIST_RMIN")) { int v = atoi(s); if (v >= 1 && v <= 128) c->persist_rmin = (uint32_t)v; }
    if (c->kernel == RT_KERNEL_PERSISTENT) {
        const size_t stack_words = (size_t)4 * c->num_cus * 1024 * (size_t)(cfg->depth > 1 ? cfg->depth - 1 : 1);   // up to 3 paths per lane
        for (int l = 0; l < c->nlanes; l++) {
            Lane& ln = c->lanes[l];
            RT_HIP_CREATE(dev_alloc(c, &ln.cursor, kCursorWords));
            RT_HIP_CREATE(hipMemset(ln.cursor, 0, kCursorWords * sizeof(uint32_t)));
            ln.cursor_clean = true;
            RT_HIP_CREATE(dev_alloc(c, &ln.pstack, stack_words));
        }
        for (int sl = 0; sl < c->nslots; sl++) {
            FrameSlot& fs = c->slots[sl];
            RT_HIP_CREATE(dev_alloc(c, &fs.wl_count, (size_t)64));   // two counters, 128 bytes apart
            RT_HIP_CREATE(hipMemset(fs.wl_count, 0, 64 * sizeof(uint32_t)));
            fs.wl_clean[0] = fs.wl_clean[1] = true;
            RT_HIP_CREATE(dev_alloc(c, &fs.worklist, (size_t)c->npix_pad));
            RT_HIP_CREATE(dev_alloc(c, &fs.phit, (size_t)c->npix_pad));
            RT_HIP_CREATE(dev_alloc(c, &fs.pacc, (size_t)c->npix_pad));
        }
        RT_HIP_CREATE(dev_alloc(c, &c->sphere_lut, (size_t)65536));
        RT_HIP_CREATE(dev_alloc(c, &c->sun_lut, (size_t)2 * 65536));
        RT_HIP_CREATE(dev_alloc(c, &c->dif_lut, (size_t)4 * 6 * 65536));
        {   // Samples per path-kernel launch: bounded by 2^31 work items and by the memory given to the per-path light records, which
            // is split between the lanes (a launch's records live until its accumulate launch has read them, and two launches are in
            // flight).  Round 3 sized one launch for 16 GiB because every launch paid its ramp-up and its drain (3840x2160 spp 256
            // depth 8: 114.1 ms per frame with 1 GiB, 109.3 with 2, 107.0 with 4, 105.6 with 8 and beyond); with the next launch
            // taking the CUs the draining one frees, a launch's size matters far less (see kDefaultLightBytes).  Halved until the
            // allocation succeeds.
            uint64_t np = c->npix_pad ? c->npix_pad : 1;
            c->light_budget_bytes = rt_light_budget_bytes(getenv("RT_PERSIST_LIGHT_GIB"));
            uint64_t B = rt_samples_per_launch(c->light_budget_bytes / (uint64_t)c->nlanes, np, (uint64_t)cfg->spp, getenv("RT_PERSIST_BATCH"));
            for (int l = 0; l < c->nlanes; l++) {
                for (;;) {
                    const hipError_t e = dev_alloc(c, &c->lanes[l].ppl, (size_t)np * B);
                    if (e == hipSuccess) break;
                    if (e != hipErrorOutOfMemory || B == 1 || l > 0) RT_HIP_CREATE(e);   // (the lanes' buffers have one size)
                    (void)hipGetLastError();
                    B = (B + 1) / 2;
                }
            }
            c->persist_batch = (uint32_t)B;
        }
        RT_HIP_CREATE(rtd::launch_sphere_lut(c->sphere_lut, c->own_stream));
        RT_HIP_CREATE(rtd::launch_dif_lut(c->sphere_lut, c->dif_lut, c->own_stream));
        RT_HIP_CREATE(hipStreamSynchronize(c->own_stream));
    }
    if (c->kernel == RT_KERNEL_WAVEFRONT) {
        uint64_t target = 4u << 20;   // paths per batch
        if (const char* s = getenv("RT_BATCH_PATHS")) { long long v = atoll(s); if (v > 0) target = (uint64_t)v; }
        uint64_t B = c->npix_pad ? target / c->npix_pad : 1;
        if (B < 1) B = 1;
        if (B > (uint64_t)cfg->spp) B = (uint64_t)cfg->spp;
        c->batch_samples = (uint32_t)B;
        uint64_t cap64 = (uint64_t)c->npix_pad * B;
        if (cap64 >= (1ull << 31)) return bail(RT_ERR_INVALID_ARG, "rt_create: batch too large");
        c->cap = (uint32_t)cap64;
        const size_t cap = c->cap;
        RT_HIP_CREATE(dev_alloc(c, &c->qox, cap)); RT_HIP_CREATE(dev_alloc(c, &c->qoy, cap)); RT_HIP_CREATE(dev_alloc(c, &c->qoz, cap));
        RT_HIP_CREATE(dev_alloc(c, &c->qdx, 2 * cap)); RT_HIP_CREATE(dev_alloc(c, &c->qdy, 2 * cap)); RT_HIP_CREATE(dev_alloc(c, &c->qdz, 2 * cap));
        RT_HIP_CREATE(dev_alloc(c, &c->qid, cap));
        RT_HIP_CREATE(dev_alloc(c, &c->hx, cap)); RT_HIP_CREATE(dev_alloc(c, &c->hy, cap)); RT_HIP_CREATE(dev_alloc(c, &c->hz, cap));
        RT_HIP_CREATE(dev_alloc(c, &c->hinfo, cap));
        RT_HIP_CREATE(dev_alloc(c, &c->sunres, cap)); RT_HIP_CREATE(dev_alloc(c, &c->pnormal, cap)); RT_HIP_CREATE(dev_alloc(c, &c->pstate, cap));
        RT_HIP_CREATE(dev_alloc(c, &c->pdx, cap)); RT_HIP_CREATE(dev_alloc(c, &c->pdy, cap)); RT_HIP_CREATE(dev_alloc(c, &c->pdz, cap));
        RT_HIP_CREATE(dev_alloc(c, &c->plx, cap)); RT_HIP_CREATE(dev_alloc(c, &c->ply, cap)); RT_HIP_CREATE(dev_alloc(c, &c->plz, cap));
        RT_HIP_CREATE(dev_alloc(c, &c->sunbits, cap));
        RT_HIP_CREATE(dev_alloc(c, &c->stack, cap * (size_t)(cfg->depth > 1 ? cfg->depth - 1 : 1)));
        RT_HIP_CREATE(dev_alloc(c, &c->acc, (size_t)c->npix_pad));
        RT_HIP_CREATE(dev_alloc(c, &c->ctrl, (size_t)2 * (RT_MAX_DEPTH + 2)));
    }
#undef RT_HIP_CREATE
    *out = c;
    return RT_OK;
}

void rt_destroy(RtContext* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->stream && ctx->stream != ctx->own_stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->lanes[1].stream) { (void)hipStreamSynchronize(ctx->lanes[1].stream); (void)hipStreamDestroy(ctx->lanes[1].stream); }
    for (Lane& ln : ctx->lanes) if (ln.ev_join) (void)hipEventDestroy(ln.ev_join);
    for (FrameSlot& fs : ctx->slots) for (hipEvent_t e : {fs.ev_prepass, fs.ev_acc, fs.ev_tail}) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {ctx->ev_fence, ctx->ev_gather}) if (e) (void)hipEventDestroy(e);
    if (ctx->upload_stream) { (void)hipStreamSynchronize(ctx->upload_stream); (void)hipStreamDestroy(ctx->upload_stream); }
    for (int k = 0; k < 2; k++) {
        if (ctx->ev_slab_copied[k]) (void)hipEventDestroy(ctx->ev_slab_copied[k]);
        if (ctx->ev_slab_applied[k]) (void)hipEventDestroy(ctx->ev_slab_applied[k]);
        if (ctx->h_slab_mat[k]) (void)hipHostFree(ctx->h_slab_mat[k]);
        if (ctx->h_slab_mine[k]) (void)hipHostFree(ctx->h_slab_mine[k]);
    }
    for (void* p : ctx->allocs) (void)hipFree(p);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (hipEvent_t e : ctx->gather_ev) (void)hipEventDestroy(e);
    if (ctx->ev_frame0) (void)hipEventDestroy(ctx->ev_frame0);
    if (ctx->ev_frame1) (void)hipEventDestroy(ctx->ev_frame1);
    if (ctx->gather_stream) { (void)hipStreamSynchronize(ctx->gather_stream); (void)hipStreamDestroy(ctx->gather_stream); }
    for (int i = 0; i < 2; i++) { if (ctx->ev_ready[i]) (void)hipEventDestroy(ctx->ev_ready[i]); if (ctx->ev_free[i]) (void)hipEventDestroy(ctx->ev_free[i]); }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
}

int rt_set_stream(RtContext* ctx, void* hip_stream) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, sync_lanes(ctx));
    // NULL selects the context's own (non-blocking) stream, NOT the legacy null stream: work a caller enqueues on the null
    // stream is not ordered against the context's frames — pass an explicit stream handle to share one.
    // On a caller's stream EVERYTHING the context does runs on that stream, in order: no second lane, one frame slot (the
    // caller's own work on the stream is ordered against the frames by the stream alone).
    ctx->stream = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
    ctx->lanes[0].stream = ctx->stream;
    ctx->user_stream = hip_stream != nullptr;
    return RT_OK;
}

int rt_upload_world(RtContext* ctx, const uint32_t* materials, const uint8_t* minefield) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!materials || !minefield) return fail(ctx, RT_ERR_INVALID_ARG, "rt_upload_world: null pointer");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, sync_lanes(ctx));
    // the caller-layout copy lives for the duration of this call only (it used to stay resident — 160 MiB at
    // R = 256, 5 GiB at R = 1024 — although nothing but this function reads it)
    uint32_t* d_mat_lin = nullptr; uint8_t* d_mine_lin = nullptr;
    hipError_t e = hipMalloc((void**)&d_mat_lin, ctx->vox * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMalloc((void**)&d_mine_lin, ctx->vox);
    if (e == hipSuccess) e = hipMemcpy(d_mat_lin, materials, ctx->vox * sizeof(uint32_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_mine_lin, minefield, ctx->vox, hipMemcpyHostToDevice);
    int rc = RT_OK;
    if (e != hipSuccess) rc = fail(ctx, e == hipErrorOutOfMemory ? RT_ERR_OOM : RT_ERR_HIP, std::string("rt_upload_world: ") + hipGetErrorString(e));
    else {
        ctx->has_world = false;
        ctx->world_resident = false;
        rc = reflatten(ctx, d_mine_lin, d_mat_lin);      // synchronises the stream before it returns
    }
    if (d_mat_lin) (void)hipFree(d_mat_lin);
    if (d_mine_lin) (void)hipFree(d_mine_lin);
    if (rc != RT_OK) return rc;
    ctx->has_world = true;
    ctx->world_resident = true;
    return RT_OK;
}

namespace {
// staging of rt_upload_slice, created on first use: pinned host memory for one slab (the reference's upload buffers are
// host-visible mapped Vulkan buffers, terrain_upload.rs:65-82), its device twin, a stream for the transfer and two events
int slab_resources(RtContext* ctx) {
    const size_t n = (size_t)RT_SLICE_SIZE * (size_t)ctx->region * (size_t)ctx->region;
    for (int k = 0; k < 2; k++) {
        if (!ctx->d_slab_mat[k]) RT_HIP(ctx, dev_alloc(ctx, &ctx->d_slab_mat[k], n));
        if (!ctx->d_slab_mine[k]) RT_HIP(ctx, dev_alloc(ctx, &ctx->d_slab_mine[k], n));
        if (!ctx->h_slab_mat[k]) RT_HIP(ctx, hipHostMalloc((void**)&ctx->h_slab_mat[k], n * sizeof(uint32_t), hipHostMallocDefault));
        if (!ctx->h_slab_mine[k]) RT_HIP(ctx, hipHostMalloc((void**)&ctx->h_slab_mine[k], n, hipHostMallocDefault));
        if (!ctx->ev_slab_copied[k]) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_slab_copied[k], hipEventDisableTiming));
        if (!ctx->ev_slab_applied[k]) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_slab_applied[k], hipEventDisableTiming));
    }
    if (!ctx->upload_stream) RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->upload_stream, hipStreamNonBlocking));
    return RT_OK;
}
}  // namespace

int rt_slice_staging(RtContext* ctx, uint32_t** materials, uint8_t** minefield) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!materials || !minefield) return fail(ctx, RT_ERR_INVALID_ARG, "rt_slice_staging: null pointer");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = slab_resources(ctx);
    if (rc != RT_OK) return rc;
    const int k = (int)(ctx->slabs & 1u);   // the set the NEXT rt_upload_slice uses
    if (ctx->slab_copy_pending[k]) { RT_HIP(ctx, hipEventSynchronize(ctx->ev_slab_copied[k])); ctx->slab_copy_pending[k] = false; }   // the slab before last has left it
    *materials = ctx->h_slab_mat[k]; *minefield = ctx->h_slab_mine[k];
    return RT_OK;
}

int rt_upload_slice(RtContext* ctx, int axis, int texel_offset, const uint32_t* materials, const uint8_t* minefield) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!materials || !minefield) return fail(ctx, RT_ERR_INVALID_ARG, "rt_upload_slice: null pointer");
    const size_t kR = (size_t)ctx->region;
    if (axis < 0 || axis > 2 || texel_offset < 0 || texel_offset + RT_SLICE_SIZE > ctx->region || texel_offset % RT_SLICE_SIZE != 0)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_upload_slice: bad axis or offset");
    if (!ctx->world_resident) return fail(ctx, RT_ERR_NOT_READY, "rt_upload_slice: upload the full region first");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    int rc = slab_resources(ctx);
    if (rc != RT_OK) return rc;
    // Incremental and asynchronous.  The slab goes pinned host staging -> device staging (5 bytes x 16 x R^2) on the upload
    // stream; the render stream waits for that event (not the host), then ONE launch re-tiles the slab's 16 R^2 voxels into the
    // brick-swizzled arrays and a second rebuilds the nibble-map words it touches.  Frames in flight keep reading the region
    // until then by stream order — the host waits for nothing on the device (the reference blocks on vkQueueWaitIdle here,
    // pipeline.rs:181-189); only a previous slab still leaving the pinned staging is waited for.
    const size_t n = (size_t)RT_SLICE_SIZE * kR * kR;
    const int k = (int)(ctx->slabs & 1u);
    if (ctx->slab_copy_pending[k]) { RT_HIP(ctx, hipEventSynchronize(ctx->ev_slab_copied[k])); ctx->slab_copy_pending[k] = false; }
    if ((ctx->cfg.flags & RT_FLAG_TRUSTED_WORLD) == 0) {
        // values above 30 are rejected as in rt_upload_world — BEFORE anything is written or transferred: a rejected slab leaves
        // the region (and what can be drawn) as it was.  The bytes are in host memory, so the check is a host loop over 16 R^2
        // bytes (~20 us at R = 256) and needs no device round trip; a host that vouches for its data skips it.
        uint8_t worst = 0;
        for (size_t i = 0; i < n; i++) worst = minefield[i] > worst ? minefield[i] : worst;
        if (worst > rtd::kMaxStepValue)
            return fail(ctx, RT_ERR_INVALID_ARG, "minefield slab holds a value above 30 (the reference writes 0..6, src/world/chunk.rs:163-183); the region is unchanged");
    }
    if (materials != ctx->h_slab_mat[k]) memcpy(ctx->h_slab_mat[k], materials, n * sizeof(uint32_t));   // borrowed buffers are released at return;
    if (minefield != ctx->h_slab_mine[k]) memcpy(ctx->h_slab_mine[k], minefield, n);                     // rt_slice_staging's pointers need no copy
    if (ctx->slab_apply_recorded[k]) RT_HIP(ctx, hipStreamWaitEvent(ctx->upload_stream, ctx->ev_slab_applied[k], 0));   // the re-tile of the slab before last has read this device staging
    RT_HIP(ctx, hipMemcpyAsync(ctx->d_slab_mine[k], ctx->h_slab_mine[k], n, hipMemcpyHostToDevice, ctx->upload_stream));
    RT_HIP(ctx, hipMemcpyAsync(ctx->d_slab_mat[k], ctx->h_slab_mat[k], n * sizeof(uint32_t), hipMemcpyHostToDevice, ctx->upload_stream));
    RT_HIP(ctx, hipEventRecord(ctx->ev_slab_copied[k], ctx->upload_stream));
    ctx->slab_copy_pending[k] = true;
    ctx->slabs++;
    RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_slab_copied[k], 0));
    RT_HIP(ctx, join_lanes_into(ctx, ctx->stream));   // frames in flight on the other lane read the region too
    {
        LaunchTimer t(ctx, 1);
        RT_HIP(ctx, rtd::launch_flatten_slab(ctx->d_slab_mine[k], ctx->d_slab_mat[k], ctx->d_mine_sw, ctx->d_mat_sw, ctx->d_coarse,
                                             ctx->d_brick, ctx->logr, axis, texel_offset, ctx->stream));
    }
    RT_HIP(ctx, hipEventRecord(ctx->ev_slab_applied[k], ctx->stream));
    ctx->slab_apply_recorded[k] = true;
    RT_HIP(ctx, fence_lanes_after(ctx, ctx->stream));   // ... and later frames, whichever lane they start on, see the slab
    return RT_OK;
}

int rt_upload_noise(RtContext* ctx, const uint8_t* rgba8) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!rgba8) return fail(ctx, RT_ERR_INVALID_ARG, "rt_upload_noise: null pointer");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, sync_lanes(ctx));
    RT_HIP(ctx, hipMemcpy(ctx->d_noise, rgba8, RT_NOISE_BYTES, hipMemcpyHostToDevice));
    ctx->has_noise = true;
    return RT_OK;
}

int rt_draw_frame(RtContext* ctx, const RtUniforms* u) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!u) return fail(ctx, RT_ERR_INVALID_ARG, "rt_draw_frame: null uniforms");
    if (!ctx->has_world || !ctx->has_noise) return fail(ctx, RT_ERR_NOT_READY, "rt_draw_frame: world and noise must be uploaded first");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const rtd::Frame f = frame_of(ctx, u);
    const bool count = (ctx->cfg.flags & RT_FLAG_COUNTERS) != 0;
    // frame_ms of rt_get_timing: two events per frame, recorded only for a context that asked for every timing — an event between
    // the last kernel of one frame and the first of the next is 4-5 us during which the GPU idles (profiles/r3_frame_timeline.txt)
    const bool frame_events = (ctx->cfg.flags & RT_FLAG_TIMING_ALL) == RT_FLAG_TIMING_ALL;
    int rc = RT_OK;
    if (ctx->kernel == RT_KERNEL_MEGA) {
        if (frame_events) RT_HIP(ctx, hipEventRecord(ctx->ev_frame0, ctx->stream));
        LaunchTimer t(ctx, 0);
        hipError_t e = rtd::launch_mega(scene_of(ctx), f, planes_of(ctx), ctx->d_counters, count, ctx->stream);
        if (e != hipSuccess) rc = fail(ctx, RT_ERR_HIP, std::string("launch_mega: ") + hipGetErrorString(e));
    } else if (ctx->kernel == RT_KERNEL_PERSISTENT) {
        const bool cache = (ctx->cfg.flags & RT_FLAG_CACHE_PRIMARY) != 0;
        const int nl = ctx->user_stream ? 1 : ctx->nlanes, nsl = ctx->user_stream ? 1 : ctx->nslots;
        hipError_t e = hipSuccess;
        // What the host put behind the previous frame since (post passes, gather, a slab) belongs to that frame's use of its slot.
        FrameSlot& prev = ctx->slots[ctx->cur_slot];
        if (ctx->frame_recorded) { e = hipEventRecord(prev.ev_tail, ctx->stream); prev.tail_recorded = e == hipSuccess; }
        // This frame's slot, and the lane its prepass and first path launch go to (launches alternate between the lanes).
        const int si = nsl == 2 ? (int)(ctx->frames_drawn & 1u) : 0;
        FrameSlot& fs = ctx->slots[si];
        // (a one-slot context whose frames are one launch each stays on lane 0: frame k + 1 cannot start before frame k has finished
        // with the slot, so alternating would only put an event wait between two streams in front of every frame)
        const bool one_lane = nsl == 1 && (uint32_t)ctx->cfg.spp <= ctx->persist_batch;
        Lane* L0 = &ctx->lanes[one_lane ? 0u : ctx->path_launches % (uint64_t)nl];
        const hipStream_t st0 = L0->stream;
        // the frame that used the slot before (frame k - 2 with two slots, k - 1 with one) has finished with it
        if (e == hipSuccess && fs.tail_recorded) e = hipStreamWaitEvent(st0, fs.ev_tail, 0);
        if (e == hipSuccess && frame_events) e = hipEventRecord(ctx->ev_frame0, st0);
        const rtd::Planes fpl = planes_of(fs.planes);
        // worklist counter of this frame: clean already if the slot's previous frame's prepass cleared it, otherwise cleared here
        uint32_t* const wlc = fs.wl_count + 32 * fs.wl_parity;
        uint32_t* const wln = fs.wl_count + 32 * (fs.wl_parity ^ 1);
        const bool prepass_clears = cache && ctx->primary_version == 2 && ctx->npix_pad != 0;
        // k_frame: the whole frame in one launch (cached primaries by construction; frames it does not cover go the persistent way)
        // (with one sample per pixel "cached primaries" is what every kernel does anyway: the flag is not needed)
        // (more than one sample: the paths' light records go to the lane's array, which must hold the whole frame's)
        const bool one_launch = (cache || ctx->cfg.spp == 1) && rtd::launch_frame_ok(f) && (ctx->cfg.spp == 1 || (uint32_t)ctx->cfg.spp <= ctx->persist_batch) &&
                                (ctx->frame_mode == 2 || (ctx->frame_mode == 1 && frame_by_size(ctx)));
        if (e == hipSuccess && cache && !one_launch) {
            if (!fs.wl_clean[fs.wl_parity]) e = hipMemsetAsync(wlc, 0, sizeof(uint32_t), st0);
            fs.wl_clean[fs.wl_parity] = false;
            LaunchTimer t(ctx, 1, st0);
            rtd::PrimaryArgs pr{};
            pr.phit = fs.phit;
            pr.worklist = fs.worklist; pr.wl_count = wlc; pr.acc = fs.pacc; pr.counters = ctx->d_counters;
            // the prepass clears what the NEXT users need zeroed instead of a memset of its own in front of them: the slot's other
            // worklist counter (the slot's next frame) and the path cursors of the lane it runs on (this frame's first launch)
            pr.zero_words = prepass_clears ? wln : nullptr; pr.zero_count = 1u;
            pr.zero_words2 = prepass_clears ? L0->cursor : nullptr; pr.zero_count2 = (uint32_t)kCursorWords;
            if (e == hipSuccess) e = rtd::launch_primary(scene_of(ctx), f, fpl, pr, count, ctx->primary_version, ctx->num_cus, st0);
            if (prepass_clears && e == hipSuccess) { fs.wl_clean[fs.wl_parity ^ 1] = true; L0->cursor_clean = true; }
            fs.wl_parity ^= 1;
        }
        // the two per-frame tables depend on the sun vector and colour only: rebuilt when those change (bit compare) — after every
        // launch that reads the old ones, on either lane, and before any launch that follows
        float lut_key[6] = {f.sunangle[0], f.sunangle[1], f.sunangle[2], f.sunlight[0], f.sunlight[1], f.sunlight[2]};
        if (e == hipSuccess && ctx->cfg.depth >= 1 && (!ctx->lut_valid || memcmp(lut_key, ctx->lut_key, sizeof(lut_key)) != 0)) {
            if (nl == 2) e = join_lanes_into(ctx, st0);
            {
                LaunchTimer t(ctx, 1, st0);
                if (e == hipSuccess) e = rtd::launch_sun_lut(f, ctx->sun_lut, st0);
                if (e == hipSuccess) e = rtd::launch_sky_lut(f, ctx->dif_lut, st0);
            }
            if (e == hipSuccess && nl == 2) e = fence_lanes_after(ctx, st0);
            ctx->lut_valid = e == hipSuccess;
            memcpy(ctx->lut_key, lut_key, sizeof(lut_key));
        }
        hipStream_t tail = st0;
        if (e == hipSuccess && one_launch) {
            // with two frame slots consecutive frames go to the two streams in turn (frame k + 1 starts while frame k's last
            // workgroups finish); with one slot a frame follows the previous one anyway, and staying on one stream spares the
            // cross-stream event wait between them
            if (nsl == 2) ctx->path_launches++;
            rtd::FrameArgs fa{};
            fa.threshold = ctx->frame_threshold; fa.tiles_per_group = ctx->frame_tiles; fa.sun_lut = ctx->sun_lut; fa.dif_lut = ctx->dif_lut; fa.pl = L0->ppl; fa.counters = ctx->d_counters;
            if (getenv("RT_DEBUG_WAVE_DUMP") && (size_t)ctx->ntiles_local * 32u <= (size_t)4 * ctx->num_cus * 1024 * sizeof(uint32_t))
                fa.dbg_waves = reinterpret_cast<unsigned long long*>(ctx->lanes[0].pstack);   // (idle while k_frame runs)
            LaunchTimer t(ctx, 0, st0);
            ctx->last_path_kernel = RT_KERNEL_FRAME;
            e = rtd::launch_frame(scene_of(ctx), f, fpl, fa, count, ctx->num_cus, st0);
        } else if (e == hipSuccess && (!cache || ctx->cfg.depth >= 1)) {
            const uint32_t spp = (uint32_t)ctx->cfg.spp, B = ctx->persist_batch;
            if (nl == 2 && spp > B) e = hipEventRecord(fs.ev_prepass, st0);   // launches on the other lane wait for the prepass
            // one sample per pixel (the reference's own frames): k_persist and k_paths store the pixel's lighting themselves, no
            // light records and no accumulate launch (raytrace.comp:352-356 has no accumulation either)
            const bool small1 = ctx->paths_by_size && (uint64_t)ctx->npix_pad < kPathsCrossover;   // spp 1: the launch goes to k_persist
            const bool on_paths = ctx->persist_version == 3 && cache && !small1;
            const bool direct = spp == 1u && (!on_paths || rtd::launch_paths_direct_ok(f));
            for (uint32_t s0 = 0; s0 < spp && e == hipSuccess; s0 += B) {
                const uint32_t ns = spp - s0 < B ? spp - s0 : B;
                // the sample batches of a frame alternate between the lanes: batch b + 1 starts on the CUs batch b's workgroups leave
                Lane* L = &ctx->lanes[one_lane ? 0u : ctx->path_launches % (uint64_t)nl];
                const hipStream_t st = L->stream;
                if (!one_lane) ctx->path_launches++;
                if (st != st0 && s0 == B) e = hipStreamWaitEvent(st, fs.ev_prepass, 0);   // (later batches on that lane follow by stream order)
                if (e == hipSuccess && !L->cursor_clean) e = hipMemsetAsync(L->cursor, 0, kCursorWords * sizeof(uint32_t), st);
                L->cursor_clean = false;
                rtd::PersistArgs pa{};
                pa.cursor = L->cursor; pa.worklist = fs.worklist; pa.wl_count = wlc; pa.direct = direct ? 1u : 0u;
                pa.npix_pad = ctx->npix_pad; pa.sample0 = s0; pa.nsamples = ns; pa.threshold = ctx->persist_threshold; pa.rmin = ctx->persist_rmin; pa.chunk = ctx->persist_chunk;
                pa.nthreads = (uint32_t)ctx->num_cus * 1024u; pa.stack = L->pstack;
                pa.phit = fs.phit;
                pa.sun_lut = ctx->sun_lut; pa.dif_lut = ctx->dif_lut;
                pa.pl = L->ppl; pa.counters = ctx->d_counters;
                // light records of this launch: streamed past the caches when there are more of them than would stay there until the
                // accumulate launch reads them (the Infinity Cache holds 256 MB; RT_PL_STREAM = 0 / 1 / 2 / 3 forces never / stores /
                // loads / both)
                const uint64_t record_bytes = (uint64_t)ctx->npix_pad * ns * sizeof(rtd::PathLight);
                const bool big_records = record_bytes > kStreamRecordBytes;
                const bool stream_st = ctx->pl_stream_mode < 0 ? big_records : (ctx->pl_stream_mode & 1) != 0;
                const bool stream_ld = ctx->pl_stream_mode < 0 ? big_records : (ctx->pl_stream_mode & 2) != 0;
                pa.pl_stream = stream_st ? 1u : 0u;
                if (e == hipSuccess) {
                    LaunchTimer t(ctx, 0, st);
                    // k_paths pays for its two contexts per lane once there is enough work to keep them filled: measured
                    // crossover against k_persist (tools/kernel_crossover.sh) between 4 M and 8 M paths per launch — 1080p spp 1:
                    // 0.28 against 0.22 ms, 256^2 spp 64: 0.35 against 0.32, 1080p spp 4: 0.47 against 0.50, spp 16: 1.29 against
                    // 1.65; the reference's own 1024^2 1-spp frame: 0.20 against 0.14.  The worklist length lives on the device;
                    // the pixel count bounds it.  (RT_KERNEL_PATHS asks for k_paths whatever the size.)
                    const bool big = !ctx->paths_by_size || (uint64_t)ctx->npix_pad * ns >= kPathsCrossover;
                    // parked lanes that trigger a pass (RT_PERSIST_THRESHOLD overrides): measured optima per kernel.  k_persist 32.
                    // k_paths (lanes with a parked context, three steps between looks), round 3, same box: 24 4.28 ms per
                    // headline launch, 28 4.22, 32 4.17, 36 4.06-4.15, 40 4.08-4.14, 44 4.12, 48 4.27; deeper frames at region 256 (depth
                    // 5..8: longer paths, fewer new ones per pass) like 40 — 3840x2160 spp 256 depth 8 launch 51.2 ms at 28, 50.7 at 32, 50.0
                    // at 36, 49.6 at 40 and 44 —, the 1024^3 frame stays at 36 (20.74 against 20.86 at 40, 21.2 at 44)
                    const bool to_paths = ctx->persist_version == 3 && cache && big;
                    if (ctx->persist_threshold == 0)
                        pa.threshold = !to_paths ? 32u : ((ctx->cfg.depth > 4 && ctx->region == 256) ? 40u : 36u);
                    if (to_paths) {
                        ctx->last_path_kernel = RT_KERNEL_PATHS;
                        e = rtd::launch_paths(scene_of(ctx), f, fpl, pa, count, ctx->num_cus, st);
                    } else {
                        ctx->last_path_kernel = RT_KERNEL_PERSISTENT;
                        e = rtd::launch_persist(scene_of(ctx), f, fpl, pa, count, cache, 1, ctx->num_cus, st);
                    }
                }
                if (e == hipSuccess && !direct) {
                    // a pixel's samples are added in sample order: batch b's accumulate follows batch b - 1's, whichever lane that ran on
                    if (nl == 2 && s0 != 0) e = hipStreamWaitEvent(st, fs.ev_acc, 0);
                    {
                        LaunchTimer t(ctx, 1, st);
                        if (e == hipSuccess)
                            e = rtd::launch_accumulate_paths(f, fpl, L->ppl, fs.worklist, wlc, ctx->npix_pad, ns, s0 == 0, s0 + B >= spp, cache, stream_ld,
                                                             fs.pacc, st);
                    }
                    if (e == hipSuccess && nl == 2 && s0 + B < spp) e = hipEventRecord(fs.ev_acc, st);
                }
                tail = st;
            }
        }
        // (no resolve launch: the prepass and the last accumulate of the frame store the lighting planes themselves)
        if (e != hipSuccess) rc = fail(ctx, RT_ERR_HIP, std::string("persistent path: ") + hipGetErrorString(e));
        // the frame ends on `tail`: post passes, gather and readback of THIS frame follow it there; its planes are the context's
        ctx->stream = tail;
        ctx->cur_slot = si;
        ctx->gbuffer = fs.gbuffer;
        for (int b = 0; b < RT_BUF_COUNT; b++) ctx->planes[b] = fs.planes[b];
        ctx->frames_drawn++;
    } else {
        if (frame_events) RT_HIP(ctx, hipEventRecord(ctx->ev_frame0, ctx->stream));
        rc = draw_wavefront(ctx, f);
    }
    if (frame_events) RT_HIP(ctx, hipEventRecord(ctx->ev_frame1, ctx->stream));
    ctx->frame_recorded = true;
    if (count) { ctx->host_noise_base += (uint64_t)ctx->cfg.spp; ctx->host_frames++; }
    return rc;
}

int rt_sync(RtContext* ctx) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, sync_lanes(ctx));
    if (ctx->gather_stream) RT_HIP(ctx, hipStreamSynchronize(ctx->gather_stream));
    return RT_OK;
}

size_t rt_buffer_bytes(RtContext* ctx, int id) {
    if (!ctx || id < 0 || id >= RT_BUF_COUNT) return 0;
    return ctx->plane_pixels * kBytesPerPixel[id];
}

void* rt_device_ptr(RtContext* ctx, int id) {
    if (!ctx || id < 0 || id >= RT_BUF_COUNT) return nullptr;
    return ctx->planes[id];
}

int rt_readback(RtContext* ctx, int id, void* dst, size_t bytes) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (id < 0 || id >= RT_BUF_COUNT || !dst) return fail(ctx, RT_ERR_INVALID_ARG, "rt_readback: bad buffer id or null destination");
    if (bytes != rt_buffer_bytes(ctx, id)) return fail(ctx, RT_ERR_INVALID_ARG, "rt_readback: size mismatch");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    RT_HIP(ctx, hipMemcpy(dst, ctx->planes[id], bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_tile_count(RtContext* ctx) { return ctx ? ctx->ntiles_local : RT_ERR_INVALID_ARG; }
int rt_tile_capacity(RtContext* ctx) { return ctx ? ctx->tile_capacity : RT_ERR_INVALID_ARG; }

int rt_untile(RtContext* ctx, int id, const void* gathered_dev, int world, void* frame_dev) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (id < 0 || id >= RT_BUF_COUNT || !gathered_dev || !frame_dev || world < 1)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_untile: bad argument");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const int capacity = (ctx->ntiles_total + world - 1) / world;
    RT_HIP(ctx, rtd::launch_untile(gathered_dev, frame_dev, world, capacity, ctx->tiles_x, ctx->tiles_y, ctx->cfg.width,
                                   ctx->cfg.height, (int)kBytesPerPixel[id], ctx->stream));
    return RT_OK;
}

int rt_denoise_planes(RtContext* ctx, void* lighting, const void* depth, const void* normal, int faithful) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!lighting || !depth || !normal) return fail(ctx, RT_ERR_INVALID_ARG, "rt_denoise_planes: null plane");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    const int sizes[6] = {1, 2, 4, 8, 8, 16};                         // pipeline.rs:103
    const int W = ctx->cfg.width, H = ctx->cfg.height;
    void** work = ctx->slots[ctx->cur_slot].denoise_work;   // per slot: the post passes of two frames in flight may overlap
    for (int i = 0; i < 2; i++)
        if (!work[i]) { uint4* p = nullptr; RT_HIP(ctx, dev_alloc(ctx, &p, (size_t)W * H)); work[i] = p; }
    {
        LaunchTimer t(ctx, 1);
        RT_HIP(ctx, rtd::launch_denoise_prepare(lighting, depth, normal, W, H, work[0], ctx->stream));
    }
    for (int pass = 0; pass < 6; pass++) {
        // pipeline.rs:104-108: the ping descriptor set on even dispatches, the pong set (normal/depth bindings swapped,
        // descriptor_sets.rs:38-39) on odd ones; the sixth dispatch writes the lighting image finalize.comp reads
        const bool odd = pass % 2 == 1;
        LaunchTimer t(ctx, 1);
        RT_HIP(ctx, rtd::launch_denoise(work[pass & 1], W, H, sizes[pass], odd && faithful != 0, pass == 5,
                                        work[(pass & 1) ^ 1], lighting, ctx->stream));
    }
    return RT_OK;
}

int rt_finalize_planes(RtContext* ctx, const void* albedo, const void* emission, const void* fog, const void* lighting,
                       const void* depth, void* out_bgra8) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!albedo || !emission || !fog || !lighting || !depth || !out_bgra8) return fail(ctx, RT_ERR_INVALID_ARG, "rt_finalize_planes: null plane");
    if (!ctx->has_noise) return fail(ctx, RT_ERR_NOT_READY, "rt_finalize_planes: noise must be uploaded first");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    LaunchTimer t(ctx, 1);
    RT_HIP(ctx, rtd::launch_finalize(albedo, emission, fog, lighting, depth, ctx->d_noise, ctx->cfg.width, ctx->cfg.height, out_bgra8,
                                     ctx->stream));
    return RT_OK;
}

int rt_denoise(RtContext* ctx, int faithful) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (ctx->cfg.tile_world != 1) return fail(ctx, RT_ERR_UNIMPLEMENTED, "rt_denoise: whole-frame contexts only (gather the tiles, then rt_denoise_planes)");
    if (!ctx->frame_recorded) return fail(ctx, RT_ERR_NOT_READY, "rt_denoise: no frame drawn yet");
    return rt_denoise_planes(ctx, ctx->planes[RT_BUF_LIGHTING_RGBA16], ctx->planes[RT_BUF_DEPTH_R16UI], ctx->planes[RT_BUF_NORMAL_R8UI], faithful);
}

int rt_finalize(RtContext* ctx) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (ctx->cfg.tile_world != 1) return fail(ctx, RT_ERR_UNIMPLEMENTED, "rt_finalize: whole-frame contexts only (gather the tiles, then rt_finalize_planes)");
    if (!ctx->frame_recorded) return fail(ctx, RT_ERR_NOT_READY, "rt_finalize: no frame drawn yet");
    return rt_finalize_planes(ctx, ctx->planes[RT_BUF_ALBEDO_RGBA8], ctx->planes[RT_BUF_EMISSION_RGBA8], ctx->planes[RT_BUF_FOG_RGBA8],
                              ctx->planes[RT_BUF_LIGHTING_RGBA16], ctx->planes[RT_BUF_DEPTH_R16UI], ctx->planes[RT_BUF_FINAL_BGRA8]);
}

void* rt_gbuffer_ptr(RtContext* ctx) { return ctx ? ctx->gbuffer : nullptr; }
size_t rt_gbuffer_bytes(RtContext* ctx) { return ctx ? ctx->gbuffer_bytes : 0; }
size_t rt_gbuffer_offset(RtContext* ctx, int id) {
    if (!ctx || id < 0 || id > RT_BUF_FOG_RGBA8) return 0;
    return ctx->gbuffer_offset[id];
}

int rt_untile_gbuffer(RtContext* ctx, const void* gathered_dev, int world, void* const* frames_dev) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (!gathered_dev || !frames_dev || world < 1) return fail(ctx, RT_ERR_INVALID_ARG, "rt_untile_gbuffer: bad argument");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    // the per-rank block layout (plane offsets, block size) is this context's own: only valid for its own split
    if (world != ctx->cfg.tile_world || world < 2)
        return fail(ctx, RT_ERR_INVALID_ARG, "rt_untile_gbuffer: world must equal the context's tile_world (>= 2)");
    const int capacity = (ctx->ntiles_total + world - 1) / world;
    for (int b = 0; b <= RT_BUF_FOG_RGBA8; b++) {
        if (!frames_dev[b]) continue;
        RT_HIP(ctx, rtd::launch_untile_strided((const uint8_t*)gathered_dev + ctx->gbuffer_offset[b], ctx->gbuffer_bytes, frames_dev[b], world,
                                               capacity, ctx->tiles_x, ctx->tiles_y, ctx->cfg.width, ctx->cfg.height,
                                               (int)kBytesPerPixel[b], ctx->stream));
    }
    return RT_OK;
}


// ---- multi-GPU: RCCL gather of the tile-split G-buffer ------------------------------------------------------
namespace {
struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error;
};
Rccl g_rccl;
std::once_flag g_rccl_once;

// librccl is 0.5 GB: loaded on first use only.  A copy the process already holds (e.g. the one PyTorch ships) is reused,
// so a communicator created by the host's own RCCL stays valid here.
bool rccl_load() {
    std::call_once(g_rccl_once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so"};
        for (const char* n : names) { if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD); }
        for (const char* n : names) { if (!g_rccl.handle) g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL); }
        if (!g_rccl.handle) { g_rccl.error = std::string("cannot load librccl: ") + dlerror(); return; }
#define RT_RCCL_SYM(name) g_rccl.name = (decltype(g_rccl.name))dlsym(g_rccl.handle, "nccl" #name); if (!g_rccl.name) g_rccl.error = "librccl lacks nccl" #name;
        RT_RCCL_SYM(GetUniqueId) RT_RCCL_SYM(CommInitRank) RT_RCCL_SYM(CommInitAll) RT_RCCL_SYM(CommDestroy) RT_RCCL_SYM(GroupStart)
        RT_RCCL_SYM(GroupEnd) RT_RCCL_SYM(Send) RT_RCCL_SYM(Recv) RT_RCCL_SYM(GetErrorString)
#undef RT_RCCL_SYM
    });
    return g_rccl.error.empty();
}
#define RT_NCCL(ctx, call)                                                                                   \
    do { ncclResult_t r_ = (call); if (r_ != ncclSuccess)                                                    \
        return fail(ctx, RT_ERR_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); } while (0)
}  // namespace

int rt_comm_unique_id(void* id_out, size_t bytes) {
    if (!id_out || bytes != NCCL_UNIQUE_ID_BYTES) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_comm_unique_id: need a 128-byte buffer");
    if (!rccl_load()) return fail(nullptr, RT_ERR_UNIMPLEMENTED, g_rccl.error);
    ncclUniqueId id;
    RT_NCCL(nullptr, g_rccl.GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return RT_OK;
}

int rt_comm_init_rank(RtContext* ctx, const void* id, size_t bytes, void** comm_out) {
    if (comm_out) *comm_out = nullptr;
    if (!ctx || !id || bytes != NCCL_UNIQUE_ID_BYTES || !comm_out) return fail(ctx, RT_ERR_INVALID_ARG, "rt_comm_init_rank: bad argument");
    if (!rccl_load()) return fail(ctx, RT_ERR_UNIMPLEMENTED, g_rccl.error);
    RT_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    ncclComm_t comm = nullptr;
    RT_NCCL(ctx, g_rccl.CommInitRank(&comm, ctx->cfg.tile_world, uid, ctx->cfg.tile_rank));
    *comm_out = comm;
    return RT_OK;
}

int rt_comm_init_all(int ndev, const int* devices, void** comms_out) {
    if (ndev < 1 || !comms_out) return fail(nullptr, RT_ERR_INVALID_ARG, "rt_comm_init_all: bad argument");
    if (!rccl_load()) return fail(nullptr, RT_ERR_UNIMPLEMENTED, g_rccl.error);
    std::vector<ncclComm_t> comms((size_t)ndev, nullptr);
    RT_NCCL(nullptr, g_rccl.CommInitAll(comms.data(), ndev, devices));
    for (int i = 0; i < ndev; i++) comms_out[i] = comms[(size_t)i];
    return RT_OK;
}

int rt_comm_destroy(void* comm) {
    if (!comm) return RT_OK;
    if (!rccl_load()) return fail(nullptr, RT_ERR_UNIMPLEMENTED, g_rccl.error);
    RT_NCCL(nullptr, g_rccl.CommDestroy((ncclComm_t)comm));
    return RT_OK;
}

int rt_gather_gbuffer(RtContext* ctx, void* comm_, int root, void* const* frames_dev, int overlapped) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    const int world = ctx->cfg.tile_world, rank = ctx->cfg.tile_rank;
    if (root < 0 || root >= world) return fail(ctx, RT_ERR_INVALID_ARG, "rt_gather_gbuffer: root out of range");
    if (!ctx->frame_recorded) return fail(ctx, RT_ERR_NOT_READY, "rt_gather_gbuffer: no frame drawn yet");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (rank == root && !frames_dev) {
        // no caller-owned planes: assemble into the library's own (allocated once; rt_frame_ptr / rt_frame_readback)
        for (int b = 0; b <= RT_BUF_FOG_RGBA8; b++)
            if (!ctx->frame_planes[b]) {
                uint8_t* p = nullptr;
                RT_HIP(ctx, dev_alloc(ctx, &p, (size_t)ctx->cfg.width * ctx->cfg.height * kBytesPerPixel[b]));
                ctx->frame_planes[b] = p;
            }
        frames_dev = ctx->frame_planes;
    }
    if (world == 1 && !comm_) {
        // one context holds the whole frame, row-major already: plain copies on the stream the frame ended on, after the previous
        // frame's copies (which sit on the other lane's stream when two frames are in flight)
        if (ctx->gather_recorded) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_gather, 0));
        for (int b = 0; b <= RT_BUF_FOG_RGBA8; b++)
            if (frames_dev[b]) RT_HIP(ctx, hipMemcpyAsync(frames_dev[b], ctx->planes[b], ctx->plane_pixels * kBytesPerPixel[b], hipMemcpyDeviceToDevice, ctx->stream));
        RT_HIP(ctx, hipEventRecord(ctx->ev_gather, ctx->stream)); ctx->gather_recorded = true;
        return RT_OK;
    }
    if (!comm_) return fail(ctx, RT_ERR_INVALID_ARG, "rt_gather_gbuffer: null communicator");
    if (!rccl_load()) return fail(ctx, RT_ERR_UNIMPLEMENTED, g_rccl.error);
    ncclComm_t comm = (ncclComm_t)comm_;
    const size_t gb = ctx->gbuffer_bytes;
    const int s = overlapped ? (int)(ctx->gathers & 1u) : 0;
    if (rank == root && !ctx->gathered[s]) RT_HIP(ctx, dev_alloc(ctx, &ctx->gathered[s], gb * (size_t)world));
    hipStream_t gs = ctx->stream;
    const uint8_t* src = (const uint8_t*)ctx->gbuffer;
    // gathers share the root's staging and the communicator: with two frames in flight the previous frame's gather sits on the
    // other lane's stream — this one follows it (the frames' rendering still overlaps)
    if (!overlapped && ctx->gather_recorded) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_gather, 0));
    if (overlapped) {
        // The frame's block is copied to one of two staging buffers on the render stream (so the next frame may overwrite the
        // planes), everything else runs on a second stream: frame k's send/recv + un-tile overlap frame k+1's kernels.
        if (!ctx->gather_stream) RT_HIP(ctx, hipStreamCreateWithFlags(&ctx->gather_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            if (!ctx->ev_ready[i]) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_ready[i], hipEventDisableTiming));
            if (!ctx->ev_free[i]) RT_HIP(ctx, hipEventCreateWithFlags(&ctx->ev_free[i], hipEventDisableTiming));
        }
        if (!ctx->stage[s]) RT_HIP(ctx, dev_alloc(ctx, &ctx->stage[s], gb));
        if (ctx->ev_free_recorded[s]) RT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_free[s], 0));   // gather k-2 has left stage[s]
        RT_HIP(ctx, hipMemcpyAsync(ctx->stage[s], ctx->gbuffer, gb, hipMemcpyDeviceToDevice, ctx->stream));
        RT_HIP(ctx, hipEventRecord(ctx->ev_ready[s], ctx->stream));
        RT_HIP(ctx, hipStreamWaitEvent(ctx->gather_stream, ctx->ev_ready[s], 0));
        gs = ctx->gather_stream;
        src = ctx->stage[s];
    }
    // RT_FLAG_TIMING: events round the transfer + un-tile on the stream they run on (rt_get_gather_timing)
    size_t gev = (size_t)-1;
    if ((ctx->cfg.flags & RT_FLAG_TIMING) != 0 && ctx->gather_ev_used + 2 <= 2 * 4096) {
        while (ctx->gather_ev.size() < ctx->gather_ev_used + 2) { hipEvent_t ev; RT_HIP(ctx, hipEventCreate(&ev)); ctx->gather_ev.push_back(ev); }
        gev = ctx->gather_ev_used; ctx->gather_ev_used += 2;
        RT_HIP(ctx, hipEventRecord(ctx->gather_ev[gev], gs));
    }
    // every rank sends its block to the root; the root posts one receive per rank (its own block included).  Each peer uses
    // its own xGMI link into the root, so the transfers run in parallel.
    RT_NCCL(ctx, g_rccl.GroupStart());
    {   // a failing call must not leave the group open: the group is always closed, the first error is reported
        ncclResult_t first = ncclSuccess;
        const char* what = "";
        if (rank == root)
            for (int r = 0; r < world && first == ncclSuccess; r++) {
                first = g_rccl.Recv(ctx->gathered[s] + (size_t)r * gb, gb, ncclUint8, r, comm, gs);
                what = "ncclRecv";
            }
        if (first == ncclSuccess) { first = g_rccl.Send(src, gb, ncclUint8, root, comm, gs); what = "ncclSend"; }
        const ncclResult_t end = g_rccl.GroupEnd();
        if (first == ncclSuccess && end != ncclSuccess) { first = end; what = "ncclGroupEnd"; }
        if (first != ncclSuccess) return fail(ctx, RT_ERR_HIP, std::string("rt_gather_gbuffer: ") + what + ": " + g_rccl.GetErrorString(first));
    }
    if (rank == root) {
        const int capacity = (ctx->ntiles_total + world - 1) / world;
        for (int b = 0; b <= RT_BUF_FOG_RGBA8; b++) {
            if (!frames_dev[b]) continue;
            if (world == 1) {   // a one-rank communicator (the transfer went to itself): the block holds row-major planes
                RT_HIP(ctx, hipMemcpyAsync(frames_dev[b], ctx->gathered[s] + ctx->gbuffer_offset[b], ctx->plane_pixels * kBytesPerPixel[b],
                                           hipMemcpyDeviceToDevice, gs));
                continue;
            }
            RT_HIP(ctx, rtd::launch_untile_strided(ctx->gathered[s] + ctx->gbuffer_offset[b], gb, frames_dev[b], world, capacity, ctx->tiles_x,
                                                   ctx->tiles_y, ctx->cfg.width, ctx->cfg.height, (int)kBytesPerPixel[b], gs));
        }
    }
    if (gev != (size_t)-1) RT_HIP(ctx, hipEventRecord(ctx->gather_ev[gev + 1], gs));
    if (overlapped) { RT_HIP(ctx, hipEventRecord(ctx->ev_free[s], gs)); ctx->ev_free_recorded[s] = true; }
    else { RT_HIP(ctx, hipEventRecord(ctx->ev_gather, gs)); ctx->gather_recorded = true; }
    ctx->gathers++;
    return RT_OK;
}

int rt_get_gather_timing(RtContext* ctx, float* ms_sum, uint32_t* calls) {
    if (!ctx || !ms_sum || !calls) return RT_ERR_INVALID_ARG;
    *ms_sum = 0.0f; *calls = 0;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, sync_lanes(ctx));
    if (ctx->gather_stream) RT_HIP(ctx, hipStreamSynchronize(ctx->gather_stream));
    for (size_t i = 0; i + 1 < ctx->gather_ev_used; i += 2) {
        float ms = 0.0f;
        RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->gather_ev[i], ctx->gather_ev[i + 1]));
        *ms_sum += ms; (*calls)++;
    }
    ctx->gather_ev_used = 0;
    return RT_OK;
}

int rt_selftest(RtContext* ctx, int which, uint64_t* result) {
    if (!ctx || !result) return RT_ERR_INVALID_ARG;
    if (which != RT_SELFTEST_DENOISE_DIVISION) return fail(ctx, RT_ERR_INVALID_ARG, "rt_selftest: unknown test");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->d_selftest) RT_HIP(ctx, dev_alloc(ctx, &ctx->d_selftest, 1));   // one word, allocated once
    unsigned long long* d = ctx->d_selftest;
    RT_HIP(ctx, hipMemsetAsync(d, 0, sizeof(*d), ctx->stream));
    RT_HIP(ctx, rtd::launch_selftest_dn_div(d, ctx->stream));
    unsigned long long h = 0;
    RT_HIP(ctx, hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    RT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *result = h;
    return RT_OK;
}

void* rt_frame_ptr(RtContext* ctx, int id) {
    if (!ctx || id < 0 || id > RT_BUF_FOG_RGBA8) return nullptr;
    return ctx->frame_planes[id];
}

int rt_frame_readback(RtContext* ctx, int id, void* dst, size_t bytes) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (id < 0 || id > RT_BUF_FOG_RGBA8 || !dst) return fail(ctx, RT_ERR_INVALID_ARG, "rt_frame_readback: bad buffer id or null destination");
    if (!ctx->frame_planes[id]) return fail(ctx, RT_ERR_NOT_READY, "rt_frame_readback: no frame assembled by rt_gather_gbuffer(frames_dev = NULL) yet");
    if (bytes != (size_t)ctx->cfg.width * ctx->cfg.height * kBytesPerPixel[id]) return fail(ctx, RT_ERR_INVALID_ARG, "rt_frame_readback: size mismatch");
    int rc = rt_sync(ctx);
    if (rc != RT_OK) return rc;
    RT_HIP(ctx, hipMemcpy(dst, ctx->frame_planes[id], bytes, hipMemcpyDeviceToHost));
    return RT_OK;
}

int rt_get_info(RtContext* ctx, RtInfo* out) {
    if (!ctx || !out) return RT_ERR_INVALID_ARG;
    if (out->struct_size != sizeof(RtInfo)) return fail(ctx, RT_ERR_INVALID_ARG, "rt_get_info: RtInfo.struct_size mismatch");
    out->num_cus = ctx->num_cus;
    out->samples_per_launch = ctx->kernel == RT_KERNEL_PERSISTENT ? ctx->persist_batch : ctx->batch_samples;
    out->launches_in_flight = (uint16_t)(ctx->user_stream ? 1 : ctx->nlanes);
    out->frames_in_flight = (uint16_t)(ctx->user_stream ? 1 : ctx->nslots);
    out->light_record_budget_bytes = ctx->light_budget_bytes;
    out->light_record_bytes = ctx->kernel == RT_KERNEL_PERSISTENT
        ? (uint64_t)sizeof(rtd::PathLight) * (ctx->npix_pad ? ctx->npix_pad : 1) * ctx->persist_batch * (uint64_t)ctx->nlanes : 0;
    out->device_bytes = ctx->device_bytes;
    return RT_OK;
}

int rt_kernel_in_use(RtContext* ctx) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    if (ctx->kernel == RT_KERNEL_PERSISTENT && ctx->last_path_kernel != 0) return ctx->last_path_kernel;   // what the last frame ran
    if (ctx->kernel == RT_KERNEL_PERSISTENT && ((ctx->cfg.flags & RT_FLAG_CACHE_PRIMARY) || ctx->cfg.spp == 1) && ctx->cfg.depth <= 8 &&
        (ctx->cfg.spp == 1 || (uint32_t)ctx->cfg.spp <= ctx->persist_batch) && (ctx->frame_mode == 2 || (ctx->frame_mode == 1 && frame_by_size(ctx))))
        return RT_KERNEL_FRAME;
    if (ctx->kernel == RT_KERNEL_PERSISTENT && ctx->persist_version == 3) return RT_KERNEL_PATHS;
    return ctx->kernel;
}

int rt_get_counters(RtContext* ctx, RtCounters* out) {
    if (!ctx || !out) return RT_ERR_INVALID_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, sync_lanes(ctx));
    rtd::DevCounters d;
    RT_HIP(ctx, hipMemcpy(&d, ctx->d_counters, sizeof(d), hipMemcpyDeviceToHost));
    out->rays = d.rays; out->rays_primary = d.rays_primary; out->rays_shadow = d.rays_shadow; out->rays_diffuse = d.rays_diffuse;
    out->iterations = d.iterations; out->minefield_fetches = d.minefield_fetches; out->material_fetches = d.material_fetches;
    out->noise_fetches = d.noise_fetches + ctx->host_noise_base;   // + the seed-base texel of each sample (raytrace.comp:302-303)
    out->hits = d.hits; out->sky_exits = d.sky_exits; out->limit_exits = d.limit_exits; out->border_fetches = d.border_fetches;
    out->pixels = d.pixels; out->frames = ctx->host_frames;
    if (const char* path = getenv("RT_DEBUG_WAVE_DUMP")) {   // -DRT_DIAG_WAVE_TIMES builds of k_paths: per-wave (start, out of paths, end, workgroup)
        if (ctx->lanes[0].pstack) {
            std::vector<unsigned long long> rec(ctx->last_path_kernel == RT_KERNEL_FRAME ? ((size_t)ctx->ntiles_local + 3u) / 4u * 16u : (size_t)ctx->num_cus * 16u * 4u);
            if (hipMemcpy(rec.data(), ctx->lanes[0].pstack, rec.size() * sizeof(rec[0]), hipMemcpyDeviceToHost) == hipSuccess)
                if (FILE* fp = fopen(path, "wb")) { fwrite(rec.data(), sizeof(rec[0]), rec.size(), fp); fclose(fp); }
        }
    }
    if (getenv("RT_DEBUG_STATS"))
        fprintf(stderr, "[rt] wave loop iters %llu | S block execs %llu (avg lanes %.1f) | F block execs %llu (avg lanes %.1f) | passes %llu "
                        "(avg lanes %.1f, sky lanes %.1f)\n", d.dbg_loop_iters, d.dbg_s_execs, d.dbg_s_execs ? (double)d.dbg_s_lanes / d.dbg_s_execs : 0.0,
                d.dbg_f_execs, d.dbg_f_execs ? (double)d.dbg_f_lanes / d.dbg_f_execs : 0.0, d.dbg_passes,
                d.dbg_passes ? (double)d.dbg_pass_lanes / d.dbg_passes : 0.0, d.dbg_passes ? (double)d.dbg_sky_lanes / d.dbg_passes : 0.0),
        fprintf(stderr, "[rt] raw: loop_iters %llu s_lanes %llu f_lanes %llu passes %llu pass_lanes %llu s_execs %llu f_execs %llu\n", d.dbg_loop_iters, d.dbg_s_lanes,
                d.dbg_f_lanes, d.dbg_passes, d.dbg_pass_lanes, d.dbg_s_execs, d.dbg_f_execs),
        fprintf(stderr, "[rt] raw2: sky_lanes %llu\n", d.dbg_sky_lanes);
    return RT_OK;
}

int rt_reset_counters(RtContext* ctx) {
    if (!ctx) return RT_ERR_INVALID_ARG;
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, sync_lanes(ctx));
    RT_HIP(ctx, hipMemset(ctx->d_counters, 0, sizeof(rtd::DevCounters)));
    ctx->host_noise_base = 0; ctx->host_frames = 0;
    return RT_OK;
}

int rt_get_timing(RtContext* ctx, RtTiming* out) {
    if (!ctx || !out) return RT_ERR_INVALID_ARG;
    memset(out, 0, sizeof(*out));
    if (!ctx->frame_recorded) return fail(ctx, RT_ERR_NOT_READY, "rt_get_timing: no frame drawn yet");
    RT_HIP(ctx, hipSetDevice(ctx->device));
    RT_HIP(ctx, sync_lanes(ctx));
    if ((ctx->cfg.flags & RT_FLAG_TIMING_ALL) == RT_FLAG_TIMING_ALL)   // 0 for a context created without RT_FLAG_TIMING_ALL
        RT_HIP(ctx, hipEventElapsedTime(&out->frame_ms, ctx->ev_frame0, ctx->ev_frame1));
    // per-launch events accumulate over every frame drawn since the previous rt_get_timing (no per-frame sync needed)
    for (size_t i = 0; i + 1 < ctx->ev_used; i += 2) {
        float ms = 0.0f;
        RT_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev_pool[i], ctx->ev_pool[i + 1]));
        if (ctx->ev_kind[i / 2] == 0) { out->trace_ms += ms; out->trace_launches++; }
        else { out->shade_ms += ms; out->other_launches++; }
    }
    ctx->ev_used = 0;
    return RT_OK;
}

}  // extern "C"
