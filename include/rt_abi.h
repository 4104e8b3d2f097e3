/*
 * rt_abi.h — C ABI of the MI355X-native ray-trace hot path (librt_amd.so).
 *
 * This is the drop-in boundary for the per-pixel path of someguynamedjosh/raytrace:
 * what the reference does in shaders/glsl/raytrace.comp behind `render::Pipeline`
 * is done here by hand-written gfx950 HIP kernels behind plain `extern "C"` entry
 * points (plain pointers and sizes only; no C++/torch types cross this line).
 * A Rust host binds these with an `extern "C"` block.
 *
 * Every entry point cites the reference interface it replaces (paths relative to
 * the reference repository root).
 *
 * Conventions
 *   - All functions returning int return RT_OK (0) or a negative RtStatus; they
 *     never throw or abort (the reference panics via .expect(): pipeline.rs:145,167).
 *   - Host pointers are borrowed for the duration of the call only; the context
 *     owns all device memory (reference: Vulkan objects owned by RenderData).
 *   - A context is used from one host thread at a time (reference: Rc<Core> is !Send,
 *     render/mod.rs:40).
 */
#ifndef RT_ABI_H
#define RT_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants restated from src/render/constants.rs:15-33 and raytrace.comp:37-58 ---- */
#define RT_CHUNK_SIZE        64      /* constants.rs:23  CHUNK_SIZE = 1 << MAX_CHUNK_LOD      */
#define RT_MAX_CHUNK_LOD     6       /* constants.rs:22                                       */
#define RT_ROOT_CHUNK_SIZE   4       /* constants.rs:26                                       */
#define RT_ROOT_BLOCK_SIZE   256     /* constants.rs:27; raytrace.comp:37 ROOT_BLOCK_WIDTH    */
#define RT_SLICE_SIZE        16      /* constants.rs:30                                       */
#define RT_SHADER_GROUP_SIZE 8       /* constants.rs:33; raytrace.comp:9 local_size 8x8       */
#define RT_PIXEL_SPREAD      16      /* raytrace.comp:54                                      */
#define RT_NOISE_SIZE        512     /* constants.rs:16-17; raytrace.comp:43                  */
#define RT_NOISE_BYTES       (512 * 512 * 4) /* constants.rs:19 BLUE_NOISE_SIZE               */
#define RT_LIGHTING_SCALE    16.0f   /* raytrace.comp:57                                      */
#define RT_TRACE_LIMIT       2048    /* raytrace.comp:109                                     */
#define RT_NORMAL_AIR        16      /* raytrace.comp:369 value stored for sky pixels         */
#define RT_DEPTH_AIR         0xFFFF  /* raytrace.comp:357                                     */
#define RT_MAX_DEPTH         16      /* build limit on the `depth` extension      */

typedef enum RtStatus {
    RT_OK = 0,
    RT_ERR_INVALID_ARG = -1,
    RT_ERR_NO_DEVICE = -2,      /* no HIP device / HIP runtime failure at create          */
    RT_ERR_HIP = -3,            /* a HIP call failed; see rt_last_error                   */
    RT_ERR_NOT_READY = -4,      /* draw before world/noise upload                         */
    RT_ERR_UNIMPLEMENTED = -5,
    RT_ERR_OOM = -6
} RtStatus;

/*
 * RtUniforms — byte-identical to `RaytraceUniformData` (src/render/pipeline/structs.rs:3-31)
 * and the GLSL std140 block `UniformData` (shaders/glsl/raytrace.comp:25-35). 192 bytes.
 * Live fields: sun_angle@0 seed@4 origin@16 forward@32 up@48 right@64 lr@160.
 * Dead in the shader (kept for layout): old_origin@80 old_transform_c0..2@96/112/128
 * region_offset@144 lso@176.
 */
typedef struct RtUniforms {
    float    sun_angle;           /*   0 */
    uint32_t seed;                /*   4 */
    uint32_t _padding0[2];        /*   8  (u64 in the reference)                       */
    float    origin[3];           /*  16 */
    uint32_t _padding1;
    float    forward[3];          /*  32 */
    uint32_t _padding2;
    float    up[3];               /*  48  pre-scaled by 0.4 on the host: pipeline.rs:198 */
    uint32_t _padding3;
    float    right[3];            /*  64  pre-scaled by 0.4 on the host: pipeline.rs:199 */
    uint32_t _padding4;
    float    old_origin[3];       /*  80  dead */
    uint32_t _padding5;
    float    old_transform_c0[3]; /*  96  dead */
    uint32_t _padding6;
    float    old_transform_c1[3]; /* 112  dead */
    uint32_t _padding7;
    float    old_transform_c2[3]; /* 128  dead */
    uint32_t _padding8;
    int32_t  region_offset[3];    /* 144  dead */
    uint32_t _padding9;
    int32_t  lr[3];               /* 160  `rotation` in structs.rs:27; `lr` in the shader */
    uint32_t _padding10;
    int32_t  lso[3];              /* 176  `space_offset`; dead */
    uint32_t _padding11;
} RtUniforms;

/* Which traversal implementation a context uses. */
typedef enum RtKernel {
    RT_KERNEL_DEFAULT = 0,    /* library picks per frame: frame (one-sample frames < 2.5 M pixels) / paths / persistent by work size */
    RT_KERNEL_MEGA = 1,       /* one thread per pixel, all rays inline, byte minefield from HBM (baseline)      */
    RT_KERNEL_WAVEFRONT = 2,  /* split stages: persistent traversal kernel fed by SoA ray/hit queues in HBM     */
    RT_KERNEL_PERSISTENT = 3, /* production kernel: persistent wave64 path kernel, a lane owns a path with its shadow
                                 and diffuse ray in two ray slots, state in registers, __ballot batched transitions,
                                 nibble map in LDS, per-XCD path cursors                                         */
    /* 4 was RT_KERNEL_PERSISTENT2 (two paths per lane, one slot each): retired in round 3, rejected by rt_create */
    RT_KERNEL_PATHS = 5,      /* a lane carries two paths with two ray slots each (four fetch chains in flight per lane) and
                                 the step loop is branch-free; frames it does not cover (no primary cache)
                                 run on RT_KERNEL_PERSISTENT                                                     */
    /* 6 was RT_KERNEL_SEQ (three paths per lane, one ray slot each): retired in round 4 (ABI 1.2), rejected by rt_create */
    RT_KERNEL_FRAME = 7,      /* (ABI 1.3) the whole frame in ONE launch: waves walk the primary rays of 8x8 tiles and then every
                                 sample's path of each of their workgroup's pixels (a pixel's samples added in order by its
                                 workgroup; no prepass, worklist or accumulate launch).  What RT_KERNEL_DEFAULT runs for frames
                                 with little work: the reference's own 1024 x 1024, 1 sample, depth 2 (one-sample frames below
                                 2.5 M pixels) and multi-sample frames below 1.5 M pixel-sample-levels.  Needs
                                 RT_FLAG_CACHE_PRIMARY (or one sample per pixel), depth <= 8 and the frame's light records in one
                                 launch's array; other frames of such a context run on RT_KERNEL_PATHS / RT_KERNEL_PERSISTENT  */
} RtKernel;

#define RT_FLAG_COUNTERS      0x1u  /* count rays/iterations/hits exactly (slower; for B_alg + parity)   */
#define RT_FLAG_CACHE_PRIMARY 0x2u  /* spp>1: trace the (seed-independent) primary ray once per pixel    */
#define RT_FLAG_TIMING        0x4u  /* bracket the traversal-kernel launches with HIP events (RtTiming.trace_ms)  */
#define RT_FLAG_TIMING_ALL    0xCu  /* ... and every other launch and the frame as well (shade_ms, frame_ms); includes RT_FLAG_TIMING */
#define RT_FLAG_TRUSTED_WORLD 0x10u /* rt_upload_slice: the host vouches that every minefield value is <= 30 (the reference
                                       writes 0..6); the slab is applied without the host-side scan of its bytes        */
#define RT_FLAG_FRAMES_IN_FLIGHT_2 0x20u /* (ABI 1.2) two frames in flight: consecutive rt_draw_frame calls render into two frame
                                       slots (two sets of output planes) on two streams, so that frame k + 1's kernels take the CUs
                                       frame k's draining path kernel leaves — the reference keeps one frame in flight behind its
                                       fence (pipeline.rs:162-172), and a host that waits (rt_sync) after every frame sees no
                                       difference.  What changes for a host that does not: rt_device_ptr / rt_gbuffer_ptr return
                                       the planes of the frame drawn LAST and alternate from frame to frame (the planes of frame k
                                       stay intact until frame k + 2 is drawn); rt_readback, rt_denoise, rt_finalize and
                                       rt_gather_gbuffer act on the frame drawn last and are ordered after it; rt_sync waits for
                                       every frame.  Persistent kernels (DEFAULT / PERSISTENT / PATHS) on the context's own stream
                                       only: ignored elsewhere and after rt_set_stream(non-NULL).                          */

/*
 * RtConfig — replaces the compile-time window constants (constants.rs:9-10) and adds the
 * build's extensions: spp, depth, and the multi-GPU tile split.
 *   spp = N   : N frames with seed, seed+1, ... (mod RT_NOISE_BYTES, pipeline.rs:201);
 *               lighting = (sum_i light_i) / N in fp32; other planes from the primary hit.
 *   depth = D : light recursion truncated at D levels; D=2 is exactly raytrace.comp:321-350.
 *   tile_rank/tile_world : this context renders only the 8x8-pixel tiles t with
 *               t % tile_world == tile_rank (t = row-major tile index); outputs are then
 *               tile-major (see rt_tile_count / rt_untile).  tile_world=1 => whole frame,
 *               row-major pixel layout, row 0 = bottom of the view (finalize.comp:60-63).
 */
typedef struct RtConfig {
    uint32_t struct_size;   /* sizeof(RtConfig), for forward compatibility */
    int32_t  width;
    int32_t  height;
    int32_t  region;        /* region edge R: 256 = the reference (ROOT_BLOCK_WIDTH); 512 and 1024 are extensions (C5) */
    int32_t  spp;           /* >= 1 */
    int32_t  depth;         /* 0..RT_MAX_DEPTH */
    int32_t  device;        /* HIP device ordinal */
    int32_t  tile_rank;
    int32_t  tile_world;
    int32_t  kernel;        /* RtKernel */
    uint32_t flags;         /* RT_FLAG_* */
    int32_t  reserved[5];
} RtConfig;

/* Output planes. Bindings cited from shaders/glsl/raytrace.comp:14-21; formats from
 * src/render/pipeline/render_data.rs:166-189. */
typedef enum RtBufferId {
    RT_BUF_LIGHTING_RGBA16 = 0, /* binding 5, R16G16B16A16_UNORM: vec4(light,1)/16            8 B/px */
    RT_BUF_DEPTH_R16UI     = 1, /* binding 8, R16_UINT: uint(|origin-pos|*32) or 0xFFFF       2 B/px */
    RT_BUF_NORMAL_R8UI     = 2, /* binding 7, R8_UINT: face id 0..5 or 16                     1 B/px */
    RT_BUF_ALBEDO_RGBA8    = 3, /* binding 2, RGBA8_UNORM                                     4 B/px */
    RT_BUF_EMISSION_RGBA8  = 4, /* binding 3, RGBA8_UNORM                                     4 B/px */
    RT_BUF_FOG_RGBA8       = 5, /* binding 4, RGBA8_UNORM: sky(dir, no sun)/2                 4 B/px */
    RT_BUF_LIGHTING_F32    = 6, /* the vec4 handed to imageStore before UNORM conversion      16 B/px */
    RT_BUF_FOG_F32         = 7, /* same for fog                                               16 B/px */
    RT_BUF_DEPTH_F32       = 8, /* |origin-pos|*32 before uint(); 65535.0 for sky             4 B/px */
    RT_BUF_FINAL_BGRA8     = 9, /* rt_finalize output = the swapchain image (finalize.comp:62): B8G8R8A8_UNORM
                                   (core_builder.rs:557-568), rows TOP-down (the shader flips Y)   4 B/px */
    RT_BUF_COUNT           = 10
} RtBufferId;

/* Exact integer counters (RT_FLAG_COUNTERS), accumulated since the last rt_reset_counters.
 * B_alg = minefield_fetches*1 + material_fetches*4 + noise_fetches*4 + pixels*23. */
typedef struct RtCounters {
    uint64_t rays;               /* trace_ray invocations (primary + shadow + diffuse)     */
    uint64_t rays_primary;
    uint64_t rays_shadow;
    uint64_t rays_diffuse;
    uint64_t iterations;         /* DDA loop iterations (raytrace.comp:113)                */
    uint64_t minefield_fetches;  /* = rays + iterations (raytrace.comp:106,137)            */
    uint64_t material_fetches;   /* = hits (raytrace.comp:150-154)                         */
    uint64_t noise_fetches;      /* noise_value + seed-base lookups (raytrace.comp:302-303,324,336) */
    uint64_t hits;
    uint64_t sky_exits;
    uint64_t limit_exits;        /* Q8: loop limit reached                                 */
    uint64_t border_fetches;     /* Q7: minefield fetch outside [0,256)^3 or NaN -> 0      */
    uint64_t pixels;             /* output pixels written (x frames)                       */
    uint64_t frames;
} RtCounters;

/* HIP-event timings, milliseconds, of a context created with a timing flag (all zero without: an event between two launches is
 * 4-5 us of GPU idle time).  RT_FLAG_TIMING: trace_ms / trace_launches.  RT_FLAG_TIMING_ALL: shade_ms / other_launches and
 * frame_ms (the last rt_draw_frame, first launch to last) too.  The per-launch sums cover every frame drawn since the previous
 * rt_get_timing call. */
typedef struct RtTiming {
    float    frame_ms;        /* whole frame on the context's stream                      */
    float    trace_ms;        /* sum of traversal-kernel launches                         */
    float    shade_ms;        /* sum of ray-gen / shade / resolve launches                */
    uint32_t trace_launches;
    uint32_t other_launches;
    uint64_t rays_traced;     /* rays pushed through the traversal kernel this frame      */
} RtTiming;

typedef struct RtContext RtContext;

/* What a context allocated (no reference counterpart: Vulkan reports this through the allocator's own statistics). */
typedef struct RtInfo {
    uint32_t struct_size;               /* = sizeof(RtInfo), set by the caller                                              */
    int32_t  num_cus;                   /* CUs the persistent kernels launch on (RT_RESERVE_CUS subtracted)                 */
    uint32_t samples_per_launch;        /* samples of every pixel one path-kernel launch covers (spp / this = launches)     */
    uint16_t launches_in_flight;        /* (ABI 1.2; was reserved = 0) path launches the context keeps in flight: 2 = its launches
                                           alternate between two streams (sample batches of a frame; frames too with
                                           RT_FLAG_FRAMES_IN_FLIGHT_2), 1 otherwise                                          */
    uint16_t frames_in_flight;          /* (ABI 1.2) frame slots: 2 with RT_FLAG_FRAMES_IN_FLIGHT_2                          */
    uint64_t light_record_budget_bytes; /* what the per-path light records were sized for, all launches in flight together:
                                           min(default, free / 10) or RT_PERSIST_LIGHT_GIB                                  */
    uint64_t light_record_bytes;        /* ... and what they take (all launches in flight)                                  */
    uint64_t device_bytes;              /* all device memory the context holds                                             */
} RtInfo;

/* render::create_instance (src/render/mod.rs:36-43) + Pipeline::new (pipeline.rs:36-76):
 * create device resources for one GPU. *out is NULL on failure; rt_last_error(NULL) explains. */
int rt_create(const RtConfig* cfg, RtContext** out);

/* impl Drop for Pipeline (pipeline.rs:258-277): wait idle, free everything. NULL is a no-op. */
void rt_destroy(RtContext* ctx);

/* Message for the last failure on this context (or the last rt_create failure on this thread
 * when ctx is NULL). Never NULL. */
const char* rt_last_error(RtContext* ctx);

/* RenderData::initialize -> full-region upload (render_data.rs:269-301; formats :54-108).
 * materials: u32[R^3], minefield: u8[R^3] (R = cfg.region, 256 in the reference), both x-fastest (util.rs:104-106),
 * texel = world + R/2 (render_data.rs:221-236). The library re-tiles into 4^3 bricks on the device. */
int rt_upload_world(RtContext* ctx, const uint32_t* materials, const uint8_t* minefield);

/* TerrainUploadManager::upload_slice (terrain_upload.rs:84-275) -> vkCmdCopyBufferToImage with an
 * offset (command_buffer.rs:262-298): replace one 16-thick slab of the region.  axis 0/1/2 = x/y/z;
 * texel_offset (multiple of 16, < 256) is the slab's start along that axis; the data is a dense box of
 * extent (16,R,R) / (R,16,R) / (R,R,16), x fastest (terrain_upload.rs:96-100).  Only the slab is re-tiled on the device (one
 * launch over its 16 R^2 voxels + the nibble-map words it touches); any region size.
 * Asynchronous: the slab is copied into pinned staging (the host's buffers are free again at return), transferred on the
 * library's upload stream and applied on the context's stream after the frames already submitted — the call does not wait
 * for them (the reference does: vkQueueWaitIdle, pipeline.rs:181-189).  There are two staging sets, used in turn: the call
 * waits (on the host) only until the slab BEFORE LAST has left its pinned buffer, and that slab's transfer waited only for the
 * re-tile of the one two before it — a host that uploads one slab per frame with one or two frames in flight never waits for
 * a frame; three slabs back to back behind a long frame do wait for it.  Without RT_FLAG_TRUSTED_WORLD the minefield
 * is checked first, on the host (values above 30 -> RT_ERR_INVALID_ARG); a rejected slab is NOT applied: the region and
 * what can be drawn stay as they were. */
int rt_upload_slice(RtContext* ctx, int axis, int texel_offset,
                    const uint32_t* materials, const uint8_t* minefield);

/* The upload buffers of TerrainUploadManager::new (terrain_upload.rs:65-82) are host-visible mapped Vulkan buffers the CPU
 * fills in place; this is their counterpart: pinned host memory for ONE slab (u32[16 R^2] materials, u8[16 R^2] minefield) —
 * the staging set the NEXT rt_upload_slice will use (there are two, used in turn, so the pointers alternate from slab to slab:
 * ask again for every slab).  A host that assembles its slab there and hands these very pointers to rt_upload_slice saves the
 * copy into the staging buffer.  The memory is the host's to write from the return of this call until that rt_upload_slice
 * (the call waits until the slab before last has left the buffer); it must not be touched afterwards and is freed by rt_destroy. */
int rt_slice_staging(RtContext* ctx, uint32_t** materials, uint8_t** minefield);

/* Allocation figures of the context (see RtInfo). */
int rt_get_info(RtContext* ctx, RtInfo* out);

/* (ABI 1.2) The launch-sizing rule behind RtInfo.samples_per_launch, as a pure function (no device needed): samples of every
 * pixel one path-kernel launch covers when `lane_bytes` are given to the 12-byte light records of ONE launch (the context's
 * budget divided by its launches in flight, RtInfo.launches_in_flight) and the frame queues `npix` pixels: at most 2^31 paths,
 * at most `spp`, at least 1; `env_batch` (the RT_PERSIST_BATCH string, or NULL) may only lower it. */
uint64_t rt_samples_per_launch(uint64_t lane_bytes, uint64_t npix, uint64_t spp, const char* env_batch);

/* Blue-noise table (render_data.rs:110-133; decoded by structures.rs:496-517): RGBA8 512x512. */
int rt_upload_noise(RtContext* ctx, const uint8_t* rgba8);

/* UBO write + queue submit of the raytrace dispatch (pipeline.rs:195-211,229-235; dispatch :86-90).
 * Asynchronous: returns after enqueue on the context's stream. */
int rt_draw_frame(RtContext* ctx, const RtUniforms* uniforms);

/* Fence wait (pipeline.rs:162-172). */
int rt_sync(RtContext* ctx);

/* Copy an output plane to host memory (synchronises first). bytes must equal rt_buffer_bytes. */
int rt_readback(RtContext* ctx, int buffer_id, void* dst, size_t bytes);

/* Size in bytes of an output plane for this context (tile-major planes are padded to whole tiles). */
size_t rt_buffer_bytes(RtContext* ctx, int buffer_id);

/* Device pointer of an output plane for zero-copy consumers (denoise/finalize, RCCL gather);
 * the images bound at descriptor_sets.rs:64-84. NULL on bad id. */
void* rt_device_ptr(RtContext* ctx, int buffer_id);

/* Run on a caller-provided hipStream_t.  NULL = the context's own non-blocking stream (NOT the legacy null stream: work on
 * the null stream is not ordered against the context's frames). */
int rt_set_stream(RtContext* ctx, void* hip_stream);

/* Multi-GPU tile split: number of 8x8 tiles this context renders, and the padded
 * per-rank tile capacity ceil(total_tiles / tile_world) used for equal-size gathers. */
int rt_tile_count(RtContext* ctx);
int rt_tile_capacity(RtContext* ctx);

/* Scatter `world` gathered tile-major planes (rank-major: world x capacity x 64 px x bpp, device
 * memory) into a row-major full-frame plane (device memory) on this context's stream. */
int rt_untile(RtContext* ctx, int buffer_id, const void* gathered_dev, int world, void* frame_dev);

/* The six reference-format planes (ids 0..5) of a context are one contiguous device block (each plane padded to 256 B),
 * so a multi-GPU host gathers a frame with ONE collective: rt_gbuffer_ptr/bytes give the block, rt_gbuffer_offset the
 * start of plane `id` inside it.  rt_untile_gbuffer scatters `world` gathered blocks (rank-major, device memory) into six
 * row-major full-frame planes frames_dev[0..5] (device pointers; NULL entries are skipped).  `world` must be the
 * context's own tile_world (>= 2): the block layout is the context's. */
void*  rt_gbuffer_ptr(RtContext* ctx);
size_t rt_gbuffer_bytes(RtContext* ctx);
size_t rt_gbuffer_offset(RtContext* ctx, int buffer_id);
int rt_untile_gbuffer(RtContext* ctx, const void* gathered_dev, int world, void* const* frames_dev);

/* The six bilateral_denoise.comp dispatches recorded at pipeline.rs:98-115 (sizes 1,2,4,8,8,16, ping/pong descriptor
 * sets): filters RT_BUF_LIGHTING_RGBA16 in place using the depth and normal planes.  faithful != 0 reproduces the
 * reference's pong descriptor set, which binds the normal and depth images swapped (descriptor_sets.rs:38-39 vs :31-32), so
 * passes 2, 4 and 6 only filter pixels whose DEPTH value is below 16; faithful == 0 binds them consistently.
 * Whole-frame contexts only (tile_world == 1): a 3*16-pixel halo is needed, so gather first. Asynchronous. */
int rt_denoise(RtContext* ctx, int faithful);

/* finalize.comp (pipeline.rs:117-123): albedo*light*16 + emission*4, distance fog, tone curve, blue-noise dither, Y flip
 * -> RT_BUF_FINAL_BGRA8.  Whole-frame contexts only. Asynchronous. */
int rt_finalize(RtContext* ctx);

/* Multi-GPU frame assembly behind the boundary (the reference is single-GPU, this replaces nothing of it): every
 * rank renders the tiles t % tile_world == tile_rank of the frame (RtConfig) and calls rt_gather_gbuffer on its context: the
 * six reference-format planes of every rank travel to `root` over RCCL (grouped ncclSend/ncclRecv on the context's stream, so
 * the transfer is ordered after the frame's kernels without any host synchronisation) and the root scatters them into the six
 * row-major full-frame device planes frames_dev[0..5] (NULL entries are skipped; ignored on the other ranks; a NULL array
 * selects planes the library owns, see rt_frame_ptr).  With
 * overlapped != 0 the transfer and the un-tiling run on a second stream from a staging copy, so the caller may draw the next
 * frame at once; rt_sync waits for both streams.  tile_world == 1: plain copies.  `comm` is an ncclComm_t over the tile_world
 * ranks in tile_rank order — from the host's own RCCL, or from the helpers below (librccl is loaded on first use only):
 *   rt_comm_unique_id   ncclGetUniqueId into a 128-byte buffer (one rank; the host distributes it)
 *   rt_comm_init_rank   ncclCommInitRank(tile_world, id, tile_rank) on the context's device (one process per GPU)
 *   rt_comm_init_all    ncclCommInitAll for one process that drives ndev devices (rt_bench --gpus N)
 *   rt_comm_destroy     ncclCommDestroy */
int rt_comm_unique_id(void* id_out, size_t bytes);
int rt_comm_init_rank(RtContext* ctx, const void* id, size_t bytes, void** comm_out);
int rt_comm_init_all(int ndev, const int* devices, void** comms_out);
int rt_comm_destroy(void* comm);
int rt_gather_gbuffer(RtContext* ctx, void* comm, int root, void* const* frames_dev, int overlapped);
/* frames_dev == NULL on the root: the frame is assembled in planes the library owns (W x H, reference formats, row-major),
 * so a host needs no device allocator of its own: rt_frame_ptr returns plane `id` (0..5; NULL before the first such gather)
 * for rt_denoise_planes / rt_finalize_planes, rt_frame_readback copies it to host memory after waiting for the gather. */
void* rt_frame_ptr(RtContext* ctx, int buffer_id);
int rt_frame_readback(RtContext* ctx, int buffer_id, void* dst, size_t bytes);

/* The same two passes on caller-owned row-major device planes of cfg.width x cfg.height pixels in the reference formats
 * (e.g. the frame rt_untile_gbuffer assembled on rank 0 from the ranks' tiles): rt_denoise_planes filters `lighting_rgba16`
 * in place, rt_finalize_planes writes `out_bgra8` (4 B/px, rows top-down).  Valid on any context, tile-split or not; the
 * context supplies the stream, the blue-noise texture and the working memory.  Asynchronous. */
int rt_denoise_planes(RtContext* ctx, void* lighting_rgba16, const void* depth_r16, const void* normal_r8, int faithful);
int rt_finalize_planes(RtContext* ctx, const void* albedo_rgba8, const void* emission_rgba8, const void* fog_rgba8,
                       const void* lighting_rgba16, const void* depth_r16, void* out_bgra8);

/* The traversal implementation the context runs: after a frame, the kernel its path launches actually ran on (RT_KERNEL_DEFAULT
 * and RT_KERNEL_PATHS choose per frame: launch size, lr, primary cache, region); before the first frame, what the
 * configuration resolved to.  Negative RtStatus on a null context. */
int rt_kernel_in_use(RtContext* ctx);

/* Device self-tests.  RT_SELFTEST_DENOISE_DIVISION: the denoise passes compute weight / (distance + normal + 1)
 * (bilateral_denoise.comp:31) with a reciprocal and one residual correction; *result = number of quotients, over the complete
 * domain of that expression (37 weights x 131072 denominators), that differ from IEEE division on this device (expected 0). */
#define RT_SELFTEST_DENOISE_DIVISION 1
int rt_selftest(RtContext* ctx, int which, uint64_t* result);

int rt_get_counters(RtContext* ctx, RtCounters* out);
int rt_reset_counters(RtContext* ctx);
int rt_get_timing(RtContext* ctx, RtTiming* out);
/* (ABI 1.2) Device time of the rt_gather_gbuffer calls since the previous call of this function: *ms_sum = sum over the calls of
 * (transfer + un-tile), from HIP events recorded round them on the stream they run on, *calls = how many.  On a rank that is
 * not the root the time includes waiting for the root to post its receive.  Contexts created with RT_FLAG_TIMING; otherwise 0, 0.
 * Waits for the context's streams. */
int rt_get_gather_timing(RtContext* ctx, float* ms_sum, uint32_t* calls);

/* Library/ABI version: (major<<16)|minor.  History of the minor version (a host bound against an older header keeps working
 * within a major version; it cannot rely on what a later minor added):
 *   1.0  rounds 1-2.
 *   1.1  round 3: rt_slice_staging, rt_get_info / RtInfo; RtKernel value 4 (PERSISTENT2) rejected by rt_create; RtTiming.frame_ms
 *        is 0 unless RT_FLAG_TIMING_ALL; rt_upload_slice asynchronous, validates on the host BEFORE applying, and a rejected slab
 *        no longer makes rt_draw_frame fail.
 *   1.2  round 4: RT_FLAG_FRAMES_IN_FLIGHT_2; RtInfo.launches_in_flight / frames_in_flight (in the former `reserved` word);
 *        rt_samples_per_launch, rt_get_gather_timing; RtKernel value 6 (SEQ) rejected by rt_create; the sample batches of a multi-launch frame run on two
 *        streams of the library (results unchanged; rt_set_stream(non-NULL) keeps everything on the caller's stream).
 *   1.3  round 4: RtKernel value 7 (RT_KERNEL_FRAME); RT_KERNEL_DEFAULT runs frames with little work on it (rt_kernel_in_use tells);
 *        results unchanged. */
#define RT_ABI_VERSION_MAJOR 1
#define RT_ABI_VERSION_MINOR 3
uint32_t rt_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RT_ABI_H */
