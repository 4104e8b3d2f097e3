#!/usr/bin/env python3
"""bench.py — headline benchmark of the ray-trace path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one frame of the workload BASELINE.json's metric is quoted on: 1920x1080, spp=64, depth=4, the procedural
256^3 region (seed 0x5EED), the reference's default pose.  Inputs are resident in HBM before the timed region.
With N > 1 the frame's 8x8 tiles are dealt round-robin over the ranks (no collective while rendering) and the six
G-buffer planes are gathered to rank 0 over RCCL and un-tiled at frame end, inside the timed step — by the library
(rt_gather_gbuffer: grouped ncclSend/ncclRecv + un-tile on the context's own stream, so the transfer is ordered after the
frame's kernels by construction); torch.distributed only carries the rendez-vous (the communicator id), the barrier and the
final reductions.

Prints ONE JSON line on rank 0 (metric/value/... + "roofline" + "cpu_baseline"; with N > 1 also "c4": the 3840x2160 spp=256
depth=8 frame BASELINE.json's 8-GPU target is quoted on, timed the same way).
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

COUNTERS_FILE = "r4_counters.json"   # written by tools/pmc_to_json.py on the GPU box (tools/profile_r4.sh)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"
KERNEL_NAMES = {"paths": "k_paths", "persistent": "k_persist", "frame": "k_frame", "wavefront": "k_trace", "mega": "k_mega"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--seed", type=int, default=1, help="seed of the frame's first sample (pipeline.rs:201 starts at 1)")
    ap.add_argument("--vary-seed", action="store_true",
                    help="frame i is drawn with seed + i (every step renders a different frame; the hash is the last one's)")
    ap.add_argument("--kernel", choices=["default", "paths", "persistent", "frame", "wavefront", "mega"], default="default")
    ap.add_argument("--no-cache-primary", dest="cache_primary", action="store_false",
                    help="re-trace the (seed-independent) primary ray for every sample, like spp reference frames would")
    ap.set_defaults(cache_primary=True)
    ap.add_argument("--region", type=int, default=256, choices=[256, 512, 1024],
                    help="region edge: 256 = the reference; 1024 = the 5 GiB stress scene of config C5")
    ap.add_argument("--lr", default="0,0,0",
                    help="render offset of a scrolled region (multiples of 16 voxels, terrain_upload.rs:84-275): the scene is the "
                         "toroidal window around it and the camera moves with it — what every frame looks like once the camera has "
                         "travelled; region 256 only")
    ap.add_argument("--pose", default=None, metavar="X,Y,Z,HEADING,PITCH",
                    help="camera pose in the region's own coordinates instead of the reference's default pose scaled with the region "
                         "(e.g. the terrain-heavy C5 pose -120,-512,160,1.5707964,-0.3: tools/bench_configs.sh)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--frames-in-flight", type=int, default=2, choices=[1, 2],
                    help="2 (default): RT_FLAG_FRAMES_IN_FLIGHT_2 — the K frames are enqueued back to back and frame k+1 starts on the CUs "
                         "frame k's draining path kernel leaves; 1: one frame slot (the reference's fence discipline, pipeline.rs:162-172). "
                         "The line reports both the throughput figure and the latency of ONE frame (draw + wait).")
    ap.add_argument("--no-reference-frame", action="store_true",
                    help="N = 1: skip the \"reference_frame\" sub-record (the reference's own frame: 1024x1024, 1 spp, depth 2, + post passes)")
    ap.add_argument("--no-c4", action="store_true", help="N > 1: skip the additional 3840x2160 spp=256 depth=8 measurement")
    ap.add_argument("--share-of", default=None, metavar="r/N",
                    help="diagnostic, N = 1 only: render rank r's tiles of an N-rank run on this one GPU (tools/scale_emulation.py); "
                         "the line's value then counts that share's rays only and config.share_of says so")
    return ap.parse_args()


def cpu_baseline(mats, mine, noise, u, width, height, spp, depth, gpu_rays_per_frame, target_s=12.0, region=256):
    """The CPU oracle (a port: the reference has no CPU renderer) timed on this host's cores on a
    bounded sample of the same workload: the full frame at as many of the workload's samples as fit in ~target_s."""
    from oracle import pyoracle as po
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    _, cn = po.render(mats, mine, noise, u, width, height, 1, depth, region=region)          # calibration: 1 sample
    dt1 = max(time.perf_counter() - t0, 1e-3)
    n = int(max(1, min(spp, target_s / dt1)))
    t0 = time.perf_counter()
    _, cn = po.render(mats, mine, noise, u, width, height, n, depth, region=region)
    dt = time.perf_counter() - t0
    # equal work: the GPU metric counts a cached primary once per pixel, the oracle re-traces it for every sample — both ray
    # counts are stated so the two rates are not read as like-for-like
    return {"value": round(cn.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "oracle (scalar fp32 C++ restatement of raytrace.comp, OpenMP dynamic over rows, %d threads): %dx%d, "
                      "%d of the %d samples (seeds %d..%d), depth %d: %d rays in %.2f s (every primary ray re-traced: the whole "
                      "frame is %d oracle rays where the GPU traces %d with cached primaries)"
                      % (cores, width, height, n, spp, u.seed, u.seed + n - 1, depth, cn.rays, dt, int(cn.rays * spp / n),
                         gpu_rays_per_frame)}


def kernel_source_sha16():
    """Hash of the kernel sources: ties a committed rocprofv3 profile to the code it was taken on."""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "raytrace_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp")) and name != "rt_api.hip":      # device code only: rt_api.hip is host-side
            h.update(open(os.path.join(d, name), "rb").read())
    h.update(open(os.path.join(ROOT, "include", "rt_math.h"), "rb").read())
    return h.hexdigest()[:16]


def profile_record(kernel, workload_key):
    """Counters of the dominant kernel from the committed rocprofv3 runs (profiles/r2_counters.json, written by
    tools/pmc_to_json.py on the GPU box): None unless the profile was taken on exactly these kernel sources."""
    path = os.path.join(ROOT, "profiles", COUNTERS_FILE)
    if not os.path.exists(path):
        return None
    try:
        prof = json.load(open(path))
        rec = prof.get("workloads", {}).get(workload_key, {}).get(kernel)
        if rec is None or prof.get("kernel_source_sha16") != kernel_source_sha16():
            return None
        return rec
    except Exception:
        return None


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world
    share_rank = share_world = 0
    if args.share_of:
        if world != 1:
            sys.exit("bench.py --share-of is a one-GPU diagnostic")
        share_rank, share_world = (int(x) for x in args.share_of.split("/"))
        if not 0 <= share_rank < share_world:
            sys.exit("bench.py --share-of r/N needs 0 <= r < N")

    import torch
    import torch.distributed as dist
    from raytrace_amd import abi, build, render, world as rt_world

    # RT_BENCH_BACKEND=gloo + RT_BENCH_SINGLE_DEVICE=1 rehearse the N>1 path on a one-GPU box (every rank on GPU 0, blocks
    # staged through host memory: RCCL needs one GPU per rank); the real run uses RCCL with one GPU per rank.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if os.environ.get("RT_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    # RT_BENCH_FORCE_DIST=1 runs the N>1 code path (process group, communicator, rt_gather_gbuffer, reductions) with a single
    # rank: the only way to exercise the RCCL calls on a one-GPU box (tests/test_gpu_parity.py::test_bench_rccl_path_single_rank)
    dist_on = world > 1 or bool(os.environ.get("RT_BENCH_FORCE_DIST"))
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
        dist.barrier()
    # the native libraries travel prebuilt; if they are stale rank 0 rebuilds them before any rank loads them
    if rank == 0:
        build.build()
    if dist_on:
        dist.barrier()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rccl = dist_on and backend == "nccl"
    # Overlapped gather (RT_BENCH_OVERLAP=1): frame k's transfer + un-tile run on a second stream while frame k+1 renders.
    # The path kernels are persistent and fill every CU (1024-thread workgroups holding ~160 KiB of LDS and all 512 VGPRs of
    # each SIMD), so an RCCL kernel beside them needs CUs of its own: RT_RESERVE_CUS keeps that many out of the path kernels'
    # grids (rt_create reads it; 8 by default in this mode).  Default: serial gather on the render stream, nothing reserved
    # — until an 8-GPU run shows that the overlap wins.
    overlap = rccl and os.environ.get("RT_BENCH_OVERLAP", "0") == "1"
    if overlap and world > 1:
        os.environ.setdefault("RT_RESERVE_CUS", "8")
    reserve_cus = int(os.environ.get("RT_RESERVE_CUS", "0") or 0)

    kernel = {"default": abi.RT_KERNEL_DEFAULT, "paths": abi.RT_KERNEL_PATHS, "persistent": abi.RT_KERNEL_PERSISTENT, "frame": abi.RT_KERNEL_FRAME,
              "wavefront": abi.RT_KERNEL_WAVEFRONT, "mega": abi.RT_KERNEL_MEGA}[args.kernel]
    xflags = abi.RT_FLAG_CACHE_PRIMARY if args.cache_primary else 0
    if args.frames_in_flight == 2:
        xflags |= abi.RT_FLAG_FRAMES_IN_FLIGHT_2
    noise = np.fromfile(os.path.join(ROOT, "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
    REGION = args.region
    LR = tuple(int(v) for v in args.lr.split(","))
    if len(LR) != 3 or any(v % 16 for v in LR) or (LR != (0, 0, 0) and REGION != 256):
        raise SystemExit("--lr takes three multiples of 16 (region 256)")
    if LR == (0, 0, 0):
        mats, mine = rt_world.generate_region(rt_world.DEFAULT_SEED, region=REGION)
    else:
        mats, mine = rt_world.toroidal_region(LR, rt_world.DEFAULT_SEED, region=REGION)
    pose = dict(render.DEFAULT_POSE)
    scale = REGION // 256            # C5 pose (-120,-512,400) = the default pose scaled with the region
    pose["origin"] = tuple(c * scale + o for c, o in zip(pose["origin"], LR))
    if args.pose:
        pv = [float(v) for v in args.pose.split(",")]
        pose["origin"], pose["heading"], pose["pitch"] = tuple(pv[:3]), pv[3], pv[4]

    def uniforms(seed):
        return render.camera_uniforms(pose["origin"], pose["heading"], pose["pitch"], pose["sun_angle"], seed=seed, lr=LR)

    uid_bytes = None
    if rccl:
        # the communicator the library gathers over: rank 0 creates the id, torch.distributed hands it round
        uid = torch.zeros(128, dtype=torch.uint8, device=dev)
        if rank == 0:
            uid.copy_(torch.frombuffer(bytearray(render.comm_unique_id()), dtype=torch.uint8))
        dist.broadcast(uid, src=0)
        uid_bytes = uid.cpu().numpy().tobytes()
    shared = {"comm": None}

    gather_ids = [abi.RT_BUF_LIGHTING_RGBA16, abi.RT_BUF_DEPTH_R16UI, abi.RT_BUF_NORMAL_R8UI, abi.RT_BUF_ALBEDO_RGBA8,
                  abi.RT_BUF_EMISSION_RGBA8, abi.RT_BUF_FOG_RGBA8]
    bpp = {abi.RT_BUF_LIGHTING_RGBA16: 8, abi.RT_BUF_DEPTH_R16UI: 2, abi.RT_BUF_NORMAL_R8UI: 1, abi.RT_BUF_ALBEDO_RGBA8: 4,
           abi.RT_BUF_EMISSION_RGBA8: 4, abi.RT_BUF_FOG_RGBA8: 4}

    def measure(W, H, SPP, D, steps, warmup):
        """Time `steps` frames of one workload (after `warmup` untimed ones) on this rank's share of the tiles; returns the
        rank's record (elapsed, launch timings, exact counts), the whole-job figures and rank 0's frame hash."""

        def make_ctx(flags, depth=D):
            cfg = render.make_config(W, H, spp=SPP, depth=depth, device=local_rank, tile_rank=share_rank if share_world else rank, tile_world=share_world or world,
                                     kernel=kernel, flags=flags | xflags, region=REGION)
            ctx = render.Context(cfg)
            ctx.upload_world(mats, mine)
            ctx.upload_noise(noise)
            return ctx

        u0 = uniforms(args.seed)
        rec = {}
        # ---- exact ray / byte counts of one frame (deterministic; outside the timed region) --------------------
        cctx = make_ctx(abi.RT_FLAG_COUNTERS)
        cctx.draw_frame(u0)
        cctx.sync()
        # the report names the kernel the frame actually ran on (RT_KERNEL_DEFAULT picks k_paths / k_persist per frame)
        rec["kernel"] = {abi.RT_KERNEL_PATHS: "paths", abi.RT_KERNEL_PERSISTENT: "persistent", abi.RT_KERNEL_FRAME: "frame",
                         abi.RT_KERNEL_WAVEFRONT: "wavefront", abi.RT_KERNEL_MEGA: "mega"}[cctx.kernel_in_use()]
        cn = cctx.counters()
        cctx_kernel = rec["kernel"]
        cctx.destroy()
        rec["rays"] = cn.rays
        # traversal algorithmic bytes: 1 B per minefield fetch + 4 B per material fetch
        # (k_frame's one launch is the whole frame: primary rays, noise and the G-buffer stores are its algorithmic bytes too)
        rec["trace_bytes"] = cn.algorithmic_bytes() if cctx_kernel == "frame" else cn.minefield_fetches + 4 * cn.material_fetches
        rec["balg"] = cn.algorithmic_bytes()
        rec["ref_rays"] = cn.rays + (SPP - 1) * cn.pixels if args.cache_primary else cn.rays
        if rec["kernel"] in ("paths", "persistent") and args.cache_primary and D >= 1:
            # the dominant kernel walks only shadow/diffuse rays; the primary prepass (k_primary2) is a separate launch,
            # untimed for the roofline: subtract its share, measured with a depth-0 counting frame
            c0 = make_ctx(abi.RT_FLAG_COUNTERS, depth=0)
            c0.draw_frame(u0)
            c0.sync()
            cn0 = c0.counters()
            c0.destroy()
            rec["trace_bytes"] -= cn0.minefield_fetches + 4 * cn0.material_fetches

        ctx = make_ctx(abi.RT_FLAG_TIMING)      # renders on the context's own stream; so does rt_gather_gbuffer
        inf = ctx.info()
        rec["samples_per_launch"], rec["light_record_bytes"], rec["light_budget_bytes"] = inf.samples_per_launch, inf.light_record_bytes, inf.light_record_budget_bytes
        rec["device_bytes"] = inf.device_bytes
        rec["launches_in_flight"], rec["frames_in_flight"] = inf.launches_in_flight, inf.frames_in_flight
        frames = None
        if dist_on and rank == 0:
            frames = {b: torch.empty(W * H * bpp[b], dtype=torch.uint8, device=dev) for b in gather_ids}
        if rccl and shared["comm"] is None:
            shared["comm"] = ctx.comm_init_rank(uid_bytes)
        comm = shared["comm"]
        if dist_on and not rccl:
            gbytes = ctx.gbuffer_bytes()
            gathered = torch.empty(world * gbytes, dtype=torch.uint8, device=dev) if rank == 0 else None
        torch.cuda.synchronize(dev)
        state = {"frame": 0}

        def step(fixed=False):
            u = uniforms(args.seed + (state["frame"] if args.vary_seed and not fixed else 0))
            if not fixed:
                state["frame"] += 1
            ctx.draw_frame(u)
            if rccl:
                ctx.gather_gbuffer(comm, 0, [frames[b].data_ptr() for b in gather_ids] if rank == 0 else None, overlapped=overlap)
            elif dist_on:      # rehearsal backend (gloo): blocks staged through host memory, un-tiled by the library on rank 0
                ctx.sync()
                # (the block of the frame drawn last: with two frame slots the pointer alternates)
                host = torch.as_tensor(_DevArray(ctx.gbuffer_ptr(), gbytes), device=dev).cpu()
                if rank == 0:
                    parts = [torch.empty_like(host) for _ in range(world)]
                    dist.gather(host, parts, dst=0)
                    gathered.copy_(torch.cat(parts))
                    torch.cuda.synchronize(dev)
                    if world > 1:
                        ctx.untile_gbuffer(gathered.data_ptr(), world, [frames[b].data_ptr() for b in gather_ids])
                    else:
                        for b in gather_ids:
                            off = ctx.gbuffer_offset(b)
                            frames[b].copy_(gathered[off:off + frames[b].numel()])
                else:
                    dist.gather(host, None, dst=0)

        def fence():
            ctx.sync()                 # the render stream and, in the overlapped mode, the gather stream
            if dist_on:
                dist.barrier()
            torch.cuda.synchronize(dev)

        for _ in range(warmup):
            step()
        fence()
        # latency of ONE frame — draw, (gather,) wait, nothing else in flight: the reference's per-frame fence (pipeline.rs:162-172).
        # Before the timed region (so that the last timed frame stays the frame that is hashed), with the first frame's seed.
        lat = []
        if warmup > 0:
            ctx.timing()
        for _ in range(3 if warmup > 0 else 0):
            t1 = time.perf_counter()
            step(fixed=True)
            ctx.sync()
            lat.append(time.perf_counter() - t1)
            fence()
        rec["latency_ms"] = sorted(lat)[len(lat) // 2] * 1e3 if lat else None
        # the path kernel's launches with nothing else in flight (their events during the latency frames): with two launches in
        # flight a launch's own begin-to-end time also holds the time it waits for the CUs of the launch in front of it
        rec["alone_trace_ms"], rec["alone_trace_launches"] = (lambda t: (t.trace_ms, t.trace_launches))(ctx.timing()) if lat else (0.0, 0)
        if warmup > 0:
            ctx.timing()      # drop the warm-up frames' launch events
        ctx.gather_timing()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        rec["elapsed"] = time.perf_counter() - t0
        # per-launch HIP events of the K timed frames (recorded on the stream the kernels run on; read after the fence)
        tm = ctx.timing()
        rec["trace_ms"], rec["trace_launches"] = tm.trace_ms, tm.trace_launches
        gt = ctx.gather_timing()
        rec["gather_ms"], rec["gathers"] = gt
        # content hash of the finished (last) frame on rank 0, outside the timed region: equal for every N
        sha = None
        if rank == 0:
            hsh = hashlib.sha256()
            for b in gather_ids:
                hsh.update(frames[b].cpu().numpy().tobytes() if dist_on else ctx.readback(b).tobytes())
            sha = hsh.hexdigest()[:16]
        ctx.destroy()
        # whole-job figures: MAX of the ranks' times, SUM of their counts
        rdev = dev if (backend == "nccl" or not dist_on) else torch.device("cpu")
        t_all = torch.tensor([rec["elapsed"]], dtype=torch.float64, device=rdev)
        sums = torch.tensor([float(rec["rays"]), float(rec["trace_bytes"]), float(rec["balg"]), float(rec["ref_rays"])],
                            dtype=torch.float64, device=rdev)
        if dist_on:
            dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
            dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        rec["elapsed_max"] = float(t_all.item())
        # per-rank spread (the elapsed time ends at a barrier and is the same everywhere; the path kernels' own time and the
        # ray counts are what differ between ranks): [min, max] over the ranks
        spread = torch.tensor([rec["trace_ms"], -rec["trace_ms"], float(rec["rays"]), -float(rec["rays"])], dtype=torch.float64, device=rdev)
        if dist_on:
            dist.all_reduce(spread, op=dist.ReduceOp.MAX)
        sp = [float(x) for x in spread.tolist()]
        rec["rank_trace_ms"] = [-sp[1], sp[0]]
        rec["rank_rays"] = [int(-sp[3]), int(sp[2])]
        rec["rays_total"], rec["trace_bytes_total"], rec["balg_total"], rec["ref_rays_total"] = [float(x) for x in sums.tolist()]
        rec["sha"] = sha
        return rec

    def roofline_of(r, steps, W_, H_, SPP_, D_):
        """Roofline record of the dominant kernel on THIS rank: algorithmic bytes of its launches / their duration (HIP events on
        the stream the kernel runs on).  Fabric/HBM bytes and SQ counters come from the committed rocprofv3 passes of the same
        command (separate --pmc runs cannot share a process with the timed run); they are attached only when taken on exactly
        these kernel sources and — being whole-frame figures — only at N = 1."""
        kname = KERNEL_NAMES[r["kernel"]]
        launches_per_frame = r["trace_launches"] // max(steps, 1)
        avg_launch_ms = r["trace_ms"] / max(r["trace_launches"], 1)
        achieved = (r["trace_bytes"] * steps) / (r["trace_ms"] * 1e-3) / 1e9 if r["trace_ms"] > 0 else 0.0
        roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": None,
                    "launches_per_frame": launches_per_frame, "avg_launch_ms": round(avg_launch_ms, 4),
                    "algorithmic_bytes_per_launch": int(r["trace_bytes"] / max(launches_per_frame, 1)),
                    "samples_per_launch": int(r["samples_per_launch"]), "launches_in_flight": int(r["launches_in_flight"])}
        if int(r["launches_in_flight"]) > 1 and r.get("elapsed", 0) > 0 and world == 1:
            # Two launches in flight: the launches' begin-to-end times overlap (each includes its wait for the CUs the launch in front
            # of it is still leaving), so their sum exceeds the wall clock and bytes / duration counts that time twice.  The bandwidth
            # the kernel's launches sustained = their bytes / the time they took TOGETHER; taken here as the elapsed time of the whole
            # timed region (prepass, accumulate and gaps included: a lower bound).  The contract's literal figure stays beside it.
            span = (r["trace_bytes"] * steps) / r["elapsed"] / 1e9
            roofline["per_launch_duration"] = {"achieved": roofline["achieved"], "frac": roofline["frac"],
                                               "note": "bytes / average begin-to-end time of a launch; launches overlap"}
            roofline["achieved"], roofline["frac"] = round(span, 2), round(span / HBM_PEAK_GBS, 5)
            roofline["basis"] = "launches overlap (launches_in_flight = 2): bytes of the timed launches / elapsed time of the timed region"
        if r.get("alone_trace_launches"):
            # `achieved` / `frac` follow the contract (launch duration over the timed region: with two launches in flight a launch's
            # begin-to-end time includes its slow start on the CUs the launch in front of it is still leaving); the same kernel with
            # nothing else in flight, from the one-frame latency runs:
            alone_ms = r["alone_trace_ms"] / r["alone_trace_launches"]
            ach1 = r["trace_bytes"] / max(launches_per_frame, 1) / (alone_ms * 1e-3) / 1e9
            roofline["one_launch_in_flight"] = {"avg_launch_ms": round(alone_ms, 4), "achieved": round(ach1, 2), "frac": round(ach1 / HBM_PEAK_GBS, 5)}
        if world > 1:
            roofline["rank"] = 0
            roofline["ranks_path_kernel_ms_per_frame"] = {"min": round(r["rank_trace_ms"][0] / max(steps, 1), 4), "max": round(r["rank_trace_ms"][1] / max(steps, 1), 4)}
            roofline["ranks_rays_per_frame"] = {"min": r["rank_rays"][0], "max": r["rank_rays"][1]}
        wkey = "%dx%d spp=%d depth=%d region=%d" % (W_, H_, SPP_, D_, REGION)
        prof = profile_record(kname, wkey) if world == 1 else None
        if prof:
            roofline["traffic"] = prof.get("hbm_bytes_per_launch")
            roofline["traffic_source"] = ("profiles/%s: rocprofv3 FETCH_SIZE x2 + WRITE_SIZE per launch (separate --pmc "
                                          "passes of this command, taken on kernel sources %s)" % (COUNTERS_FILE, kernel_source_sha16()))
            if prof.get("hbm_bytes_per_launch") and avg_launch_ms > 0:
                gbs = prof["hbm_bytes_per_launch"] / (avg_launch_ms * 1e-3) / 1e9
                roofline["measured_hbm"] = {"GB/s": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 5),
                                            "bytes_per_launch": prof["hbm_bytes_per_launch"], "l2_hit_rate": prof.get("l2_hit_rate")}
            if "valu" in prof:
                roofline["valu"] = prof["valu"]
            # the launch sizing depends on the memory free at rt_create; say so if this run's differs from the profiled one
            if prof.get("samples_per_launch") not in (None, int(r["samples_per_launch"])):
                roofline["profile_samples_per_launch_differs"] = {"profile": prof.get("samples_per_launch"), "run": int(r["samples_per_launch"])}
        return roofline

    def reference_frame():
        """The reference's own frame through the drop-in entry point: 1024x1024, 1 spp, depth 2
        (constants.rs:9-10) — Pipeline::draw_frame of the C++ mirror with the post passes enabled enqueues what the reference
        records into its one command buffer (pipeline.rs:86-123): ray trace, six denoise dispatches, finalize.  Wall clock per
        frame, frames enqueued back to back behind the mirror's per-frame fence (one frame in flight, as the reference), and the
        stages by difference (ray trace only / + denoise / + finalize)."""
        g = render.Game()
        g.generate_world(rt_world.DEFAULT_SEED)
        out = {"workload": "1024x1024 spp=1 depth=2 + denoise x6 + finalize (the reference's frame, pipeline.rs:86-123)", "frames": 200}

        def run(post, frames=200):
            cfg = render.make_config(1024, 1024, spp=1, depth=2, device=local_rank, kernel=kernel, flags=abi.RT_FLAG_CACHE_PRIMARY)
            pipe = render.create_instance(cfg, g, noise)
            if post:
                pipe.enable_post_passes(faithful=True)
            for _ in range(10):
                pipe.draw_frame(g)
            pipe.wait()
            t0 = time.perf_counter()
            for _ in range(frames):
                pipe.draw_frame(g)       # waits for the previous frame first (pipeline.rs:162-172)
            pipe.wait()
            ms = (time.perf_counter() - t0) * 1e3 / frames
            sha = hashlib.sha256(pipe.context.readback(abi.RT_BUF_FINAL_BGRA8).tobytes()).hexdigest()[:16] if post else None
            pipe.close()
            return ms, sha

        rt_ms, _ = run(False)
        full_ms, sha = run(True)
        out["ms_per_frame"] = round(full_ms, 4)
        out["raytrace_ms"] = round(rt_ms, 4)
        out["post_passes_ms"] = round(full_ms - rt_ms, 4)
        out["final_image_sha256_16"] = sha
        # the same frames through the C ABI without the mirror's fence: two frames in flight, post passes on the frame's own stream
        cfg = render.make_config(1024, 1024, spp=1, depth=2, device=local_rank, kernel=kernel, flags=abi.RT_FLAG_CACHE_PRIMARY | abi.RT_FLAG_FRAMES_IN_FLIGHT_2)
        with render.Context(cfg) as c2:
            c2.upload_world(mats, mine)
            c2.upload_noise(noise)
            u1 = render.camera_uniforms(render.DEFAULT_POSE["origin"], render.DEFAULT_POSE["heading"], render.DEFAULT_POSE["pitch"], render.DEFAULT_POSE["sun_angle"], seed=1)
            for post in (False, True):
                for it in range(2):
                    t0 = time.perf_counter()
                    for _ in range(200):
                        c2.draw_frame(u1)
                        if post:
                            c2.denoise(True)
                            c2.finalize()
                    c2.sync()
                    ms = (time.perf_counter() - t0) * 1e3 / 200
                out["two_in_flight_%s_ms" % ("full" if post else "raytrace")] = round(ms, 4)
        g.close()
        return out

    W, H, SPP, D = args.width, args.height, args.spp, args.depth
    rec = measure(W, H, SPP, D, args.steps, args.warmup)
    c4 = None
    if world > 1 and not args.no_c4 and REGION == 256:
        # the configuration BASELINE.json's ">= 6x at 8 GPUs" is quoted on
        r4 = measure(3840, 2160, 256, 8, 3, 1)
        c4 = {"workload": "3840x2160 spp=256 depth=8", "steps": 3, "warmup": 1, "ms_per_step": round(r4["elapsed_max"] / 3 * 1e3, 3),
              "value": round(r4["rays_total"] * 3 / r4["elapsed_max"] / 1e6, 2), "unit": "Mrays/s", "rays_per_frame": int(r4["rays_total"]),
              "frame_sha256_16": r4["sha"]}
        if rank == 0:
            c4["roofline"] = roofline_of(r4, 3, 3840, 2160, 256, 8)

    if rank == 0:
        elapsed = rec["elapsed_max"]
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rec["rays_total"] * args.steps / elapsed / 1e6
        roofline = roofline_of(rec, args.steps, W, H, SPP, D)
        out = {
            "metric": "Mrays/s at %dx%d spp=%d (rays actually traced: primary + shadow + diffuse)" % (W, H, SPP),
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d spp=%d depth=%d, procedural %d^3 region seed 0x5EED, pose (%g,%g,%g) h=%s p=%g sun=0"
                                   % ((W, H, SPP, D, REGION) + tuple(pose["origin"]) + ("pi/2" if not args.pose else "%g" % pose["heading"], pose["pitch"])),
                       "kernel": rec["kernel"],
                       "rays_per_frame": int(rec["rays_total"]), "reference_equivalent_rays_per_frame": int(rec["ref_rays_total"]),
                       "algorithmic_bytes_per_frame": int(rec["balg_total"]), "parallelism": "tiles%d" % world, **({"share_of": args.share_of} if args.share_of else {}),
                       "primary_cache": bool(args.cache_primary), "lr": list(LR), "frame_sha256_16": rec["sha"],
                       "seed": args.seed + ((args.steps + args.warmup - 1) if args.vary_seed else 0),
                       "gather": None if not dist_on else (("rt_gather_gbuffer over RCCL, " + ("overlapped with the next frame" if overlap
                                                                                              else "serial on the render stream"))
                                                           if rccl else "host-staged (%s rehearsal)" % backend),
                       "reserve_cus": reserve_cus,
                       # frames the K timed steps keep in flight (frame slots) and path launches in flight (streams the library
                       # alternates its launches between); ms_per_step is the throughput figure, latency_ms_one_frame = draw + wait alone
                       "frames_in_flight": int(rec["frames_in_flight"]), "launches_in_flight": int(rec["launches_in_flight"]),
                       "latency_ms_one_frame": None if rec["latency_ms"] is None else round(rec["latency_ms"], 4),
                       # footprint of rank 0's context: samples one path-kernel launch covers and what its light records take
                       # (sized for min(16 GiB, a tenth of the free memory) unless RT_PERSIST_LIGHT_GIB says otherwise)
                       "samples_per_launch": int(rec["samples_per_launch"]), "light_record_bytes": int(rec["light_record_bytes"]),
                       "light_record_budget_bytes": int(rec["light_budget_bytes"]), "context_device_bytes": int(rec["device_bytes"])},
            "roofline": roofline,
        }
        if c4 is not None:
            out["c4"] = c4
        if dist_on and rec["gathers"]:
            # rank 0's rt_gather_gbuffer calls alone (events round transfer + un-tile on their stream): separates render, drain and gather
            out["gather"] = {"ms_per_frame": round(rec["gather_ms"] / rec["gathers"], 4), "calls": int(rec["gathers"]), "rank": 0,
                             "bytes_to_root_per_frame": int(W * H * 23)}
        if world == 1 and not dist_on and not args.no_reference_frame and not args.share_of and REGION == 256 and LR == (0, 0, 0):
            out["reference_frame"] = reference_frame()
        if not args.no_cpu_baseline:
            # rank 0's host cores, after the timed region (the other ranks wait at the closing barrier); a shorter sample at N > 1
            out["cpu_baseline"] = cpu_baseline(mats, mine, noise, uniforms(args.seed), W, H, SPP, D, int(rec["rays_total"]),
                                               target_s=12.0 if world == 1 else 6.0, region=REGION)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    if shared["comm"]:
        render.comm_destroy(shared["comm"])
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


class _DevArray:
    """Zero-copy view of a device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


if __name__ == "__main__":
    main()
