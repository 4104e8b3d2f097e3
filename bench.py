#!/usr/bin/env python3
"""bench.py — headline benchmark of the ray-trace path on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = one frame of the workload BASELINE.json's metric is quoted on: 1920x1080, spp=64, depth=4, the procedural
256^3 region (seed 0x5EED), the reference's default pose.  Inputs are resident in HBM before the timed region.
With N > 1 the frame's 8x8 tiles are dealt round-robin over the ranks (no collective while rendering) and the six
G-buffer planes are gathered to rank 0 over RCCL and un-tiled at frame end, inside the timed step.

Prints ONE JSON line on rank 0 (metric/value/... + "roofline" + "cpu_baseline").
"""
import argparse
import ctypes
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "Chip-level parameters"


class _DevArray:
    """Zero-copy view of a device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--depth", type=int, default=4)
    ap.add_argument("--kernel", choices=["default", "paths", "persistent", "persistent2", "wavefront", "mega"], default="default")
    ap.add_argument("--no-cache-primary", dest="cache_primary", action="store_false",
                    help="re-trace the (seed-independent) primary ray for every sample, like spp reference frames would")
    ap.set_defaults(cache_primary=True)
    ap.add_argument("--region", type=int, default=256, choices=[256, 512, 1024],
                    help="region edge: 256 = the reference; 1024 = the 5 GiB stress scene of config C5")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    return ap.parse_args()


def cpu_baseline(mats, mine, noise, u, width, height, spp, depth, target_s=12.0, region=256):
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
    return {"value": round(cn.rays / dt / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "oracle (scalar fp32 C++ restatement of raytrace.comp, OpenMP dynamic over rows, %d threads): %dx%d, "
                      "%d of the %d samples (seeds %d..%d), depth %d: %d rays in %.2f s (every primary ray re-traced)"
                      % (cores, width, height, n, spp, u.seed, u.seed + n - 1, depth, cn.rays, dt)}


def main():
    args = parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus %d must be launched with torch.distributed.run (one rank per GPU)" % args.gpus)
        args.gpus = world

    import torch
    import torch.distributed as dist
    from raytrace_amd import abi, build, render, world as rt_world

    # RT_BENCH_BACKEND=gloo + RT_BENCH_SINGLE_DEVICE=1 rehearse the N>1 path on a one-GPU box (every rank on GPU 0, planes
    # staged through host memory for the gather); the real run uses RCCL ("nccl") with one GPU per rank.
    backend = os.environ.get("RT_BENCH_BACKEND", "nccl")
    if os.environ.get("RT_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    # RT_BENCH_FORCE_DIST=1 runs the N>1 code path (process group, gather, reductions) with a single rank: the only way
    # to exercise the RCCL calls on a one-GPU box (tests/test_gpu_parity.py::test_bench_rccl_path_single_rank)
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

    W, H, SPP, D = args.width, args.height, args.spp, args.depth
    kernel = {"default": abi.RT_KERNEL_DEFAULT, "paths": abi.RT_KERNEL_PATHS, "persistent": abi.RT_KERNEL_PERSISTENT, "persistent2": abi.RT_KERNEL_PERSISTENT2,
              "wavefront": abi.RT_KERNEL_WAVEFRONT, "mega": abi.RT_KERNEL_MEGA}[args.kernel]
    xflags = abi.RT_FLAG_CACHE_PRIMARY if args.cache_primary else 0
    noise = np.fromfile(os.path.join(ROOT, "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
    REGION = args.region
    mats, mine = rt_world.generate_region(rt_world.DEFAULT_SEED, region=REGION)
    pose = dict(render.DEFAULT_POSE)
    scale = REGION // 256            # C5 pose (-120,-512,400) = the default pose scaled with the region
    pose["origin"] = tuple(c * scale for c in pose["origin"])
    u = render.camera_uniforms(pose["origin"], pose["heading"], pose["pitch"], pose["sun_angle"], seed=1)

    def make_ctx(flags):
        cfg = render.make_config(W, H, spp=SPP, depth=D, device=local_rank, tile_rank=rank, tile_world=world,
                                 kernel=kernel, flags=flags | xflags, region=REGION)
        ctx = render.Context(cfg)
        ctx.upload_world(mats, mine)
        ctx.upload_noise(noise)
        return ctx

    # ---- exact ray / byte counts of one frame (deterministic; outside the timed region) --------------------
    cctx = make_ctx(abi.RT_FLAG_COUNTERS)
    # the report names the kernel the library actually runs (RT_KERNEL_DEFAULT picks by workload size, rt_create)
    args.kernel = {abi.RT_KERNEL_PATHS: "paths", abi.RT_KERNEL_PERSISTENT: "persistent", abi.RT_KERNEL_PERSISTENT2: "persistent2",
                   abi.RT_KERNEL_WAVEFRONT: "wavefront", abi.RT_KERNEL_MEGA: "mega"}[cctx.kernel_in_use()]
    cctx.draw_frame(u)
    cctx.sync()
    cn = cctx.counters()
    cctx.destroy()
    rays_local = cn.rays
    # traversal algorithmic bytes: 1 B per minefield fetch + 4 B per material fetch
    trace_bytes_local = cn.minefield_fetches + 4 * cn.material_fetches
    balg_local = cn.algorithmic_bytes()
    ref_equiv_rays_local = cn.rays + (SPP - 1) * cn.pixels if args.cache_primary else cn.rays
    if args.kernel in ("paths", "persistent", "persistent2") and args.cache_primary and D >= 1:
        # the dominant kernel (k_persist) walks only shadow/diffuse rays; the primary prepass (k_primary) is a separate,
        # untimed-for-roofline launch: subtract its share, measured with a depth-0 counting frame
        cfg0 = render.make_config(W, H, spp=SPP, depth=0, device=local_rank, tile_rank=rank, tile_world=world,
                                  kernel=kernel, flags=abi.RT_FLAG_COUNTERS | xflags, region=REGION)
        c0 = render.Context(cfg0)
        c0.upload_world(mats, mine)
        c0.upload_noise(noise)
        c0.draw_frame(u)
        c0.sync()
        cn0 = c0.counters()
        c0.destroy()
        trace_bytes_local -= cn0.minefield_fetches + 4 * cn0.material_fetches

    # RCCL gather overlapped with the next frame (RT_BENCH_OVERLAP=0: serial).  The path kernels are persistent and fill
    # every CU with a 1024-thread workgroup holding 130+ KiB of LDS, so a concurrent RCCL kernel is given CUs of its own:
    # RT_RESERVE_CUS keeps that many out of the path kernels' grids (rt_create reads it).
    overlap = dist_on and backend == "nccl" and os.environ.get("RT_BENCH_OVERLAP", "1") != "0"
    if overlap and world > 1:
        os.environ.setdefault("RT_RESERVE_CUS", "8")
    ctx = make_ctx(abi.RT_FLAG_TIMING)
    stream = torch.cuda.current_stream(dev)
    ctx.set_stream(stream.cuda_stream)

    gather_ids = [abi.RT_BUF_LIGHTING_RGBA16, abi.RT_BUF_DEPTH_R16UI, abi.RT_BUF_NORMAL_R8UI, abi.RT_BUF_ALBEDO_RGBA8,
                  abi.RT_BUF_EMISSION_RGBA8, abi.RT_BUF_FOG_RGBA8]
    bpp = {abi.RT_BUF_LIGHTING_RGBA16: 8, abi.RT_BUF_DEPTH_R16UI: 2, abi.RT_BUF_NORMAL_R8UI: 1, abi.RT_BUF_ALBEDO_RGBA8: 4,
           abi.RT_BUF_EMISSION_RGBA8: 4, abi.RT_BUF_FOG_RGBA8: 4}
    if dist_on:
        # one collective per frame: the six planes are one contiguous block on every rank (rt_gbuffer_ptr)
        gbytes = ctx.gbuffer_bytes()
        local_view = torch.as_tensor(_DevArray(ctx.gbuffer_ptr(), gbytes), device=dev)
        gathered = torch.empty(world * gbytes, dtype=torch.uint8, device=dev) if rank == 0 else None
        frames = {b: torch.empty(W * H * bpp[b], dtype=torch.uint8, device=dev) if rank == 0 else None for b in gather_ids}
    if overlap:
        # Pipelined gather (RCCL only): frame k's block is copied to one of two staging tensors and gathered asynchronously
        # while frame k+1 renders; its scatter into the row-major frame runs one step later.  The fence flushes the pipe.
        stage = [torch.empty(gbytes, dtype=torch.uint8, device=dev) for _ in range(2)]
        gathered2 = [torch.empty(world * gbytes, dtype=torch.uint8, device=dev) if rank == 0 else None for _ in range(2)]
        pending = [None, None]
        pipe = {"frame": 0, "open": None}

    def scatter(block):
        if rank == 0 and world > 1:
            ctx.untile_gbuffer(block.data_ptr(), world, [frames[b].data_ptr() for b in gather_ids])
        elif rank == 0:   # forced single-rank run: the block already holds row-major planes
            for b in gather_ids:
                off = ctx.gbuffer_offset(b)
                frames[b].copy_(block[off:off + frames[b].numel()])

    def finish(slot):
        pending[slot].wait()          # the current stream waits for that gather (no host block)
        pending[slot] = None
        scatter(gathered2[slot])

    def flush():
        if overlap and pipe["open"] is not None:
            finish(pipe["open"])
            pipe["open"] = None

    def step():
        ctx.draw_frame(u)
        if overlap:
            s = pipe["frame"] & 1
            stage[s].copy_(local_view)           # after the frame on the current stream; frees the context's block
            pending[s] = dist.gather(stage[s], list(gathered2[s].chunk(world)) if rank == 0 else None, dst=0, async_op=True)
            if pipe["open"] is not None:
                finish(pipe["open"])             # the previous frame: its gather had this frame's render time to complete
            pipe["open"] = s
            pipe["frame"] += 1
        elif dist_on:
            if backend == "nccl":
                if rank == 0:
                    dist.gather(local_view, list(gathered.chunk(world)), dst=0)
                else:
                    dist.gather(local_view, None, dst=0)
            else:   # rehearsal backend: stage through host memory
                host = local_view.cpu()
                if rank == 0:
                    parts = [torch.empty_like(host) for _ in range(world)]
                    dist.gather(host, parts, dst=0)
                    gathered.copy_(torch.cat(parts))
                else:
                    dist.gather(host, None, dst=0)
            scatter(gathered)

    def fence():
        flush()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    if args.warmup > 0:
        ctx.timing()      # drop the warm-up frames' launch events
    trace_ms = 0.0
    trace_launches = 0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    # per-launch HIP events of the K timed frames (recorded on the stream the kernels run on; read after the fence)
    tm = ctx.timing()
    trace_ms += tm.trace_ms
    trace_launches += tm.trace_launches

    t_all = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    sums = torch.tensor([float(rays_local), float(trace_bytes_local), float(balg_local), float(ref_equiv_rays_local)],
                        dtype=torch.float64, device=dev)
    if dist_on:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
    elapsed = float(t_all.item())
    rays_total, trace_bytes_total, balg_total, ref_rays_total = [float(x) for x in sums.tolist()]

    # content hash of the finished frame on rank 0 (outside the timed region): equal for every N
    frame_sha = None
    if rank == 0:
        import hashlib
        hsh = hashlib.sha256()
        for b in gather_ids:
            if dist_on:
                hsh.update(frames[b].cpu().numpy().tobytes())
            else:
                hsh.update(ctx.readback(b).tobytes())
        frame_sha = hsh.hexdigest()[:16]
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        mrays = rays_total * args.steps / elapsed / 1e6
        # roofline of the dominant kernel (k_trace) on this rank: algorithmic bytes of its launches / their duration
        achieved = (trace_bytes_local * args.steps) / (trace_ms * 1e-3) / 1e9 if trace_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_r1.json")
        if os.path.exists(tpath) and (W, H, SPP, D, REGION) == (1920, 1080, 64, 4, 256) and world == 1:
            try:
                traffic = json.load(open(tpath)).get("k_%s_hbm_bytes_per_launch" % {"paths": "paths", "persistent": "persist", "persistent2": "persist2", "wavefront": "trace", "mega": "mega"}[args.kernel])
            except Exception:
                traffic = None
        out = {
            "metric": "Mrays/s at %dx%d spp=%d (rays actually traced: primary + shadow + diffuse)" % (W, H, SPP),
            "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d spp=%d depth=%d, procedural %d^3 region seed 0x5EED, pose (%g,%g,%g) h=pi/2 p=0 sun=0"
                                   % ((W, H, SPP, D, REGION) + tuple(pose["origin"])), "kernel": args.kernel, "rays_per_frame": int(rays_total), "reference_equivalent_rays_per_frame": int(ref_rays_total),
                       "algorithmic_bytes_per_frame": int(balg_total), "parallelism": "tiles%d" % world,
                       "primary_cache": bool(args.cache_primary), "frame_sha256_16": frame_sha,
                       "gather": None if not dist_on else ("overlapped with the next frame" if overlap else "serial")},
            "roofline": {"bound": "hbm", "kernel": {"paths": "k_paths", "persistent": "k_persist", "persistent2": "k_persist2", "wavefront": "k_trace", "mega": "k_mega"}[args.kernel], "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "launches_per_frame": trace_launches // max(args.steps, 1),
                         "avg_launch_ms": round(trace_ms / max(trace_launches, 1), 4),
                         "algorithmic_bytes_per_launch": int(trace_bytes_local / max(trace_launches // max(args.steps, 1), 1))},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(mats, mine, noise, u, W, H, SPP, D, region=REGION)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)

    ctx.destroy()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
