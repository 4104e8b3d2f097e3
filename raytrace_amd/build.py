"""Build driver: compiles the gfx950 library and the C++ host mirror in-tree.

    python -m raytrace_amd.build            # build what is stale
    python -m raytrace_amd.build --force

Artifacts (git-ignored, shipped to the GPU box by gpurun):
    raytrace_amd/librt_amd.so    HIP kernels + the C ABI of include/rt_abi.h   (hipcc --offload-arch=gfx950)
    raytrace_amd/librt_host.so   C++ host mirror of the reference's render/world/game API (g++)
    raytrace_amd/rt_bench        headless counterpart of src/bin/main.rs
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
INC = os.path.join(ROOT, "include")

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CXX = os.environ.get("CXX", "g++")

# -ffp-contract=off is part of the arithmetic contract (include/rt_math.h).
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
             "-Wall", "-Wno-unused-function"]
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-mfma", "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wextra"]

AMD_SRCS = [os.path.join(CSRC, "rt_kernels.hip"), os.path.join(CSRC, "rt_persist.hip"), os.path.join(CSRC, "rt_paths.hip"), os.path.join(CSRC, "rt_frame.hip"), os.path.join(CSRC, "rt_post.hip"),
            os.path.join(CSRC, "rt_api.hip")]
AMD_DEPS = AMD_SRCS + [os.path.join(CSRC, "rt_device.hpp"), os.path.join(CSRC, "rt_kernels.hpp"), os.path.join(CSRC, "rt_dda.hpp"), os.path.join(CSRC, "rt_pslot.hpp"),
                       os.path.join(INC, "rt_abi.h"), os.path.join(INC, "rt_math.h")]
HOST_SRCS = [os.path.join(HOST, f) for f in ("world.cpp", "chunk_storage.cpp", "terrain_upload.cpp", "render.cpp", "host_capi.cpp")]
HOST_DEPS = HOST_SRCS + [os.path.join(HOST, f) for f in ("world.hpp", "render.hpp", "chunk_storage.hpp", "terrain_upload.hpp")] + [os.path.join(INC, "rt_abi.h")]
BENCH_SRCS = [os.path.join(HOST, "rt_bench.cpp")]

LIB_AMD = os.path.join(HERE, "librt_amd.so")
LIB_HOST = os.path.join(HERE, "librt_host.so")
BENCH = os.path.join(HERE, "rt_bench")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _run(cmd):
    print("+", " ".join(cmd), file=sys.stderr, flush=True)   # stderr: bench.py's stdout carries one JSON line only
    subprocess.check_call(cmd)


def build(force=False, verbose=True):
    built = []
    if force or _stale(LIB_AMD, AMD_DEPS):
        objs = []
        for src in AMD_SRCS:
            obj = os.path.join(CSRC, os.path.basename(src) + ".o")
            _run([HIPCC] + HIP_FLAGS + ["-I", INC, "-c", src, "-o", obj])
            objs.append(obj)
        _run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_AMD] + objs)
        built.append(LIB_AMD)
    if force or _stale(LIB_HOST, HOST_DEPS + [LIB_AMD]):
        _run([CXX] + CXX_FLAGS + ["-shared", "-o", LIB_HOST] + HOST_SRCS +
             ["-L", HERE, "-lrt_amd", "-ldl", "-Wl,-rpath,$ORIGIN"])
        built.append(LIB_HOST)
    if all(os.path.exists(s) for s in BENCH_SRCS) and (force or _stale(BENCH, BENCH_SRCS + [LIB_HOST, LIB_AMD])):
        _run([CXX] + CXX_FLAGS + ["-o", BENCH] + BENCH_SRCS + ["-L", HERE, "-lrt_host", "-lrt_amd", "-lpthread", "-Wl,-rpath,$ORIGIN"])
        built.append(BENCH)
    return built


if __name__ == "__main__":
    build(force="--force" in sys.argv)
