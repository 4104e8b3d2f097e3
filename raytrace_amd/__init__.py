"""raytrace_amd — MI355X-native implementation of the per-pixel ray-trace path of someguynamedjosh/raytrace.

Layout:  csrc/   HIP kernels + the C ABI (include/rt_abi.h)        -> librt_amd.so
         host/   C++ mirror of the reference's render/world/game API -> librt_host.so, rt_bench
         *.py    ctypes bindings used by tests and bench.py (no compute in Python, no CPU fallback)
"""
from . import abi  # noqa: F401
