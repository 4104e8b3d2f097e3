"""CPU sanitizer run: the oracle and the C++ host mirror rebuilt with AddressSanitizer + UBSan
(`make -C oracle asan`) and driven through the known-answer, golden-frame, world and streaming tests in a child process.
CPU only: the GPU box never runs sanitizers."""
import os
import subprocess
import sys

from tests.conftest import ROOT


def test_oracle_and_host_mirror_under_asan_ubsan(native_built):
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    libasan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    assert os.path.isabs(libasan) and os.path.exists(libasan), libasan
    env = dict(os.environ,
               LD_PRELOAD=libasan,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",      # the interpreter itself is not leak-clean
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               RT_ORACLE_LIB=os.path.join(ROOT, "oracle", "_asan", "librt_oracle.so"),
               RT_HOST_LIB=os.path.join(ROOT, "oracle", "_asan", "librt_host.so"))
    tests = ["tests/test_oracle_kat.py", "tests/test_golden.py", "tests/test_world.py", "tests/test_host_mirror.py",
             "tests/test_streaming.py", "tests/test_post_passes.py"]
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu", "-p", "no:cacheprovider",
                        "-k", "not region_512"] + tests,      # (the 512^3 case is 0.7 GB of shadow-mapped arrays; 256^3 covers the code)
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (r.stdout + r.stderr)[-4000:]
    assert r.returncode == 0, tail
    assert "passed" in r.stdout and "AddressSanitizer" not in tail and "runtime error" not in tail, tail
