"""Committed golden frames (tests/golden/frames.npz, written by tests/golden/make_golden.py with the CPU oracle).

CPU: the oracle still reproduces them bit for bit (guards the arithmetic contract against drift).
GPU: the HIP path reproduces them through the C ABI without running the oracle at all."""
import os

import numpy as np
import pytest

from raytrace_amd import abi, render
from oracle import pyoracle as po
from tests.conftest import GOLDEN
from tests.golden.make_golden import CASES, scene_arrays

FRAMES = np.load(os.path.join(GOLDEN, "frames.npz"))
COUNTER_FIELDS = [f for f, _ in abi.RtCounters._fields_]


@pytest.fixture(scope="module")
def scenes_cache(native_built):
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = scene_arrays(name)
        return cache[name]
    return get


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_reproduces_golden(case, scenes_cache, blue_noise):
    scene, W, H, spp, depth, origin, heading, pitch, sun, seed, lr = CASES[case]
    mats, mine = scenes_cache(scene)
    u = po.camera_uniforms(origin, heading, pitch, sun, seed, lr)
    planes, cn = po.render(mats, mine, blue_noise, u, W, H, spp, depth)
    for name, arr in planes.items():
        assert np.array_equal(arr, FRAMES["%s/%s" % (case, name)], equal_nan=True), name
    assert [getattr(cn, f) for f in COUNTER_FIELDS] == FRAMES["%s/counters" % case].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [abi.RT_KERNEL_PERSISTENT, abi.RT_KERNEL_WAVEFRONT, abi.RT_KERNEL_MEGA])
@pytest.mark.parametrize("case", sorted(CASES))
def test_gpu_reproduces_golden(case, kernel, scenes_cache, blue_noise):
    scene, W, H, spp, depth, origin, heading, pitch, sun, seed, lr = CASES[case]
    mats, mine = scenes_cache(scene)
    u = render.camera_uniforms(origin, heading, pitch, sun, seed, lr)      # product-side uniform fill
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=abi.RT_FLAG_COUNTERS)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        planes = ctx.readback_all()
        cn = ctx.counters()
    for name, arr in planes.items():
        assert np.array_equal(arr, FRAMES["%s/%s" % (case, name)], equal_nan=True), name
    assert [getattr(cn, f) for f in COUNTER_FIELDS] == FRAMES["%s/counters" % case].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [abi.RT_KERNEL_PATHS])
@pytest.mark.parametrize("case", sorted(CASES))
def test_gpu_paths_kernel_reproduces_golden(case, kernel, scenes_cache, blue_noise):
    """RT_KERNEL_PATHS (cached primaries: the configuration RT_KERNEL_DEFAULT runs) on the golden frames; frames with lr != 0
    take its k_persist fallback."""
    scene, W, H, spp, depth, origin, heading, pitch, sun, seed, lr = CASES[case]
    mats, mine = scenes_cache(scene)
    u = render.camera_uniforms(origin, heading, pitch, sun, seed, lr)
    cfg = render.make_config(W, H, spp=spp, depth=depth, kernel=kernel, flags=abi.RT_FLAG_CACHE_PRIMARY)
    with render.Context(cfg) as ctx:
        ctx.upload_world(mats, mine)
        ctx.upload_noise(blue_noise)
        ctx.draw_frame(u)
        ctx.sync()
        planes = ctx.readback_all()
    for name, arr in planes.items():
        assert np.array_equal(arr, FRAMES["%s/%s" % (case, name)], equal_nan=True), name
