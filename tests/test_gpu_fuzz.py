"""Randomised parity: random poses, sun angles, seeds, region offsets, depths and image shapes on several scenes —
the HIP path through the C ABI must equal the CPU oracle bit for bit (planes and counters).  Seeds are fixed, so a
failure is reproducible; the case list is printed in the assertion message."""
import numpy as np
import pytest

from raytrace_amd import abi, render, world
from oracle import pyoracle as po
from tests import scenes

pytestmark = pytest.mark.gpu


def _cases(rng, n):
    out = []
    for _ in range(n):
        W = int(rng.integers(3, 20)) * 8 + int(rng.choice([0, 0, 3, 5]))
        H = int(rng.integers(3, 16)) * 8 + int(rng.choice([0, 0, 1, 6]))
        out.append(dict(
            W=W, H=H, spp=int(rng.integers(1, 5)), depth=int(rng.integers(0, 7)),
            origin=tuple(float(x) for x in rng.uniform(-125, 125, size=3)),
            heading=float(rng.uniform(-3.2, 3.2)), pitch=float(rng.uniform(-1.5, 1.5)),
            sun=float(rng.uniform(-1.5, 1.5)), seed=int(rng.integers(0, abi.NOISE_BYTES)),
            lr=tuple(int(v) * 16 for v in rng.integers(-3, 4, size=3)) if rng.random() < 0.3 else (0, 0, 0)))
    return out


@pytest.mark.parametrize("scene_name", ["procedural", "blocks", "stairs"])
def test_random_cases_match_oracle(scene_name, procedural_region, blue_noise):
    if scene_name == "procedural":
        mats, mine = procedural_region
    else:
        mats, mine = world.region_from_ids({"blocks": scenes.random_blocks_ids, "stairs": scenes.staircase_ids}[scene_name]())
    rng = np.random.default_rng({"procedural": 101, "blocks": 202, "stairs": 303}[scene_name])
    for case in _cases(rng, 10):
        if scene_name == "procedural" and case["origin"][2] < 0 and rng.random() < 0.7:
            case["origin"] = (case["origin"][0], case["origin"][1], abs(case["origin"][2]))   # mostly above ground
        u = po.camera_uniforms(case["origin"], case["heading"], case["pitch"], case["sun"], case["seed"], case["lr"])
        cpu, ccn = po.render(mats, mine, blue_noise, u, case["W"], case["H"], case["spp"], case["depth"])
        for kernel, flags in ((abi.RT_KERNEL_PERSISTENT, abi.RT_FLAG_COUNTERS),
                              (abi.RT_KERNEL_PERSISTENT, abi.RT_FLAG_CACHE_PRIMARY),
                              (abi.RT_KERNEL_PATHS, abi.RT_FLAG_CACHE_PRIMARY),
                              (abi.RT_KERNEL_FRAME, abi.RT_FLAG_CACHE_PRIMARY),
                              (abi.RT_KERNEL_MEGA, abi.RT_FLAG_COUNTERS)):
            cfg = render.make_config(case["W"], case["H"], spp=case["spp"], depth=case["depth"], kernel=kernel, flags=flags)
            with render.Context(cfg) as ctx:
                ctx.upload_world(mats, mine)
                ctx.upload_noise(blue_noise)
                ctx.draw_frame(u)
                ctx.sync()
                gpu = ctx.readback_all()
                gcn = ctx.counters()
            for name in cpu:
                assert np.array_equal(gpu[name], cpu[name], equal_nan=True), (name, kernel, flags, case)
            if flags & abi.RT_FLAG_COUNTERS and not flags & abi.RT_FLAG_CACHE_PRIMARY:
                assert gcn.as_dict() == ccn.as_dict(), (kernel, case)
