#!/usr/bin/env python3
"""Regenerates tests/golden/frames.npz: small G-buffers + counters rendered by the CPU oracle.

The reference cannot be built or run in this image (Rust + Vulkan + glslc absent) and holds no golden vectors for this
path, so these frames are produced by the build's own oracle (PARITY UNPINNED by the reference; see oracle/rt_oracle.cpp).
They pin the oracle against drift of the arithmetic contract and give the GPU tests a fixture that does not depend on
running the oracle at test time.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raytrace_amd import world  # noqa: E402
from oracle import pyoracle as po  # noqa: E402
from tests import scenes  # noqa: E402

# name -> (scene, W, H, spp, depth, origin, heading, pitch, sun, seed, lr)
CASES = {
    "reference_frame": ("procedural", 48, 48, 1, 2, (-30.0, -128.0, 100.0), np.pi / 2, 0.0, 0.0, 1, (0, 0, 0)),
    "deep_multi_sample": ("procedural", 40, 24, 3, 4, (100.0, 100.0, 60.0), -2.0, -0.1, 0.7, 77, (0, 0, 0)),
    "stairs": ("stairs", 32, 32, 2, 3, (-40.0, -100.0, 90.0), 1.1, -0.5, 0.3, 5, (0, 0, 0)),
    "scrolled": ("procedural", 32, 32, 1, 2, (-14.0, -100.0, 100.0), np.pi / 2, 0.0, 0.0, 9, (16, 32, 0)),
    "inside_ground": ("procedural", 24, 24, 2, 2, (10.0, 10.0, 5.0), 1.0, 0.3, 1.2, 3, (0, 0, 0)),
}


def scene_arrays(name):
    if name == "procedural":
        return world.generate_region(world.DEFAULT_SEED)
    return world.region_from_ids({"stairs": scenes.staircase_ids}[name]())


def main():
    noise = np.fromfile(os.path.join(ROOT, "tests", "golden", "blue_noise_512.rgba"), dtype=np.uint8)
    out = {}
    cache = {}
    for case, (scene, W, H, spp, depth, origin, heading, pitch, sun, seed, lr) in CASES.items():
        if scene not in cache:
            cache[scene] = scene_arrays(scene)
        mats, mine = cache[scene]
        u = po.camera_uniforms(origin, heading, pitch, sun, seed, lr)
        planes, cn = po.render(mats, mine, noise, u, W, H, spp, depth)
        for k, v in planes.items():
            out["%s/%s" % (case, k)] = v
        out["%s/counters" % case] = np.array([getattr(cn, f) for f, _ in cn._fields_], dtype=np.uint64)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "frames.npz"), **out)
    print("wrote %d arrays" % len(out))


if __name__ == "__main__":
    main()
