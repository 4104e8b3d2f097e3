"""C++ host mirror of the reference's render/game API (raytrace_amd/host) — the parts that need no GPU."""
import ctypes as C
import math

import numpy as np
import pytest

from raytrace_amd import abi, render
from oracle import pyoracle as po

pytestmark = pytest.mark.usefixtures("native_built")


def test_compute_triple_euler_vector_matches_util_rs():
    # src/util.rs:9-22; K6
    f, u, r = render.compute_triple_euler_vector(math.pi / 2, 0.0)
    assert abs(f[0] + 4.371139e-8) < 1e-12 and f[1] == 1.0 and f[2] == 0.0
    assert abs(u[2] - 1.0) < 1e-7 and abs(r[0] - 1.0) < 1e-7
    for heading, pitch in ((0.3, -0.2), (-2.0, -0.1), (3.0, 0.7), (1.0, 1.4)):
        f, u, r = (v.astype(np.float64) for v in render.compute_triple_euler_vector(heading, pitch))
        assert abs(np.dot(f, u)) < 1e-6 and abs(np.dot(f, r)) < 1e-6 and abs(np.dot(u, r)) < 1e-6
        assert np.allclose(np.cross(f, u), r, atol=1e-6)
        assert np.allclose(f, [math.cos(heading) * math.cos(pitch), math.sin(heading) * math.cos(pitch), math.sin(pitch)], atol=1e-6)


def test_product_uniform_fill_equals_the_oracle_restatement():
    # pipeline.rs:191-207 written twice (C++ mirror / oracle); they must agree byte for byte
    for args in (((-30.0, -128.0, 100.0), math.pi / 2, 0.0, 0.0, 1, (0, 0, 0)),
                 ((100.0, 200.0, 60.0), -3.0, -0.1, 1.2, 4242, (16, -32, 0))):
        a = render.camera_uniforms(*args)
        b = po.camera_uniforms(*args)
        assert bytes(a) == bytes(b)


def test_game_defaults_and_cli_arguments():
    # src/game/mod.rs:37-58
    g = render.Game()
    assert g.camera.origin == (-30.0, -128.0, 100.0)
    assert abs(g.camera.heading - math.pi / 2) < 1e-6 and g.camera.pitch == 0.0 and g.get_sun_angle() == 0.0
    g2 = render.Game(args=(100, 200, 60, -2, -0.1, 0.7))       # capture_training_data.py argument order
    assert g2.camera.origin == (100.0, 200.0, 60.0)
    assert abs(g2.camera.heading + 2.0) < 1e-6 and abs(g2.camera.pitch + 0.1) < 1e-6 and abs(g2.get_sun_angle() - 0.7) < 1e-6
    g2.camera.set(origin=(1, 2, 3), pitch=0.5)
    assert g2.camera.origin == (1.0, 2.0, 3.0) and g2.camera.pitch == 0.5
    g2.set_sun_angle(-1.2)
    assert abs(g2.get_sun_angle() + 1.2) < 1e-6


def test_create_instance_without_gpu_reports_the_error(blue_noise):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    g = render.Game()
    g.set_world(np.zeros(256 ** 3, dtype=np.uint32), np.full(256 ** 3, 6, dtype=np.uint8))
    with pytest.raises(render.RtError) as e:
        render.create_instance(render.make_config(32, 32), g, blue_noise)
    assert "rt_create" in str(e.value)


def test_create_instance_never_replaces_a_world_the_caller_set(blue_noise):
    """a world set by the caller used to be regenerated silently when the config's region differed."""
    g = render.Game()
    g.set_world(np.zeros(256 ** 3, dtype=np.uint32), np.full(256 ** 3, 6, dtype=np.uint8))
    with pytest.raises(render.RtError) as e:
        render.create_instance(render.make_config(32, 32, region=512), g, blue_noise)
    assert "region 256" in str(e.value) and "512" in str(e.value)
    with pytest.raises(ValueError):
        g.set_world(np.zeros(256 ** 3, dtype=np.uint32), np.full(256 ** 3, 6, dtype=np.uint8), region=512)
