"""Fixtures the path depends on."""
import hashlib
import os

import numpy as np

from tests.conftest import GOLDEN


def test_blue_noise_fixture_is_the_reference_asset(blue_noise):
    # src/render/pipeline/blue_noise_512.png decoded to raw RGBA8 (last row)
    assert blue_noise.size == 512 * 512 * 4
    assert hashlib.sha256(blue_noise.tobytes()).hexdigest() == \
        "8db1dbee3ae75367dec24b715e64afb13e167243ef968ce0c25f4ba6143d74a2"
    n = blue_noise.reshape(512, 512, 4)
    assert tuple(n[0, 1, :2]) == (168, 91)      # texel (x=1, y=0): SURVEY K8


def test_blue_noise_channels_cover_all_bytes(blue_noise):
    n = blue_noise.reshape(-1, 4)
    for c in range(4):
        assert n[:, c].min() == 0 and n[:, c].max() == 255
