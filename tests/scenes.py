"""Analytic test scenes as 256^3 material-id arrays [z,y,x] in texel space (texel = world + 128)."""
import numpy as np


def empty_ids():
    return np.zeros((256, 256, 256), dtype=np.uint8)


def floor_ids(world_z_top=0, material=2):
    """Solid below world z = world_z_top."""
    ids = empty_ids()
    ids[: 128 + world_z_top, :, :] = material
    return ids


def single_voxel_ids(texel=(128, 128, 128), material=4):
    ids = empty_ids()
    x, y, z = texel
    ids[z, y, x] = material
    return ids


def staircase_ids():
    """Steps rising along +x, plus a wall and a floating block — exercises all six face ids."""
    ids = empty_ids()
    for i in range(16):
        ids[: 100 + 4 * i, :, 64 + 8 * i: 72 + 8 * i] = 2 + (i % 3) * 2 if (i % 3) != 2 else 5
    ids[:200, 230:240, :] = 6          # wall across +y
    ids[180:190, 100:110, 100:110] = 4  # floating block
    return ids


def random_blocks_ids(seed=7, density=0.02):
    rng = np.random.default_rng(seed)
    ids = empty_ids()
    mask = rng.random((64, 64, 64)) < density
    mats = rng.integers(1, 7, size=(64, 64, 64), dtype=np.uint8)
    coarse = np.where(mask, mats, 0).astype(np.uint8)
    ids[:] = np.repeat(np.repeat(np.repeat(coarse, 4, 0), 4, 1), 4, 2)
    # carve fine detail so 4^3 bricks are not all uniform
    fine = rng.random((256, 256, 256)) < 0.15
    ids[fine & (ids != 0)] = 0
    ids[:20, :, :] = 2
    return ids
