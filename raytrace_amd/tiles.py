"""Multi-GPU framebuffer split: 8x8-pixel tiles dealt round-robin over the ranks.

Global tile t (row-major over ceil(W/8) x ceil(H/8)) belongs to rank t % world and is that rank's local tile
t // world.  Every rank's planes are tile-major [capacity*64 pixels] with capacity = ceil(tiles / world), so the
frame-end gather moves equal-size messages; rank 0 scatters them back with rt_untile (HIP) — `untile_numpy` is
the same mapping on the host, used by the CPU tests.
"""
import numpy as np


def tile_grid(width, height):
    return (width + 7) // 8, (height + 7) // 8


def tile_capacity(width, height, world):
    tx, ty = tile_grid(width, height)
    return (tx * ty + world - 1) // world


def tile_count(width, height, rank, world):
    tx, ty = tile_grid(width, height)
    n = tx * ty
    return max(0, (n - rank + world - 1) // world)


def tiles_of_rank(width, height, rank, world):
    tx, ty = tile_grid(width, height)
    return np.arange(rank, tx * ty, world)


def tile_major_from_frame(frame, rank, world):
    """Extract rank's tiles from a row-major [H,W(,C)] plane into the tile-major [capacity*64(,C)] layout."""
    h, w = frame.shape[:2]
    tx, _ = tile_grid(w, h)
    cap = tile_capacity(w, h, world)
    out = np.zeros((cap * 64,) + frame.shape[2:], dtype=frame.dtype)
    for j, t in enumerate(tiles_of_rank(w, h, rank, world)):
        x0, y0 = (t % tx) * 8, (t // tx) * 8
        tile = np.zeros((8, 8) + frame.shape[2:], dtype=frame.dtype)
        sub = frame[y0:y0 + 8, x0:x0 + 8]
        tile[:sub.shape[0], :sub.shape[1]] = sub
        out[j * 64:(j + 1) * 64] = tile.reshape((64,) + frame.shape[2:])
    return out


def untile_numpy(gathered, width, height, world):
    """gathered: [world, capacity*64(,C)] -> row-major [H,W(,C)] (what rt_untile does on the GPU)."""
    tx, ty = tile_grid(width, height)
    frame = np.zeros((height, width) + gathered.shape[2:], dtype=gathered.dtype)
    for r in range(world):
        for j, t in enumerate(tiles_of_rank(width, height, r, world)):
            x0, y0 = (t % tx) * 8, (t // tx) * 8
            tile = gathered[r, j * 64:(j + 1) * 64].reshape((8, 8) + gathered.shape[2:])
            hh, ww = min(8, height - y0), min(8, width - x0)
            frame[y0:y0 + hh, x0:x0 + ww] = tile[:hh, :ww]
    return frame
