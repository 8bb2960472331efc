"""Host-side screen-tile sharding math (numpy), identical to the C ABI's
``VrFrame`` tile semantics (include/volrend_hip.h): the frame is cut into
``tile_w x tile_h`` tiles in row-major tile order; rank ``r`` of ``world`` owns the
tiles ``t`` with ``t % world == r``; its k-th tile (``k = t // world``) sits densely at
offset ``k * tile_w * tile_h`` pixels of the rank's COMPACT buffer.

Used by the multi-process tests (gloo, no GPU) and available to callers that want
to assemble gathered buffers on the host.
"""
from __future__ import annotations

import numpy as np


def tile_geometry(width: int, height: int, tile_w: int, tile_h: int):
    if tile_w == 0 and tile_h == 0:
        tile_w, tile_h = (width + 7) // 8 * 8, (height + 7) // 8 * 8
    if tile_w <= 0 or tile_h <= 0 or tile_w % 8 or tile_h % 8:
        raise ValueError("tile size must be positive multiples of 8")
    return tile_w, tile_h, -(-width // tile_w), -(-height // tile_h)


def tiles_per_rank(width, height, tile_w, tile_h, world) -> int:
    tw, th, tx, ty = tile_geometry(width, height, tile_w, tile_h)
    return -(-(tx * ty) // max(world, 1))


def compact_pixels(width, height, tile_w, tile_h, world) -> int:
    tw, th, _, _ = tile_geometry(width, height, tile_w, tile_h)
    return tiles_per_rank(width, height, tile_w, tile_h, world) * tw * th


def owner_map(width, height, tile_w, tile_h, world) -> np.ndarray:
    """[H, W] int32: the rank that renders each pixel."""
    tw, th, tx, _ = tile_geometry(width, height, tile_w, tile_h)
    ys, xs = np.mgrid[0:height, 0:width]
    return (((ys // th) * tx + xs // tw) % max(world, 1)).astype(np.int32)


def frame_to_compact(frame: np.ndarray, tile_w, tile_h, rank, world) -> np.ndarray:
    """Extract rank's pixels of a full [H, W, C] frame into its COMPACT buffer
    (pixels of edge tiles outside the frame stay zero)."""
    h, w = frame.shape[:2]
    tw, th, tx, ty = tile_geometry(w, h, tile_w, tile_h)
    n = tiles_per_rank(w, h, tile_w, tile_h, world)
    out = np.zeros((n, th, tw) + frame.shape[2:], dtype=frame.dtype)
    for t in range(rank, tx * ty, max(world, 1)):
        k = t // max(world, 1)
        y0, x0 = (t // tx) * th, (t % tx) * tw
        blk = frame[y0:y0 + th, x0:x0 + tw]
        out[k, :blk.shape[0], :blk.shape[1]] = blk
    return out.reshape((n * th * tw,) + frame.shape[2:])


def assemble_tiles(gathered: np.ndarray, width, height, tile_w, tile_h, world) -> np.ndarray:
    """Inverse: ``gathered`` = [world, compact_pixels, C] (rank-major) -> [H, W, C]."""
    tw, th, tx, ty = tile_geometry(width, height, tile_w, tile_h)
    n = tiles_per_rank(width, height, tile_w, tile_h, world)
    g = gathered.reshape((max(world, 1), n, th, tw) + gathered.shape[2:])
    frame = np.zeros((height, width) + gathered.shape[2:], dtype=gathered.dtype)
    for t in range(tx * ty):
        r, k = t % max(world, 1), t // max(world, 1)
        y0, x0 = (t // tx) * th, (t % tx) * tw
        hh, ww = min(th, height - y0), min(tw, width - x0)
        frame[y0:y0 + hh, x0:x0 + ww] = g[r, k, :hh, :ww]
    return frame
