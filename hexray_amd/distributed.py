"""One-process-per-GPU frame sharding.

The frame is cut into square tiles; tile NUMBER ``t`` belongs to rank ``t % world`` and is that rank's local tile
``t // world`` -- round-robin, so expensive image regions spread over all GPUs (the reference spreads 64x64 buckets
over threads with an atomic cursor, src/main.cpp:273-275).  Numbers run row-major, each tile row rotated by
``TILE_SKEW`` more columns than the one above it (``tile_position``), so that a rank does not own the same columns
in every row.
Every rank renders its tiles into one tile-major buffer (``LAYOUT_TILES``); the buffers are collected with
ONE gather (RCCL over xGMI when the tensors live on GPUs; gloo in the CPU tests) and rank 0 scatters them
into the row-major frame.  There is no other communication: the scene is replicated and the random
numbers are keyed on (pixel, sample), so the image does not depend on the world size.
"""
from __future__ import annotations

import numpy as np


def tiles_in_frame(width: int, height: int, tile: int):
    return (width - 1) // tile + 1, (height - 1) // tile + 1


TILE_SKEW = 3   # HXR_TILE_SKEW (hexray_amd/csrc/hxr_device_scene.h)


def tile_position(t: int, tiles_x: int, world: int):
    """Column and row of tile NUMBER t (hxr_tile_position): with more than one shard the numbers of tile row ty are rotated by TILE_SKEW * ty
    columns, so that a shard does not own the same columns in every row."""
    ty, v = divmod(t, tiles_x)
    return ((v - TILE_SKEW * ty) % tiles_x if world > 1 else v), ty


def shard_tile_count(width: int, height: int, tile: int, rank: int, world: int) -> int:
    tx, ty = tiles_in_frame(width, height, tile)
    total = tx * ty
    return (total - rank + world - 1) // world if total > rank else 0


def padded_tiles_per_rank(width: int, height: int, tile: int, world: int) -> int:
    """Every rank contributes the same number of tile slots to the gather (rank 0's count is the largest)."""
    return shard_tile_count(width, height, tile, 0, world)


def gather_tiles(local, world: int, rank: int, group=None, out=None):
    """Collect every rank's tile buffer on rank 0.  ``local`` is a 1-D float32 torch tensor of the padded
    per-rank size; returns the rank-major concatenation on rank 0, ``None`` elsewhere.  ``out`` (rank 0): a
    reusable 1-D tensor of ``world * local.numel()`` elements that receives the ranks' buffers in place (the
    collective writes straight into its slices: no per-frame allocation, no concatenation pass)."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local
    if rank == 0:
        if out is None:
            out = torch.empty(world * local.numel(), dtype=local.dtype, device=local.device)
        parts = list(out.view(world, local.numel()).unbind(0))
        dist.gather(local, gather_list=parts, dst=0, group=group)
        return out
    dist.gather(local, gather_list=None, dst=0, group=group)
    return None


def assemble_tiles_numpy(gathered: np.ndarray, width: int, height: int, tile: int, world: int) -> np.ndarray:
    """Host mirror of hxr_assemble_frame (hexray_amd/csrc): rank-major tile buffers -> (H, W, 3) frame."""
    tx, ty = tiles_in_frame(width, height, tile)
    per_rank = padded_tiles_per_rank(width, height, tile, world) * tile * tile * 3
    g = np.asarray(gathered, dtype=np.float32).reshape(world, per_rank)
    frame = np.zeros((height, width, 3), dtype=np.float32)
    for t in range(tx * ty):
        r, lt = t % world, t // world
        block = g[r, lt * tile * tile * 3:(lt + 1) * tile * tile * 3].reshape(tile, tile, 3)
        cx, cy = tile_position(t, tx, world)
        x0, y0 = cx * tile, cy * tile
        x1, y1 = min(x0 + tile, width), min(y0 + tile, height)
        frame[y0:y1, x0:x1] = block[:y1 - y0, :x1 - x0]
    return frame
