"""Screen-space tile sharding used for multi-GPU rendering (host-side arithmetic, numpy).

The frame is cut into 16x16-pixel tiles numbered row-major and dealt in groups of world_size
consecutive tiles, one to each rank, the deal rotating by one from group to group: local tile q of
rank r is tile q * world_size + (r + q) % world_size (so a rank's tiles are not the same columns of
every row).  A shard is the rank's tiles in local order,
[local_tile][16][16][3], padded to ceil(total / world_size) tiles so that all shards have equal
size for an all-gather.  The GPU kernels (k_render / k_untile) use the same rule.
"""
import numpy as np

TILE = 16


def shard_tile(local, rank, world_size):
    """The frame tile that is local tile `local` of `rank`."""
    return local * world_size + (rank + local) % world_size


def tile_counts(width, height, world_size):
    tx, ty = -(-width // TILE), -(-height // TILE)
    total = tx * ty
    return tx, ty, total, -(-total // world_size)


def shard_from_frame(frame, rank, world_size):
    """Cuts rank's shard out of a row-major frame [h, w, 3]."""
    h, w = frame.shape[:2]
    tx, ty, total, per = tile_counts(w, h, world_size)
    shard = np.zeros((per, TILE, TILE, 3), frame.dtype)
    for local in range(per):
        t = shard_tile(local, rank, world_size)
        if t >= total:
            break
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        blk = frame[y0:y0 + TILE, x0:x0 + TILE]
        shard[local, :blk.shape[0], :blk.shape[1]] = blk
    return shard


def frame_from_shards(shards, width, height):
    """Reassembles [world_size][per_rank][16][16][3] (rank-major, as an all-gather delivers it)."""
    world_size = shards.shape[0]
    tx, ty, total, per = tile_counts(width, height, world_size)
    frame = np.zeros((height, width, 3), shards.dtype)
    for t in range(total):
        local = t // world_size
        r = (t - local) % world_size  # t = local * world_size + (r + local) % world_size
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        bh, bw = min(TILE, height - y0), min(TILE, width - x0)
        frame[y0:y0 + bh, x0:x0 + bw] = shards[r, local, :bh, :bw]
    return frame
