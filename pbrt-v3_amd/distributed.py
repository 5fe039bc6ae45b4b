"""Film-tile sharding across ranks (one process per GPU) and the gather of the per-rank film.

The reference shards a frame across machines by disjoint crop windows stitched with
`imgtool assemble` (main/pbrt.cpp:94-100, tools/imgtool.cpp:190-285).  Here every rank renders the
16x16 tiles t with t % world == rank of the FULL-frame tiling (sampler and tile indices unchanged, so
samples are identical to a single-process render) and one gather moves each rank's packed
(RGB sum, weight) tile buffer to rank 0 -- RCCL over xGMI when the tensors are on GPUs, gloo on CPU.
No reduction is needed on the device: with a box filter of radius 0.5 every pixel is owned by exactly one tile, and for
wider filters a tile's block carries its halo (PgRenderDesc.tile_pixels entries per tile) and rank 0's host Film adds the
overlapping blocks in tile order (Film::MergeFilmTile).
"""
import numpy as np
import torch
import torch.distributed as dist


def shard_buffers(n_tiles_max, device, tile_pixels=256):
    """Fixed-size per-rank buffers (equal on every rank so one gather suffices); tile_pixels = PgRenderDesc.tile_pixels."""
    max_strays = n_tiles_max * tile_pixels // 8 + 1024
    film = torch.zeros((n_tiles_max * tile_pixels, 4), dtype=torch.float32, device=device)
    strays = torch.zeros((max_strays, 8), dtype=torch.int32, device=device)  # PgStraySample = 8 x 4 bytes
    nstrays = torch.zeros(1, dtype=torch.int32, device=device)
    return film, strays, nstrays, max_strays


def gather_lists(film, strays, nstrays, dst=0):
    """Receive buffers for gather_film on dst (None on the other ranks); allocate once, reuse every frame."""
    if dist.get_rank() != dst:
        return None
    world = dist.get_world_size()
    return ([torch.empty_like(film) for _ in range(world)], [torch.empty_like(strays) for _ in range(world)],
            [torch.empty_like(nstrays) for _ in range(world)])


def gather_film(film, strays, nstrays, lists=None, dst=0):
    """dist.gather of the three shard buffers; returns the lists on dst, None elsewhere."""
    if lists is None:
        lists = gather_lists(film, strays, nstrays, dst)
    dist.gather(film, lists[0] if lists else None, dst=dst)
    dist.gather(strays, lists[1] if lists else None, dst=dst)
    dist.gather(nstrays, lists[2] if lists else None, dst=dst)
    return lists


def merge_shards(pkg, scene, tile_count, shards):
    """Film::MergeFilmTile for every rank's shard on the host; shards = [(film, strays, n), ...] per rank.
    Returns the final (h, w, 3) image."""
    world = len(shards)
    scene.film_clear()
    for r, (film, strays, n) in enumerate(shards):
        rd = scene.render_desc(tile_first=r, tile_step=world)
        nt = tile_count(rd)
        f = np.ascontiguousarray(film.detach().cpu().numpy()[:nt * rd.tile_pixels]).view(pkg.FILM_PIXEL_DTYPE).reshape(-1)
        s = np.ascontiguousarray(strays.detach().cpu().numpy()[:int(n)]).view(pkg.STRAY_DTYPE).reshape(-1)
        scene.film_merge(rd, f, s)
    return scene.film_image()
