"""The N>1 path on CPU: two gloo ranks each own the tiles t % 2 == rank of the full-frame tiling, render them
(the CPU oracle stands in for the HIP kernels here -- it writes the same PgFilmPixel/PgStraySample buffers),
gather to rank 0 with the product's pbrt_v3_amd.distributed code and merge.  The merged image must equal the
single-process render bit for bit and the reference golden image."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import GOLD, ROOT


def _worker(rank, world, port, name, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, ROOT)
    from __graft_entry__ import load_package
    pkg = load_package()
    from oracle import oracle
    from pbrt_v3_amd import distributed as pdist
    import ctypes as C
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pkg.HostScene(os.path.join(GOLD, name + ".pbrt"))
    tile_count = lambda rd: oracle.lib().oracle_render_tile_count(C.byref(rd))
    rd = scene.render_desc(tile_first=rank, tile_step=world)
    film, strays, nstrays, max_strays = pdist.shard_buffers(tile_count(scene.render_desc(0, world)), "cpu", rd.tile_pixels)
    f, s, _ = oracle.render(scene.desc, rd, max_strays=max_strays)
    film[:len(f)] = torch.from_numpy(f.view(np.float32).reshape(-1, 4))
    strays[:len(s)] = torch.from_numpy(s.view(np.int32).reshape(-1, 8))
    nstrays[0] = len(s)
    lists = pdist.gather_film(film, strays, nstrays, dst=0)
    if rank == 0:
        shards = [(lists[0][r], lists[1][r], int(lists[2][r].item())) for r in range(world)]
        np.save(out, pdist.merge_shards(pkg, scene, tile_count, shards))
    dist.barrier()
    dist.destroy_process_group()


def _run(name, world, tmp_path, port):
    out = str(tmp_path / f"{name}_{world}.npy")
    mp.spawn(_worker, args=(world, port, name, out), nprocs=world, join=True)
    return np.load(out)


def test_two_rank_sharded_render_matches_golden(pkg, tmp_path):
    # box filter; a crop window; a deep BVH; the volumetric integrator with the Sobol' sampler (sample indices depend on the
    # full-frame sample bounds, not on the shard); moving instances and shapes whose motion rotates (every rank builds the same MotionBounds boxes)
    for i, name in enumerate(("cornell_40x24", "cornell_crop", "synthetic_n40", "sobol_vol_smoke", "sobol_round_crop", "motion_rotate_instances",
                              "nest_motion_moving_instances")):  # (and moving shapes inside object definitions under moving instances, a shutter inside the motion)
        img = _run(name, 2, tmp_path, 29531 + i)
        assert np.array_equal(img, pkg.read_pfm(os.path.join(GOLD, name + ".pfm"))), name


def test_three_rank_uneven_shards(pkg, tmp_path):
    img = _run("cornell_32", 3, tmp_path, 29541)  # 4 tiles over 3 ranks: 2/1/1
    assert np.array_equal(img, pkg.read_pfm(os.path.join(GOLD, "cornell_32.pfm")))


def test_sharded_frame_whose_film_positions_round_onto_the_next_pixel(pkg, tmp_path):
    """filter_box_round_up (tests/test_film_round_up.py): a box-filter frame on the gathering film path -- every rank's tile blocks
    carry their one-pixel halo (the sample that rounds up at a tile's last column lands in the NEIGHBOURING rank's pixel), and rank
    0's merge adds the blocks as the reference merges its FilmTiles: the reference binary's image bit for bit, from 2 ranks."""
    img = _run("filter_box_round_up", 2, tmp_path, 29551)
    assert np.array_equal(img, pkg.read_pfm(os.path.join(GOLD, "filter_box_round_up.pfm")))


def test_wide_filter_shards_merge_in_the_frames_tile_order(pkg, tmp_path):
    """Filters wider than half a pixel: the tile blocks of neighbouring tiles overlap and the film adds them as floats, so the ORDER of
    the tiles is part of the result.  The shards are merged in the frame's own tile order (tile t from shard t % world,
    Film::MergeShards), not rank by rank: the images of 2 and 3 ranks are the reference binary's (single-threaded: tiles in order),
    bit for bit -- on the tile-border pixels too."""
    for i, (name, world) in enumerate((("filter_gaussian", 2), ("filter_mitchell_crop", 2), ("filter_gaussian", 3), ("filter_sinc", 3), ("filter_sobol_gaussian", 2))):
        img = _run(name, world, tmp_path, 29561 + i)
        assert np.array_equal(img, pkg.read_pfm(os.path.join(GOLD, name + ".pfm"))), (name, world)


def test_rank_by_rank_merge_would_differ(pkg, oracle):
    """The guard of the test above: merging the same shards rank by rank (what the round-4 code did) changes tile-border pixels of a
    wide-filter frame in the last bits -- if this ever stops being true the test above proves nothing."""
    import ctypes as C
    scene = pkg.HostScene(os.path.join(GOLD, "filter_gaussian.pbrt"))
    world = 3
    shards = []
    for r in range(world):
        rd = scene.render_desc(tile_first=r, tile_step=world)
        f, s, _ = oracle.render(scene.desc, rd)
        shards.append((rd, f, s))
    scene.film_clear()
    for rd, f, s in shards:
        scene.film_merge(rd, f, s)
    by_rank = scene.film_image()
    scene.film_clear()
    scene.film_merge_shards(scene.render_desc(0, 1), [(f, s) for _, f, s in shards])
    in_order = scene.film_image()
    gold = pkg.read_pfm(os.path.join(GOLD, "filter_gaussian.pfm"))
    assert np.array_equal(in_order, gold)
    assert not np.array_equal(by_rank, gold) and np.abs(by_rank - gold).max() < 1e-4
