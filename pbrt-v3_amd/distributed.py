"""Film-tile sharding across ranks (one process per GPU) and the gather of the per-rank film.

The reference shards a frame across machines by disjoint crop windows stitched with
`imgtool assemble` (main/pbrt.cpp:94-100, tools/imgtool.cpp:190-285).  Here every rank renders the
16x16 tiles t with t % world == rank of the FULL-frame tiling (sampler and tile indices unchanged, so
samples are identical to a single-process render) and ONE gather per frame moves each rank's packed
shard -- (RGB sum, weight) tile buffer, stray-sample list and its length in one buffer -- to rank 0:
RCCL over xGMI when the tensors are on GPUs (a gather is point-to-point sends into rank 0, one xGMI
hop each), gloo on CPU.
No reduction is needed on the device: with a box filter of radius 0.5 every pixel is owned by exactly one tile, and for
wider filters a tile's block carries its halo (PgRenderDesc.tile_pixels entries per tile) and rank 0's host Film adds the
overlapping blocks in the FRAME's tile order (Film::MergeShards: tile t from shard t % world), i.e. the order a one-GPU
render merges them in: the N-GPU image is the one-GPU image bit for bit for every filter.
"""
import numpy as np
import torch
import torch.distributed as dist


class ShardBuffer:
    """One rank's shard of a frame as ONE buffer of 4-byte words, equal in size on every rank so that one gather suffices:
    [ film: n_tiles_max * tile_pixels x (r, g, b, weight) float32 | strays: max_strays x PgStraySample (8 words) | count, pad ].
    `film`, `strays` and `nstrays` are views; their data_ptr()s are what pg_render writes through."""

    def __init__(self, n_tiles_max, device, tile_pixels=256, max_strays=None):
        self.max_strays = n_tiles_max * tile_pixels // 8 + 1024 if max_strays is None else int(max_strays)
        fw, sw = n_tiles_max * tile_pixels * 4, self.max_strays * 8
        self.words = torch.zeros(fw + sw + 4, dtype=torch.int32, device=device)
        self.film = self.words[:fw].view(torch.float32).view(-1, 4)
        self.strays = self.words[fw:fw + sw].view(-1, 8)
        self.nstrays = self.words[fw + sw:fw + sw + 1]
        self._fw, self._sw = fw, sw

    def views_of(self, words):
        """(film, strays, n) views of another rank's gathered words."""
        fw, sw = self._fw, self._sw
        return words[:fw].view(torch.float32).view(-1, 4), words[fw:fw + sw].view(-1, 8), int(words[fw + sw].item())


class FilmGather:
    """The per-frame gather to `dst`.  With `async_op` the collective is only enqueued (RCCL orders it behind the render on the
    current stream and runs it on its own stream): the next frame -- rendered into the OTHER ShardBuffer of a pair -- overlaps it,
    and wait() is called before a buffer is rendered into again.  On `dst` every ShardBuffer has its OWN set of receive buffers
    (allocated on first use, kept): what shards(b) returns stays valid while the next frame is gathered into the other buffer's
    set -- until `b` itself is gathered again."""

    def __init__(self, shard, dst=0, comm_device=None):
        self.dst, self.rank, self.world = dst, dist.get_rank(), dist.get_world_size()
        self.comm_device = torch.device(comm_device) if comm_device is not None else shard.words.device
        self.staged = self.comm_device != shard.words.device  # pre-flight only: ranks sharing one GPU gather host copies through gloo
        self.pending = None
        self._recv_for(shard)

    def _recv_for(self, shard):
        if self.rank != self.dst:
            return None
        # the receive set lives ON the ShardBuffer (a dict keyed by id() would hand a new buffer of another shape the set of a dead one)
        r = getattr(shard, "_recv_set", None)
        if r is None or r[0].shape != shard.words.shape or r[0].device != self.comm_device or len(r) != self.world:
            r = shard._recv_set = [torch.empty(shard.words.shape, dtype=torch.int32, device=self.comm_device) for _ in range(self.world)]
        return r

    def start(self, shard, async_op=True):
        self.wait()
        src = shard.words.to(self.comm_device) if self.staged else shard.words
        self.pending = dist.gather(src, self._recv_for(shard), dst=self.dst, async_op=async_op)
        return self.pending

    def wait(self):
        if self.pending is not None:
            self.pending.wait()
            self.pending = None

    def shards(self, shard):
        """[(film, strays, n)] per rank on dst after wait(): views of `shard`'s own receive set."""
        self.wait()
        return [shard.views_of(w) for w in self._recv_for(shard)]


# --- the older three-buffer interface (tests, small tools): same transport, one gather per buffer
def shard_buffers(n_tiles_max, device, tile_pixels=256):
    """Fixed-size per-rank buffers (views of one ShardBuffer); tile_pixels = PgRenderDesc.tile_pixels."""
    b = ShardBuffer(n_tiles_max, device, tile_pixels)
    return b.film, b.strays, b.nstrays, b.max_strays


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
    dist.gather(film.contiguous(), lists[0] if lists else None, dst=dst)
    dist.gather(strays.contiguous(), lists[1] if lists else None, dst=dst)
    dist.gather(nstrays.contiguous(), lists[2] if lists else None, dst=dst)
    return lists


def merge_shards(pkg, scene, tile_count, shards):
    """Film::MergeFilmTile for the frame's tiles in tile order on the host (tile t from shard t % world: Film::MergeShards);
    shards = [(film, strays, n), ...] per rank.  Returns the final (h, w, 3) image."""
    world = len(shards)
    scene.film_clear()
    arrays = []
    for r, (film, strays, n) in enumerate(shards):
        rd = scene.render_desc(tile_first=r, tile_step=world)
        nt = tile_count(rd)
        f = np.ascontiguousarray(film.detach().cpu().numpy()[:nt * rd.tile_pixels]).view(pkg.FILM_PIXEL_DTYPE).reshape(-1)
        s = np.ascontiguousarray(strays.detach().cpu().numpy()[:int(n)]).view(pkg.STRAY_DTYPE).reshape(-1)
        arrays.append((f, s))
    scene.film_merge_shards(scene.render_desc(tile_first=0, tile_step=1), arrays)
    return scene.film_image()
