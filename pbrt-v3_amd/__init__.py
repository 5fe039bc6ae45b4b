"""pbrt-v3 path-tracing hot path, MI355X-native: Python plumbing.

The product is two native libraries built in-tree by ``make -C pbrt-v3_amd``:

* ``libpbrt_host.so`` -- C++ mirror of pbrt-v3's scene-file front end and API
  state machine (parser, graphics state, BVHAccel build, Film); flattens a
  ``.pbrt`` file into ``PgSceneDesc`` / ``PgRenderDesc``.
* ``libpbrt_gpu.so``  -- hand-written HIP kernels for gfx950 behind the C ABI
  of ``include/pbrt_gpu.h`` (wavefront PathIntegrator, BVH traversal).

This module only binds them with ctypes so tests, ``bench.py`` and
``torch.distributed`` sharding can drive the same entry points the
``pbrt_amd`` CLI uses.  There is no Python or CPU fallback for the device
path: if ``libpbrt_gpu.so`` is missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

from . import abi
from .abi import (PgCounters, PgFilmPixel, PgRenderDesc, PgSceneDesc, PgStraySample, PG_MEM_DEVICE, PG_MEM_HOST,
                  PG_OK)

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libpbrt_host.so")
GPU_LIB_PATH = os.environ.get("PBRT_GPU_LIB") or os.path.join(_HERE, "libpbrt_gpu.so")  # override: kernel A/B builds only

FILM_PIXEL_DTYPE = np.dtype([("rgb", np.float32, 3), ("weight", np.float32)])
STRAY_DTYPE = np.dtype([("px", np.int32), ("py", np.int32), ("src_px", np.int32), ("src_py", np.int32),
                        ("rgb", np.float32, 3), ("weight", np.float32)])

_host = None
_gpu = None


class PbrtGpuError(RuntimeError):
    pass


def host_lib():
    global _host
    if _host is None:
        if not os.path.exists(HOST_LIB_PATH):
            raise PbrtGpuError(f"{HOST_LIB_PATH} not built; run __graft_entry__.build() or make -C pbrt-v3_amd")
        _host = abi.bind(C.CDLL(HOST_LIB_PATH), abi.HOST_SYMBOLS)
    return _host


def gpu_lib():
    """The HIP back end.  Fails loudly when the extension is missing."""
    global _gpu
    if _gpu is None:
        if not os.path.exists(GPU_LIB_PATH):
            raise PbrtGpuError(f"{GPU_LIB_PATH} not built; run __graft_entry__.build() (there is no CPU fallback)")
        _gpu = abi.bind(C.CDLL(GPU_LIB_PATH), abi.GPU_SYMBOLS)
    return _gpu


def _check(status, what):
    if status != PG_OK:
        msg = gpu_lib().pg_last_error()
        raise PbrtGpuError(f"{what} failed ({status}): {msg.decode() if msg else ''}")


class HostScene:
    """A parsed .pbrt scene: pbrtInit/pbrtParseFile/pbrtCleanup up to (not including) Render."""

    def __init__(self, filename=None, text=None, quick=False, crop=None):
        lib = host_lib()
        cropv = (C.c_float * 4)(*crop) if crop is not None else None
        if filename is not None:
            self._h = lib.pbrt_host_load_file(os.fsencode(filename), int(quick), cropv)
        else:
            self._h = lib.pbrt_host_load_string(text.encode(), int(quick), cropv)
        if not self._h:
            raise PbrtGpuError(f"scene {filename or '<string>'} produced no renderable scene")
        self.desc = lib.pbrt_host_scene_desc(self._h).contents

    def close(self):
        if getattr(self, "_h", None):
            host_lib().pbrt_host_free(self._h)
            self._h = None

    __del__ = close

    def render_desc(self, tile_first=0, tile_step=1):
        rd = PgRenderDesc()
        host_lib().pbrt_host_render_desc(self._h, C.byref(rd))
        rd.tile_first, rd.tile_step = tile_first, tile_step
        return rd

    @property
    def film_size(self):
        w, h = C.c_int(), C.c_int()
        host_lib().pbrt_host_film_size(self._h, C.byref(w), C.byref(h))
        return w.value, h.value

    def film_clear(self):
        host_lib().pbrt_host_film_clear(self._h)

    def film_merge(self, rd, film, strays):
        """Film::MergeFilmTile for one shard (numpy arrays of FILM_PIXEL_DTYPE / STRAY_DTYPE)."""
        film = np.ascontiguousarray(film)
        strays = np.ascontiguousarray(strays)
        host_lib().pbrt_host_film_merge(self._h, C.byref(rd), film.ctypes.data, strays.ctypes.data, len(strays))

    def film_merge_shards(self, full_rd, shards):
        """The n shards [(film, strays), ...] of ONE frame (shard r = the tiles t = r (mod n)), merged in the frame's own tile order:
        Film::MergeShards.  `full_rd` describes the whole frame (tile_first 0, tile_step 1)."""
        n = len(shards)
        films = [np.ascontiguousarray(f) for f, _ in shards]
        strays = [np.ascontiguousarray(s) for _, s in shards]
        fp = (C.c_void_p * n)(*[f.ctypes.data for f in films])
        sp = (C.c_void_p * n)(*[s.ctypes.data for s in strays])
        ns = (C.c_int * n)(*[len(s) for s in strays])
        host_lib().pbrt_host_film_merge_shards(self._h, C.byref(full_rd), n, fp, sp, ns)

    def film_image(self):
        """Final RGB image as Film::WriteImage computes it: (h, w, 3) float32, top row first."""
        w, h = self.film_size
        img = np.empty((h, w, 3), np.float32)
        host_lib().pbrt_host_film_image(self._h, img.ctypes.data)
        return img

    # numpy views of the flattened scene (for tests / diagnostics)
    def nodes(self):
        return np.ctypeslib.as_array(C.cast(self.desc.nodes, C.POINTER(C.c_uint8)), (self.desc.n_nodes * 32,)).view(
            np.dtype([("bmin", np.float32, 3), ("bmax", np.float32, 3), ("offset", np.int32), ("nprims", np.uint16),
                      ("axis", np.uint8), ("pad", np.uint8)]))

    def indices(self):
        return np.ctypeslib.as_array(self.desc.indices, (self.desc.n_tris, 3))

    def positions(self):
        return np.ctypeslib.as_array(self.desc.P, (self.desc.n_verts, 3))


NODE_DTYPE = np.dtype([("bmin", np.float32, 3), ("bmax", np.float32, 3), ("offset", np.int32), ("nprims", np.uint16),
                       ("axis", np.uint8), ("pad", np.uint8)])


def hlbvh_build(bounds, max_prims_in_node=4, device=True):
    """BVHAccel::HLBVHBuild + flattenBVHTree over bare primitive bounds (n x 6 float32: pMin, pMax): on the device through
    pg_hlbvh_build, or with the host front end's builder (device=False).  Returns (nodes, ordered primitive indices)."""
    bounds = np.ascontiguousarray(bounds, np.float32).reshape(-1, 6)
    n = len(bounds)
    nodes = np.zeros(max(1, 2 * n), NODE_DTYPE)
    order = np.zeros(max(1, n), np.int32)
    if device:
        nn = C.c_int32(0)
        _check(gpu_lib().pg_hlbvh_build(n, bounds.ctypes.data, max_prims_in_node, nodes.ctypes.data, C.byref(nn), order.ctypes.data), "pg_hlbvh_build")
    else:
        nn = C.c_int(0)
        host_lib().pbrt_host_hlbvh_build(n, bounds.ctypes.data, max_prims_in_node, nodes.ctypes.data, C.byref(nn), order.ctypes.data)
    return nodes[:nn.value], order[:n]


def set_device_bvh(on):
    """Scenes loaded afterwards build their "hlbvh" accelerators on the device (pbrt_amd --devicebvh)."""
    host_lib().pbrt_host_set_device_bvh(1 if on else 0)


def write_pfm(filename, img):
    img = np.ascontiguousarray(img, np.float32)
    h, w, _ = img.shape
    if host_lib().pbrt_host_write_pfm(os.fsencode(filename), img.ctypes.data, w, h) != 0:
        raise PbrtGpuError(f"cannot write {filename}")


def read_pfm(filename):
    """PFM reader (core/imageio.cpp ReadImagePFM semantics): returns (h, w, 3) float32, top row first."""
    with open(filename, "rb") as f:
        if f.readline().strip() != b"PF":
            raise ValueError("not a colour PFM")
        w, h = map(int, f.readline().split())
        scale = float(f.readline())
        data = np.frombuffer(f.read(w * h * 12), "<f4" if scale < 0 else ">f4").reshape(h, w, 3)
    return np.ascontiguousarray(data[::-1]).astype(np.float32)


def default_max_strays(rd, n_tiles):
    """Room for the box filter's stray samples (a sample whose offset inside its pixel is exactly 0 also lands in the previous
    pixel, film.h:127-132): rare, except under the MaxMinDistSampler, whose first sample of every pixel is (0, 0) by construction
    and lands in up to three neighbours."""
    per_tile = 256 * 4 if rd.sampler == abi.PG_SAMPLER_MAXMINDIST else 256 // 8
    return n_tiles * per_tile + 1024


class GpuScene:
    """Device-resident scene (pg_scene_create) and the render / intersect entry points."""

    def __init__(self, desc, device=0):
        lib = gpu_lib()
        _check(lib.pg_set_device(device), "pg_set_device")
        self.device = device
        h = C.c_void_p()
        _check(lib.pg_scene_create(C.byref(desc), C.byref(h)), "pg_scene_create")
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            gpu_lib().pg_scene_destroy(self._h)
            self._h = None

    __del__ = close

    @staticmethod
    def tile_count(rd):
        return gpu_lib().pg_render_tile_count(C.byref(rd))

    def render(self, rd, max_strays=None, stream=None):
        """Integrator::Render for the shard in rd into host numpy buffers (film, strays)."""
        n = self.tile_count(rd)
        if max_strays is None:
            max_strays = default_max_strays(rd, n)
        film = np.zeros(n * rd.tile_pixels, FILM_PIXEL_DTYPE)
        strays = np.zeros(max_strays, STRAY_DTYPE)
        ns = C.c_int32(0)
        _check(gpu_lib().pg_render(self._h, C.byref(rd), film.ctypes.data, strays.ctypes.data, max_strays, C.byref(ns),
                                   PG_MEM_HOST, stream), "pg_render")
        return film, strays[:ns.value]

    def render_device(self, rd, film_ptr, strays_ptr, max_strays, nstrays_ptr, stream=None):
        """Same, into caller-owned device buffers (raw pointers; e.g. torch tensors' data_ptr())."""
        _check(gpu_lib().pg_render(self._h, C.byref(rd), film_ptr, strays_ptr, max_strays, nstrays_ptr, PG_MEM_DEVICE,
                                   stream), "pg_render")

    def intersect(self, o, d, tmax):
        """Batched Scene::Intersect on host arrays: returns prim (int32), t, bary (n,3)."""
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = len(tmax)
        prim = np.empty(n, np.int32)
        t = np.empty(n, np.float32)
        bary = np.empty((n, 3), np.float32)
        _check(gpu_lib().pg_intersect(self._h, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, prim.ctypes.data,
                                      t.ctypes.data, bary.ctypes.data, PG_MEM_HOST, None), "pg_intersect")
        return prim, t, bary

    def intersect_p(self, o, d, tmax):
        o = np.ascontiguousarray(o, np.float32)
        d = np.ascontiguousarray(d, np.float32)
        tmax = np.ascontiguousarray(tmax, np.float32)
        n = len(tmax)
        occ = np.empty(n, np.uint8)
        _check(gpu_lib().pg_intersect_p(self._h, n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, occ.ctypes.data,
                                        PG_MEM_HOST, None), "pg_intersect_p")
        return occ

    def counters(self):
        c = PgCounters()
        _check(gpu_lib().pg_counters(self._h, C.byref(c)), "pg_counters")
        return c.as_dict()

    def counters_reset(self):
        _check(gpu_lib().pg_counters_reset(self._h), "pg_counters_reset")

    def set_option(self, option, value):
        """pg_scene_set_option (abi.PG_OPT_*): per-scene switches that never change an image."""
        _check(gpu_lib().pg_scene_set_option(self._h, int(option), int(value)), "pg_scene_set_option")


def render_sharded(gpu_scenes, rd, max_strays=None):
    """pg_render_sharded: the frame of rd (tile_first 0, tile_step 1) over the devices of gpu_scenes -- one host thread per
    device, the packed shards gathered on the first device by one ncclGather (RCCL) or, where RCCL cannot run, peer copies
    (shard_transport() tells which).  Returns [(shard rd, film, strays)] per rank."""
    n = len(gpu_scenes)
    shards = []
    for r in range(n):
        srd = PgRenderDesc.from_buffer_copy(rd)
        srd.tile_first, srd.tile_step = r, n
        shards.append(srd)
    counts = [GpuScene.tile_count(s) for s in shards]
    if max_strays is None:
        max_strays = default_max_strays(rd, max(counts))
    films = [np.zeros(c * rd.tile_pixels, FILM_PIXEL_DTYPE) for c in counts]
    strays = [np.zeros(max_strays, STRAY_DTYPE) for _ in range(n)]
    handles = (C.c_void_p * n)(*[g._h for g in gpu_scenes])
    fptr = (C.c_void_p * n)(*[f.ctypes.data for f in films])
    sptr = (C.c_void_p * n)(*[s.ctypes.data for s in strays])
    ns = (C.c_int32 * n)()
    _check(gpu_lib().pg_render_sharded(handles, n, C.byref(rd), fptr, sptr, max_strays, ns), "pg_render_sharded")
    return [(shards[r], films[r], strays[r][:ns[r]]) for r in range(n)]


def shard_transport():
    """How the last render_sharded gathered: "rccl" or "peer (<reason>)"."""
    return gpu_lib().pg_shard_transport().decode()


def render_scene(scene, device=0):
    """Whole-frame render of a HostScene on one GPU; returns the final (h, w, 3) image."""
    gs = GpuScene(scene.desc, device)
    try:
        rd = scene.render_desc()
        film, strays = gs.render(rd)
        scene.film_clear()
        scene.film_merge(rd, film, strays)
        return scene.film_image(), gs.counters()
    finally:
        gs.close()
