"""Parity of the HIP path (through the C ABI) against the CPU oracle, the reference's golden images and the
reference's own test vectors.  Bars: bit-exact for hits / indices / counts of the traversal kernels AND for images: every
pixel of every golden equals the unmodified reference binary's (BASELINE.json asks for |delta| <= 1e-4; the device computes
libm's float functions as glibc does, csrc/pg_libm.h, so nothing is left to tolerate)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import GOLD, INTEGRATOR_STATS, check_integrator_stats, golden_names
from kat_util import jittered_sphere  # noqa: F401
from test_reference_kats import quadric_reintersect_run, reintersect_cases, sphere_scene, watertight_rays

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel_err(img, ref):
    return np.abs(img - ref) / np.maximum(1.0, np.abs(ref))


# Wide filters: a pixel near a tile border sums contributions of several FilmTiles, and the reference merges tiles in the order its
# threads finish them (film.cpp:117-130) -- its own image is reproducible to the last bit only in that order.  The goldens of these
# scenes were rendered with one thread (oracle/make_golden.py), i.e. in tile order, which is the order the device's film merge uses.
@pytest.mark.parametrize("name", golden_names())
def test_golden_images(gpu, oracle, name):
    """The device image against the image the UNMODIFIED reference binary rendered: IDENTICAL, every pixel, every bit -- and its
    statistics (camera / regular / shadow rays, ray-triangle tests, the integrator's path statistics) equal the reference's printed counters exactly.  No tolerance and
    no exempted pixels: until round 3 the device evaluated sinf / cosf / acosf / atan2f / logf / expf in double (correctly rounded),
    glibc's float versions are not, and a last-bit difference there could tip a discrete event of a sample a few bounces later
    (2 pixels per image were exempted, 0.5 - 2 % in the noise-bump and subsurface scenes, whole tiles under the tile-serial samplers).
    csrc/pg_libm.h now computes those functions as the reference's libm does (tests/test_libm_restated.py: all 2^32 arguments)."""
    scene = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    img, cn = gpu.render_scene(scene)
    ref = gpu.read_pfm(os.path.join(GOLD, name + ".pfm"))
    assert img.shape == ref.shape
    same = img.view(np.uint32) == ref.view(np.uint32)
    assert same.all(), f"{(~same).any(axis=2).sum()} of {same.shape[0] * same.shape[1]} pixels differ from the reference binary's image, max rel err {rel_err(img, ref).max():.3e}"
    stats = json.load(open(os.path.join(GOLD, name + ".json")))
    for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests"):
        assert cn[k] == stats[k], (k, cn[k], stats[k])
    check_integrator_stats(cn, stats)  # zero-radiance paths, path length, volume / surface interactions as the reference printed them


@pytest.mark.parametrize("name", golden_names())
def test_film_bit_identical_to_oracle(gpu, oracle, name):
    """Below the image: every film pixel (the sums before the final division), every stray sample and the node-visit count of the
    device equal, bit for bit, the CPU restatement that tests/test_oracle_vs_reference.py pins against the reference binary."""
    scene = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    gs = gpu.GpuScene(scene.desc)
    rd = scene.render_desc()
    film, strays = gs.render(rd)
    cn = gs.counters()
    ofilm, ostrays, ocn = oracle.render(scene.desc, rd)
    assert np.array_equal(film["weight"], ofilm["weight"])
    assert np.array_equal(film["rgb"], ofilm["rgb"]), f"{(film['rgb'] != ofilm['rgb']).any(axis=1).sum()} pixels differ"
    key = lambda s: np.lexsort((s["src_px"], s["src_py"], s["px"], s["py"]))
    a, b = strays[key(strays)], ostrays[key(ostrays)]
    assert len(a) == len(b)
    for f in ("px", "py", "src_px", "src_py", "weight", "rgb"):
        assert np.array_equal(a[f], b[f]), f
    for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests", "node_visits") + INTEGRATOR_STATS:  # the reference's counters, exactly
        assert cn[k] == ocn[k], (k, cn[k], ocn[k])
    gs.close()


@pytest.mark.parametrize("name", ["cornell_32", "cornell_plastic", "plastic_topdown", "cornell_mirror_glass", "cornell_glass_eta", "cornell_orennayar",
                                  "cornell_tangents", "cornell_spot_power", "synthetic_n40"])
def test_general_kernels_equal_specialised_kernels(gpu, name, monkeypatch):
    """matte / plastic / mirror / glass run in shading kernels specialised to those materials' BxDF lists; PG_FORCE_EXT=1 sends
    the same scenes through the general kernels (BxDF-list BSDF, sphere / infinite-light support compiled in): same film, bit
    for bit, same counters."""
    scene = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    rd = scene.render_desc()
    gs = gpu.GpuScene(scene.desc)
    film, strays = gs.render(rd)
    cn = gs.counters()
    gs.close()
    monkeypatch.setenv("PG_FORCE_EXT", "1")
    gs2 = gpu.GpuScene(scene.desc)
    film2, strays2 = gs2.render(rd)
    cn2 = gs2.counters()
    gs2.close()
    assert np.array_equal(film["rgb"], film2["rgb"]) and np.array_equal(film["weight"], film2["weight"])
    assert len(strays) == len(strays2)
    for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests", "node_visits"):
        assert cn[k] == cn2[k], k


def random_rays(scene, n, seed):
    rng = np.random.default_rng(seed)
    nodes = scene.nodes()
    lo, hi = nodes["bmin"][0], nodes["bmax"][0]
    ext = hi - lo
    o = (lo - 0.2 * ext + 1.4 * ext * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    # axis-parallel and single-zero-component directions: invDir = +-inf paths of Bounds3::IntersectP
    d[: n // 20, 0] = 0
    d[n // 20: n // 10, 1:] = 0
    d[n // 10: n // 8] *= np.float32(1e-3)
    return o, d


@pytest.mark.parametrize("n", [0, 1, 63, 256, 257, 10000])
def test_intersect_bit_exact(gpu, oracle, n):
    scene = gpu.HostScene(os.path.join(GOLD, "synthetic_n40.pbrt"))
    gs = gpu.GpuScene(scene.desc)
    o, d = random_rays(scene, n, 7 + n)
    tmax = np.full(n, np.inf, np.float32)
    tmax[::7] = 0.75
    gs.counters_reset()
    prim, t, bary = gs.intersect(o, d, tmax)
    oprim, ot, obary, ocn = oracle.intersect(scene.desc, o, d, tmax)
    assert np.array_equal(prim, oprim)
    assert np.array_equal(t, ot) and np.array_equal(bary, obary)
    cn = gs.counters()
    assert cn["closest_node_visits"] == ocn["node_visits"] and cn["closest_tri_tests"] == ocn["tri_tests"]
    occ = gs.intersect_p(o, d, tmax)
    oocc, ocn2 = oracle.intersect_p(scene.desc, o, d, tmax)
    assert np.array_equal(occ, oocc)
    cn = gs.counters()
    assert cn["shadow_node_visits"] == ocn2["node_visits"] and cn["shadow_tri_tests"] == ocn2["tri_tests"]
    if n > 1:
        assert (prim >= 0).any() and (prim < 0).any()
        assert ((prim >= 0) == (occ == 1)).all()  # closest-hit and any-hit agree on hit/miss
    gs.close()


def test_alpha_mask_rays_bit_exact(gpu, oracle):
    """Alpha / shadow-alpha textures inside the traversal: Triangle::Intersect rejects hits whose alpha texture evaluates to
    0, IntersectP also those whose shadow alpha does -- same hits, same counters as the oracle for every ray."""
    scene = gpu.HostScene(os.path.join(GOLD, "alpha_masks.pbrt"))
    assert scene.desc.n_alphas >= 4
    gs = gpu.GpuScene(scene.desc)
    n = 30000
    o, d = random_rays(scene, n, 321)
    tmax = np.full(n, np.inf, np.float32)
    gs.counters_reset()
    prim, t, bary = gs.intersect(o, d, tmax)
    oprim, ot, obary, ocn = oracle.intersect(scene.desc, o, d, tmax)
    assert np.array_equal(prim, oprim) and np.array_equal(t, ot) and np.array_equal(bary, obary)
    cn = gs.counters()
    assert cn["closest_node_visits"] == ocn["node_visits"] and cn["closest_tri_tests"] == ocn["tri_tests"]
    occ = gs.intersect_p(o, d, tmax)
    oocc, ocn2 = oracle.intersect_p(scene.desc, o, d, tmax)
    assert np.array_equal(occ, oocc)
    assert ((prim >= 0) & (occ == 0)).sum() > 20  # rays stopped by a surface that only the shadow-alpha mask lets through
    cn = gs.counters()
    assert cn["shadow_node_visits"] == ocn2["node_visits"] and cn["shadow_tri_tests"] == ocn2["tri_tests"]
    gs.close()


@pytest.mark.parametrize("name", ["instance_boxes", "instance_accel"])
def test_instance_rays_bit_exact(gpu, oracle, name):
    """Object instances (TransformedPrimitive over an object definition's own BVH, or over a lone primitive): the ray is
    carried into instance space, traverses the second-level BVH and comes back with r.tMax = the instance-space tHit --
    primitive, t, barycentrics and the node-visit / triangle-test counters equal the oracle's for every ray."""
    scene = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    assert scene.desc.n_instances >= 7 and scene.desc.n_objects == 3 and scene.desc.n_prims_all > scene.desc.n_tris
    gs = gpu.GpuScene(scene.desc)
    n = 20000
    o, d = random_rays(scene, n, 123)
    tmax = np.full(n, np.inf, np.float32)
    tmax[::5] = 300.0
    gs.counters_reset()
    prim, t, bary = gs.intersect(o, d, tmax)
    oprim, ot, obary, ocn = oracle.intersect(scene.desc, o, d, tmax)
    assert (prim >= scene.desc.n_tris).sum() > 500  # hits on primitives of object definitions
    assert np.array_equal(prim, oprim) and np.array_equal(t, ot) and np.array_equal(bary, obary)
    cn = gs.counters()
    assert cn["closest_node_visits"] == ocn["node_visits"] and cn["closest_tri_tests"] == ocn["tri_tests"]
    occ = gs.intersect_p(o, d, tmax)
    oocc, ocn2 = oracle.intersect_p(scene.desc, o, d, tmax)
    assert np.array_equal(occ, oocc)
    cn = gs.counters()
    assert cn["shadow_node_visits"] == ocn2["node_visits"] and cn["shadow_tri_tests"] == ocn2["tri_tests"]
    gs.close()


@pytest.mark.parametrize("name", ["sphere_light", "sphere_partial", "sphere_enclosing", "quadrics", "quadric_lights"])
def test_sphere_rays_bit_exact(gpu, oracle, name):
    """Shape "sphere" in the BVH next to triangles: Sphere::Intersect / IntersectP (error-bounded quadratic, partial-sphere
    clipping, transforms) decide every ray as the oracle does, tHit included; a sphere hit reports (tHit, 0, 0) as its bary."""
    scene = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    assert scene.desc.n_spheres >= 1
    gs = gpu.GpuScene(scene.desc)
    n = 20000
    o, d = random_rays(scene, n, 99)
    flags = np.ctypeslib.as_array(scene.desc.tri_flags, (scene.desc.n_tris,))
    tmax = np.full(n, np.inf, np.float32)
    tmax[::5] = 300.0
    gs.counters_reset()
    prim, t, bary = gs.intersect(o, d, tmax)
    oprim, ot, obary, ocn = oracle.intersect(scene.desc, o, d, tmax)
    on_sphere = (prim >= 0) & ((flags[np.maximum(prim, 0)] & 32) != 0)
    assert on_sphere.sum() > 50
    assert np.array_equal(prim, oprim) and np.array_equal(t, ot) and np.array_equal(bary, obary)
    assert np.array_equal(bary[on_sphere, 0], t[on_sphere])
    cn = gs.counters()
    assert cn["closest_node_visits"] == ocn["node_visits"] and cn["closest_tri_tests"] == ocn["tri_tests"]
    occ = gs.intersect_p(o, d, tmax)
    oocc, ocn2 = oracle.intersect_p(scene.desc, o, d, tmax)
    assert np.array_equal(occ, oocc)
    cn = gs.counters()
    assert cn["shadow_node_visits"] == ocn2["node_visits"] and cn["shadow_tri_tests"] == ocn2["tri_tests"]
    gs.close()


def test_watertight_and_degenerate(gpu, oracle):
    """Triangle.Watertight (tests/shapes.cpp:28-129) through the BVH kernel, plus zero-area triangles
    (Triangle::Intersect's 'bogus intersection' rejection, triangle.cpp:309-317) and tMax edge cases."""
    verts, idx = jittered_sphere()
    scene = gpu.HostScene(text=sphere_scene(verts, idx))
    gs = gpu.GpuScene(scene.desc)
    o, d = watertight_rays(verts, 20000)
    inf = np.full(len(o), np.inf, np.float32)
    prim, t, bary = gs.intersect(o, d, inf)
    assert (prim >= 0).all()
    oprim, ot, obary, _ = oracle.intersect(scene.desc, o, d, inf)
    assert np.array_equal(prim, oprim) and np.array_equal(t, ot) and np.array_equal(bary, obary)
    zero = np.zeros(len(o), np.float32)
    assert (gs.intersect(o, d, zero)[0] < 0).all() and not gs.intersect_p(o, d, zero).any()
    gs.close()
    degenerate = ('Film "image" "integer xresolution" [8] "integer yresolution" [8] "string filename" "d.pfm"\nWorldBegin\n'
                  'Shape "trianglemesh" "integer indices" [ 0 1 2  3 4 5 ] "point P" [ 0 0 1  1 1 1  2 2 1   -1 -1 2  1 -1 2  0 1 2 ]\nWorldEnd\n')
    scene = gpu.HostScene(text=degenerate)
    gs = gpu.GpuScene(scene.desc)
    o = np.array([[1, 1, 0], [0, 0, 0], [0.5, 0.5, 0]], np.float32)
    d = np.array([[0, 0, 1]] * 3, np.float32)
    prim, t, _ = gs.intersect(o, d, np.full(3, np.inf, np.float32))
    oprim, ot, _, _ = oracle.intersect(scene.desc, o, d, np.full(3, np.inf, np.float32))
    assert np.array_equal(prim, oprim) and np.array_equal(t, ot)
    assert (t[prim >= 0] == 2).all()  # only the real triangle at z = 2 is ever hit
    gs.close()


def test_triangle_reintersect_on_device(gpu, oracle):
    """Triangle.Reintersect (src/tests/shapes.cpp:154-205) on the HIP traversal kernels: rays spawned from a hit point with the
    reference's error bounds and OffsetRayOrigin (SpawnRay, and SpawnRayTo with tMax = 1 - ShadowEpsilon) never hit their own triangle
    again -- 24 triangles with coordinates from 1e-8 to 1e8, 240 rays each, closest-hit and any-hit kernels, and the answers equal the
    oracle's (tests/test_reference_kats.py::test_triangle_reintersect_through_the_scene is the same check without a GPU)."""
    n = 0
    for text, o, d, tmax in reintersect_cases(oracle):
        scene = gpu.HostScene(text=text)
        gs = gpu.GpuScene(scene.desc)
        prim, t, bary = gs.intersect(o, d, tmax)
        occ = gs.intersect_p(o, d, tmax)
        gs.close()
        oprim, ot, obary, _ = oracle.intersect(scene.desc, o, d, tmax)
        oocc, _ = oracle.intersect_p(scene.desc, o, d, tmax)
        assert (prim < 0).all() and not np.asarray(occ).any()
        assert np.array_equal(prim, oprim) and np.array_equal(np.asarray(occ, bool), np.asarray(oocc, bool))
        n += len(tmax)
    assert n >= 24 * 240


@pytest.mark.parametrize("kind", ["full_sphere", "partial_sphere", "cylinder"])
def test_quadric_reintersect_on_device(gpu, oracle, kind):
    """FullSphere / PartialSphere / Cylinder .Reintersect (src/tests/shapes.cpp:372-513) on the HIP kernels: 40 random quadrics each
    (radius 1e-4 .. 1e4, clipped in z and phi), hit points found with the oracle, 300 rays spawned from each into the normal's
    hemisphere -- none may hit the shape again, on the closest-hit and the any-hit kernel.  (tests/test_reference_kats.py runs the
    same rays through the oracle and through its correctly-rounded-libm build, whose arithmetic is the device's.)"""
    quadric_reintersect_run(gpu, oracle, kind, device=gpu)


def test_sharded_render_equals_whole(gpu):
    scene = gpu.HostScene(os.path.join(GOLD, "cornell_40x24.pbrt"))
    gs = gpu.GpuScene(scene.desc)
    whole, _ = gpu.render_scene(scene)
    for world in (2, 3, 5):
        scene.film_clear()
        for r in range(world):
            rd = scene.render_desc(r, world)
            film, strays = gs.render(rd)
            scene.film_merge(rd, film, strays)
        assert np.array_equal(scene.film_image(), whole), world
    gs.close()


def test_device_buffers_and_determinism(gpu):
    import torch
    scene = gpu.HostScene(os.path.join(GOLD, "synthetic_n40.pbrt"))
    gs = gpu.GpuScene(scene.desc)
    rd = scene.render_desc()
    film_h, strays_h = gs.render(rd)
    n = gs.tile_count(rd)
    film = torch.zeros((n * 256, 4), device="cuda")
    strays = torch.zeros((4096, 8), dtype=torch.int32, device="cuda")
    ns = torch.zeros(1, dtype=torch.int32, device="cuda")
    for _ in range(2):
        gs.render_device(rd, film.data_ptr(), strays.data_ptr(), 4096, ns.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(film.cpu().numpy().reshape(-1), film_h.view(np.float32).reshape(-1))
        assert int(ns.item()) == len(strays_h)
    # small batches (several sample passes and tile groups) give the same film
    os.environ["PG_BATCH_PATHS"] = "1024"
    try:
        film_b, strays_b = gs.render(rd)
    finally:
        del os.environ["PG_BATCH_PATHS"]
    assert np.array_equal(film_b, film_h) and len(strays_b) == len(strays_h)
    gs.close()


def test_unsupported_inputs_fail_loudly(gpu):
    scene = gpu.HostScene(os.path.join(GOLD, "cornell_32.pbrt"))
    desc = scene.desc
    mats = (gpu.abi.PgMaterial * desc.n_materials)(*[desc.materials[i] for i in range(desc.n_materials)])
    mats[0].type = 7
    bad = gpu.abi.PgSceneDesc.from_buffer_copy(desc)
    bad.materials = mats
    with pytest.raises(gpu.PbrtGpuError, match="unknown type 7"):
        gpu.GpuScene(bad)
    mats[0].type = 1
    mats[0].first_bxdf = desc.n_bxdfs  # a BxDF list that runs off the end of the scene's table
    mats[0].n_bxdfs = 2
    with pytest.raises(gpu.PbrtGpuError, match="BxDF list"):
        gpu.GpuScene(bad)
    gs = gpu.GpuScene(desc)
    rd = scene.render_desc()
    rd.filter_radius[0] = 2.0  # a wide filter must come with filter_general / its tile block geometry
    with pytest.raises(gpu.PbrtGpuError, match="filter_general"):
        gs.render(rd)
    gs.close()


def _chain_bvh(gpu, desc, k):
    """A BVH in the reference's array layout (bvh.cpp:640-658) that is a chain of k interior nodes -- what a "middle" or "equal"
    split degenerates to on hostile input: the first child of interior node j is interior node j + 1, every second child a leaf
    with the scene's first primitive."""
    nodes = (gpu.abi.PgBVHNode * (2 * k + 1))()
    root = desc.nodes[0]
    for i in range(2 * k + 1):
        for c in range(3):
            nodes[i].bmin[c], nodes[i].bmax[c] = root.bmin[c], root.bmax[c]
        if i < k:
            nodes[i].nprims, nodes[i].axis = 0, i % 3
            nodes[i].offset = k + 1 + (k - 1 - i)
        else:
            nodes[i].nprims, nodes[i].offset = 1, 0
    return nodes


def test_malformed_or_too_deep_bvh_is_refused(gpu, oracle):
    """pg_scene_create checks the caller's node array before it indexes anything by it: children out of range, shared by two
    parents or unreachable from the root are PG_ERR_INVALID; a tree with more than 64 pending entries -- the reference's
    nodesToVisit[64] (bvh.cpp:670, :708), which the reference itself does not guard -- is PG_ERR_UNSUPPORTED instead of a write past
    the kernel's stack.  The deepest tree that is accepted (64 entries: 11 in LDS, 53 behind them) traverses like the oracle."""
    scene = gpu.HostScene(os.path.join(GOLD, "cornell_32.pbrt"))
    desc = scene.desc
    nn = desc.n_nodes
    own = (gpu.abi.PgBVHNode * nn)(*[desc.nodes[i] for i in range(nn)])
    interior = [i for i in range(nn) if own[i].nprims == 0]
    bad = gpu.abi.PgSceneDesc.from_buffer_copy(desc)
    bad.nodes = own
    # the second child of the root is made the second child of the root's first child too: shared, and a subtree unreachable
    first = interior[1]
    saved = own[first].offset
    own[first].offset = own[0].offset
    with pytest.raises(gpu.PbrtGpuError, match="not a tree"):
        gpu.GpuScene(bad)
    own[first].offset = first + 1  # second child = first child
    with pytest.raises(gpu.PbrtGpuError, match="not a tree"):
        gpu.GpuScene(bad)
    own[first].offset = nn  # out of range
    with pytest.raises(gpu.PbrtGpuError, match="not a tree"):
        gpu.GpuScene(bad)
    own[first].offset = saved
    gpu.GpuScene(bad).close()
    deep = gpu.abi.PgSceneDesc.from_buffer_copy(desc)
    deep.nodes, deep.n_nodes = _chain_bvh(gpu, desc, 65), 131
    with pytest.raises(gpu.PbrtGpuError, match="65 levels deep"):
        gpu.GpuScene(deep)
    deep.nodes, deep.n_nodes = _chain_bvh(gpu, desc, 64), 129
    gs = gpu.GpuScene(deep)
    o, d = random_rays(scene, 2000, 5)
    tmax = np.full(2000, np.inf, np.float32)
    prim, t, bary = gs.intersect(o, d, tmax)
    oprim, ot, obary, ocn = oracle.intersect(deep, o, d, tmax)
    assert np.array_equal(prim, oprim) and np.array_equal(t, ot) and np.array_equal(bary, obary)
    occ = gs.intersect_p(o, d, tmax)
    oocc, _ = oracle.intersect_p(deep, o, d, tmax)
    assert np.array_equal(occ, oocc) and (prim >= 0).any()
    cn = gs.counters()
    assert cn["closest_node_visits"] == ocn["node_visits"]
    gs.close()


def test_transformed_primitives_inside_object_definitions_are_validated(gpu):
    """ABI 29: PG_PRIM_INSTANCE may stand among an object definition's primitives, ONE level -- what such a primitive wraps holds shapes only (the
    reference has no ObjectInstance inside a definition, api.cpp:1549-1552).  A third level, a definition that contains itself and an instance index
    out of range are refused by pg_scene_create before anything is indexed by them; the scene as the front end flattened it is accepted."""
    scene = gpu.HostScene(os.path.join(GOLD, "nest_motion.pbrt"))
    desc = scene.desc
    n = desc.n_prims_all
    nested = [k for k in range(desc.n_tris, n) if desc.tri_flags[k] & gpu.abi.PG_PRIM_INSTANCE]
    assert len(nested) == 3 and desc.n_prims_all > desc.n_tris  # the tall box, the second sphere, the lone triangle
    gpu.GpuScene(desc).close()
    k = nested[0]
    inner = desc.objects[desc.instances[desc.indices[3 * k]].object]
    flags = (C.c_uint32 * n)(*[desc.tri_flags[i] for i in range(n)])
    idx = (C.c_int32 * (3 * n))(*[desc.indices[i] for i in range(3 * n)])
    # what the nested primitive wraps contains a TransformedPrimitive itself: three levels
    flags[inner.first_prim] = gpu.abi.PG_PRIM_INSTANCE
    idx[3 * inner.first_prim] = desc.indices[3 * nested[1]]
    bad = gpu.abi.PgSceneDesc.from_buffer_copy(desc)
    bad.tri_flags, bad.indices = flags, idx
    with pytest.raises(gpu.PbrtGpuError, match="more than two levels"):
        gpu.GpuScene(bad)
    # an instance index out of range inside a definition
    idx2 = (C.c_int32 * (3 * n))(*[desc.indices[i] for i in range(3 * n)])
    idx2[3 * k] = desc.n_instances
    bad2 = gpu.abi.PgSceneDesc.from_buffer_copy(desc)
    bad2.indices = idx2
    with pytest.raises(gpu.PbrtGpuError, match="its object out of range"):
        gpu.GpuScene(bad2)
    # objects without an objects array
    bad3 = gpu.abi.PgSceneDesc.from_buffer_copy(desc)
    bad3.objects = None
    with pytest.raises(gpu.PbrtGpuError, match="without an objects array"):
        gpu.GpuScene(bad3)


def test_invalid_media_and_sampler_descriptions_fail_loudly(gpu):
    """The v15 / v16 additions to the ABI are validated like the rest: out-of-range medium indices, an unknown integrator or
    sampler, a Sobol' render on a scene created without the tables, inconsistent Sobol' resolutions."""
    scene = gpu.HostScene(os.path.join(GOLD, "vol_smoke.pbrt"))
    desc = scene.desc
    n = max(desc.n_tris, desc.n_prims_all)
    inside = (C.c_int32 * n)(*[desc.tri_medium_inside[i] for i in range(n)])
    inside[0] = desc.n_media
    bad = gpu.abi.PgSceneDesc.from_buffer_copy(desc)
    bad.tri_medium_inside = inside
    with pytest.raises(gpu.PbrtGpuError, match="medium index out of range"):
        gpu.GpuScene(bad)
    gs = gpu.GpuScene(desc)
    rd = scene.render_desc()
    rd.camera_medium = desc.n_media
    with pytest.raises(gpu.PbrtGpuError, match="camera_medium"):
        gs.render(rd)
    rd = scene.render_desc()
    rd.integrator = 2
    with pytest.raises(gpu.PbrtGpuError, match="integrator 2"):
        gs.render(rd)
    rd = scene.render_desc()
    rd.sampler = 1  # vol_smoke uses the Halton sampler: the scene carries no Sobol' matrices
    rd.sobol_resolution, rd.sobol_log2_resolution = 64, 6
    with pytest.raises(gpu.PbrtGpuError, match="without the Sobol"):
        gs.render(rd)
    gs.close()
    sob = gpu.HostScene(os.path.join(GOLD, "sobol_cornell.pbrt"))
    gs = gpu.GpuScene(sob.desc)
    rd = sob.render_desc()
    rd.sobol_resolution = 48
    with pytest.raises(gpu.PbrtGpuError, match="sobol_resolution"):
        gs.render(rd)
    rd = sob.render_desc()
    rd.sampler = 7
    with pytest.raises(gpu.PbrtGpuError, match="sampler 7"):
        gs.render(rd)
    rd = sob.render_desc()
    rd.sampler, rd.strat_samples[0], rd.strat_samples[1] = 3, 3, 2  # stratified: spp must be the product of the strata
    with pytest.raises(gpu.PbrtGpuError, match="stratified sampler 3 x 2"):
        gs.render(rd)
    rd = sob.render_desc()
    rd.sampler, rd.sampler_dims = 5, 2  # maxmindist without its generator matrices
    with pytest.raises(gpu.PbrtGpuError, match="cmaxmin"):
        gs.render(rd)
    gs.close()
    with pytest.raises(gpu.PbrtGpuError, match="null argument"):
        gpu._check(gpu.gpu_lib().pg_hlbvh_build(4, None, 4, None, None, None), "pg_hlbvh_build")


def test_full_size_properties(gpu, oracle, tmp_path):
    """BASELINE.json config 3 geometry at full frame size (1920x1080, ~1M triangles), 1 spp: size-independent
    properties -- every pixel receives exactly its samples, two renders are bit-identical, primary hits of a
    random subset equal the oracle's, and the image agrees with a sharded render."""
    import gen_synthetic
    path = str(tmp_path / "full.pbrt")
    gen_synthetic.write_scene(path, n=708, xres=1920, yres=1080, spp=1)
    scene = gpu.HostScene(path)
    assert scene.desc.n_tris == 999698 + 12
    gs = gpu.GpuScene(scene.desc)
    rd = scene.render_desc()
    film, strays = gs.render(rd)
    cn = gs.counters()
    assert cn["camera_rays"] == 1920 * 1080
    w = film["weight"].reshape(-1, 16, 16)
    nx, ny = 120, 68
    assert w.shape[0] == nx * ny
    assert (w[: nx * (ny - 1)] == 1).all()              # full tiles: one sample per pixel
    assert (w[nx * (ny - 1):, :8] == 1).all() and (w[nx * (ny - 1):, 8:] == 0).all()  # 1080 = 67*16 + 8
    assert np.isfinite(film["rgb"]).all() and (film["rgb"] >= 0).all()
    film2, strays2 = gs.render(rd)
    assert np.array_equal(film, film2) and len(strays) == len(strays2)
    scene.film_clear(); scene.film_merge(rd, film, strays); whole = scene.film_image()
    scene.film_clear()
    for r in range(2):
        rdr = scene.render_desc(r, 2)
        f, s = gs.render(rdr)
        scene.film_merge(rdr, f, s)
    assert np.array_equal(scene.film_image(), whole)
    rng = np.random.default_rng(3)
    n = 50000
    o = np.tile(np.array([0, -2.6, 1.4], np.float32), (n, 1))
    tgt = np.stack([rng.uniform(-1.1, 1.1, n), rng.uniform(-1.1, 1.1, n), rng.uniform(-0.25, 0.3, n)], 1)
    d = (tgt - o).astype(np.float32)
    inf = np.full(n, np.inf, np.float32)
    prim, t, bary = gs.intersect(o, d, inf)
    oprim, ot, obary, _ = oracle.intersect(scene.desc, o, d, inf)
    assert np.array_equal(prim, oprim) and np.array_equal(t, ot) and np.array_equal(bary, obary)
    gs.close()


@pytest.mark.parametrize("name", ["cornell_40x24", "filter_gaussian", "cornell_mirror_glass"])
def test_path_batching_does_not_change_the_film(gpu, name, monkeypatch):
    """pg_render splits a frame into batches of paths when it exceeds the work-buffer budget (by samples for the box filter,
    by tiles for the other filters).  Any split must give the film of the single-batch render, bit for bit."""
    scene = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    gs = gpu.GpuScene(scene.desc)
    rd = scene.render_desc()
    whole, strays = gs.render(rd)
    for budget in ("256", "768", "5000"):
        monkeypatch.setenv("PG_BATCH_PATHS", budget)
        part, strays2 = gs.render(rd)
        assert np.array_equal(part["rgb"], whole["rgb"]) and np.array_equal(part["weight"], whole["weight"]), budget
        assert len(strays2) == len(strays)
    gs.close()


@pytest.mark.parametrize("name", ["cornell_32", "cornell_spot_power", "sphere_light", "instance_boxes", "tex_materials", "vol_smoke", "vol_fog", "sobol_vol_smoke", "filter_gaussian"])
def test_sparse_light_tables_equal_dense(gpu, name, monkeypatch):
    """The "spatial" light distribution filled on first touch (PG_SPARSE_LIGHTS=1 forces the sparse tables that large light
    counts switch on): lanes that meet a voxel without a distribution are re-shaded once it exists.  Film, strays and the
    reference's counters must equal the dense-table render bit for bit, in the first (cold) frame and in the second."""
    scene = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    rd = scene.render_desc()
    dense = gpu.GpuScene(scene.desc)
    film, strays = dense.render(rd)
    cn = dense.counters()
    dense.close()
    monkeypatch.setenv("PG_SPARSE_LIGHTS", "1")
    sparse = gpu.GpuScene(scene.desc)
    for frame in range(2):
        sparse.counters_reset()
        f2, s2 = sparse.render(rd)
        c2 = sparse.counters()
        assert np.array_equal(film["rgb"], f2["rgb"]) and np.array_equal(film["weight"], f2["weight"]), (name, frame)
        assert len(strays) == len(s2)
        for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests", "node_visits"):
            assert cn[k] == c2[k], (name, frame, k)
    sparse.close()


def test_exact_fallback_when_a_ray_outruns_the_cull_margin(gpu, oracle, monkeypatch):
    """k_trace culls far children early with a margin that is provably exact for up to 4096 accepted hits per ray; a ray
    that accepts more raises a guard and the call is repeated without the margin.  PG_TRACE_MAXACC=1 makes nearly every ray
    raise it: film, hits and the reference's counters (incl. node visits) must still equal the oracle's bit for bit."""
    monkeypatch.setenv("PG_TRACE_MAXACC", "1")
    scene = gpu.HostScene(os.path.join(GOLD, "synthetic_n40.pbrt"))
    gs = gpu.GpuScene(scene.desc)
    rd = scene.render_desc()
    film, strays = gs.render(rd)
    cn = gs.counters()
    ofilm, ostrays, ocn = oracle.render(scene.desc, rd)
    assert np.array_equal(film["rgb"], ofilm["rgb"]) and np.array_equal(film["weight"], ofilm["weight"])
    for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests", "node_visits"):
        assert cn[k] == ocn[k], k
    rng = np.random.default_rng(5)
    nodes = scene.nodes()
    lo, hi = nodes["bmin"][0], nodes["bmax"][0]
    o = (lo + (hi - lo) * rng.random((4096, 3))).astype(np.float32)
    d = rng.normal(size=(4096, 3)).astype(np.float32)
    inf = np.full(4096, np.inf, np.float32)
    gs.counters_reset()
    prim, t, bary = gs.intersect(o, d, inf)
    op, ot, ob, oc = oracle.intersect(scene.desc, o, d, inf)
    assert np.array_equal(prim, op) and np.array_equal(t, ot) and np.array_equal(bary, ob)
    assert gs.counters()["closest_node_visits"] == oc["node_visits"]
    gs.close()


@pytest.mark.parametrize("name", ["cornell_32", "vol_fog"])
def test_integrator_statistics_run_on_across_frames(gpu, name):
    """PgCounters' integrator statistics (ABI 28) are folded from the device's shards after every frame: a second frame doubles the sums, leaves the
    shortest / longest path where they were, pg_counters_reset zeroes all of them (the reference's statistics likewise run on until they are printed)."""
    scene = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    gs = gpu.GpuScene(scene.desc)
    rd = scene.render_desc()
    gs.render(rd)
    one = gs.counters()
    stats = json.load(open(os.path.join(GOLD, name + ".json")))
    check_integrator_stats(one, stats)
    gs.render(rd)
    two = gs.counters()
    for k in ("paths_total", "paths_zero_radiance", "path_length_sum", "path_length_count", "volume_interactions", "surface_interactions"):
        assert two[k] == 2 * one[k], k
    assert (two["path_length_min"], two["path_length_max"]) == (one["path_length_min"], one["path_length_max"])
    gs.counters_reset()
    assert all(gs.counters()[k] == 0 for k in INTEGRATOR_STATS)
    gs.render(rd)
    check_integrator_stats(gs.counters(), stats)
    gs.close()
