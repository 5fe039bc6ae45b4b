"""GPU checks at BASELINE.json's sizes.  The oracle cannot render these in seconds, so the full-size runs are checked through
size-independent properties of the path (determinism, tile-shard invariance, exact linearity in the emitted radiance, film
weights, closest-hit / any-hit agreement, the reference's ray accounting), and the full-size GEOMETRY is checked against
the unmodified reference at a small resolution (tests/golden_large, rendered by oracle/make_golden.py) and against the CPU
oracle on a ray subset (bit-exact)."""
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT, check_integrator_stats

pytestmark = pytest.mark.gpu
LARGE = os.path.join(ROOT, "tests", "golden_large")


def rel_err(img, ref):
    return np.abs(img - ref) / np.maximum(1.0, np.abs(ref))


@pytest.fixture(scope="module")
def synthetic_dir(tmp_path_factory):
    import gen_synthetic
    d = tmp_path_factory.mktemp("synthetic_1m")
    gen_synthetic.write_scene(str(d / "small.pbrt"), n=708, xres=96, yres=54, spp=4, filename="synthetic_1m.pfm")
    return d


def scene_text(synthetic_dir, xres, yres, spp, light_scale=None):
    txt = open(synthetic_dir / "small.pbrt").read()
    txt = re.sub(r'"integer xresolution" \[ \d+ \]', f'"integer xresolution" [ {xres} ]', txt)
    txt = re.sub(r'"integer yresolution" \[ \d+ \]', f'"integer yresolution" [ {yres} ]', txt)
    txt = re.sub(r'"integer pixelsamples" \[ \d+ \]', f'"integer pixelsamples" [ {spp} ]', txt)
    txt = txt.replace('Include "small_mesh.pbrt"', f'Include "{synthetic_dir / "small_mesh.pbrt"}"')
    if light_scale is not None:
        assert '"rgb L" [ 17 12 4 ]' in txt
        txt = txt.replace('"rgb L" [ 17 12 4 ]', f'"rgb L" [ 17 12 4 ] "rgb scale" [ {light_scale} {light_scale} {light_scale} ]')
    return txt


def test_million_triangle_geometry_matches_reference(gpu, synthetic_dir):
    """999 710 triangles, SAH BVH of 1.48 M nodes: the image of the unmodified reference and its ray counters."""
    scene = gpu.HostScene(str(synthetic_dir / "small.pbrt"))
    assert scene.desc.n_tris == 999710
    img, cn = gpu.render_scene(scene)
    ref = gpu.read_pfm(os.path.join(LARGE, "synthetic_1m.pfm"))
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{(img != ref).any(axis=2).sum()} pixels differ, max rel err {rel_err(img, ref).max():.3e}"
    stats = json.load(open(os.path.join(LARGE, "synthetic_1m.json")))
    for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests"):
        assert cn[k] == stats[k], (k, cn[k], stats[k])
    check_integrator_stats(cn, stats)


def test_million_triangle_rays_bit_exact_vs_oracle(gpu, oracle, synthetic_dir):
    scene = gpu.HostScene(str(synthetic_dir / "small.pbrt"))
    gs = gpu.GpuScene(scene.desc)
    rng = np.random.default_rng(11)
    n = 1 << 20
    nodes = scene.nodes()
    lo, hi = nodes["bmin"][0], nodes["bmax"][0]
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[: n // 16, 2] *= np.float32(0.02)  # grazing rays along the heightfield: long traversals, deep stacks
    inf = np.full(n, np.inf, np.float32)
    prim, t, bary = gs.intersect(o, d, inf)
    occ = gs.intersect_p(o, d, inf)
    assert ((prim >= 0) == (occ == 1)).all()  # BVHAccel::Intersect and IntersectP agree on hit / miss for every ray
    assert (t[prim >= 0] > 0).all() and np.isinf(t[prim < 0]).all()
    sub = slice(0, 40000)  # the CPU oracle on a subset (first 1/16 grazing), bit-exact incl. counters
    gs.counters_reset()
    p2, t2, b2 = gs.intersect(o[sub], d[sub], inf[sub])
    op, ot, ob, ocn = oracle.intersect(scene.desc, o[sub], d[sub], inf[sub])
    assert np.array_equal(p2, prim[sub]) and np.array_equal(t2, t[sub])  # same ray, same answer regardless of batch
    assert np.array_equal(p2, op) and np.array_equal(t2, ot) and np.array_equal(b2, ob)
    cn = gs.counters()
    assert cn["closest_node_visits"] == ocn["node_visits"] and cn["closest_tri_tests"] == ocn["tri_tests"]
    gs.close()


def test_config3_full_size_properties(gpu, synthetic_dir):
    """BASELINE.json config 3 itself: 1920x1080 @ 64 spp on the 999 710-triangle scene."""
    xres, yres, spp = 1920, 1080, 64
    scene = gpu.HostScene(text=scene_text(synthetic_dir, xres, yres, spp))
    gs = gpu.GpuScene(scene.desc)
    rd = scene.render_desc()
    film, strays = gs.render(rd)
    cn = gs.counters()
    # the reference's accounting: one camera ray per pixel sample, every pixel weighted spp times by the box filter
    assert cn["camera_rays"] == xres * yres * spp
    ntx, nty = (xres + 15) // 16, (yres + 15) // 16
    w = film["weight"].reshape(nty, ntx, 16, 16).transpose(0, 2, 1, 3).reshape(nty * 16, ntx * 16)
    assert (w[:yres, :xres] == spp).all() and (w[yres:] == 0).all()
    assert np.isfinite(film["rgb"]).all() and (film["rgb"] >= 0).all()
    assert 5.0 < (cn["closest_rays"] + cn["shadow_rays"]) / cn["camera_rays"] < 8.0  # closed box, maxdepth 5
    scene.film_clear(); scene.film_merge(rd, film, strays)
    whole = scene.film_image()
    # determinism: a second render is bit-identical
    film2, strays2 = gs.render(rd)
    assert np.array_equal(film["rgb"], film2["rgb"]) and len(strays) == len(strays2)
    # tile sharding (the multi-GPU decomposition) does not change a single bit
    scene.film_clear()
    for r in range(3):
        rdr = scene.render_desc(r, 3)
        f, s = gs.render(rdr)
        scene.film_merge(rdr, f, s)
    assert np.array_equal(scene.film_image(), whole)
    gs.close()
    # exact linearity in emitted radiance: scaling every light by 2 (a power of two) scales every pixel by exactly 2
    scene2 = gpu.HostScene(text=scene_text(synthetic_dir, xres, yres, spp, light_scale=2))
    img2, _ = gpu.render_scene(scene2)
    assert np.array_equal(img2, whole * np.float32(2))


def test_config2_cornell_quarter_size_vs_reference(gpu):
    """BASELINE.json config 2 (Cornell box) at 128x128 @ 64 spp against the unmodified reference's image."""
    scene = gpu.HostScene(os.path.join(LARGE, "cornell_128.pbrt"))
    img, cn = gpu.render_scene(scene)
    ref = gpu.read_pfm(os.path.join(LARGE, "cornell_128.pfm"))
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{(img != ref).any(axis=2).sum()} pixels differ, max rel err {rel_err(img, ref).max():.3e}"
    stats = json.load(open(os.path.join(LARGE, "cornell_128.json")))
    for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests"):
        assert cn[k] == stats[k], (k, cn[k], stats[k])
    check_integrator_stats(cn, stats)


CONFIG0 = os.path.join(LARGE, "config0", "config0.pbrt")


@pytest.mark.parametrize("order", ["reference", "free"])
def test_config0_killeroo_simple_vs_reference(gpu, oracle, order, monkeypatch):
    """BASELINE.json config 0: scenes/killeroo-simple.pbrt as the reference ships it (66 532 Loop-subdivided triangles, plastic,
    uv'd planes, a SPHERE area light: every ray runs the quadric instantiation of k_trace) at 400x400 @ 8 spp, against the image
    and the statistics of the unmodified reference binary (tests/golden_large/config0/, oracle/make_golden.py) and, bit for bit,
    against the CPU restatement.  Twice: shadow rays in the reference's visiting order (its triangle-test statistic reproduced
    exactly) and in the product's default free order (same film, same ray counts)."""
    monkeypatch.setenv("PG_ANYHIT_ORDER", order)
    scene = gpu.HostScene(CONFIG0)
    assert scene.desc.n_spheres == 1 and scene.desc.n_tris == 66532 + 1  # (primitives: the sphere counts)
    gs = gpu.GpuScene(scene.desc)
    rd = scene.render_desc()
    film, strays = gs.render(rd)
    cn = gs.counters()
    scene.film_clear(); scene.film_merge(rd, film, strays)
    img = scene.film_image()
    gs.close()
    ref = gpu.read_pfm(os.path.join(LARGE, "config0", "config0.pfm"))
    assert img.shape == ref.shape == (400, 400, 3)
    assert np.array_equal(img, ref), f"{(img != ref).any(axis=2).sum()} pixels differ from the reference binary's image, max rel err {rel_err(img, ref).max():.3e}"
    stats = json.load(open(os.path.join(LARGE, "config0", "config0.json")))
    for k in ("camera_rays", "closest_rays", "shadow_rays") + (("tri_tests",) if order == "reference" else ()):
        assert cn[k] == stats[k], (k, cn[k], stats[k])
    check_integrator_stats(cn, stats)
    ofilm, ostrays, ocn = oracle.render(scene.desc, rd)
    assert np.array_equal(film["rgb"], ofilm["rgb"]) and np.array_equal(film["weight"], ofilm["weight"]) and len(strays) == len(ostrays)
    if order == "reference":
        assert cn["node_visits"] == ocn["node_visits"]


def test_config2_cornell_full_size_properties(gpu):
    """Cornell box at its full 512x512 @ 256 spp: accounting, determinism and shard invariance."""
    scene = gpu.HostScene(os.path.join(ROOT, "scenes", "cornell.pbrt"))
    gs = gpu.GpuScene(scene.desc)
    rd = scene.render_desc()
    film, strays = gs.render(rd)
    cn = gs.counters()
    assert cn["camera_rays"] == 512 * 512 * 256
    assert (film["weight"] == 256).all()
    scene.film_clear(); scene.film_merge(rd, film, strays)
    whole = scene.film_image()
    assert np.isfinite(whole).all() and whole.min() >= 0 and 0.1 < whole.mean() < 2.0
    scene.film_clear()
    for r in range(8):
        rdr = scene.render_desc(r, 8)
        f, s = gs.render(rdr)
        scene.film_merge(rdr, f, s)
    assert np.array_equal(scene.film_image(), whole)
    gs.close()


def test_native_sharded_render_equals_single_device(gpu, tmp_path, monkeypatch):
    """pg_render_sharded -- the in-process multi-GPU path of `pbrt_amd --gpus N`: one host thread per device, peer-to-peer
    gather of the film shards on the first device.  On a single-GPU box the device id repeats (three shards on GPU 0): the
    merged film must equal the single-device render bit for bit, and every shard must equal pg_render of that shard."""
    scene = gpu.HostScene(os.path.join(LARGE, "cornell_128.pbrt"))
    whole, _ = gpu.render_scene(scene)
    rd = scene.render_desc()
    scenes = [gpu.GpuScene(scene.desc, device=0) for _ in range(3)]
    shards = gpu.render_sharded(scenes, rd)
    scene.film_clear()
    for srd, film, strays in shards:
        alone_film, alone_strays = scenes[0].render(srd)
        assert np.array_equal(film["rgb"], alone_film["rgb"]) and np.array_equal(film["weight"], alone_film["weight"]) and len(strays) == len(alone_strays)
        scene.film_merge(srd, film, strays)
    assert np.array_equal(scene.film_image(), whole)
    assert gpu.shard_transport().startswith("peer (a device appears twice") or os.environ.get("PBRT_EMULATED_DEVICE") == "1", gpu.shard_transport()
    # the RCCL transport itself on this box's one GPU: a one-rank communicator (ncclCommInitAll + ncclGather through the lazily opened
    # librccl); PG_SHARD_GATHER=rccl turns "RCCL could not be used" into an error instead of the peer-copy fallback
    if os.environ.get("PBRT_EMULATED_DEVICE") != "1":
        monkeypatch.setenv("PG_SHARD_GATHER", "rccl")
        (srd, film, strays), = gpu.render_sharded(scenes[:1], rd)
        assert gpu.shard_transport() == "rccl"
        scene.film_clear(); scene.film_merge(srd, film, strays)
        assert np.array_equal(scene.film_image(), whole)
        monkeypatch.delenv("PG_SHARD_GATHER")
    for s in scenes:
        s.close()
    # the same through the CLI: pbrt_amd --gpu-ids 0,0 writes the image of pbrt_amd --gpu 0
    import subprocess
    exe = os.path.join(ROOT, "pbrt-v3_amd", "pbrt_amd")
    for flags, out in ((["--gpu", "0"], "one.pfm"), (["--gpu-ids", "0,0"], "two.pfm")):
        p = subprocess.run([exe, "--quiet", *flags, "--outfile", str(tmp_path / out), os.path.join(LARGE, "cornell_128.pbrt")], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
    assert np.array_equal(gpu.read_pfm(str(tmp_path / "one.pfm")), gpu.read_pfm(str(tmp_path / "two.pfm")))
    assert np.array_equal(gpu.read_pfm(str(tmp_path / "one.pfm")), whole)


def test_sharded_render_with_more_devices_than_tiles(gpu, tmp_path):
    """More shards than film tiles (a 24x16 image is two 16x16 tiles; four shards): the ranks that own no tile take part with an
    empty film, the merged frame equals the single-device render bit for bit -- from Python (empty arrays) and through the CLI,
    whose empty std::vector hands pg_render_sharded a null film pointer for those ranks."""
    import subprocess
    text = (open(os.path.join(ROOT, "tests", "golden", "cornell_40x24.pbrt")).read()
            .replace('"integer xresolution" [ 40 ]', '"integer xresolution" [ 24 ]').replace('"integer yresolution" [ 24 ]', '"integer yresolution" [ 16 ]'))
    assert '[ 24 ]' in text and '[ 16 ]' in text
    path = str(tmp_path / "tiny.pbrt")
    open(path, "w").write(text)
    scene = gpu.HostScene(path)
    whole, _ = gpu.render_scene(scene)
    rd = scene.render_desc()
    scenes = [gpu.GpuScene(scene.desc, device=0) for _ in range(4)]
    shards = gpu.render_sharded(scenes, rd)
    assert [len(f) for _, f, _ in shards] == [256, 256, 0, 0]
    scene.film_clear()
    for srd, film, strays in shards:
        scene.film_merge(srd, film, strays)
    assert np.array_equal(scene.film_image(), whole)
    for s in scenes:
        s.close()
    exe = os.path.join(ROOT, "pbrt-v3_amd", "pbrt_amd")
    p = subprocess.run([exe, "--quiet", "--gpu-ids", "0,0,0,0", "--outfile", str(tmp_path / "four.pfm"), path], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert np.array_equal(gpu.read_pfm(str(tmp_path / "four.pfm")), whole)
