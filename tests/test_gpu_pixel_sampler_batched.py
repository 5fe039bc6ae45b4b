"""The PixelSamplers (stratified, 02sequence, maxmindist) touch their tile's RNG stream after StartPixel only for draws beyond their
"dimensions" (core/sampler.cpp:108-134).  When the sampled dimensions cover every draw a path of the PathIntegrator can make
(1 + 2 maxdepth one-dimensional, 2 + 3 maxdepth two-dimensional), the device generates every pixel's arrays ahead -- one lane per tile
walking its pixels in the reference's order -- and traces all pixels as one wavefront instead of one path per tile (DESIGN.md
section 4, "Samplers").  Both forms must give the same film, stray samples and ray counts; the goldens of these scenes are compared
with the reference's images in tests/test_gpu_parity.py like all others."""
import os

import numpy as np
import pytest

from conftest import GOLD

pytestmark = pytest.mark.gpu
NAMES = ["sampler_stratified_dims", "sampler_stratified_dims_tex", "filter_02sequence_dims", "sampler_maxmindist_dims"]
COUNTERS = ("camera_rays", "closest_rays", "shadow_rays", "mis_rays", "shade_items", "closest_node_visits", "closest_tri_tests", "shadow_node_visits",
            "shadow_tri_tests", "light_tri_tests")


def render(gpu, scene):
    gs = gpu.GpuScene(scene.desc)
    film, strays = gs.render(scene.render_desc())
    cn = gs.counters()
    gs.close()
    return film, strays, cn


@pytest.mark.parametrize("name", NAMES)
def test_batched_and_serial_forms_agree(gpu, monkeypatch, name):
    scene = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    monkeypatch.setenv("PG_TS_BATCHED", "0")
    fa, sa, ca = render(gpu, scene)
    monkeypatch.delenv("PG_TS_BATCHED")
    fb, sb, cb = render(gpu, scene)
    assert np.array_equal(fa["rgb"], fb["rgb"]) and np.array_equal(fa["weight"], fb["weight"])
    key = lambda s: np.lexsort((s["src_px"], s["src_py"], s["px"], s["py"]))
    sa, sb = sa[key(sa)], sb[key(sb)]
    assert len(sa) == len(sb) and all(np.array_equal(sa[f], sb[f]) for f in ("px", "py", "src_px", "src_py", "weight", "rgb"))
    for k in COUNTERS:
        assert ca[k] == cb[k], (k, ca[k], cb[k])
    # one wavefront of all pixels: a handful of launches, against one path per tile and sample (thousands)
    assert cb["closest_launches"] <= 16 < ca["closest_launches"]


def test_one_dimension_short_takes_the_serial_form(gpu):
    """maxdepth 3 needs 11 two-dimensional samples; with 10 the last vertex's next direction comes from the stream."""
    text = open(os.path.join(GOLD, "sampler_stratified_dims.pbrt")).read()
    assert '"integer dimensions" [ 14 ]' in text
    old = os.getcwd()
    os.chdir(GOLD)
    try:
        launches = {}
        for dims in (10, 11):
            scene = gpu.HostScene(text=text.replace('"integer dimensions" [ 14 ]', '"integer dimensions" [ %d ]' % dims))
            launches[dims] = render(gpu, scene)[2]["closest_launches"]
    finally:
        os.chdir(old)
    assert launches[11] <= 16 < launches[10]


@pytest.mark.parametrize("skew", [1, -3])
def test_wrong_first_guess_of_the_stream_positions_is_corrected(gpu, monkeypatch, skew):
    """k_ts_start_tile starts pixel p of a tile at (pixels before it) x (numbers a StartPixel nominally takes) and re-runs the pixels
    whose true position -- the prefix sum of what the pixels before them really took -- differs (a repeated rejection loop of
    RNG::UniformUInt32(b) shifts everything behind it).  A deliberately wrong nominal count makes every pixel but the first re-run."""
    scene = gpu.HostScene(os.path.join(GOLD, "filter_02sequence_dims.pbrt"))
    fa, sa, ca = render(gpu, scene)
    monkeypatch.setenv("PG_TS_GUESS_SKEW", str(skew))
    fb, sb, cb = render(gpu, scene)
    assert np.array_equal(fa["rgb"], fb["rgb"]) and np.array_equal(fa["weight"], fb["weight"]) and ca["closest_rays"] == cb["closest_rays"]


@pytest.mark.parametrize("name", ["sampler_stratified_dims_tex", "filter_02sequence_dims"])
def test_batched_form_under_tile_sharding(gpu, monkeypatch, name):
    """The multi-GPU decomposition (tiles t = r mod N per rank) with a batched PixelSampler: a tile's sample arrays depend on the tile's
    own stream only.  Box filter: three shards merged give the unsharded image bit for bit.  Wide filter (overlapping tile blocks are
    summed in merge order): the three shards of the batched form equal the three shards of the serial form, merged the same way."""
    scene = gpu.HostScene(os.path.join(GOLD, name + ".pbrt"))
    gs = gpu.GpuScene(scene.desc)

    def merged_shards():
        scene.film_clear()
        for r in range(3):
            rdr = scene.render_desc(r, 3)
            f, s = gs.render(rdr)
            scene.film_merge(rdr, f, s)
        return scene.film_image().copy()

    batched = merged_shards()
    assert gs.counters()["closest_launches"] <= 3 * 16
    if name.startswith("filter_"):
        monkeypatch.setenv("PG_TS_BATCHED", "0")
        assert np.array_equal(merged_shards(), batched)
    else:
        rd = scene.render_desc()
        film, strays = gs.render(rd)
        scene.film_clear(); scene.film_merge(rd, film, strays)
        assert np.array_equal(scene.film_image(), batched)
    gs.close()
