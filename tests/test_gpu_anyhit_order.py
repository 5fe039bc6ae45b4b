"""Shadow rays (BVHAccel::IntersectP, accelerators/bvh.cpp:702-738) answer "is anything hit", and ray.tMax does not change while they are
traced: which nodes pass their box test -- and so which primitives can be met at all -- does not depend on the order of the visits.  The
product therefore visits the child the ray enters first (k_trace<2, .>), and only the reference-statistics tests ask for the reference's
order (PG_ANYHIT_ORDER=reference, tests/conftest.py).  Here: every golden scene and 40 random scenes rendered in BOTH orders -- films,
stray samples and ray counts must be bit-identical; the free order must not test more triangles in total."""
import os

import numpy as np
import pytest

from conftest import GOLD, golden_names
from test_gpu_fuzz import random_scene, random_scene_ext, random_scene_vol

pytestmark = pytest.mark.gpu
FAST = [n for n in golden_names() if not n.startswith(("sampler_", "filter_02sequence"))]  # (tile-serial samplers: minutes each; one below)


def render_both(gpu, monkeypatch, scene):
    out = {}
    for order in ("reference", "free"):
        monkeypatch.setenv("PG_ANYHIT_ORDER", order)
        gs = gpu.GpuScene(scene.desc)  # the order is read when the scene is created
        film, strays = gs.render(scene.render_desc())
        out[order] = (film, strays, gs.counters())
        gs.close()
    (fa, sa, ca), (fb, sb, cb) = out["reference"], out["free"]
    assert np.array_equal(fa["rgb"], fb["rgb"]) and np.array_equal(fa["weight"], fb["weight"])
    key = lambda s: np.lexsort((s["src_px"], s["src_py"], s["px"], s["py"]))  # (stray samples are appended in whatever order the blocks finish)
    sa, sb = sa[key(sa)], sb[key(sb)]
    assert len(sa) == len(sb) and all(np.array_equal(sa[f], sb[f]) for f in ("px", "py", "src_px", "src_py", "weight", "rgb"))
    for k in ("camera_rays", "closest_rays", "shadow_rays", "mis_rays", "shade_items", "closest_node_visits", "closest_tri_tests", "light_tri_tests"):
        assert ca[k] == cb[k], (k, ca[k], cb[k])
    return ca, cb


@pytest.mark.parametrize("name", FAST + ["sampler_stratified"])
def test_golden_scene_same_film_in_both_orders(gpu, monkeypatch, name):
    ca, cb = render_both(gpu, monkeypatch, gpu.HostScene(os.path.join(GOLD, name + ".pbrt")))
    assert cb["shadow_tri_tests"] <= ca["shadow_tri_tests"] * 1.5 + 64  # (a different order may test a few more before the first hit)


@pytest.mark.parametrize("seed", range(40))
def test_random_scene_same_film_in_both_orders(gpu, monkeypatch, seed):
    text = (random_scene, random_scene_ext, random_scene_vol)[seed % 3](seed // 3)
    render_both(gpu, monkeypatch, gpu.HostScene(text=text))
