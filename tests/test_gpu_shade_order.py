"""Scenes with textured / BxDF-list materials are shaded grouped by material class: k_shade_order sorts the entry indices of every
window of the main queue by the class of the entry's hit, and the shading threads take the entries in that order (DESIGN.md section 4,
"Shading order").  Which thread shades a vertex must not matter: every golden scene that has such materials, and random scenes of the
extended generator, rendered with PG_SHADE_ORDER=0 (queue order) and with the product's default -- films, stray samples and
every counter must be bit-identical."""
import os

import numpy as np
import pytest

from conftest import GOLD, golden_names
from test_gpu_fuzz import random_scene_ext, random_scene_vol

pytestmark = pytest.mark.gpu


def textured(name):
    text = open(os.path.join(GOLD, name + ".pbrt")).read()
    return "Texture " in text and not name.startswith(("sampler_", "filter_02sequence"))  # (tile-serial samplers: minutes each)


NAMES = [n for n in golden_names() if textured(n)]
COUNTERS = ("camera_rays", "closest_rays", "shadow_rays", "mis_rays", "shade_items", "closest_node_visits", "closest_tri_tests", "shadow_node_visits",
            "shadow_tri_tests", "light_tri_tests")


def render_both(gpu, monkeypatch, scene):
    out = []
    for order in ("0", "1"):
        monkeypatch.setenv("PG_SHADE_ORDER", order)
        gs = gpu.GpuScene(scene.desc)  # read when the scene is created
        film, strays = gs.render(scene.render_desc())
        out.append((film, strays, gs.counters()))
        gs.close()
    (fa, sa, ca), (fb, sb, cb) = out
    assert np.array_equal(fa["rgb"], fb["rgb"]) and np.array_equal(fa["weight"], fb["weight"])
    key = lambda s: np.lexsort((s["src_px"], s["src_py"], s["px"], s["py"]))  # (stray samples are appended in whatever order the blocks finish)
    sa, sb = sa[key(sa)], sb[key(sb)]
    assert len(sa) == len(sb) and all(np.array_equal(sa[f], sb[f]) for f in ("px", "py", "src_px", "src_py", "weight", "rgb"))
    for k in COUNTERS:
        assert ca[k] == cb[k], (k, ca[k], cb[k])


def test_there_are_textured_goldens():
    assert len(NAMES) >= 15 and "divergent_small" in NAMES and "divergent_small_vol" in NAMES


@pytest.mark.parametrize("name", NAMES)
def test_golden_scene_same_film_in_both_orders(gpu, monkeypatch, name):
    render_both(gpu, monkeypatch, gpu.HostScene(os.path.join(GOLD, name + ".pbrt")))


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_same_film_in_both_orders(gpu, monkeypatch, seed):
    text = (random_scene_ext, random_scene_vol)[seed % 2](seed // 2)
    render_both(gpu, monkeypatch, gpu.HostScene(text=text))


def test_more_materials_than_classes(gpu, monkeypatch):
    """Beyond 13 materials the classes are shared by materials that run the same code (type, kind, bump): 24 small quads, each with a material
    of its own (matte / plastic / metal / uber / substrate, textured and plain), in front of the textured Cornell box -- both orders, same film."""
    text = open(os.path.join(GOLD, "tex_materials.pbrt")).read()
    kinds = ['Material "matte" "texture Kd" "chk" "float sigma" [ %g ]', 'Material "plastic" "texture Kd" "uvt" "float roughness" [ %g ]',
             'Material "metal" "texture k" "bil" "float roughness" [ %g ]', 'Material "uber" "texture Kd" "mixed" "float roughness" [ %g ]',
             'Material "substrate" "texture Kd" "uvt" "float uroughness" [ %g ]', 'Material "matte" "rgb Kd" [ 0.5 %g 0.3 ]']
    quads = []
    for k in range(24):
        x, y = 60 + 85 * (k % 6), 60 + 110 * (k // 6)
        quads.append('AttributeBegin\n  %s\n  Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ %d %d 120  %d %d 120  %d %d 125  %d %d 125 ] '
                     '"float uv" [ 0 0 1 0 1 1 0 1 ]\nAttributeEnd' % (kinds[k % len(kinds)] % (0.05 + 0.03 * k), x, y, x + 60, y, x + 60, y + 80, x, y + 80))
    old = os.getcwd()
    os.chdir(GOLD)  # (the scene's image textures)
    try:
        scene = gpu.HostScene(text=text.replace("WorldEnd", "\n".join(quads) + "\nWorldEnd"))
    finally:
        os.chdir(old)
    assert scene.desc.n_materials > 13
    render_both(gpu, monkeypatch, scene)
