"""Materials with textured parameters are evaluated AHEAD of the shading launch (k_material writes Material::ComputeScatteringFunctions'
outputs -- BxDF list, BSDF::eta, the bump-mapped shading frame -- per main-queue entry; k_shade<3, .> shades from them; DESIGN.md section 4,
"k_material").  PG_MAT_PRE=0 evaluates them inside the shading kernel (k_shade<2, .>), which is also what a device without room for the
lists does.  Both must render the same film, strays and counters bit for bit: every golden scene with textures, random scenes of the
extended generators, and a scene whose lists need the full BSDF::MaxBxDFs."""
import os

import numpy as np
import pytest

from conftest import GOLD, golden_names
from test_gpu_fuzz import random_scene_ext, random_scene_vol
from test_gpu_shade_order import COUNTERS, textured

pytestmark = pytest.mark.gpu

NAMES = [n for n in golden_names() if textured(n)]


def render_both(gpu, monkeypatch, scene):
    out = []
    for pre in ("0", "1"):
        monkeypatch.setenv("PG_MAT_PRE", pre)
        gs = gpu.GpuScene(scene.desc)  # read when the scene is created
        film, strays = gs.render(scene.render_desc())
        out.append((film, strays, gs.counters()))
        gs.close()
    (fa, sa, ca), (fb, sb, cb) = out
    assert np.array_equal(fa["rgb"], fb["rgb"]) and np.array_equal(fa["weight"], fb["weight"])
    key = lambda s: np.lexsort((s["src_px"], s["src_py"], s["px"], s["py"]))
    sa, sb = sa[key(sa)], sb[key(sb)]
    assert len(sa) == len(sb) and all(np.array_equal(sa[f], sb[f]) for f in ("px", "py", "src_px", "src_py", "weight", "rgb"))
    for k in COUNTERS:
        assert ca[k] == cb[k], (k, ca[k], cb[k])


@pytest.mark.parametrize("name", NAMES)
def test_golden_scene_same_film_with_and_without_the_material_pass(gpu, monkeypatch, name):
    render_both(gpu, monkeypatch, gpu.HostScene(os.path.join(GOLD, name + ".pbrt")))


@pytest.mark.parametrize("seed", range(16))
def test_random_scene_same_film_with_and_without_the_material_pass(gpu, monkeypatch, seed):
    text = (random_scene_ext, random_scene_vol)[seed % 2](seed // 2)
    render_both(gpu, monkeypatch, gpu.HostScene(text=text))


def test_longest_lists(gpu, monkeypatch):
    """A mix of two uber materials builds up to 8 BxDFs (BSDF::MaxBxDFs) with two ScaledBxDF levels: the per-hit room k_material is given
    (the scene's longest list) is exactly what ComputeScatteringFunctions adds."""
    text = open(os.path.join(GOLD, "tex_materials.pbrt")).read()
    extra = ('MakeNamedMaterial "ua" "string type" "uber" "texture Kd" "chk" "rgb Ks" [ 0.2 0.2 0.2 ] "rgb Kr" [ 0.1 0.1 0.1 ] "rgb Kt" [ 0.1 0.1 0.1 ] '
             '"rgb opacity" [ 0.8 0.8 0.8 ] "float roughness" [ 0.1 ]\n'
             'MakeNamedMaterial "ub" "string type" "uber" "texture Kd" "uvt" "rgb Ks" [ 0.3 0.3 0.3 ] "rgb Kr" [ 0.2 0.1 0.1 ] "rgb Kt" [ 0.1 0.2 0.1 ] '
             '"rgb opacity" [ 0.7 0.7 0.7 ] "float roughness" [ 0.2 ]\n'
             'AttributeBegin\n  Material "mix" "string namedmaterial1" "ua" "string namedmaterial2" "ub" "texture amount" "chk"\n'
             '  Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ 100 100 150  450 100 150  450 450 160  100 450 160 ] "float uv" [ 0 0 1 0 1 1 0 1 ]\nAttributeEnd\n')
    old = os.getcwd()
    os.chdir(GOLD)  # (the scene's image textures)
    try:
        scene = gpu.HostScene(text=text.replace("WorldEnd", extra + "WorldEnd"))
    finally:
        os.chdir(old)
    render_both(gpu, monkeypatch, scene)
