"""Pins the CPU restatement (oracle/pbrt_oracle.c) and the host front end against output of the
UNMODIFIED reference binary: tests/golden/*.pfm + *.json were rendered by oracle/_ref/pbrt_oracle
(oracle/make_golden.py).  Bit-exact images and identical ray statistics are required."""
import json
import os

import numpy as np
import pytest

from conftest import GOLD, check_integrator_stats, golden_names


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_image_and_stats(pkg, oracle, name):
    scene = pkg.HostScene(os.path.join(GOLD, name + ".pbrt"))
    img, cn = oracle.render_image(scene)
    ref = pkg.read_pfm(os.path.join(GOLD, name + ".pfm"))
    assert img.shape == ref.shape
    assert np.array_equal(img, ref), f"max |diff| {np.abs(img - ref).max()}"
    stats = json.load(open(os.path.join(GOLD, name + ".json")))
    for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests"):  # the reference's own STAT_COUNTERs (scene.cpp:40-42, integrator.cpp:48, triangle.cpp:45)
        assert cn[k] == stats[k], k
    check_integrator_stats(cn, stats)  # path.cpp:45-46, volpath.cpp:45-47


def test_reference_binary_live_when_present(pkg, oracle, tmp_path):
    """Where oracle/_ref exists (build container and GPU box), re-render one scene with it now."""
    if not os.path.exists(oracle.REF_BINARY):
        pytest.skip("oracle/_ref/pbrt_oracle not built here")
    name = "cornell_40x24"
    out = str(tmp_path / "live.pfm")
    oracle.run_reference(os.path.join(GOLD, name + ".pbrt"), out, nthreads=2)
    assert np.array_equal(pkg.read_pfm(out), pkg.read_pfm(os.path.join(GOLD, name + ".pfm")))


def test_config0_killeroo_simple_unmodified(pkg, oracle, tmp_path):
    """BASELINE.json config 0, scenes/killeroo-simple.pbrt exactly as the reference ships it (sphere area light, two
    Loop-subdivided killeroos, plastic, uv'd planes) at a reduced resolution: the unmodified reference and this
    repository's front end + oracle produce bit-identical images.  Reads /root/reference (build container only)."""
    src = "/root/reference/scenes"
    if not os.path.exists(os.path.join(src, "killeroo-simple.pbrt")) or not os.path.exists(oracle.REF_BINARY):
        pytest.skip("reference scenes / binary not available here")
    os.makedirs(tmp_path / "geometry")
    os.symlink(os.path.join(src, "geometry", "killeroo.pbrt"), tmp_path / "geometry" / "killeroo.pbrt")
    s = open(os.path.join(src, "killeroo-simple.pbrt")).read()
    s = s.replace('"integer xresolution" [700] "integer yresolution" [700]', '"integer xresolution" [100] "integer yresolution" [100]').replace("killeroo-simple.exr", "kroo.pfm")
    scene_file = str(tmp_path / "kroo.pbrt")
    open(scene_file, "w").write(s)
    out = str(tmp_path / "ref.pfm")
    oracle.run_reference(scene_file, out, nthreads=4)
    scene = pkg.HostScene(scene_file)
    assert scene.desc.n_tris == 2 * 33264 + 4 + 1 and scene.desc.n_spheres == 1
    img, _ = oracle.render_image(scene)
    assert np.array_equal(img, pkg.read_pfm(out))


def test_killeroo_geometry_live_when_reference_present(pkg, oracle, tmp_path):
    """BASELINE.json config 0's scene (scenes/killeroo-simple.pbrt: two Loop-subdivided killeroos, 66 532 triangles, plastic,
    uv'd planes) with its sphere light swapped for an emissive quad -- spheres are outside the closed set.  The scene data
    belongs to the reference, so it is read from /root/reference where that exists (build container only) and rendered both
    by the unmodified reference and by this repository's front end + oracle: the images must be bit-identical."""
    src = "/root/reference/scenes"
    if not os.path.exists(os.path.join(src, "killeroo-simple.pbrt")) or not os.path.exists(oracle.REF_BINARY):
        pytest.skip("reference scenes / binary not available here")
    os.makedirs(tmp_path / "geometry")
    os.symlink(os.path.join(src, "geometry", "killeroo.pbrt"), tmp_path / "geometry" / "killeroo.pbrt")
    s = open(os.path.join(src, "killeroo-simple.pbrt")).read()
    s = (s.replace('"integer xresolution" [700] "integer yresolution" [700]', '"integer xresolution" [64] "integer yresolution" [64]')
          .replace("killeroo-simple.exr", "kroo.pfm").replace('"integer pixelsamples" [8]', '"integer pixelsamples" [2]')
          .replace('Shape "sphere" "float radius" [3]', 'Shape "trianglemesh" "integer indices" [0 1 2 0 2 3] "point P" [ -3 -3 0  3 -3 0  3 3 0  -3 3 0 ]')
          .replace('"integer nsamples" [8]', '"integer nsamples" [8] "bool twosided" "true"'))
    scene_file = str(tmp_path / "kroo.pbrt")
    open(scene_file, "w").write(s)
    out = str(tmp_path / "ref.pfm")
    oracle.run_reference(scene_file, out, nthreads=4)
    scene = pkg.HostScene(scene_file)
    assert scene.desc.n_tris == 2 * 33264 + 4 + 2
    img, _ = oracle.render_image(scene)
    assert np.array_equal(img, pkg.read_pfm(out))


def test_fuzz_scenes_live_when_reference_present(pkg, oracle, tmp_path):
    """The random scenes of tests/test_gpu_fuzz.py (degenerate soups, every material / light / filter / strategy, spheres,
    object instances, participating media under the volpath integrator) rendered by the unmodified reference and by front end + oracle: bit-identical images.  Needs the
    reference binary (build container); on the GPU box those scenes are checked against the oracle only."""
    if not os.path.exists(oracle.REF_BINARY):
        pytest.skip("reference binary not available here")
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_scenes", os.path.join(os.path.dirname(__file__), "test_gpu_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    scene_file, out = str(tmp_path / "fuzz.pbrt"), str(tmp_path / "ref.pfm")
    # (the last four: the same scenes under a moving camera; with moving shapes and object instances; those beside subsurface materials / grid media;
    # with motions that rotate -- the front end's own MotionBounds in the top-level BVH and the world bound)
    # (and: moving shapes INSIDE object definitions -- a TransformedPrimitive under every ObjectInstance's TransformedPrimitive, ABI 29)
    for gen in (fz.random_scene, fz.random_scene_ext, fz.random_scene_vol, fz.random_scene_moving_camera, fz.random_scene_motion, fz.random_scene_motion_sss_grid,
                fz.random_scene_rotating_motion, fz.random_scene_nested_motion):
        for seed in (range(8) if gen is fz.random_scene_nested_motion else range(16) if gen in (fz.random_scene_vol, fz.random_scene_moving_camera) else (range(24) if gen in (fz.random_scene_motion, fz.random_scene_rotating_motion) else (range(12) if gen is fz.random_scene_motion_sss_grid else range(0, 24, 2)))):
            open(scene_file, "w").write(gen(seed))
            txt = oracle.run_reference(scene_file, out, nthreads=1, timeout=600)  # one thread: overlapping FilmTiles merge in tile order
            img, _ = oracle.render_image(pkg.HostScene(scene_file))
            assert np.array_equal(img, pkg.read_pfm(out)), (gen.__name__, seed)
            if gen in (fz.random_scene_rotating_motion, fz.random_scene_nested_motion):  # the reference's own statistics too: the triangle tests count what the BVHs -- the top-level one
                import re                               # over MotionBounds' boxes -- made the rays visit
                g = lambda pat: int(re.search(pat, txt).group(1)) if re.search(pat, txt) else 0
                sc = pkg.HostScene(scene_file)
                _, _, cn = oracle.render(sc.desc, sc.render_desc())
                assert (cn["closest_rays"], cn["shadow_rays"], cn["tri_tests"]) == (g(r"Regular ray intersection tests\s+(\d+)"), g(r"Shadow ray intersection tests\s+(\d+)"),
                                                                                  g(r"Ray-triangle intersection tests\s+\d+ /\s+(\d+)")), (gen.__name__, seed)
