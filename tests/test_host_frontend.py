"""Host logic: .pbrt parser / API state machine, BVHAccel build invariants, Halton set-up, Film merge,
tile sharding.  CPU only."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLD, ROOT

MINI = '''
LookAt 0 0 -5  0 0 0  0 1 0
Camera "perspective" "float fov" [ 45 ]
Film "image" "integer xresolution" [ %d ] "integer yresolution" [ %d ] "string filename" "t.pfm"
Sampler "halton" "integer pixelsamples" [ %d ]
WorldBegin
AttributeBegin
  AreaLightSource "diffuse" "rgb L" [ 5 5 5 ]
  Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ -1 2 -1  1 2 -1  0 2 1 ]
AttributeEnd
Shape "trianglemesh" "integer indices" [ 0 1 2 0 2 3 ] "point P" [ -2 -1 -2  2 -1 -2  2 -1 2  -2 -1 2 ]
WorldEnd
'''


def test_defaults_and_descs(pkg):
    s = pkg.HostScene(text=MINI % (33, 17, 4))
    d, rd = s.desc, s.render_desc()
    assert (d.n_tris, d.n_lights, d.n_materials) == (3, 1, 1)
    assert d.light_strategy == pkg.abi.PG_LIGHTS_UNIFORM  # one light => uniform (lightdistrib.cpp:50)
    assert d.materials[0].type == pkg.abi.PG_MAT_MATTE and tuple(d.materials[0].kd) == (0.5, 0.5, 0.5)  # api.cpp default
    assert (rd.spp, rd.max_depth, rd.rr_threshold) == (4, 5, 1.0)  # path.cpp:193-209
    assert tuple(rd.filter_radius) == (0.5, 0.5) and tuple(rd.sample_bounds) == (0, 0, 33, 17)
    # HaltonSampler ctor (halton.cpp:75-92): 33 -> 64 = 2^6, 17 -> 27 = 3^3
    assert tuple(rd.base_scales) == (64, 27) and tuple(rd.base_exponents) == (6, 3) and rd.sample_stride == 64 * 27
    assert (rd.mult_inverse[0] * 27) % 64 == 1 and (rd.mult_inverse[1] * 64) % 27 == 1
    assert pkg.GpuScene.tile_count is not None
    light = d.lights[0]
    assert d.tri_light[light.prim] == 0 and abs(light.area - 2.0) < 1e-6


def test_halton_permutations_are_permutations(pkg):
    s = pkg.HostScene(text=MINI % (16, 16, 1))
    d = s.desc
    sums = np.ctypeslib.as_array(d.perm_sums, (d.n_perm_dims + 1,))
    perms = np.ctypeslib.as_array(d.perms, (sums[-1],))
    assert d.n_perm_dims >= 5 + 8 * 6
    for i in range(d.n_perm_dims):
        p = perms[sums[i]:sums[i + 1]]
        assert sorted(p.tolist()) == list(range(len(p)))
    # first bases of the default-seeded table (RNG(), lowdiscrepancy.cpp:2490-2504) are not the identity
    assert perms[:2].tolist() in ([0, 1], [1, 0]) and not np.array_equal(perms[sums[5]:sums[6]], np.arange(13))


def test_bvh_invariants(pkg):
    s = pkg.HostScene(os.path.join(GOLD, "synthetic_n40.pbrt"))
    nodes, idx, P = s.nodes(), s.indices(), s.positions()
    n_tris = s.desc.n_tris
    seen = np.zeros(n_tris, int)
    stack = [0]
    while stack:
        i = stack.pop()
        nd = nodes[i]
        if nd["nprims"] > 0:
            for k in range(nd["offset"], nd["offset"] + nd["nprims"]):
                seen[k] += 1
                v = P[idx[k]]
                assert (v >= nd["bmin"]).all() and (v <= nd["bmax"]).all()
            assert nd["nprims"] <= 4  # maxnodeprims default (bvh.cpp:757)
        else:
            for c in (i + 1, nd["offset"]):  # first child adjacent, second at offset (bvh.cpp:640-658)
                assert (nodes[c]["bmin"] >= nd["bmin"]).all() and (nodes[c]["bmax"] <= nd["bmax"]).all()
                stack.append(c)
            assert nd["axis"] in (0, 1, 2)
    assert (seen == 1).all()


@pytest.mark.parametrize("split", ["sah", "middle", "equal", "hlbvh"])
def test_split_methods_give_same_image(pkg, oracle, split):
    txt = open(os.path.join(GOLD, "cornell_40x24.pbrt")).read().replace('Accelerator "bvh"', f'Accelerator "bvh" "string splitmethod" "{split}" "integer maxnodeprims" [ 2 ]')
    img, _ = oracle.render_image(pkg.HostScene(text=txt))
    assert np.array_equal(img, pkg.read_pfm(os.path.join(GOLD, "cornell_40x24.pfm")))


def test_errors_are_reported_not_thrown(pkg):
    before = pkg.host_lib().pbrt_host_error_count()
    s = pkg.HostScene(text=(MINI % (16, 16, 1)).replace("WorldEnd", 'Shape "teapot"\nLightSource "infinite" "string mapname" "sky.exr"\nTexture "t" "spectrum" "imagemap"\nWorldEnd'))
    assert s.desc.n_tris == 3  # an unknown shape and unreadable assets are reported as the reference reports them, the scene still loads (error.cpp:62-102 semantics)
    assert pkg.host_lib().pbrt_host_error_count() >= before + 2  # (a shape the REFERENCE has and this build has not -- "curve" -- refuses the frame: the last test of this file)
    with pytest.raises(pkg.PbrtGpuError):
        pkg.HostScene(text="WorldBegin\nWorldEnd\n" if False else "Camera \"perspective\"\n")  # no WorldEnd => nothing to render


def test_plymesh_loader(pkg, tmp_path):
    """Shape "plymesh" (plymesh.cpp:158-290): quads split as (a b c)(d a c), other polygons skipped, bad files reported."""
    ply = tmp_path / "m.ply"
    ply.write_text("ply\nformat ascii 1.0\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\n"
                   "element face 3\nproperty list uchar int vertex_indices\nend_header\n"
                   "0 0 0\n1 0 0\n1 1 0\n0 1 0\n2 2 2\n4 0 1 2 3\n3 0 1 4\n5 0 1 2 3 4\n")
    mini = (MINI % (16, 16, 1)).replace("WorldEnd", f'Shape "plymesh" "string filename" "{ply}"\nWorldEnd')
    s = pkg.HostScene(text=mini)
    assert s.desc.n_tris == 3 + 3  # the scene's own 3 + quad (2) + triangle (1); the pentagon is ignored
    before = pkg.host_lib().pbrt_host_error_count()
    bad = tmp_path / "bad.ply"
    bad.write_text("ply\nformat ascii 1.0\nelement vertex 3\nproperty float x\nproperty float y\nproperty float z\n"
                   "element face 1\nproperty list uchar int vertex_indices\nend_header\n0 0 0\n1 0 0\n0 1 0\n3 0 1 7\n")
    for fn in (bad, tmp_path / "missing.ply"):
        s2 = pkg.HostScene(text=(MINI % (16, 16, 1)).replace("WorldEnd", f'Shape "plymesh" "string filename" "{fn}"\nWorldEnd'))
        assert s2.desc.n_tris == 3  # out-of-range index / unreadable file: reported, shape dropped, scene still loads
    assert pkg.host_lib().pbrt_host_error_count() >= before + 2


def test_empty_world_loads(pkg, oracle):
    s = pkg.HostScene(text='Film "image" "integer xresolution" [ 8 ] "integer yresolution" [ 8 ] "string filename" "e.pfm"\nWorldBegin\nWorldEnd\n')
    assert s.desc.n_tris == 0 and s.desc.n_nodes == 0
    img, cn = oracle.render_image(s)
    assert img.shape == (8, 8, 3) and not img.any() and cn["camera_rays"] == 64 * 16


def test_tile_sharding_partitions_frame(pkg, oracle):
    s = pkg.HostScene(text=MINI % (70, 37, 1))
    total = oracle.lib().oracle_render_tile_count(C.byref(s.render_desc()))
    assert total == 5 * 3
    for world in (1, 2, 3, 4, 8, 16):
        counts = [oracle.lib().oracle_render_tile_count(C.byref(s.render_desc(r, world))) for r in range(world)]
        assert sum(counts) == total and max(counts) - min(counts) <= 1


def test_film_merge_stray_samples(pkg):
    """A sample whose film offset is exactly 0 also lands in the previous pixel (film.h:127-132): same-tile strays add
    in RGB after the pixel's own samples, cross-tile strays add in XYZ."""
    s = pkg.HostScene(text=MINI % (32, 16, 1))
    rd = s.render_desc()
    film = np.zeros(2 * 256, pkg.FILM_PIXEL_DTYPE)
    film["rgb"] = 1.0
    film["weight"] = 1.0
    strays = np.zeros(2, pkg.STRAY_DTYPE)
    strays[0] = (4, 4, 5, 4, (0.5, 0.25, 0.125), 1.0)     # same tile
    strays[1] = (15, 3, 16, 3, (2.0, 2.0, 2.0), 1.0)       # from tile 1 into tile 0
    s.film_clear(); s.film_merge(rd, film, strays)
    img = s.film_image()
    to_xyz = lambda c: np.array([0.412453 * c[0] + 0.357580 * c[1] + 0.180423 * c[2], 0.212671 * c[0] + 0.715160 * c[1] + 0.072169 * c[2],
                                 0.019334 * c[0] + 0.119193 * c[1] + 0.950227 * c[2]], np.float32)
    to_rgb = lambda x: np.array([3.240479 * x[0] - 1.537150 * x[1] - 0.498535 * x[2], -0.969256 * x[0] + 1.875991 * x[1] + 0.041556 * x[2],
                                 0.055648 * x[0] - 0.204043 * x[1] + 1.057311 * x[2]], np.float32)
    np.testing.assert_allclose(img[4, 4], to_rgb(to_xyz(np.array([1.5, 1.25, 1.125], np.float32))) / 2, rtol=1e-6)
    np.testing.assert_allclose(img[3, 15], to_rgb(to_xyz(np.ones(3, np.float32)) + to_xyz(np.full(3, 2.0, np.float32))) / 2, rtol=1e-6)
    np.testing.assert_allclose(img[0, 0], to_rgb(to_xyz(np.ones(3, np.float32))), rtol=1e-6)


def test_cli_refuses_without_backend(pkg, tmp_path):
    """pbrt_amd never renders on the CPU: with no HIP back end reachable it exits non-zero."""
    import subprocess
    exe = os.path.join(ROOT, "pbrt-v3_amd", "pbrt_amd")
    scene = tmp_path / "m.pbrt"
    scene.write_text(MINI % (16, 16, 1))
    r = subprocess.run([exe, "--quiet", str(scene)], env=dict(os.environ, PBRT_GPU_LIB="/nonexistent.so"), capture_output=True, text=True)
    assert r.returncode != 0 and "no HIP back end" in r.stderr


def test_bvh_build_does_not_depend_on_the_thread_count(pkg, tmp_path, monkeypatch):
    """The SAH build hands subtrees of >= 32768 primitives to separate threads (--nthreads / PBRT_NTHREADS); leaves write their
    primitives to the positions the reference's depth-first order gives them, so nodes and primitive order are the same for
    any thread count."""
    import gen_synthetic
    path = str(tmp_path / "s.pbrt")
    gen_synthetic.write_scene(path, n=220, xres=32, yres=18, spp=1, filename="s.pfm")  # 95 922 triangles
    built = []
    for threads in ("1", "3", "16"):
        monkeypatch.setenv("PBRT_NTHREADS", threads)
        s = pkg.HostScene(path)
        built.append((s.nodes().tobytes(), s.indices().tobytes()))
    assert built[0] == built[1] == built[2]


def test_malformed_scene_raises_instead_of_exiting(pkg):
    """A fatal parse error (where the reference calls exit(1)) unwinds to the C entry point: the host process survives, the load
    returns no scene, and the library is usable afterwards."""
    for bad in ['Shape "trianglemesh" "integer indices" [ 0 1 2', "LookAt 0 0 0 x", 'WorldBegin\nShape trianglemesh\n']:
        with pytest.raises(pkg.PbrtGpuError):
            pkg.HostScene(text=bad)
    assert pkg.HostScene(os.path.join(GOLD, "cornell_32.pbrt")).desc.n_tris == 36


def test_hostile_geometry_is_reported_not_read_out_of_bounds(pkg):
    """Inputs a fuzzer found (mutated golden scenes under AddressSanitizer, tools/fuzz_host_frontend.py): the reference reads out of
    bounds, trips a CHECK or hangs on them; the library reports them and stays usable."""
    head = MINI.split("WorldBegin")[0] % (8, 8, 1) + "WorldBegin\n"
    # a negative vertex index (the reference checks only the upper bound): the mesh is refused, the scene has no geometry
    s = pkg.HostScene(text=head + 'Shape "trianglemesh" "integer indices" [ 0 -1 2 ] "point P" [ 0 0 0 1 0 0 0 1 0 ]\nWorldEnd\n')
    assert s.desc.n_tris == 0
    # non-finite vertices: no SAH bucket holds the centroid (bvh.cpp:323-324 CHECKs in the reference)
    tris = " ".join(f"{i} 0 0  {i} 1 0  {i} 0 1" for i in range(6))
    idx = " ".join(str(k) for k in range(18))
    with pytest.raises(pkg.PbrtGpuError):
        pkg.HostScene(text=head + f'Shape "trianglemesh" "integer indices" [ {idx} ] "point P" [ nan 0 0 {tris[5:]} ]\nWorldEnd\n')
    # a file that ends inside a directive after its last tokenizer is gone: the error message must not read the freed location
    with pytest.raises(pkg.PbrtGpuError):
        pkg.HostScene(text=head + 'Shape "trianglemesh" "integer indices"')
    # a control mesh that is not a manifold (three faces on one edge): the reference aborts in SDFace::vnum() or walks forever
    with pytest.raises(pkg.PbrtGpuError):
        pkg.HostScene(text=head + 'Shape "loopsubdiv" "integer levels" [ 1 ] "integer indices" [ 0 1 2  0 1 3  0 1 4  1 0 2 ] '
                                  '"point P" [ 0 0 0  1 0 0  0 1 0  0 0 1  1 1 1 ]\nWorldEnd\n')
    assert pkg.HostScene(os.path.join(GOLD, "cornell_32.pbrt")).desc.n_tris == 36


def test_file_that_includes_itself_is_an_error(pkg, tmp_path):
    f = tmp_path / "self.pbrt"
    f.write_text(f'Include "{f}"\n')
    with pytest.raises(pkg.PbrtGpuError):
        pkg.HostScene(str(f))


def test_transform_end_inside_attribute_block_does_not_read_an_empty_stack(pkg):
    """AttributeBegin pushes a transform too; a stray TransformEnd inside the block takes it, and the reference's AttributeEnd then
    reads back() of the empty stack (api.cpp:1141-1163).  Here: reported, the scene still loads with its geometry."""
    one = open(os.path.join(GOLD, "cornell_32.pbrt")).read()
    head, world = one.split("WorldBegin", 1)
    s = pkg.HostScene(text=head + "WorldBegin\nAttributeBegin\nTransformEnd\nAttributeEnd\n" + world)
    assert s.desc.n_tris == 36


def test_hostile_maxdepth_does_not_overflow_the_dimension_count(pkg):
    """"integer maxdepth" is the file's to choose: the Halton table's size is computed in 64 bits and never below the camera's five
    dimensions (fuzz seed 101: 8 * (maxdepth + 2) overflowed an int)."""
    one = open(os.path.join(GOLD, "cornell_32.pbrt")).read()
    for depth, dims in ((-2147483648, 5), (-7, 5), (2147483647, 1000), (5, 61)):
        s = pkg.HostScene(text=one.replace('"integer maxdepth" [ 5 ]', f'"integer maxdepth" [ {depth} ]'))
        assert s.desc.n_perm_dims == dims, depth


def test_second_frame_starts_from_fresh_render_options(pkg):
    """Several WorldBegin / WorldEnd frames in one file: pbrtWorldEnd resets RenderOptions (api.cpp:1630-1640), so frame 2 does not
    inherit frame 1's film, sampler, integrator or material tables.  (A load keeps the last frame.)"""
    one = open(os.path.join(GOLD, "cornell_32.pbrt")).read()
    head, world = one.split("WorldBegin", 1)
    two = head + "WorldBegin" + world + 'Film "image" "integer xresolution" [ 16 ] "integer yresolution" [ 8 ] "string filename" "b.pfm"\nLookAt 278 273 -800 278 273 0 0 1 0\nCamera "perspective" "float fov" [ 39 ]\nWorldBegin' + world
    s1, s2 = pkg.HostScene(text=one), pkg.HostScene(text=two)
    assert s2.film_size == (16, 8)
    rd = s2.render_desc()
    assert rd.spp == 16 and rd.max_depth == 5  # the defaults again, not frame 1's "pixelsamples" 8 / its Integrator line
    assert s2.desc.n_materials == s1.desc.n_materials and s2.desc.n_bxdfs == s1.desc.n_bxdfs  # tables not accumulated across frames


def test_every_golden_scene_loads_without_an_error_message(pkg):
    """A scene asset that cannot be read (a texture, a PLY mesh, an environment map) is an Error() and a 1x1 / empty fallback in the
    reference and here (imageio.cpp:60-78, mipmap.h / imagemap.cpp:55-75): the render goes on and LOOKS plausible.  Round 3's divergent
    stand-ins rendered that way -- their alpha, bump and sky files were written under other names than the scene asked for -- so the
    goldens are held to loading with no Error() at all."""
    import glob
    dirs = [GOLD, os.path.join(ROOT, "tests", "golden_sss"), os.path.join(ROOT, "tests", "golden_grid"), os.path.join(ROOT, "tests", "golden_large")]
    files = sorted(f for d in dirs for f in glob.glob(os.path.join(d, "*.pbrt")) if os.path.exists(f[:-5] + ".json"))
    assert len(files) > 100
    # scenes that ask for the error path on purpose: an absent map name / texture file / named material, with the reference's fallback
    on_purpose = {"light_gonio_power", "light_projection", "mat_mix", "sampler_stratified_dims_tex", "sobol_tex_lens", "tex_image", "tex_image_lens",
                  "camanim_lens_tex", "camanim_sobol_tex", "camanim_strat_dims_tex", "camanim_random_tex", "camanim_env", "motion_tex"}
    for f in files:
        before = pkg.host_lib().pbrt_host_error_count()
        pkg.HostScene(f).close()
        reported = pkg.host_lib().pbrt_host_error_count() != before
        name = os.path.basename(f)[:-5]
        assert name in on_purpose or not reported, f"{name}: the host front end reported an Error() while loading"


def test_moving_shapes_and_instances_become_animated_instances(pkg):
    """The reference interpolates an AnimatedTransform per ray inside TransformedPrimitive::Intersect (primitive.cpp:76-103).  A motion -- with or
    without rotation (Dot(R[0], R[1]) < 0.9995; its bounds: host/motion_bounds.cpp, tests/test_motion_bounds.py) -- becomes a PgInstance with the two
    ends' decompositions (a moving SHAPE: an anonymous object created at the identity, api.cpp:1386-1419).  A moving shape inside an object
    definition is a TransformedPrimitive among that object's primitives (PG_PRIM_INSTANCE inside the object's run, ABI 29: the nest_motion* goldens).
    A motion that MIRRORS is REFUSED -- an Error, no frame, never an image with one end of the motion.  A moving CAMERA is rendered (the camanim_* goldens);
    textures and lights take the start transform in the reference itself (api.cpp WARN_IF_ANIMATED_TRANSFORM): a Warning."""
    anim = 'ActiveTransform EndTime\nTranslate 0.3 0 0\nActiveTransform All\n'
    spin = 'ActiveTransform EndTime\nRotate 40 0 1 0\nActiveTransform All\n'
    mini = MINI % (16, 16, 1)
    tri = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\n'
    shape = lambda a: mini.replace("WorldEnd", "AttributeBegin\nTranslate 0 0 1\n" + a + tri + "AttributeEnd\nWorldEnd")
    inst = lambda a: mini.replace("WorldEnd", 'ObjectBegin "o"\n' + tri + 'ObjectEnd\nAttributeBegin\n' + a + 'ObjectInstance "o"\nAttributeEnd\nWorldEnd')
    before = pkg.host_lib().pbrt_host_error_count()
    for what, txt in (("shape", shape(anim)), ("instance", inst(anim))):
        s = pkg.HostScene(text=txt)
        d = s.desc
        moving = [d.instances[i] for i in range(d.n_instances) if d.instances[i].animated]
        assert len(moving) == 1 and pkg.host_lib().pbrt_host_error_count() == before, what
        m = moving[0]
        assert list(m.time) == [0.0, 1.0] and abs((m.T[1][0] - m.T[0][0]) - 0.3) < 1e-6 and list(m.R[0]) == list(m.R[1]) and m.i2w_end[3] - m.i2w[3] == pytest.approx(0.3)
        if what == "shape":  # created at the identity: the object's vertices are the file's, the transform carries the CTM
            assert m.i2w[11] == 1.0 and d.objects[m.object].n_prims == 1 and d.objects[m.object].n_nodes == 0
        s.close()
    for what, txt in (("rotating shape", shape(spin)), ("rotating instance", inst(spin))):
        s = pkg.HostScene(text=txt)
        d = s.desc
        moving = [d.instances[i] for i in range(d.n_instances) if d.instances[i].animated]
        assert len(moving) == 1 and pkg.host_lib().pbrt_host_error_count() == before, what
        q0, q1 = list(moving[0].R[0]), list(moving[0].R[1])
        assert sum(a * b for a, b in zip(q0, q1)) < 0.9995, what
        s.close()
    nested = mini.replace("WorldEnd", 'ObjectBegin "o"\n' + anim + tri + 'ObjectEnd\nObjectInstance "o"\nWorldEnd')
    # (a mirrored motion: the reference slerps the non-unit quaternion of an improper rotation, and its own renders abort or do not terminate)
    s = pkg.HostScene(text=nested)  # the object "o" holds ONE primitive: the moving shape's TransformedPrimitive over an object of its own
    d = s.desc
    assert d.n_instances == 2 and d.n_objects == 2 and pkg.host_lib().pbrt_host_error_count() == before
    outer = [k for k in range(d.n_tris) if d.tri_flags[k] & pkg.abi.PG_PRIM_INSTANCE]
    inner = [k for k in range(d.n_tris, d.n_prims_all) if d.tri_flags[k] & pkg.abi.PG_PRIM_INSTANCE]
    assert len(outer) == 1 and len(inner) == 1
    io, ii = d.instances[d.indices[3 * outer[0]]], d.instances[d.indices[3 * inner[0]]]
    assert not io.animated and ii.animated and d.objects[io.object].first_prim == inner[0] and d.objects[io.object].n_prims == 1
    assert d.objects[ii.object].n_prims == 1 and not (d.tri_flags[d.objects[ii.object].first_prim] & pkg.abi.PG_PRIM_INSTANCE)
    s.close()
    for what, txt in (("mirrored instance", inst("Scale 1 -1 1\n" + anim)), ("mirrored rotating shape", shape("Scale -1 1 1\n" + spin))):
        before = pkg.host_lib().pbrt_host_error_count()
        with pytest.raises(pkg.PbrtGpuError):
            pkg.HostScene(text=txt)
        assert pkg.host_lib().pbrt_host_error_count() > before, what
    before = pkg.host_lib().pbrt_host_error_count()
    s = pkg.HostScene(text=mini.replace("WorldEnd", "AttributeBegin\n" + anim + 'LightSource "point"\nTexture "t" "float" "checkerboard"\nAttributeEnd\nWorldEnd'))
    assert s.desc.n_lights >= 1 and pkg.host_lib().pbrt_host_error_count() == before  # the reference's own behaviour: warnings only
    s = pkg.HostScene(text=mini.replace("Camera ", anim + "Camera ", 1))  # a moving camera: AnimatedTransform CameraToWorld, rendered
    rd = s.render_desc()
    assert rd.camera_animated == 1 and pkg.host_lib().pbrt_host_error_count() == before
    assert list(rd.camera_time) == [0.0, 1.0] and abs(rd.camera_T[1][0] - rd.camera_T[0][0]) > 0.2 and list(rd.camera_R[0]) == list(rd.camera_R[1])


def test_features_of_the_reference_outside_the_closed_set_refuse_the_frame(pkg):
    """What the REFERENCE renders and this build does not -- hair / disney / fourier materials, curve shapes, Ptex textures, a textured subsurface component
    inside a mix -- is an Error() and NO frame: never an image with a stand-in (matte for the material, the shape left out, the BSSRDF dropped).  A name the
    reference does not know either gets the reference's own Warning and fallback (api.cpp:531, :590, :640, :676, :750)."""
    mini = MINI % (16, 16, 1)
    tri = 'Shape "trianglemesh" "integer indices" [0 1 2] "point P" [0 0 0 1 0 0 0 1 0]\n'
    sss_mix = ('Texture "chk" "spectrum" "checkerboard"\nMakeNamedMaterial "wax" "string type" "kdsubsurface" "texture Kd" "chk"\nMakeNamedMaterial "paint" "string type" "matte"\n'
               'Material "mix" "string namedmaterial1" "wax" "string namedmaterial2" "paint"\n' + tri)
    for what, extra in (("disney", 'Material "disney"\n' + tri), ("hair", 'Material "hair"\n' + tri), ("fourier", 'Material "fourier" "string bsdffile" "none.bsdf"\n' + tri),
                        ("curve", 'Shape "curve" "point P" [0 0 0 1 0 0 1 1 0 0 1 0]\n'), ("ptex", 'Texture "p" "spectrum" "ptex" "string filename" "none.ptx"\n' + tri),
                        ("textured subsurface in a mix", sss_mix)):
        before = pkg.host_lib().pbrt_host_error_count()
        with pytest.raises(pkg.PbrtGpuError):
            pkg.HostScene(text=mini.replace("WorldEnd", "AttributeBegin\n" + extra + "AttributeEnd\nWorldEnd"))
        assert pkg.host_lib().pbrt_host_error_count() > before, what
    # names nobody knows: the reference's warnings, the scene loads (matte for the material, nothing for the shape / texture)
    before = pkg.host_lib().pbrt_host_error_count()
    s = pkg.HostScene(text=mini.replace("WorldEnd", 'AttributeBegin\nMaterial "velvet"\n' + tri + 'Shape "teapot"\nTexture "t" "float" "plaid"\nAttributeEnd\nWorldEnd'))
    assert pkg.host_lib().pbrt_host_error_count() == before and s.desc.n_tris >= 1
    s.close()
