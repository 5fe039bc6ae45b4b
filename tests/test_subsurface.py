"""Subsurface scattering: `Material "subsurface"` / `"kdsubsurface"` (/root/reference/src/materials/subsurface.cpp, kdsubsurface.cpp),
TabulatedBSSRDF and the BSSRDF branch of PathIntegrator::Li / VolPathIntegrator::Li (core/bssrdf.cpp, integrators/path.cpp:152-174,
volpath.cpp:150-177).  The host front end computes the photon-beam-diffusion table as the material's constructor does
(pbrt-v3_amd/host/bssrdf.cpp), hands it over in the ABI-24 tables (PgBSSRDF), the CPU oracle renders it bit-identically to the
UNMODIFIED reference (tests/golden/sss_*, rendered by oracle/_ref/pbrt_oracle through oracle/make_golden.py).  All of this runs
without a GPU.  The device half -- k_shade<., ., SSS>, k_sss_probe, k_sss_exit (pbrt-v3_amd/csrc/pg_kernels.hip) -- is checked by the
GPU parity suite on the same goldens (tests/test_gpu_parity.py: film and counters bit-identical to the correctly-rounded oracle)."""
import ctypes as C
import glob
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, check_integrator_stats
from test_gpu_binding import BINDING

SSS = os.path.join(ROOT, "tests", "golden")
NAMES = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(SSS, "sss_*.json")))
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")


def test_goldens_present():
    assert len(NAMES) >= 14


@pytest.mark.parametrize("name", NAMES)
def test_oracle_matches_reference_image_and_stats(pkg, oracle, name):
    scene = pkg.HostScene(os.path.join(SSS, name + ".pbrt"))
    img, cn = oracle.render_image(scene)
    ref = pkg.read_pfm(os.path.join(SSS, name + ".pfm"))
    assert img.shape == ref.shape and np.array_equal(img, ref), f"max |diff| {np.abs(img - ref).max()}"
    stats = json.load(open(os.path.join(SSS, name + ".json")))
    for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests"):  # the probe rays count as regular intersection tests
        assert cn[k] == stats[k], k
    check_integrator_stats(cn, stats)  # path.cpp:45-46, volpath.cpp:45-47 (the average as the reference prints it)


def host_table(pkg, material):
    text = open(os.path.join(SSS, "sss_subsurface.pbrt")).read()
    old = [l for l in text.splitlines() if l.startswith('Material "subsurface"')][0]
    scene = pkg.HostScene(text=text.replace(old, material))
    d = scene.desc
    b = d.bssrdfs[0]
    n = b.n_rho + b.n_radius + 2 * b.n_rho * b.n_radius + b.n_rho
    table = np.ctypeslib.as_array(d.bssrdf_tables, shape=(d.n_bssrdf_floats,))[b.table:b.table + n].copy()
    return table, np.array(list(b.sigma_t), dtype=np.float32), np.array(list(b.rho), dtype=np.float32), b.eta


def test_beam_diffusion_table_equals_the_references(pkg):
    """BSSRDFTable(100, 64) after ComputeBeamDiffusionBSSRDF(0, 1.33) (bssrdf.cpp:149-180), printed by the unmodified reference through
    oracle/ref_probe.cpp and committed as bit patterns: rhoSamples, radiusSamples, profile, rhoEff, profileCDF -- all 13 064 floats."""
    table, sigma_t, rho, eta = host_table(pkg, 'Material "subsurface" "float eta" [ 1.33 ]')
    ref = np.load(os.path.join(SSS, "sss_table_g0_eta1.33.npy"))
    assert table.size == ref.size == 13064 and np.array_equal(table.view(np.uint32), ref)
    # the defaults of CreateSubsurfaceMaterial (subsurface.cpp:94-97) through the TabulatedBSSRDF constructor (bssrdf.h:146-150)
    sa, ss = np.array([.0011, .0024, .014], dtype=np.float32), np.array([2.55, 3.21, 3.77], dtype=np.float32)
    assert np.array_equal(sigma_t, sa + ss) and np.array_equal(rho, ss / (sa + ss)) and eta == np.float32(1.33)


@pytest.mark.parametrize("g,eta", [(0.4, 1.5), (-0.3, 1.1), (0.0, 0.8)])
def test_beam_diffusion_table_live(pkg, g, eta):
    """Further (g, eta) pairs against the reference itself where it is built (eta < 1 runs the branch of FresnelMoment1 with the
    double-precision literal, bssrdf.cpp:48, and a NaN critical angle in the single-scattering term -- both reproduced)."""
    if not os.path.exists(PROBE):
        pytest.skip("oracle/_ref/ref_probe is built only where /root/reference exists")
    out = subprocess.run([PROBE, "bssrdf", str(g), str(eta)], capture_output=True, text=True, check=True).stdout
    ref = np.concatenate([np.array([int(x, 16) for x in line.split()[1:]], dtype=np.uint32) for line in out.splitlines()])
    table, _, _, _ = host_table(pkg, f'Material "subsurface" "float g" [ {g} ] "float eta" [ {eta} ]')
    assert np.array_equal(table.view(np.uint32), ref)


def test_kdsubsurface_inversion_live(pkg):
    """SubsurfaceFromDiffuse (bssrdf.cpp:182-191): InvertCatmullRom of the effective albedo, per channel."""
    if not os.path.exists(PROBE):
        pytest.skip("oracle/_ref/ref_probe is built only where /root/reference exists")
    out = subprocess.run([PROBE, "bssrdf", "0", "1.33", "0.5", "0.4", "0.3", "1", "2", "3"], capture_output=True, text=True, check=True).stdout
    vals = {l.split()[0]: np.array([int(x, 16) for x in l.split()[1:]], dtype=np.uint32).view(np.float32) for l in out.splitlines()}
    _, sigma_t, rho, _ = host_table(pkg, 'Material "kdsubsurface" "rgb Kd" [ 0.5 0.4 0.3 ] "rgb mfp" [ 1 2 3 ] "float eta" [ 1.33 ]')
    st = vals["sigma_a"] + vals["sigma_s"]
    assert np.array_equal(sigma_t, st) and np.array_equal(rho, vals["sigma_s"] / st)


def test_material_identity_follows_the_directives(pkg):
    """Sample_Sp's probe rays accept hits on the SAME Material object (bssrdf.cpp:301): one `Material` directive shared by both
    boxes is one table entry, the same text written twice is two, and materials of these types are never merged."""
    one = pkg.HostScene(os.path.join(SSS, "sss_subsurface.pbrt"))
    two = pkg.HostScene(os.path.join(SSS, "sss_two_materials.pbrt"))
    assert one.desc.n_bssrdfs == 1 and two.desc.n_bssrdfs == 2 and two.desc.n_materials == one.desc.n_materials + 1
    assert two.desc.n_bssrdf_floats == one.desc.n_bssrdf_floats == 13064  # equal (g, eta): one shared table
    black = pkg.HostScene(os.path.join(SSS, "sss_black_no_bssrdf.pbrt"))
    assert black.desc.n_bssrdfs == 0 and not black.desc.material_bssrdf  # the early return of subsurface.cpp:55


def test_textured_parameters_are_evaluated_per_hit(pkg):
    """A texture or a bump map among the parameters: the BSDF becomes a per-hit material with glass's parameter slots, the BSSRDF entry
    keeps the two coefficient parameters as texture references (goldens sss_textured_*)."""
    scene = pkg.HostScene(os.path.join(SSS, "sss_textured_kd_bump.pbrt"))
    d = scene.desc
    assert d.n_bssrdfs == 1 and d.bssrdfs[0].textured == 2 and d.bssrdfs[0].a.tex >= 0 and d.bssrdfs[0].b.tex == -1
    mats = [d.materials[i] for i in range(d.n_materials) if d.material_bssrdf[i] >= 0]
    assert len(mats) == 1 and mats[0].type == 6 and d.textured[mats[0].textured_index].kind == 4 and d.textured[mats[0].textured_index].has_bump == 1
    plain = pkg.HostScene(os.path.join(SSS, "sss_subsurface.pbrt"))
    assert plain.desc.bssrdfs[0].textured == 0


@pytest.mark.parametrize("name", ["sss_subsurface", "sss_kd_rough", "sss_preset_volpath", "sss_two_materials", "sss_mix_component"])
def test_reference_side_binding_flattens_the_reference_bssrdf(pkg, name, tmp_path):
    """The reference's own SubsurfaceMaterial / KdSubsurfaceMaterial objects (its constructor's table, its TabulatedBSSRDF's sigma_t and
    rho), flattened by the compiled binding and rendered by the oracle behind the C ABI: bit-identical to the reference's image."""
    if os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-j8", "_ref/pbrt_gpubind"])
    if not os.path.exists(BINDING):
        pytest.skip("oracle/_ref/pbrt_gpubind is built only where /root/reference exists")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_abi_shim.so"])
    out = str(tmp_path / "bound.pfm")
    p = subprocess.run([BINDING, "--outfile", out, os.path.join(SSS, name + ".pbrt")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PBRT_GPU_LIB=os.path.join(ROOT, "oracle", "liboracle_abi_shim.so")))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert np.array_equal(pkg.read_pfm(out), pkg.read_pfm(os.path.join(SSS, name + ".pfm")))


@pytest.mark.parametrize("seed", [4, 7, 11])
def test_random_scenes_live_when_reference_present(pkg, oracle, seed, tmp_path):
    """Random scenes mixing both CPU-only features into the volumetric fuzz scenes (tests/test_gpu_fuzz.py::random_scene_sss_grid, swept by tools/fuzz_oracle_vs_reference.py: a random
    GridDensityMedium, random subsurface / kdsubsurface materials on degenerate triangle soups and quadrics), rendered now by the
    unmodified reference and by front end + oracle.  Seed 4 has probe segments that collect more than 256 hits on their own
    material.  40 seeds were swept with the tool: 0 mismatches."""
    if not os.path.exists(oracle.REF_BINARY):
        pytest.skip("oracle/_ref/pbrt_oracle not built here")
    import importlib.util
    def load(name, path):
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    fz = load("fuzz_scenes", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
    scene_file, out = str(tmp_path / "fuzz.pbrt"), str(tmp_path / "ref.pfm")
    open(scene_file, "w").write(fz.random_scene_sss_grid(seed))
    oracle.run_reference(scene_file, out, nthreads=1)
    img, _ = oracle.render_image(pkg.HostScene(scene_file))
    assert np.array_equal(img, pkg.read_pfm(out))


def test_find_interval_basics(oracle):
    """The reference's own FindInterval.Basics (src/tests/find_interval.cpp) on the restatement (predicate a[i] <= x)."""
    a = np.arange(10, dtype=np.float32)
    fi = lambda x: oracle.lib().oracle_find_interval_le(a.size, a.ctypes.data, x)
    assert fi(-1) == 0 and fi(100) == a.size - 2
    for i in range(a.size - 1):
        assert fi(i) == i and fi(i + 0.5) == i
        if i > 0:
            assert fi(i - 0.5) == i - 1


@pytest.mark.parametrize("g,eta,seed", [(0.0, 1.33, 7), (0.4, 1.5, 12345)])
def test_spline_routines_equal_the_references_live(pkg, oracle, g, eta, seed):
    """CatmullRomWeights, SampleCatmullRom2D, InvertCatmullRom (core/interpolation.cpp) and FresnelMoment1 (bssrdf.cpp:43-52) of the
    restatement against the reference's own functions (oracle/ref_probe.cpp `spline`) on 400 pseudo-random queries over the beam-
    diffusion table -- including albedo 0 and 1 (the spline's ends, where the profile is all zero) and queries exactly on a node."""
    if not os.path.exists(PROBE):
        pytest.skip("oracle/_ref/ref_probe is built only where /root/reference exists")
    n = 400
    out = subprocess.run([PROBE, "spline", str(g), str(eta), str(seed), str(n)], capture_output=True, text=True, check=True).stdout.splitlines()
    assert len(out) == n
    table, _, _, _ = host_table(pkg, f'Material "subsurface" "float g" [ {g} ] "float eta" [ {eta} ]')
    rho, radius = table[:100].copy(), table[100:164].copy()
    profile, rho_eff, cdf = table[164:164 + 6400].copy(), table[6564:6664].copy(), table[6664:].copy()
    L = oracle.lib()
    f32 = lambda bits: np.array([int(bits, 16)], dtype=np.uint32).view(np.float32)[0]
    b32 = lambda v: int(np.array([v], dtype=np.float32).view(np.uint32)[0])
    for line in out:
        q, cw, s2, inv, fm = [part.split() for part in line.split("|")]
        alpha, u, x = (f32(t) for t in q)
        off, w = C.c_int(-7), np.zeros(4, np.float32)
        ok = L.oracle_catmull_rom_weights(100, rho.ctypes.data, alpha, C.byref(off), w.ctypes.data)
        assert [ok, off.value] + [b32(v) for v in w] == [int(cw[0]), int(cw[1])] + [int(t, 16) for t in cw[2:]], line
        got = L.oracle_sample_catmull_rom_2d(100, 64, rho.ctypes.data, radius.ctypes.data, profile.ctypes.data, cdf.ctypes.data, alpha, u)
        assert b32(got) == int(s2[0], 16) or (np.isnan(got) and np.isnan(f32(s2[0]))), line
        assert b32(L.oracle_invert_catmull_rom(100, rho.ctypes.data, rho_eff.ctypes.data, x)) == int(inv[0], 16), line
        assert b32(L.oracle_fresnel_moment1(np.float32(0.5) + np.float32(1.5) * alpha)) == int(fm[0], 16), line
