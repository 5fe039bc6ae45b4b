#!/usr/bin/env python3
"""Wider sweep of the random scenes of tests/test_gpu_fuzz.py: every seed is rendered by the unmodified reference binary
(oracle/_ref/pbrt_oracle) and by front end + oracle; any image that is not bit-identical is reported.  Build container only.
usage: python tools/fuzz_oracle_vs_reference.py FIRST_SEED LAST_SEED [GENERATOR]"""
import importlib.util
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from __graft_entry__ import load_package  # noqa: E402


def random_scene_sss_grid(fz, seed):
    """The volumetric / extended random scenes with the two CPU-only features mixed in: one of the media becomes a GridDensityMedium
    (random dimensions, densities with empty voxels, box and CTM), and the glass / mirror / plastic / matte materials become
    subsurface / kdsubsurface materials with random coefficients (smooth and rough, presets, textured Kd)."""
    rng = np.random.default_rng(77000 + seed)
    f = lambda a: " ".join(f"{x:.9g}" for x in np.asarray(a, np.float32).ravel())
    text = fz.random_scene_vol(seed) if seed % 3 else fz.random_scene_ext(seed)
    if 'MakeNamedMedium "m1"' in text:
        line = [l for l in text.splitlines() if l.startswith('MakeNamedMedium "m1"')][0]
        nx, ny, nz = (int(v) for v in rng.integers(1, 6, size=3))
        den = rng.random(nx * ny * nz) * (rng.random(nx * ny * nz) > 0.25) * (0.5 + 3 * rng.random())
        if den.max() == 0: den[0] = 1
        sa, ss = 0.3 * rng.random(), 0.5 + 3 * rng.random()
        p0 = rng.normal(size=3) - 1.0
        grid = ('MakeNamedMedium "m1" "string type" "heterogeneous" "rgb sigma_a" [ %s ] "rgb sigma_s" [ %s ] "float g" [ %.4g ] "integer nx" [ %d ] "integer ny" [ %d ] "integer nz" [ %d ] '
                '"point p0" [ %s ] "point p1" [ %s ] "float density" [ %s ]' % (f([sa] * 3), f([ss] * 3), 1.4 * rng.random() - 0.7, nx, ny, nz, f(p0), f(p0 + 1 + 2 * rng.random(3)), f(den)))
        text = text.replace(line, grid, 1)
    def sss():
        k = rng.integers(0, 4)
        rough = ' "float uroughness" [ %.4g ] "float vroughness" [ %.4g ]' % (0.3 * rng.random(), 0.3 * rng.random()) if rng.random() < 0.4 else ""
        if k == 0: return 'Material "subsurface" "rgb sigma_a" [ %s ] "rgb sigma_s" [ %s ] "float eta" [ %.4g ] "float g" [ %.4g ]%s' % (f(rng.random(3)), f(0.2 + 5 * rng.random(3)), 1.1 + 0.5 * rng.random(), 1.2 * rng.random() - 0.5, rough)
        if k == 1: return 'Material "kdsubsurface" "rgb Kd" [ %s ] "rgb mfp" [ %s ] "float eta" [ %.4g ]%s' % (f(rng.random(3)), f(0.05 + rng.random(3)), 1.2 + 0.4 * rng.random(), rough)
        if k == 2: return 'Material "subsurface" "string name" "%s" "float scale" [ %.4g ]%s' % (["Skin1", "Marble", "Ketchup", "Wholemilk"][seed % 4], 0.5 + 20 * rng.random(), rough)
        return 'Material "kdsubsurface" "rgb Kd" [ %s ] "rgb mfp" [ %s ] "rgb Kr" [ %s ] "float scale" [ %.4g ]' % (f(rng.random(3)), f(0.2 + rng.random(3)), f(rng.random(3)), 0.5 + rng.random())
    out = []
    for l in text.splitlines():
        st = l.strip()
        if (st.startswith('Material "glass"') or st.startswith('Material "mirror"') or st.startswith('Material "plastic"') or st.startswith('Material "matte"')) and "texture" not in st and rng.random() < 0.6:
            l = l[:len(l) - len(l.lstrip())] + sss()
        out.append(l)
    return "\n".join(out) + "\n"


def main():
    a, b = int(sys.argv[1]), int(sys.argv[2])
    pkg = load_package()
    from oracle import oracle
    spec = importlib.util.spec_from_file_location("fuzz_scenes", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    bad = 0
    with tempfile.TemporaryDirectory() as d:
        scene_file, out = os.path.join(d, "fuzz.pbrt"), os.path.join(d, "ref.pfm")
        gens = (fz.random_scene, fz.random_scene_ext, fz.random_scene_vol, lambda seed: random_scene_sss_grid(fz, seed))
        gens[-1].__name__ = "random_scene_sss_grid"
        if len(sys.argv) > 3: gens = [g for g in gens if g.__name__ == sys.argv[3]]
        for gen in gens:
            for seed in range(a, b):
                open(scene_file, "w").write(gen(seed))
                try:
                    oracle.run_reference(scene_file, out, nthreads=1)
                except Exception as e:  # the reference aborts on some degenerate inputs (CHECK failures)
                    print(gen.__name__, seed, "reference failed:", str(e)[:80]); continue
                img, _ = oracle.render_image(pkg.HostScene(scene_file))
                ref = pkg.read_pfm(out)
                if not np.array_equal(img, ref):
                    bad += 1
                    print(gen.__name__, seed, "DIFFERS: pixels", int((img != ref).any(axis=2).sum()), "max", float(np.abs(img - ref).max()))
    print("done, mismatches:", bad)


if __name__ == "__main__":
    main()
