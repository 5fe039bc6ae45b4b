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
        gens = (fz.random_scene, fz.random_scene_ext, fz.random_scene_vol, fz.random_scene_sss_grid, fz.random_scene_moving_camera, fz.random_scene_motion,
                fz.random_scene_motion_sss_grid, fz.random_scene_rotating_motion, fz.random_scene_nested_motion, fz.random_scene_nested_motion_sss_grid)
        if len(sys.argv) > 3: gens = [g for g in gens if g.__name__ == sys.argv[3]]
        for gen in gens:
            for seed in range(a, b):
                open(scene_file, "w").write(gen(seed))
                try:
                    oracle.run_reference(scene_file, out, nthreads=1, timeout=300)
                except Exception as e:  # the reference aborts on some degenerate inputs (CHECK failures) and does not terminate on others (TimeoutExpired)
                    print(gen.__name__, seed, "reference failed:", str(e)[:80]); continue
                img, _ = oracle.render_image(pkg.HostScene(scene_file))
                ref = pkg.read_pfm(out)
                if not np.array_equal(img, ref):
                    bad += 1
                    print(gen.__name__, seed, "DIFFERS: pixels", int((img != ref).any(axis=2).sum()), "max", float(np.abs(img - ref).max()))
    print("done, mismatches:", bad)


if __name__ == "__main__":
    main()
