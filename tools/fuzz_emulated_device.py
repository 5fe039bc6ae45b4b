#!/usr/bin/env python3
"""Wider sweep of the random scenes of tests/test_gpu_fuzz.py through the device kernels: every seed is rendered by the device library
(the SIMT emulator build of tests/emu/ when PBRT_EMULATED_DEVICE=1, or a real GPU) and checked like the tests check theirs (check_scene) --
film, stray samples and counters bit-identical to the oracle in the reference's shadow-ray order, film and ray counts in the product's
default free order, rays through the same soup bit-exact with the reference's counters.
usage: PBRT_GPU_LIB=/tmp/emu/libpbrt_gpu_emulated.so PBRT_EMULATED_DEVICE=1 python tools/fuzz_emulated_device.py FIRST LAST [GENERATOR]"""
import importlib.util
import os
import sys
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PG_ANYHIT_ORDER", "reference")  # as tests/conftest.py: the counters compared are the reference's (check_scene renders the free order too)
from __graft_entry__ import load_package  # noqa: E402


def main():
    a, b = int(sys.argv[1]), int(sys.argv[2])
    pkg = load_package()
    from oracle import oracle
    oracle.lib()
    spec = importlib.util.spec_from_file_location("fuzz_scenes", os.path.join(ROOT, "tests", "test_gpu_fuzz.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    gens = {"random_scene": fz.random_scene, "random_scene_ext": fz.random_scene_ext, "random_scene_vol": fz.random_scene_vol,
            "random_scene_sss": lambda s: fz.random_scene_sss_grid(s, "sss"), "random_scene_grid": lambda s: fz.random_scene_sss_grid(s, "grid"),
            "random_scene_pixel_sampler": fz.random_scene_pixel_sampler, "random_scene_moving_camera": fz.random_scene_moving_camera,
            "random_scene_motion": fz.random_scene_motion, "random_scene_motion_sss_grid": fz.random_scene_motion_sss_grid,
            "random_scene_rotating_motion": fz.random_scene_rotating_motion,
            "random_scene_nested_motion": fz.random_scene_nested_motion, "random_scene_nested_motion_sss_grid": fz.random_scene_nested_motion_sss_grid}
    if len(sys.argv) > 3: gens = {k: v for k, v in gens.items() if k == sys.argv[3]}
    bad = 0
    for name, gen in gens.items():
        for seed in range(a, b):
            try:
                fz.check_scene(pkg, oracle, gen(seed), seed)  # film, strays, counters and rays bit for bit, in both shadow-ray orders
            except AssertionError as e:
                bad += 1
                print(name, seed, "FAILS:", str(e)[:300], flush=True)
            except Exception:
                bad += 1
                print(name, seed, "ERROR", flush=True)
                traceback.print_exc()
        print(f"{name}: seeds {a} .. {b - 1} done, {bad} failures so far", flush=True)
    print(f"{(b - a) * len(gens)} scenes, {bad} failures")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
