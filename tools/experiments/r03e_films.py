#!/usr/bin/env python3
"""Renders a set of golden scenes with the device library PBRT_GPU_LIB points at and saves the films (rgb, weight) to OUT.npz:
two runs with two libraries must agree bit for bit when the libraries differ only in the any-hit visiting order."""
import glob, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
out = {}
names = sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(ROOT, "tests", "golden", "*.json")))
for name in names:
    if name.startswith(("sampler_", "filter_02")): continue
    scene = pkg.HostScene(os.path.join(ROOT, "tests", "golden", name + ".pbrt"))
    gs = pkg.GpuScene(scene.desc)
    film, strays = gs.render(scene.render_desc())
    out[name + ".rgb"], out[name + ".w"] = film["rgb"].copy(), film["weight"].copy()
    gs.close()
np.savez(sys.argv[1], **out)
print(len(out) // 2, "scenes")
