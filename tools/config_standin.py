#!/usr/bin/env python3
"""Stand-ins for BASELINE.json configs 3 and 4, whose scene files (crown, San Miguel) are not in the reference repository:
the synthetic heightfield-in-a-box of config 2 scaled to the same triangle counts.
  config 3: ~5 M triangles, PathIntegrator, 1920x1080
  config 4: ~10 M triangles inside a HomogeneousMedium that also surrounds the camera, VolPathIntegrator, 1920x1080
For each: (a) parity at full geometry size -- a 64x36-pixel window (Integrator "pixelbounds") rendered by the device and by
the CPU oracle built with correctly rounded libm: film and ray counters must be identical; (b) whole-frame device throughput
at the configuration's samples per pixel (--spp=N overrides).  One JSON line per config.  Run on the GPU box: python tools/config_standin.py [3] [4]"""
import importlib.util
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes"))
import gen_synthetic  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

FOG = ('MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [ 0.02 0.03 0.04 ] "rgb sigma_s" [ 0.15 0.12 0.1 ] "float g" [ 0.4 ]\n'
       'MediumInterface "" "fog"\n')


def run(config, spp):
    pkg = load_package()
    from oracle import oracle
    n = {3: 1582, 4: 2237}[config]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "standin.pbrt")
        t0 = time.perf_counter()
        gen_synthetic.write_scene(path, n=n, xres=1920, yres=1080, spp=spp, filename="standin.pfm")
        if config == 4:
            s = open(path).read()
            s = s.replace("Camera ", FOG + "Camera ", 1).replace('Integrator "path"', 'Integrator "volpath"', 1)
            s = s.replace("WorldBegin\n", 'WorldBegin\nMediumInterface "fog" "fog"\n', 1)
            open(path, "w").write(s)
        t1 = time.perf_counter()
        scene = pkg.HostScene(path)
        t2 = time.perf_counter()
    gs = pkg.GpuScene(scene.desc)
    t3 = time.perf_counter()
    # (a) parity on a window in the middle of the frame
    rd = scene.render_desc()
    rd.pixel_bounds[0], rd.pixel_bounds[1], rd.pixel_bounds[2], rd.pixel_bounds[3] = 928, 522, 992, 558
    film, strays = gs.render(rd)
    cn = gs.counters()
    ofilm, ostrays, ocn = oracle.render(scene.desc, rd)
    same_film = bool(np.array_equal(film["rgb"], ofilm["rgb"]) and np.array_equal(film["weight"], ofilm["weight"]))
    same_counts = all(cn[k] == ocn[k] for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests", "node_visits"))
    ref, _, _ = oracle.render(scene.desc, rd)  # glibc libm = the reference's arithmetic
    spp_eff = rd.spp
    err = np.abs(film["rgb"] - ref["rgb"]) / spp_eff / np.maximum(1.0, np.abs(ref["rgb"]) / spp_eff)
    # (b) whole frame
    gs.counters_reset()
    full = scene.render_desc()
    gs.render(full)  # warm-up: buffers, light tables
    gs.counters_reset()
    gs.render(full)
    c = gs.counters()
    rays = c["closest_rays"] + c["shadow_rays"]
    out = {"config": config, "triangles": int(scene.desc.n_tris), "bvh_nodes": int(scene.desc.n_nodes),
           "integrator": "volpath + HomogeneousMedium" if config == 4 else "path", "frame": "1920x1080", "spp": int(full.spp),
           "parity_window": "64x36 px, %d camera rays" % cn["camera_rays"], "film_identical_to_cr_oracle": same_film, "counters_identical": same_counts,
           "max_rel_err_vs_reference_arithmetic": float(err.max()),
           "Mrays_per_s": round(rays / (c["render_ms"] * 1e-3) / 1e6, 1), "Msamples_per_s": round(c["camera_rays"] / (c["render_ms"] * 1e-3) / 1e6, 2),
           "render_ms": round(c["render_ms"], 1), "rays_per_sample": round(rays / c["camera_rays"], 2),
           "host_generate_s": round(t1 - t0, 1), "host_parse_and_bvh_s": round(t2 - t1, 1), "upload_s": round(t3 - t2, 2)}
    gs.close()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    spp = None  # BASELINE.json: config 3 at 256 spp, config 4 at 128 spp
    for a in sys.argv[1:]:
        if a.startswith("--spp="): spp = int(a[6:])
    for c in ([int(a) for a in args] or [3, 4]):
        run(c, spp or {3: 256, 4: 128}[c])
