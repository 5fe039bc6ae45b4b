#!/usr/bin/env python3
"""Full-size parity of the device path against the UNMODIFIED reference binary (oracle/_ref/pbrt_oracle), on the GPU box.

  config 2: Cornell box, 512x512 @ 256 spp, whole frame
  config 3: synthetic 999 710-triangle scene, 1920x1080 @ 64 spp, whole frame
  config 4: the 5 M-triangle stand-in of scenes/gen_divergent.py (PLY meshes under > 100 object instances, image / alpha
            textures, eight-material palette, environment light), path, 1920x1080 @ 256 spp -- a 256x144 window of the full
            frame (Integrator "pixelbounds")
  config 5: the same at 10 M triangles inside a HomogeneousMedium, volpath, 1920x1080 @ 128 spp -- the same window
  config 40 / 50: round 2's stand-ins (config 3's matte heightfield scaled to 5 M / 10 M triangles; 50 in fog under volpath)

Both renderers read the same .pbrt file; the reference writes a PFM (core/imageio.cpp:437-482), the device film goes through
the host Film (MergeFilmTile + WriteImage arithmetic).  Reported per config: the share of bit-identical pixels (the bar: all of
them -- the device computes libm's float functions as the reference's glibc does, csrc/pg_libm.h), max / 99.99th percentile / count
of pixels with |d| > 1e-4 * max(1, |ref|) (BASELINE.json's tolerance: none), and the ray counters of both (the bar: equal).
TEST INFRASTRUCTURE: runs oracle/_ref/pbrt_oracle.  One JSON line per config.

usage: python tools/fullsize_parity.py [2] [3] [4] [5] [--out FILE.json]"""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes"))
import gen_synthetic  # noqa: E402
import gen_divergent  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402

TOL = 1e-4
FOG = ('MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [ 0.02 0.03 0.04 ] "rgb sigma_s" [ 0.15 0.12 0.1 ] "float g" [ 0.4 ]\n'
       'MediumInterface "" "fog"\n')
WINDOW = (832, 468, 1088, 612)  # 256x144 pixels in the middle of the 1920x1080 frame


def write_config(config, d):
    path = os.path.join(d, f"config{config}.pbrt")
    window = None
    if config == 2:
        open(path, "w").write(open(os.path.join(ROOT, "scenes", "cornell.pbrt")).read())
    elif config in (4, 5):
        dry = float(os.environ.get("PBRT_PARITY_DRY_RUN_SCALE", "1"))  # < 1: a CPU dry run of this code path under tests/emu
        gen_divergent.write_scene(path, tris=int({4: 5000000, 5: 10000000}[config] * dry), xres=1920, yres=1080, spp=max(1, int({4: 256, 5: 128}[config] * dry)),
                                  volumetric=config == 5, filename=f"config{config}.pfm")
        window = WINDOW
        s = re.sub(r'(Integrator "(?:vol)?path")', r'\1 "integer pixelbounds" [ %d %d %d %d ]' % (window[0], window[2], window[1], window[3]), open(path).read(), count=1)
        open(path, "w").write(s)
    else:
        n, spp = {3: (708, 64), 40: (1582, 256), 50: (2237, 128)}[config]
        gen_synthetic.write_scene(path, n=n, xres=1920, yres=1080, spp=spp, filename=f"config{config}.pfm")
        s = open(path).read()
        if config == 50:
            s = s.replace("Camera ", FOG + "Camera ", 1).replace('Integrator "path"', 'Integrator "volpath"', 1)
            s = s.replace("WorldBegin\n", 'WorldBegin\nMediumInterface "fog" "fog"\n', 1)
        if config >= 40:
            window = WINDOW
            s = re.sub(r'(Integrator "(?:vol)?path")', r'\1 "integer pixelbounds" [ %d %d %d %d ]' % (window[0], window[2], window[1], window[3]), s, count=1)
        open(path, "w").write(s)
    return path, window


def usable_cpus():
    """CPUs this process may use: the affinity mask, cut by the cgroup quota (the GPU box: 256 logical CPUs, a quota of 16 -- the
    reference with 256 threads on 16 CPUs' worth of time runs at a third of its 16-thread rate)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max": n = max(1, min(n, int(int(q) / int(per) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


def reference_render(path, out_pfm):
    ref = os.path.join(ROOT, "oracle", "_ref", "pbrt_oracle")
    t0 = time.time()
    out = subprocess.run([ref, "--nthreads", str(usable_cpus()), "--outfile", out_pfm, path], capture_output=True, text=True, check=True).stdout
    wall = time.time() - t0
    def stat(pat):  # a counter that stayed 0 is not printed (core/stats.cpp)
        m = re.search(pat + r"\s+(\d+)", out)
        return int(m.group(1)) if m else 0
    cn = {"camera_rays": stat(r"Camera rays traced"), "closest_rays": stat(r"Regular ray intersection tests"),
          "shadow_rays": stat(r"Shadow ray intersection tests")}
    m = re.search(r"Ray-triangle intersection tests\s+(\d+) /\s+(\d+)", out)
    if m:
        cn["tri_tests"] = int(m.group(2))
    secs = re.findall(r"\((\d+\.\d+)s\)", out)
    return cn, (float(secs[-1]) if secs else None), wall


def run(config):
    pkg = load_package()
    with tempfile.TemporaryDirectory() as d:
        path, window = write_config(config, d)
        scene = pkg.HostScene(path)
        gs = pkg.GpuScene(scene.desc)
        rd = scene.render_desc()
        film, strays = gs.render(rd)
        cn = gs.counters()
        scene.film_clear(); scene.film_merge(rd, film, strays)
        img = scene.film_image()
        ref_pfm = os.path.join(d, "ref.pfm")
        rcn, ref_render_s, ref_wall_s = reference_render(path, ref_pfm)
        ref = pkg.read_pfm(ref_pfm)
    assert ref.shape == img.shape, (ref.shape, img.shape)
    # "pixelbounds" restricts the pixels that are SAMPLED; both renderers write the whole frame, black outside the window but for
    # the rim a box-filter sample with offset 0 reaches (film.h:127-132).  The comparison is over the whole frame.
    x0 = y0 = 0
    img_w, ref_w = img, ref
    if window:
        wx0, wy0, wx1, wy1 = window
        mask = np.ones(img.shape[:2], bool)
        mask[max(0, wy0 - 1):wy1 + 1, max(0, wx0 - 1):wx1 + 1] = False
        outside_black = bool((img[mask] == 0).all() and (ref[mask] == 0).all())
    else:
        outside_black = True
    err = (np.abs(img_w - ref_w) / np.maximum(1.0, np.abs(ref_w))).max(axis=2)
    bad = np.argwhere(err > TOL)
    out = {"config": config, "triangles": int(max(scene.desc.n_tris, scene.desc.n_prims_all)), "object_instances": int(scene.desc.n_instances),
           "integrator": "volpath" if config in (5, 50) else "path",
           "frame": f"{img.shape[1]}x{img.shape[0]}", "spp": int(rd.spp), "compared_pixels": int(err.size),
           "sampled_pixels": int((window[2] - window[0]) * (window[3] - window[1])) if window else int(err.size),
           "window": list(window) if window else None, "outside_window_black": outside_black,
           "max_rel_err": float(err.max()), "p9999_rel_err": float(np.percentile(err, 99.99)), "pixels_over_tol": int(len(bad)), "tol": TOL,
           "pixels_differing": int((img_w.view(np.uint32) != ref_w.view(np.uint32)).any(axis=2).sum()), "bit_identical_pixel_share": float((img_w == ref_w).all(axis=2).mean()),
           "mean_ref": float(ref_w.mean()), "mean_device": float(img_w.mean()),
           "device_counters": {k: int(cn[k]) for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests")},
           "reference_counters": rcn,
           "counter_rel_delta": {k: (cn[k] - rcn[k]) / max(1, rcn[k]) for k in rcn},
           "device_render_ms": float(cn["render_ms"]), "reference_render_s": ref_render_s, "reference_wall_s": round(ref_wall_s, 1),
           "reference_threads": usable_cpus()}
    gs.close()
    return out


if __name__ == "__main__":
    outfile = None
    cfgs = []
    for a in sys.argv[1:]:
        if a.startswith("--out="): outfile = a[6:]
        else: cfgs.append(int(a))
    res = []
    for c in cfgs or [2, 3]:
        r = run(c)
        res.append(r)
        print(json.dumps(r), flush=True)
    if outfile:
        json.dump(res, open(outfile, "w"), indent=1)
