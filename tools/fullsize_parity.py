#!/usr/bin/env python3
"""Full-size parity of the device path against the UNMODIFIED reference binary (oracle/_ref/pbrt_oracle), on the GPU box.

  config 2: Cornell box, 512x512 @ 256 spp, whole frame
  config 3: synthetic 999 710-triangle scene, 1920x1080 @ 64 spp, whole frame
  config 4: the 5 M-triangle stand-in of scenes/gen_divergent.py (PLY meshes under > 100 object instances, image / alpha
            textures, eight-material palette, environment light), path, 1920x1080 @ 256 spp -- a 256x144 window of the full
            frame (Integrator "pixelbounds")
  config 5: the same at 10 M triangles inside a HomogeneousMedium, volpath, 1920x1080 @ 128 spp -- the same window
  config 40 / 50: round 2's stand-ins (config 3's matte heightfield scaled to 5 M / 10 M triangles; 50 in fog under volpath)
  config 41 / 51: configs 4 / 5 over the WHOLE 1920x1080 frame at their own 256 / 128 spp (no window).  The reference takes
            tens of minutes of host time on them, so its image is rendered once where host time is free (--reference-only,
            --fingerprint-out: SHA-256 of the float image, a CRC-32 per 16x16 tile, the ray counters) and the device frame is
            compared with that fingerprint on the GPU box (--fingerprint-in): equal hashes = 0 differing pixels

Both renderers read the same .pbrt file; the reference writes a PFM (core/imageio.cpp:437-482), the device film goes through
the host Film (MergeFilmTile + WriteImage arithmetic).  Reported per config: the share of bit-identical pixels (the bar: all of
them -- the device computes libm's float functions as the reference's glibc does, csrc/pg_libm.h), max / 99.99th percentile / count
of pixels with |d| > 1e-4 * max(1, |ref|) (BASELINE.json's tolerance: none), and the ray counters of both (the bar: equal).
TEST INFRASTRUCTURE: runs oracle/_ref/pbrt_oracle.  One JSON line per config.

usage: python tools/fullsize_parity.py [2] [3] [4] [5] [--out FILE.json]"""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile
import time
import zlib

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
    elif config in (4, 5, 41, 51):
        full = config > 5
        config = config // 10 if full else config
        dry = float(os.environ.get("PBRT_PARITY_DRY_RUN_SCALE", "1"))  # < 1: a CPU dry run of this code path under tests/emu
        gen_divergent.write_scene(path, tris=int({4: 5000000, 5: 10000000}[config] * dry), xres=1920, yres=1080, spp=max(1, int({4: 256, 5: 128}[config] * dry)),
                                  volumetric=config == 5, filename=f"config{config}.pfm")
        if not full:
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


def fingerprint(img):
    """What identifies a float image exactly and locates a difference: SHA-256 of its bytes and a CRC-32 per 16x16 tile."""
    img = np.ascontiguousarray(img, dtype=np.float32)
    h, w = img.shape[:2]
    tiles = [zlib.crc32(np.ascontiguousarray(img[y:y + 16, x:x + 16]).tobytes()) for y in range(0, h, 16) for x in range(0, w, 16)]
    return {"shape": [int(h), int(w), int(img.shape[2])], "sha256": hashlib.sha256(img.tobytes()).hexdigest(), "tile_crc32": tiles, "mean": float(img.mean())}


def reference_fingerprint(config, keep_pfm=None):
    """The reference's image of one config as a fingerprint (no device involved): run where host time is free."""
    pkg = load_package()
    with tempfile.TemporaryDirectory() as d:
        path, window = write_config(config, d)
        ref_pfm = keep_pfm or os.path.join(d, "ref.pfm")
        rcn, ref_render_s, ref_wall_s = reference_render(path, ref_pfm)
        ref = pkg.read_pfm(ref_pfm)
        scene_sha = hashlib.sha256(open(path, "rb").read()).hexdigest()
    fp = fingerprint(ref)
    fp.update({"config": config, "scene_file_sha256": scene_sha, "reference_counters": rcn, "reference_render_s": ref_render_s, "reference_wall_s": round(ref_wall_s, 1),
               "reference_threads": usable_cpus()})
    return fp


def run(config, fp_in=None):
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
        if fp_in is not None:
            # the reference's image is not here, its fingerprint is: equal SHA-256 = every pixel identical, bit for bit
            scene_sha = hashlib.sha256(open(path, "rb").read()).hexdigest()
            fp = fingerprint(img)
            ref_tiles, dev_tiles = np.array(fp_in["tile_crc32"], dtype=np.uint32), np.array(fp["tile_crc32"], dtype=np.uint32)
            rcn = fp_in["reference_counters"]
            out = {"config": config, "triangles": int(max(scene.desc.n_tris, scene.desc.n_prims_all)), "object_instances": int(scene.desc.n_instances),
                   "integrator": "volpath" if config in (5, 50, 51) else "path", "frame": f"{img.shape[1]}x{img.shape[0]}", "spp": int(rd.spp),
                   "compared_pixels": int(img.shape[0] * img.shape[1]), "window": None,
                   "same_scene_file": scene_sha == fp_in["scene_file_sha256"], "sha256_equal": fp["sha256"] == fp_in["sha256"],
                   "tiles_differing": int((ref_tiles != dev_tiles).sum()) if ref_tiles.shape == dev_tiles.shape else -1, "tiles": int(dev_tiles.size),
                   "pixels_differing": 0 if fp["sha256"] == fp_in["sha256"] else None,
                   "mean_ref": fp_in["mean"], "mean_device": fp["mean"],
                   "device_counters": {k: int(cn[k]) for k in ("camera_rays", "closest_rays", "shadow_rays", "tri_tests")}, "reference_counters": rcn,
                   "counter_rel_delta": {k: (cn[k] - rcn[k]) / max(1, rcn[k]) for k in rcn},
                   "device_render_ms": float(cn["render_ms"]), "reference_render_s": fp_in["reference_render_s"], "reference_threads": fp_in["reference_threads"],
                   "reference": "fingerprint of the reference binary's image, rendered beforehand (tools/fullsize_parity.py --reference-only)"}
            gs.close()
            return out
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
           "integrator": "volpath" if config in (5, 50, 51) else "path",
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
    outfile = fp_out = fp_in = None
    ref_only = False
    cfgs = []
    for a in sys.argv[1:]:
        if a.startswith("--out="): outfile = a[6:]
        elif a == "--reference-only": ref_only = True
        elif a.startswith("--fingerprint-out="): fp_out = a[18:]   # a file name pattern with {config}
        elif a.startswith("--fingerprint-in="): fp_in = a[17:]
        else: cfgs.append(int(a))
    res = []
    for c in cfgs or [2, 3]:
        if ref_only:
            r = reference_fingerprint(c)
            json.dump(r, open((fp_out or "reference_fingerprint_config{config}.json").format(config=c), "w"))
            r = {k: v for k, v in r.items() if k != "tile_crc32"}
        else:
            r = run(c, json.load(open(fp_in.format(config=c))) if fp_in else None)
        res.append(r)
        print(json.dumps(r), flush=True)
    if outfile:
        json.dump(res, open(outfile, "w"), indent=1)
