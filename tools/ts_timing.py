#!/usr/bin/env python3
"""Frame time of the tile-serial samplers (one path per tile in flight; all pixels at once when "dimensions" covers every draw of a path)
next to the Halton sampler on the same scene and sample count: python tools/ts_timing.py [xres yres spp [fast]]  ->  one JSON line.  GPU box."""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes"))
import gen_synthetic  # noqa: E402
from __graft_entry__ import load_package  # noqa: E402


def main():
    xres, yres, spp = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (960, 540, 16)
    only_fast = len(sys.argv) >= 5 and sys.argv[4] == "fast"  # halton and the "dimensions 17" cases only (the serial forms take ~13 s each at 960x540 @ 16)
    pkg = load_package()
    out = {"frame": f"{xres}x{yres}", "spp": spp, "triangles": None, "samplers": {}}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "s.pbrt")
        gen_synthetic.write_scene(path, n=708, xres=xres, yres=yres, spp=spp)
        base = open(path).read()
        for name, spec in (("halton", f'"halton" "integer pixelsamples" [ {spp} ]'), ("random", f'"random" "integer pixelsamples" [ {spp} ]'),
                           ("stratified", f'"stratified" "integer xsamples" [ 4 ] "integer ysamples" [ {spp // 4} ]'),
                           ("02sequence", f'"02sequence" "integer pixelsamples" [ {spp} ]'), ("maxmindist", f'"maxmindist" "integer pixelsamples" [ {spp} ]'),
                           # "dimensions" covering every draw of a maxdepth-5 path (2 + 3 * 5): the arrays of all pixels are generated ahead, one wavefront
                           ("stratified, dimensions 17", f'"stratified" "integer xsamples" [ 4 ] "integer ysamples" [ {spp // 4} ] "integer dimensions" [ 17 ]'),
                           ("02sequence, dimensions 17", f'"02sequence" "integer pixelsamples" [ {spp} ] "integer dimensions" [ 17 ]'),
                           ("maxmindist, dimensions 17", f'"maxmindist" "integer pixelsamples" [ {spp} ] "integer dimensions" [ 17 ]')):
            if only_fast and name != "halton" and "dimensions" not in name: continue
            open(path, "w").write(base.replace(f'Sampler "halton" "integer pixelsamples" [ {spp} ]', "Sampler " + spec))
            scene = pkg.HostScene(path)
            out["triangles"] = int(scene.desc.n_tris)
            gs = pkg.GpuScene(scene.desc)
            rd = scene.render_desc()
            gs.render(rd)
            gs.counters_reset()
            gs.render(rd)
            c = gs.counters()
            rays = c["closest_rays"] + c["shadow_rays"]
            out["samplers"][name] = {"render_ms": round(c["render_ms"], 1), "Mrays_per_s": round(rays / c["render_ms"] / 1e3, 1)}
            gs.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
