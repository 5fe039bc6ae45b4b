#!/usr/bin/env python3
"""Ad-hoc GPU-vs-oracle comparison (development aid; the formal checks live in tests/)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from oracle import oracle
sys.path.insert(0, os.path.join(ROOT, "scenes"))
import gen_synthetic

def cornell(res, spp):
    return (open(os.path.join(ROOT, "scenes", "cornell.pbrt")).read()
            .replace('"integer xresolution" [ 512 ]', f'"integer xresolution" [ {res} ]')
            .replace('"integer yresolution" [ 512 ]', f'"integer yresolution" [ {res} ]')
            .replace('"integer pixelsamples" [ 256 ]', f'"integer pixelsamples" [ {spp} ]'))

def compare(name, scene):
    t = time.time(); gs = pkg.GpuScene(scene.desc); tcreate = time.time() - t
    rd = scene.render_desc()
    t = time.time(); film, strays = gs.render(rd); tg = time.time() - t
    cg = gs.counters()
    t = time.time(); ofilm, ostrays, co = oracle.render(scene.desc, rd); to = time.time() - t
    d = np.abs(film["rgb"] - ofilm["rgb"])
    nbad = int((d.max(axis=1) > 0).sum())
    print(f"[{name}] create {tcreate:.2f}s gpu {tg:.3f}s oracle {to:.2f}s | film max|d| {d.max():.3e} differing px {nbad}/{len(film)} "
          f"weights equal {np.array_equal(film['weight'], ofilm['weight'])} strays gpu/oracle {len(strays)}/{len(ostrays)}")
    for k in ("camera_rays", "closest_rays", "shadow_rays", "node_visits", "tri_tests"):
        print(f"    {k}: gpu {cg[k]} oracle {co[k]}")
    print(f"    closest_ms {cg['closest_ms']:.3f} shadow_ms {cg['shadow_ms']:.3f} render_ms {cg['render_ms']:.3f}")
    scene.film_clear(); scene.film_merge(rd, film, strays); img = scene.film_image()
    scene.film_clear(); scene.film_merge(rd, ofilm, ostrays); oimg = scene.film_image()
    rel = np.abs(img - oimg) / np.maximum(1, np.abs(oimg))
    print(f"    image: max rel diff {rel.max():.3e}, 99.99pct {np.percentile(rel, 99.99):.3e}, px > 1e-4: {int((rel.max(axis=2) > 1e-4).sum())}")
    # ray-level: primary camera rays through pg_intersect vs oracle
    rng = np.random.default_rng(1)
    n = 20000
    nodes = scene.nodes()
    lo, hi = nodes["bmin"][0], nodes["bmax"][0]
    o = (lo + (hi - lo) * rng.random((n, 3))).astype(np.float32)
    dd = rng.normal(size=(n, 3)).astype(np.float32)
    tm = np.full(n, np.inf, np.float32)
    gp, gt, gb = gs.intersect(o, dd, tm)
    op, ot, ob, _ = oracle.intersect(scene.desc, o, dd, tm)
    print(f"    intersect: prim equal {np.array_equal(gp, op)} t equal {np.array_equal(gt, ot)} bary equal {np.array_equal(gb, ob)} hits {int((op>=0).sum())}")
    tm2 = rng.random(n).astype(np.float32) * 2
    go = gs.intersect_p(o, dd, tm2)
    oo, _ = oracle.intersect_p(scene.desc, o, dd, tm2)
    print(f"    intersect_p: equal {np.array_equal(go, oo)} occluded {int(oo.sum())}")
    gs.close()

if __name__ == "__main__":
    compare("cornell 64x64@16", pkg.HostScene(text=cornell(64, 16)))
    os.makedirs("/tmp/pgwork", exist_ok=True)
    gen_synthetic.write_scene("/tmp/pgwork/syn.pbrt", n=60, xres=96, yres=54, spp=8)
    compare("synthetic 7k 96x54@8", pkg.HostScene("/tmp/pgwork/syn.pbrt"))
    gen_synthetic.write_scene("/tmp/pgwork/syn1m.pbrt", n=708, xres=192, yres=108, spp=4)
    compare("synthetic 1M 192x108@4", pkg.HostScene("/tmp/pgwork/syn1m.pbrt"))
