import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from __graft_entry__ import load_package
pkg = load_package()
from oracle import oracle
from kat_util import jittered_sphere
from test_reference_kats import sphere_scene, watertight_rays
verts, idx = jittered_sphere()
scene = pkg.HostScene(text=sphere_scene(verts, idx))
gs = pkg.GpuScene(scene.desc)
o, d = watertight_rays(verts, 20000)
inf = np.full(len(o), np.inf, np.float32)
prim, t, bary = gs.intersect(o, d, inf)
oprim, ot, obary, ocn = oracle.intersect(scene.desc, o, d, inf)
bad = np.nonzero((prim != oprim) | (t != ot) | (bary != obary).any(axis=1))[0]
print("n_nodes", scene.desc.n_nodes, "n_tris", scene.desc.n_tris, "mismatches", len(bad), "counters", gs.counters()["closest_node_visits"], ocn["node_visits"], gs.counters()["closest_tri_tests"], ocn["tri_tests"])
for i in bad[:10]:
    print(i, "gpu", prim[i], t[i], bary[i], "oracle", oprim[i], ot[i], obary[i], "o", o[i], "d", d[i])
