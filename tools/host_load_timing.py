import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "scenes"))
os.environ["PBRT_HOST_TIMING"] = "1"
import gen_synthetic
from __graft_entry__ import load_package
pkg = load_package()
t = time.time(); gen_synthetic.write_scene("/tmp/s5m.pbrt", n=1582, spp=1); print("scene file written in", round(time.time() - t, 1), "s", flush=True)
for k in range(2):
    t = time.time(); sc = pkg.HostScene("/tmp/s5m.pbrt"); print("HostScene 5M:", round(time.time() - t, 2), "s", sc.desc.n_tris, "tris", flush=True); del sc
