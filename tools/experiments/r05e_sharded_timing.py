"""Where does tests/test_gpu_fullsize.py::test_native_sharded_render_equals_single_device spend its time?  (r05d: 336 s)"""
import os, sys, time, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pkg = load_package()
LARGE = os.path.join(ROOT, "tests", "golden_large")
t = time.time()
def lap(what):
    global t
    print(f"{what}: {time.time() - t:.2f} s", flush=True); t = time.time()
scene = pkg.HostScene(os.path.join(LARGE, "cornell_128.pbrt"))
whole, _ = pkg.render_scene(scene); lap("single render")
rd = scene.render_desc()
scenes = [pkg.GpuScene(scene.desc, device=0) for _ in range(3)]; lap("3 scenes")
shards = pkg.render_sharded(scenes, rd); lap("render_sharded x3 (peer)")
shards = pkg.render_sharded(scenes, rd); lap("render_sharded x3 again")
os.environ["PG_SHARD_GATHER"] = "rccl"
pkg.render_sharded(scenes[:1], rd); lap("render_sharded one rank rccl (first: comm init)")
pkg.render_sharded(scenes[:1], rd); lap("render_sharded one rank rccl again")
del os.environ["PG_SHARD_GATHER"]
exe = os.path.join(ROOT, "pbrt-v3_amd", "pbrt_amd")
for flags in (["--gpu", "0"], ["--gpu-ids", "0,0"]):
    subprocess.run([exe, "--quiet", *flags, "--outfile", "/tmp/x.pfm", os.path.join(LARGE, "cornell_128.pbrt")], check=True); lap("cli " + " ".join(flags))
