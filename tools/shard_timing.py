#!/usr/bin/env python3
"""What one rank of an N-GPU run has to do, timed on ONE GPU: the frame time of shard 0 of N (tiles t = 0 mod N of BASELINE
config 3) for N = 1, 2, 4, 8, per kernel, and the speed-up an N-GPU run can reach before the gather (t(1) / t(N)).
python tools/shard_timing.py [spp]  ->  one JSON line.  GPU box."""
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
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    pkg = load_package()
    out = {"workload": f"config 3, 1920x1080 @ {spp} spp", "shards": {}}
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "s.pbrt")
        gen_synthetic.write_scene(path, n=708, xres=1920, yres=1080, spp=spp)
        scene = pkg.HostScene(path)
        gs = pkg.GpuScene(scene.desc)
        if os.environ.get("SHARD_OVERLAP") == "1":  # each any-hit launch beside the next closest-hit launch, as every bench.py run does
            gs.set_option(pkg.abi.PG_OPT_OVERLAP_SHADOW, 1)
            out["overlap"] = True
        t1 = None
        for n in (1, 2, 4, 8):
            rd = scene.render_desc(tile_first=0, tile_step=n)
            gs.render(rd)
            best = None
            for _ in range(3):
                gs.counters_reset()
                gs.render(rd)
                c = gs.counters()
                if best is None or c["render_ms"] < best["render_ms"]:
                    best = c
            if n == 1:
                t1 = best["render_ms"]
            kernels = {k: round(best[k], 2) for k in ("closest_ms", "shadow_ms", "shade_ms", "resolve_ms", "generate_ms", "film_ms") if k in best}
            out["shards"][str(n)] = {"tiles": int(gs.tile_count(rd)), "render_ms": round(best["render_ms"], 2), "kernels_ms": kernels,
                                     "launch_and_sync_ms": round(best["render_ms"] - sum(kernels.values()), 2),
                                     "speedup_bound": round(t1 / best["render_ms"], 2)}
        gs.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
