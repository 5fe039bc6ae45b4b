#!/usr/bin/env python3
"""Times pg_hlbvh_build (device) against the host front end's HLBVHBuild on random triangle-sized bounds; prints one JSON
line per size.  Run on the GPU box: python tools/hlbvh_timing.py [n ...]"""
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("pbrt_v3_amd", os.path.join(ROOT, "pbrt-v3_amd", "__init__.py"), submodule_search_locations=[os.path.join(ROOT, "pbrt-v3_amd")])
pkg = importlib.util.module_from_spec(spec)
sys.modules["pbrt_v3_amd"] = pkg
spec.loader.exec_module(pkg)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [1_000_000, 10_000_000]
    rng = np.random.default_rng(5)
    pkg.hlbvh_build(np.zeros((4, 6), np.float32), 4, device=True)  # context + module load
    for n in sizes:
        c = (rng.random((n, 3)) * 100).astype(np.float32)
        h = (rng.random((n, 3)) * 0.05).astype(np.float32)
        b = np.concatenate([c - h, c + h], axis=1)
        t0 = time.perf_counter(); dn, do = pkg.hlbvh_build(b, 4, device=True); t1 = time.perf_counter()
        hn, ho = pkg.hlbvh_build(b, 4, device=False); t2 = time.perf_counter()
        print(json.dumps({"n_prims": n, "n_nodes": int(len(dn)), "device_s": round(t1 - t0, 4), "host_s": round(t2 - t1, 4),
                          "identical": bool(dn.tobytes() == hn.tobytes() and np.array_equal(do, ho))}))


if __name__ == "__main__":
    main()
