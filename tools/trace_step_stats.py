#!/usr/bin/env python3
"""Where the lanes of a k_trace wave go: wave steps and lanes per step kind (interior record, triangle, alpha lookup, instance entry / exit,
refill) and what the other lanes wait for meanwhile.  Needs a device library built with -DTR_STATS (tools/build_variant.sh stats "-DTR_STATS");
the product library has no such counters.

    PBRT_GPU_LIB=$PWD/gpurun_in_libpbrt_gpu_stats.so python tools/trace_step_stats.py --workload divergent --tris 5000000 --spp 16
"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

NAMES = ["iter", "int_steps", "int_lanes", "tri_steps", "tri_lanes", "alpha_steps", "alpha_lanes", "enter_steps", "enter_lanes", "exit_steps",
         "exit_lanes", "refill_steps", "refill_lanes", "wait_enter", "wait_exit", "idle", "inst_prim_lanes", "rays", "in_inst_int", "in_inst_tri",
         "alpha_wait"]


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="divergent")
    ap.add_argument("--grid", type=int, default=708)
    ap.add_argument("--tris", type=int, default=5000000)
    ap.add_argument("--xres", type=int, default=1920)
    ap.add_argument("--yres", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=16)
    ap.add_argument("--filter", default="box")
    args = ap.parse_args()
    args.gpus = 1
    pkg = bench.load_package()
    lib = C.CDLL(pkg.GPU_LIB_PATH)
    if not hasattr(lib, "pg_debug_trace_stats"):
        raise SystemExit("this library was not built with -DTR_STATS")
    ctx = bench.Context(args)
    n = lib.pg_debug_trace_stats(None, 1)
    m = bench.run_workload(ctx, args, steps=1, warmup=0)
    buf = (C.c_ulonglong * (3 * n))()
    lib.pg_debug_trace_stats(buf, 0)
    out = {"workload": bench.describe(args, m.scene), "counters": {k: int(m.cn[k]) for k in ("closest_rays", "shadow_rays", "node_visits", "tri_tests") if k in m.cn}}
    for kind, label in ((0, "k_trace<0> closest hit"), (1, "k_trace<1> any hit, reference order"), (2, "k_trace<2> any hit, free order")):
        v = dict(zip(NAMES, buf[kind * n:(kind + 1) * n]))
        if not v["iter"]:
            continue
        rays = max(1, v["refill_lanes"])
        d = {"rays": rays, "wave_iterations": v["iter"]}
        for step in ("int", "tri", "alpha", "enter", "exit", "refill"):
            s, l = v[step + "_steps"], v[step + "_lanes"]
            d[step] = {"steps": s, "lanes_per_step": round(l / s, 2) if s else 0, "per_ray": round(l / rays, 3), "share_of_iterations": round(s / v["iter"], 4)}
        d["lanes_per_iteration"] = {"waiting_for_entry": round(v["wait_enter"] / v["iter"], 2), "waiting_for_exit": round(v["wait_exit"] / v["iter"], 2),
                                    "idle": round(v["idle"] / v["iter"], 2)}
        d["inside_an_instance"] = {"of_interior_lanes": round(v["in_inst_int"] / max(1, v["int_lanes"]), 3), "of_triangle_lanes": round(v["in_inst_tri"] / max(1, v["tri_lanes"]), 3)}
        d["instance_prims_met_per_ray"] = round(v["inst_prim_lanes"] / rays, 3)
        out[label] = d
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
