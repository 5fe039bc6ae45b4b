#!/usr/bin/env python3
"""Headline benchmark: PathIntegrator on the synthetic ~1M-triangle scene, 1920x1080 @ 64 spp
(BASELINE.json configs[2]), film tiles sharded across the visible ranks, one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one full Integrator::Render of the frame on device-resident scene data: camera-ray
generation, every bounce's BVH traversal + shading, film accumulation and (N>1) the RCCL gather of
the per-rank film tiles to rank 0.  Scene parsing, BVH construction, upload and the final
Film::WriteImage are outside the timed region, as in the reference's own accounting
(SURVEY.md section 8d).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes"))
from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy ceiling
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth, 8 XCDs x 4 MiB (MI355X_MICROARCH.md "L2 (per XCD)")
INFINITY_CACHE_BYTES = 256 << 20
FOG = ('MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [ 0.02 0.03 0.04 ] "rgb sigma_s" [ 0.15 0.12 0.1 ] "float g" [ 0.4 ]\n'
       'MediumInterface "" "fog"\n')


def make_scene_file(workdir, args):
    import gen_synthetic
    path = os.path.join(workdir, "synthetic.pbrt")
    if args.workload == "cornell":
        txt = open(os.path.join(ROOT, "scenes", "cornell.pbrt")).read()
        txt = (txt.replace('"integer xresolution" [ 512 ]', f'"integer xresolution" [ {args.xres} ]')
                  .replace('"integer yresolution" [ 512 ]', f'"integer yresolution" [ {args.yres} ]')
                  .replace('"integer pixelsamples" [ 256 ]', f'"integer pixelsamples" [ {args.spp} ]'))
        open(path, "w").write(txt)
    else:
        # (PBRT_BENCH_MAXDEPTH: kernel diagnostics only; the benchmark is maxdepth 5)
        gen_synthetic.write_scene(path, n=args.grid, xres=args.xres, yres=args.yres, spp=args.spp,
                                  maxdepth=int(os.environ.get("PBRT_BENCH_MAXDEPTH", "5")))
        if args.workload == "synthetic-vol":  # BASELINE config 4's stand-in: the mesh inside a HomogeneousMedium, VolPathIntegrator
            txt = open(path).read()
            txt = txt.replace("Camera ", FOG + "Camera ", 1).replace('Integrator "path"', 'Integrator "volpath"', 1)
            txt = txt.replace("WorldBegin\n", 'WorldBegin\nMediumInterface "fog" "fog"\n', 1)
            open(path, "w").write(txt)
    if args.filter != "box":
        txt = open(path).read().replace('PixelFilter "box"', f'PixelFilter "{args.filter}"')
        open(path, "w").write(txt)
    return path


def cpu_baseline(workdir, args):
    """The unmodified reference (oracle/_ref/pbrt_oracle) on the host cores, bounded sample of the same workload."""
    ref = os.path.join(ROOT, "oracle", "_ref", "pbrt_oracle")
    cores = os.cpu_count() or 1
    model = "unknown CPU"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    # the whole frame at a quarter of the samples per pixel (the rate does not depend on spp): every one of the 8160 tiles
    # is rendered, so the reference's thread pool is loaded as in the full job; --cpu-full renders all samples
    xres, yres, spp = args.xres, args.yres, (args.spp if args.cpu_full else max(1, args.spp // 4))
    sample = f"same scene, {xres}x{yres} @ {spp} spp ({xres * yres * spp / 1e6:.2f} Msamples) on {model}"
    if os.path.exists(ref):
        src = open(os.path.join(workdir, "synthetic.pbrt")).read()
        src = re.sub(r'"integer xresolution" \[ \d+ \]', f'"integer xresolution" [ {xres} ]', src)
        src = re.sub(r'"integer yresolution" \[ \d+ \]', f'"integer yresolution" [ {yres} ]', src)
        src = re.sub(r'"integer pixelsamples" \[ \d+ \]', f'"integer pixelsamples" [ {spp} ]', src)
        small = os.path.join(workdir, "cpu_sample.pbrt")
        open(small, "w").write(src)
        try:
            out = subprocess.run([ref, "--nthreads", str(cores), "--outfile", os.path.join(workdir, "cpu.pfm"), small],
                                 capture_output=True, text=True, timeout=1500).stdout
            reg = int(re.search(r"Regular ray intersection tests\s+(\d+)", out).group(1))
            sh = int(re.search(r"Shadow ray intersection tests\s+(\d+)", out).group(1))
            secs = float(re.findall(r"\((\d+\.\d+)s\)", out)[-1])  # ProgressReporter's final elapsed time = time in Render()
            if secs > 0:
                return {"value": (reg + sh) / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "reference",
                        "sample": sample + f"; {reg + sh} rays in {secs:.1f} s of Integrator::Render()"}
        except Exception as e:  # fall through to the port
            sys.stderr.write(f"bench: reference CPU run failed ({e}); timing the C port instead\n")
    from oracle import oracle
    pkg = load_package()
    src = open(os.path.join(workdir, "synthetic.pbrt")).read()
    scene = pkg.HostScene(filename=os.path.join(workdir, "cpu_sample.pbrt")) if os.path.exists(os.path.join(workdir, "cpu_sample.pbrt")) \
        else pkg.HostScene(text=src)
    rd = scene.render_desc()
    t0 = time.time()
    _, _, cn = oracle.render(scene.desc, rd)
    secs = time.time() - t0
    return {"value": (cn["closest_rays"] + cn["shadow_rays"]) / secs / 1e6, "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": sample}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="synthetic", choices=["synthetic", "synthetic-vol", "cornell"],
                    help="synthetic: BASELINE config 3 (with --grid 1582 --spp 256: the 5 M-triangle stand-in of config 4); "
                         "synthetic-vol (--grid 2237 --spp 128): the 10 M-triangle volpath stand-in of config 5; cornell: config 2")
    ap.add_argument("--grid", type=int, default=708, help="heightfield vertices per side (708 -> 999 698 triangles)")
    ap.add_argument("--xres", type=int, default=1920)
    ap.add_argument("--yres", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--filter", default="box", help='PixelFilter of the scene (BASELINE config: "box")')
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline renders every sample of the frame (minutes)")
    ap.add_argument("--out", default=None, help="write the rendered image (PFM) here")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible (the HIP path has no CPU fallback)")
    # one rank per GPU; PBRT_BENCH_OVERSUBSCRIBE=1 lets several ranks share a device (a functional pre-flight of the N > 1
    # path on a single-GPU box -- RCCL permitting; never a measurement)
    if local_rank >= torch.cuda.device_count() and os.environ.get("PBRT_BENCH_OVERSUBSCRIBE") != "1":
        raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
    local_rank %= torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the single-GPU pre-flight of the N > 1 path
    # (PBRT_BENCH_OVERSUBSCRIBE=1) moves the film shards through gloo on host copies instead; a real run is RCCL over xGMI
    backend = "gloo" if os.environ.get("PBRT_BENCH_OVERSUBSCRIBE") == "1" else "nccl"
    comm_dev = dev if backend == "nccl" else torch.device("cpu")
    # PBRT_BENCH_FORCE_DIST=1: run the collective path with however many ranks there are -- with ONE rank on a single-GPU box
    # this is the only way to execute the RCCL transport (communicator, gather, all_reduce, barrier) before an 8-GPU run
    multi = world > 1 or os.environ.get("PBRT_BENCH_FORCE_DIST") == "1"
    if multi:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    pkg = load_package()
    workdir = tempfile.mkdtemp(prefix=f"pbrt_bench_r{rank}_")
    scene_file = make_scene_file(workdir, args)
    t0 = time.time()
    scene = pkg.HostScene(scene_file)
    t_parse = time.time() - t0
    gs = pkg.GpuScene(scene.desc, device=local_rank)
    rd = scene.render_desc(tile_first=rank, tile_step=world)
    n_tiles = gs.tile_count(rd)
    max_tiles = gs.tile_count(scene.render_desc(0, world))  # rank 0 owns the most
    from pbrt_v3_amd import distributed as pdist
    film, strays, nstrays, max_strays = pdist.shard_buffers(max_tiles, dev, rd.tile_pixels)
    to_comm = (lambda t: t) if backend == "nccl" else (lambda t: t.cpu())
    gathered = [pdist.gather_lists(to_comm(film), to_comm(strays), to_comm(nstrays)) if multi else None]

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        gs.render_device(rd, film.data_ptr(), strays.data_ptr(), max_strays, nstrays.data_ptr(), stream=stream)
        if multi:  # Film gather over xGMI: every rank's packed tile buffer to rank 0
            pdist.gather_film(to_comm(film), to_comm(strays), to_comm(nstrays), lists=gathered[0], dst=0)

    def sync():
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    gs.counters_reset()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    cn = gs.counters()
    stats = torch.tensor([elapsed, float(cn["closest_rays"] + cn["shadow_rays"]), float(cn["camera_rays"])], dtype=torch.float64, device=comm_dev)
    if multi:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        stats[0] = tmax[0]
    elapsed, rays, samples = (float(x) for x in stats.tolist())

    if rank == 0:
        # final image (outside the timed region): MergeFilmTile per shard + WriteImage arithmetic on the host
        img = None
        if args.out:
            if multi:
                shards = [(gathered[0][0][r], gathered[0][1][r], int(gathered[0][2][r].item())) for r in range(world)]
            else:
                shards = [(film, strays, int(nstrays.item()))]
            img = pdist.merge_shards(pkg, scene, gs.tile_count, shards)
            pkg.write_pfm(args.out, img)
        # Rooflines of the three hot kernels on rank 0.  Algorithmic bytes:
        #   k_trace: 32 B per reference node fetch + 48 B per triangle test + 32 B per ray in + 16 B (closest) / 4 B (any) out
        #   k_shade: per path vertex 16 B ray direction + 16 B hit + 48 B state in + 48 B triangle + 48 B state out + 48 B
        #            pending direct-light terms, + 32 B per ray pushed (+ 16 B of MIS terms per MIS ray)
        # Time: HIP events around every launch on the stream it runs on (pg_render), so each kernel is timed alone.
        # Bound: the BVH + triangle working set against the 256 MiB Infinity Cache -- below it the node/triangle gathers
        # are served on-die and the binding bandwidth is the L2's; above it they reach HBM.
        integ = "VolPathIntegrator + HomogeneousMedium" if args.workload == "synthetic-vol" else "PathIntegrator"
        workload = ((f"synthetic heightfield-in-a-box, {scene.desc.n_tris} triangles" if args.workload != "cornell"
                     else "Cornell box, 36 triangles") +
                    f", {integ} maxdepth 5, halton, {args.filter} filter, {args.xres}x{args.yres} @ {args.spp} spp")
        n_interior = max(0, (scene.desc.n_nodes - 1) // 2)
        working_set = 64 * n_interior + 64 * scene.desc.n_tris  # child-pair records + triangle records, one 64-B line each
        bound = "l2" if working_set < INFINITY_CACHE_BYTES else "hbm"
        peak = L2_PEAK_GBS if bound == "l2" else HBM_PEAK_GBS
        pmc_kernels = {}  # HBM-side bytes per launch from the committed PMC passes of this same workload (tools/pmc_traffic.sh)
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                pmc_kernels = json.load(open(pmc)).get("workloads", {}).get(workload, {}).get("kernels", {})
            except Exception:
                pass

        def kernel_roofline(name, tag, alg_bytes, ms, launches, units, unit_name):
            launches = max(1, launches)
            achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            traffic = next((v["hbm_bytes_per_launch"] for k, v in pmc_kernels.items() if k.startswith(tag)), None)
            r = {"kernel": name, "bound": bound, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                 "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes / launches, "avg_launch_ms": ms / launches,
                 "launches": launches, "total_ms": ms, f"bytes_per_{unit_name}": alg_bytes / max(1, units)}
            if traffic is not None and ms > 0:  # what the memory side of the L2 actually moved, against the HBM peak
                hb = traffic * launches / (ms * 1e-3) / 1e9
                r["hbm_side"] = {"achieved": hb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hb / HBM_PEAK_GBS}
            return r

        n_close, n_shadow, n_mis, n_items = cn["closest_rays"], cn["shadow_rays"], cn["mis_rays"], cn["shade_items"]
        n_next = max(0, n_close - cn["camera_rays"] - n_mis)
        kernels = [
            kernel_roofline("k_trace<false> (BVHAccel::Intersect + Triangle::Intersect)", "void k_trace<false",
                            32 * cn["closest_node_visits"] + 48 * cn["closest_tri_tests"] + 48 * n_close,
                            cn["closest_ms"], cn["closest_launches"], n_close, "ray"),
            kernel_roofline("k_trace<true> (BVHAccel::IntersectP + Triangle::IntersectP)", "void k_trace<true",
                            32 * cn["shadow_node_visits"] + 48 * cn["shadow_tri_tests"] + 36 * n_shadow,
                            cn["shadow_ms"], cn["shadow_launches"], n_shadow, "ray"),
            kernel_roofline("k_shade (PathIntegrator::Li loop body + EstimateDirect set-up)", "void k_shade",
                            224 * n_items + 32 * (n_next + n_shadow + n_mis) + 16 * n_mis,
                            cn["shade_ms"], cn["shade_launches"], n_items, "vertex"),
        ]
        # The traversal kernels are gathers of 64-B child-pair records and 48-B triangle records, one per lane and step.  What bounds
        # them is the chip's rate of random record fetches at this working-set size, measured live by tools/ubench_gather.hip
        # (pbrt-v3_amd/ubench_gather: every lane chases its own chain of records through a table of that size): DESIGN.md section 5.
        # Closest hit: an interior step counts two reference node visits (both children), the root one per ray, so
        # record fetches = (node visits - rays) / 2 + triangle tests.
        gather = None
        ub = os.path.join(ROOT, "pbrt-v3_amd", "ubench_gather")
        if os.path.exists(ub) and cn["closest_ms"] > 0:
            try:
                ws_mb = max(3, int(round(working_set / 2**20)))
                ceil = json.loads(subprocess.run([ub, "--json", "2", str(ws_mb)], capture_output=True, text=True, timeout=120).stdout)
                c_l2, c_ws = ceil["2"]["together"], ceil[str(ws_mb)]["together"]
                fetches = max(0, cn["closest_node_visits"] - n_close) / 2 + cn["closest_tri_tests"]
                rate = fetches / (cn["closest_ms"] * 1e-3)
                # A traversal is not a uniformly random walk: the top of the tree stays in the L2s.  With the kernel's measured L2 hit
                # rate h (committed PMC pass of this workload) the ceiling is the harmonic blend of the L2-resident rate and the rate
                # at the working set's size; without h only the L2-resident rate is a safe upper bound.
                h = next((v.get("l2_hit_rate") for k, v in pmc_kernels.items() if k.startswith("void k_trace<false")), None)
                ceiling = 1.0 / (h / c_l2 + (1.0 - h) / c_ws) if h is not None else c_l2
                gather = {"kernel": "k_trace<false>", "record_fetches_per_s": rate, "ceiling_records_per_s": ceiling, "frac": rate / ceiling,
                          "ceiling_kind": ("1 / (h / C(2 MiB) + (1 - h) / C(working set)), h = L2 hit rate of the kernel from profiles/pmc_traffic.json"
                                           if h is not None else "C(2 MiB): L2-resident table (no PMC pass of this workload committed: upper bound)"),
                          "l2_hit_rate": h, "table_MiB": ws_mb, "ceiling_l2_resident": c_l2, "ceiling_at_working_set": c_ws,
                          "note": "C(x) = random 64-B record fetches per second measured live by pbrt-v3_amd/ubench_gather: one chain per lane, table of x MiB"}
            except Exception as e:  # the measurement tool is optional; the bench line is not
                gather = {"error": str(e)}
        kernels = [k for k in kernels if k["total_ms"] > 0]
        kernels.sort(key=lambda k: -k["total_ms"])  # dominant = the most time, each kernel timed alone
        roofline = dict(kernels[0]) if kernels else {"kernel": None, "bound": bound, "achieved": 0.0, "peak": peak, "unit": "GB/s", "frac": 0.0, "traffic": None}
        roofline["working_set_bytes"] = working_set
        if gather is not None:
            roofline["gather"] = gather
        roofline["bound_reason"] = (f"BVH + triangle records {working_set / 2**20:.0f} MiB " +
                                    ("fit the 256 MiB Infinity Cache: gathers are served on-die, L2 bandwidth is the ceiling"
                                     if bound == "l2" else "exceed the 256 MiB Infinity Cache: gathers reach HBM"))
        other_ms = {k: cn[k] for k in ("resolve_ms", "generate_ms", "film_ms")}
        result = {
            "metric": "Mrays/s", "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "samples_per_s": samples / elapsed,
            "config": {"workload": workload,
                       "sharding": f"16x16 film tiles round-robin over {world} GPU(s), " + ("RCCL gather to rank 0" if backend == "nccl" else "PRE-FLIGHT: ranks share a GPU, gloo gather of host copies"),
                       "rays_per_sample": rays / max(1.0, samples), "host_parse_and_bvh_s": t_parse},
            "roofline": roofline,
            "roofline_kernels": kernels,
            "kernel_ms_per_step": {**{k["kernel"].split(" ")[0]: k["total_ms"] / args.steps for k in kernels},
                                   **{k[:-3]: v / args.steps for k, v in other_ms.items()}},
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(workdir, args)
        print(json.dumps(result), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
