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

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy ceiling


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
        gen_synthetic.write_scene(path, n=args.grid, xres=args.xres, yres=args.yres, spp=args.spp)
    if args.filter != "box":
        txt = open(path).read().replace('PixelFilter "box"', f'PixelFilter "{args.filter}"')
        open(path, "w").write(txt)
    return path


def cpu_baseline(workdir, args):
    """The unmodified reference (oracle/_ref/pbrt_oracle) on the host cores, bounded sample of the same workload."""
    ref = os.path.join(ROOT, "oracle", "_ref", "pbrt_oracle")
    cores = os.cpu_count() or 1
    xres, yres, spp = max(16, args.xres // 2), max(16, args.yres // 2), max(1, args.spp // 4)
    sample = f"same scene, {xres}x{yres} @ {spp} spp ({xres * yres * spp / 1e6:.2f} Msamples)"
    if os.path.exists(ref):
        src = open(os.path.join(workdir, "synthetic.pbrt")).read()
        src = re.sub(r'"integer xresolution" \[ \d+ \]', f'"integer xresolution" [ {xres} ]', src)
        src = re.sub(r'"integer yresolution" \[ \d+ \]', f'"integer yresolution" [ {yres} ]', src)
        src = re.sub(r'"integer pixelsamples" \[ \d+ \]', f'"integer pixelsamples" [ {spp} ]', src)
        small = os.path.join(workdir, "cpu_sample.pbrt")
        open(small, "w").write(src)
        try:
            out = subprocess.run([ref, "--nthreads", str(cores), "--outfile", os.path.join(workdir, "cpu.pfm"), small],
                                 capture_output=True, text=True, timeout=900).stdout
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
    ap.add_argument("--workload", default="synthetic", choices=["synthetic", "cornell"])
    ap.add_argument("--grid", type=int, default=708, help="heightfield vertices per side (708 -> 999 698 triangles)")
    ap.add_argument("--xres", type=int, default=1920)
    ap.add_argument("--yres", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=64)
    ap.add_argument("--filter", default="box", help='PixelFilter of the scene (BASELINE config: "box")')
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--out", default=None, help="write the rendered image (PFM) here")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py: no GPU visible (the HIP path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)  # "nccl" is RCCL on ROCm

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
    gathered = [pdist.gather_lists(film, strays, nstrays) if world > 1 else None]

    def step():
        stream = torch.cuda.current_stream().cuda_stream
        gs.render_device(rd, film.data_ptr(), strays.data_ptr(), max_strays, nstrays.data_ptr(), stream=stream)
        if world > 1:  # Film gather over xGMI: every rank's packed tile buffer to rank 0
            pdist.gather_film(film, strays, nstrays, lists=gathered[0], dst=0)

    def sync():
        if world > 1:
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
    stats = torch.tensor([elapsed, float(cn["closest_rays"] + cn["shadow_rays"]), float(cn["camera_rays"])], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        stats[0] = tmax[0]
    elapsed, rays, samples = (float(x) for x in stats.tolist())

    if rank == 0:
        # final image (outside the timed region): MergeFilmTile per shard + WriteImage arithmetic on the host
        img = None
        if args.out:
            if world > 1:
                shards = [(gathered[0][0][r], gathered[0][1][r], int(gathered[0][2][r].item())) for r in range(world)]
            else:
                shards = [(film, strays, int(nstrays.item()))]
            img = pdist.merge_shards(pkg, scene, gs.tile_count, shards)
            pkg.write_pfm(args.out, img)
        # roofline of the dominant kernel (k_trace<false>, BVHAccel::Intersect) on rank 0:
        # algorithmic bytes = 32 B/node fetch + 48 B/triangle test + 32 B/ray in + 16 B/hit out
        n_ray = cn["closest_rays"]
        alg_bytes = 32 * cn["closest_node_visits"] + 48 * cn["closest_tri_tests"] + 32 * n_ray + 16 * n_ray
        launches = max(1, cn["closest_launches"])
        achieved = alg_bytes / (cn["closest_ms"] * 1e-3) / 1e9 if cn["closest_ms"] > 0 else 0.0
        workload = ((f"synthetic heightfield-in-a-box, {scene.desc.n_tris} triangles" if args.workload == "synthetic"
                     else "Cornell box, 36 triangles") +
                    f", PathIntegrator maxdepth 5, halton, {args.filter} filter, {args.xres}x{args.yres} @ {args.spp} spp")
        traffic = None  # HBM-side bytes per launch from the committed PMC passes of this same workload (tools/pmc_traffic.sh)
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                j = json.load(open(pmc))
                if j.get("workload") == workload:
                    traffic = j.get("hbm_bytes_per_launch")
            except Exception:
                pass
        roofline = {"bound": "hbm", "kernel": "k_trace<false> (BVHAccel::Intersect + Triangle::Intersect)",
                    "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                    "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes / launches,
                    "avg_launch_ms": cn["closest_ms"] / launches, "launches": launches,
                    "bytes_per_ray": alg_bytes / max(1, n_ray)}
        result = {
            "metric": "Mrays/s", "value": rays / elapsed / 1e6, "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "samples_per_s": samples / elapsed,
            "config": {"workload": workload,
                       "sharding": f"16x16 film tiles round-robin over {world} GPU(s), RCCL gather to rank 0",
                       "rays_per_sample": rays / max(1.0, samples), "host_parse_and_bvh_s": t_parse},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(workdir, args)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
