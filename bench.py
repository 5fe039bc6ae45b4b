#!/usr/bin/env python3
"""Headline benchmark: PathIntegrator on the synthetic ~1M-triangle scene, 1920x1080 @ 64 spp
(BASELINE.json configs[2]), film tiles sharded across the ranks, one process per GPU.

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no launcher around it (WORLD_SIZE unset) the script starts its own N ranks
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...`); under
torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.

A "step" is one full Integrator::Render of the frame on device-resident scene data: camera-ray
generation, every bounce's BVH traversal + shading, film accumulation and (N>1) ONE RCCL gather of
the rank's packed film shard to rank 0, enqueued behind the render and overlapped with the next frame
(two shard buffers alternate).  Scene parsing, BVH construction, upload and the final Film::WriteImage
are outside the timed region, as in the reference's own accounting.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import re
import socket
import subprocess
import sys
import tempfile
import time
import types

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes"))
from __graft_entry__ import load_package  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E datasheet peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy ceiling
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 2  # MI355X: 256 CUs x 4 SIMD-32 units, a 64-wide vector instruction issues over 2 cycles (MI355X_MICROARCH.md "Wave scheduling"), 2.4 GHz: 1.23e12 per second
L2_PEAK_GBS = 34500.0   # aggregate L2 bandwidth, 8 XCDs x 4 MiB (MI355X_MICROARCH.md "L2 (per XCD)")
INFINITY_CACHE_BYTES = 256 << 20
FOG = ('MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [ 0.02 0.03 0.04 ] "rgb sigma_s" [ 0.15 0.12 0.1 ] "float g" [ 0.4 ]\n'
       'MediumInterface "" "fog"\n')
OPTIONAL_PMC = ("GRBM_TA_BUSY", "TCP_PENDING_STALL_CYCLES_sum")
EMULATED = os.environ.get("PBRT_EMULATED_DEVICE") == "1"  # tests/emu: the device library compiled for the host (a functional check, never a measurement)


def make_scene_file(workdir, args):
    import gen_synthetic
    path = os.path.join(workdir, "synthetic.pbrt")
    if args.workload == "config0":  # BASELINE config 0: the reference's own scenes/killeroo-simple.pbrt (fixture: tests/golden_large/config0)
        import shutil
        src = os.path.join(ROOT, "tests", "golden_large", "config0")
        os.makedirs(os.path.join(workdir, "geometry"), exist_ok=True)
        shutil.copy(os.path.join(src, "geometry", "killeroo.pbrt"), os.path.join(workdir, "geometry", "killeroo.pbrt"))
        txt = open(os.path.join(src, "config0.pbrt")).read()
        txt = (txt.replace('"integer xresolution" [400]', f'"integer xresolution" [ {args.xres} ]').replace('"integer yresolution" [400]', f'"integer yresolution" [ {args.yres} ]')
                  .replace('"integer pixelsamples" [8]', f'"integer pixelsamples" [ {args.spp} ]'))
        open(path, "w").write(txt)
    elif args.workload == "cornell":
        txt = open(os.path.join(ROOT, "scenes", "cornell.pbrt")).read()
        txt = (txt.replace('"integer xresolution" [ 512 ]', f'"integer xresolution" [ {args.xres} ]')
                  .replace('"integer yresolution" [ 512 ]', f'"integer yresolution" [ {args.yres} ]')
                  .replace('"integer pixelsamples" [ 256 ]', f'"integer pixelsamples" [ {args.spp} ]'))
        open(path, "w").write(txt)
    elif args.workload in ("divergent", "divergent-vol"):
        import gen_divergent  # BASELINE configs 4 / 5 stand-ins: instanced PLY meshes, textures, alpha masks, a material palette
        gen_divergent.write_scene(path, tris=args.tris, xres=args.xres, yres=args.yres, spp=args.spp, volumetric=args.workload.endswith("-vol"))
    else:
        # (PBRT_BENCH_MAXDEPTH: kernel diagnostics only; the benchmark is maxdepth 5)
        gen_synthetic.write_scene(path, n=args.grid, xres=args.xres, yres=args.yres, spp=args.spp,
                                  maxdepth=int(os.environ.get("PBRT_BENCH_MAXDEPTH", "5")))
        if args.workload == "synthetic-vol":  # the mesh inside a HomogeneousMedium, VolPathIntegrator
            txt = open(path).read()
            txt = txt.replace("Camera ", FOG + "Camera ", 1).replace('Integrator "path"', 'Integrator "volpath"', 1)
            txt = txt.replace("WorldBegin\n", 'WorldBegin\nMediumInterface "fog" "fog"\n', 1)
            open(path, "w").write(txt)
    if args.filter != "box":
        txt = open(path).read().replace('PixelFilter "box"', f'PixelFilter "{args.filter}"')
        open(path, "w").write(txt)
    return path


def host_cpus():
    """What this process may actually use: logical CPUs, its affinity mask, the cgroup CPU quota (v2 cpu.max / v1 cfs_quota_us)."""
    info = {"logical": os.cpu_count() or 1, "affinity": len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else None, "cgroup_quota_cpus": None,
            "model": "unknown CPU"}
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                info["model"] = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max": info["cgroup_quota_cpus"] = round(int(q) / int(per), 2)
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0: info["cgroup_quota_cpus"] = round(q / per, 2)
        except (OSError, ValueError):
            pass
    usable = info["affinity"] or info["logical"]
    if info["cgroup_quota_cpus"]: usable = max(1, min(usable, int(info["cgroup_quota_cpus"] + 0.5)))
    info["usable"] = usable
    return info


def cpu_baseline(workdir, args):
    """The unmodified reference (oracle/_ref/pbrt_oracle) on the host cores, bounded samples of the same workload: a thread sweep
    (1 / 16 / 64 / all usable CPUs, a few seconds each: the rate per thread and how it scales can be read from the line), then the
    whole frame at a quarter of the samples per pixel with the best thread count, which is `value`."""
    ref = os.path.join(ROOT, "oracle", "_ref", "pbrt_oracle")
    cpus = host_cpus()
    usable, model = cpus["usable"], cpus["model"]
    src0 = open(os.path.join(workdir, "synthetic.pbrt")).read()

    def reference_run(nthreads, xres, yres, spp, timeout):
        src = re.sub(r'"integer xresolution" \[ ?\d+ ?\]', f'"integer xresolution" [ {xres} ]', src0)
        src = re.sub(r'"integer yresolution" \[ ?\d+ ?\]', f'"integer yresolution" [ {yres} ]', src)
        src = re.sub(r'"integer pixelsamples" \[ ?\d+ ?\]', f'"integer pixelsamples" [ {spp} ]', src)
        small = os.path.join(workdir, "cpu_sample.pbrt")
        open(small, "w").write(src)
        out = subprocess.run([ref, "--nthreads", str(nthreads), "--outfile", os.path.join(workdir, "cpu.pfm"), small],
                             capture_output=True, text=True, timeout=timeout).stdout
        reg = int(re.search(r"Regular ray intersection tests\s+(\d+)", out).group(1))
        m = re.search(r"Shadow ray intersection tests\s+(\d+)", out)
        sh = int(m.group(1)) if m else 0
        secs = float(re.findall(r"\((\d+\.\d+)s\)", out)[-1])  # ProgressReporter's final elapsed time = time in Render()
        return reg + sh, secs

    # the whole frame at a quarter of the samples per pixel (the rate does not depend on spp): every one of the 8160 tiles
    # is rendered, so the reference's thread pool is loaded as in the full job; --cpu-full renders all samples
    xres, yres, spp = args.xres, args.yres, (args.spp if args.cpu_full else max(1, args.spp // 4))
    if os.path.exists(ref):
        try:
            sweep = {}
            counts = sorted({n for n in (1, 16, 64, usable) if n <= usable})
            for n in counts:
                # sized for seconds: one thread gets a quarter-size frame at 1 spp, the others the whole frame at 1 spp (<= 16 threads) or 4
                sx, sy, sspp = (max(64, xres // 2), max(64, yres // 2), 1) if n == 1 else (xres, yres, 1 if n <= 16 else min(4, spp))
                rays, secs = reference_run(n, sx, sy, sspp, 600)
                if secs > 0: sweep[n] = {"mrays_per_s": round(rays / secs / 1e6, 3), "per_thread": round(rays / secs / 1e6 / n, 4), "sample": f"{sx}x{sy} @ {sspp} spp, {secs:.1f} s"}
            best = max(sweep, key=lambda n: sweep[n]["mrays_per_s"]) if sweep else usable
            rays, secs = reference_run(best, xres, yres, spp, 1500)
            if secs > 0:
                return {"value": rays / secs / 1e6, "unit": "Mrays/s", "cores": best, "kind": "reference",
                        "host": {"cpu_model": model, "logical_cpus": cpus["logical"], "affinity_cpus": cpus["affinity"], "cgroup_quota_cpus": cpus["cgroup_quota_cpus"]},
                        "thread_sweep": {str(n): v for n, v in sweep.items()},
                        "sample": f"same scene, {xres}x{yres} @ {spp} spp ({xres * yres * spp / 1e6:.2f} Msamples) on {model}, --nthreads {best} (the fastest of the sweep; "
                                  f"`cores` = threads used); {rays} rays in {secs:.1f} s of Integrator::Render()"}
        except Exception as e:  # fall through to the port
            sys.stderr.write(f"bench: reference CPU run failed ({e}); timing the C port instead\n")
    from oracle import oracle
    pkg = load_package()
    src = re.sub(r'"integer pixelsamples" \[ ?\d+ ?\]', f'"integer pixelsamples" [ {spp} ]', src0)
    open(os.path.join(workdir, "cpu_sample.pbrt"), "w").write(src)
    scene = pkg.HostScene(filename=os.path.join(workdir, "cpu_sample.pbrt"))
    rd = scene.render_desc()
    t0 = time.time()
    _, _, cn = oracle.render(scene.desc, rd)
    secs = time.time() - t0
    return {"value": (cn["closest_rays"] + cn["shadow_rays"]) / secs / 1e6, "unit": "Mrays/s", "cores": usable, "kind": "port",
            "sample": f"same scene, {xres}x{yres} @ {spp} spp on {model} (OpenMP, {usable} threads)"}


def self_launch(args):
    """`python bench.py --gpus N` with no launcher: start N ranks of this same command under torch.distributed.run."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), PBRT_BENCH_SELF_LAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd, env=env))


class Context:
    """Rank, device and transport of this process."""

    def __init__(self, args):
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={self.world}")
        self.oversubscribed = os.environ.get("PBRT_BENCH_OVERSUBSCRIBE") == "1"
        if EMULATED:
            self.dev, self.device_index = torch.device("cpu"), 0
        else:
            if not torch.cuda.is_available():
                raise SystemExit("bench.py: no GPU visible (the HIP path has no CPU fallback)")
            # one rank per GPU; PBRT_BENCH_OVERSUBSCRIBE=1 lets several ranks share a device (a functional pre-flight of the
            # N > 1 path on a single-GPU box; never a measurement)
            if local_rank >= torch.cuda.device_count() and not self.oversubscribed:
                raise SystemExit(f"bench.py: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible")
            self.device_index = local_rank % torch.cuda.device_count()
            torch.cuda.set_device(self.device_index)
            self.dev = torch.device("cuda", self.device_index)
        # RCCL refuses two ranks on one device ("Duplicate GPU detected"), so the single-GPU pre-flight moves host copies of the
        # shards through gloo; so does the emulated device.  A real run is RCCL over xGMI ("nccl" is RCCL on ROCm).
        self.backend = "gloo" if (EMULATED or self.oversubscribed) else "nccl"
        self.comm_dev = self.dev if self.backend == "nccl" else torch.device("cpu")
        # PBRT_BENCH_FORCE_DIST=1: run the collective path with ONE rank (the only way to execute the RCCL transport on a 1-GPU box)
        self.multi = self.world > 1 or os.environ.get("PBRT_BENCH_FORCE_DIST") == "1"
        if self.multi:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group("gloo")

    def device_sync(self):
        if not EMULATED:
            torch.cuda.synchronize()

    def sync(self):
        """barrier + torch.cuda.synchronize(): both sides of the timed region."""
        self.device_sync()
        if self.multi:
            dist.barrier()
        self.device_sync()


def run_workload(ctx, args, steps, warmup, keep_image=False, serial_frame=False):
    """W untimed + K timed frames of one workload on this rank's tile shard.  Returns the measurements (and, on rank 0, the image)."""
    pkg = load_package()
    from pbrt_v3_amd import distributed as pdist
    workdir = tempfile.mkdtemp(prefix=f"pbrt_bench_r{ctx.rank}_")
    scene_file = make_scene_file(workdir, args)
    t0 = time.time()
    scene = pkg.HostScene(scene_file)
    t_parse = time.time() - t0
    gs = pkg.GpuScene(scene.desc, device=ctx.device_index)
    gs.set_option(pkg.abi.PG_OPT_OVERLAP_SHADOW, 1 if getattr(args, "overlap", False) else 0)
    rd = scene.render_desc(tile_first=ctx.rank, tile_step=ctx.world)
    max_tiles = gs.tile_count(scene.render_desc(0, ctx.world))  # rank 0 owns the most
    # two shard buffers alternate, so that frame i's gather overlaps frame i+1's render (N = 1: one buffer, no gather)
    bufs = [pdist.ShardBuffer(max_tiles, ctx.dev, rd.tile_pixels) for _ in range(2 if ctx.multi else 1)]
    gather = pdist.FilmGather(bufs[0], dst=0, comm_device=ctx.comm_dev) if ctx.multi else None
    frame = [0]

    def step():
        b = bufs[frame[0] % len(bufs)]
        frame[0] += 1
        stream = None if EMULATED else torch.cuda.current_stream().cuda_stream
        gs.render_device(rd, b.film.data_ptr(), b.strays.data_ptr(), b.max_strays, b.nstrays.data_ptr(), stream=stream)
        if gather:  # Film gather over xGMI: the packed shard to rank 0, one collective, behind the render on this stream
            gather.start(b, async_op=True)
        return b

    last = None
    for _ in range(warmup):
        last = step()
    if gather:
        gather.wait()
    gs.counters_reset()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        last = step()
    if gather:
        gather.wait()
    ctx.device_sync()
    local = time.perf_counter() - t0  # this rank's own frames + gathers done
    ctx.sync()
    elapsed = time.perf_counter() - t0
    cn = gs.counters()
    # Per-kernel times (HIP events around every launch) only mean something while kernels do not share the chip.  The timed frames of a
    # one-GPU run overlap each any-hit launch with the next closest-hit launch (PG_OVERLAP_SHADOW=1: +1.2 % Mrays/s,
    # profiles/r04b_bench_overlap*.json); the figures the rooflines are made of come from ONE more frame, serialised, after the timed region.
    serial = None
    if serial_frame and steps > 0:
        gs.set_option(pkg.abi.PG_OPT_OVERLAP_SHADOW, 0)
        gs.counters_reset()
        ctx.device_sync()
        ts = time.perf_counter()
        step()
        ctx.device_sync()
        serial = types.SimpleNamespace(ms=(time.perf_counter() - ts) * 1e3, cn=gs.counters())
        gs.set_option(pkg.abi.PG_OPT_OVERLAP_SHADOW, 1)
    stats = torch.tensor([elapsed, float(cn["closest_rays"] + cn["shadow_rays"]), float(cn["camera_rays"])], dtype=torch.float64, device=ctx.comm_dev)
    per_rank = [local]
    if ctx.multi:
        tmax = stats[:1].clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        stats[0] = tmax[0]
        mine = torch.tensor([local], dtype=torch.float64, device=ctx.comm_dev)
        every = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)
        per_rank = [float(t.item()) for t in every]
    # every rank's kernel times (its own HIP events): which rank is the straggler, and in which kernel
    kn = ("closest_ms", "shadow_ms", "shade_ms", "resolve_ms", "generate_ms", "film_ms")
    per_rank_kernels = [[float(cn[k]) / max(1, steps) for k in kn]]
    if ctx.multi:
        mine_k = torch.tensor(per_rank_kernels[0], dtype=torch.float64, device=ctx.comm_dev)
        every_k = [torch.zeros_like(mine_k) for _ in range(dist.get_world_size())]
        dist.all_gather(every_k, mine_k)
        per_rank_kernels = [[float(x) for x in t.tolist()] for t in every_k]
    elapsed, rays, samples = (float(x) for x in stats.tolist())
    m = types.SimpleNamespace(elapsed=elapsed, rays=rays, samples=samples, cn=cn, scene=scene, gs=gs, t_parse=t_parse, workdir=workdir, serial=serial,
                              per_rank_ms=[t / max(1, steps) * 1e3 for t in per_rank], steps=steps, image=None,
                              per_rank_kernel_ms=[{k[:-3]: round(v, 3) for k, v in zip(kn, row)} for row in per_rank_kernels])
    if ctx.rank == 0 and keep_image and last is not None:  # final image (outside the timed region): MergeFilmTile per shard + WriteImage arithmetic
        shards = gather.shards(last) if gather else [(last.film, last.strays, int(last.nstrays.item()))]
        m.image = pdist.merge_shards(pkg, scene, gs.tile_count, shards)
    return m


def shading_mode(cn):
    """PgCounters.shading_modes in words: which k_shade<MODE> the timed frames ran -- and whether a textured scene ran the slower k_shade<2> only
    because k_material's lists found no room in device memory (the silent cliff of pg_abi.hip's ensureWorkBuffers)."""
    bits = int(cn.get("shading_modes", 0))
    names = {0: "k_shade<0> (baked-in BxDF shapes)", 1: "k_shade<1> (BxDF lists)", 3: "k_material lists + k_shade<3>", 2: "k_shade<2> (material evaluators inside the shading kernel)"}
    ran = [names[m] for m in (0, 1, 3, 2) if bits >> m & 1]
    out = " + ".join(ran) if ran else "none"
    if bits & 0x200: out += " -- FALLBACK: k_material's lists did not fit in device memory"
    return out


def describe(args, scene):
    integ = "VolPathIntegrator + HomogeneousMedium" if args.workload.endswith("-vol") else "PathIntegrator"
    what = {"cornell": "Cornell box, 36 triangles",
            "config0": "scenes/killeroo-simple.pbrt of the reference (66 532 Loop-subdivided triangles, plastic + matte, one sphere area light)"}.get(args.workload)
    if what is None:
        d = scene.desc
        if args.workload.startswith("divergent"):
            unique = max(d.n_tris, d.n_prims_all)
            inst = sum(d.objects[d.instances[i].object].n_prims for i in range(d.n_instances))
            what = (("stand-in for BASELINE config 5 (San Miguel + medium: not in the reference mount): " if args.workload.endswith("-vol") else
                     "stand-in for BASELINE config 4 (crown: not in the reference mount): ") +
                    f"PLY meshes under {d.n_instances} object instances ({d.n_tris - d.n_instances + inst} triangles after instancing, {unique - d.n_instances} unique), "
                    "image / bump / alpha-mask textures, 8-material palette, environment + area light")
        else:
            what = f"synthetic heightfield-in-a-box, {d.n_tris} triangles"
    return f"{what}, {integ} maxdepth 5, halton, {args.filter} filter, {args.xres}x{args.yres} @ {args.spp} spp"


def live_pmc(bench_args, passes, timeout=300):
    """Hardware counters of ONE frame of this workload, collected by THIS run: one child `rocprofv3 --pmc <counters> -- python bench.py
    <same workload> --steps 1` per pass -- FETCH_SIZE and WRITE_SIZE in passes of their own, no trace domain beside them, as
    MI355X_MICROARCH.md prescribes -- summed per kernel and divided by its dispatches.  Returns ({kernel name: {...}}, None) in the
    layout of profiles/pmc_traffic.json, or (None, why) where rocprofv3 cannot run (then the committed passes are replayed, labelled)."""
    import csv
    import glob
    import shutil
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    tot = {}
    for counters in passes:
        d = tempfile.mkdtemp(prefix="pbrt_pmc_", dir="/tmp")
        cmd = [exe, "--pmc", *counters, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable, os.path.abspath(__file__), *bench_args,
               "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-hbm-regime", "--no-live-pmc", "--no-overlap"]
        try:
            p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"))
        except Exception as e:
            return None, f"rocprofv3 --pmc {' '.join(counters)}: {e}"
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if p.returncode != 0 or not files:
            if counters[0] in OPTIONAL_PMC:  # a diagnostic pass this rocprofv3 refuses does not cost the run its traffic counters
                sys.stderr.write(f"bench: optional PMC pass {' '.join(counters)} failed (rc {p.returncode}); skipped\n")
                shutil.rmtree(d, ignore_errors=True)
                continue
            return None, f"rocprofv3 --pmc {' '.join(counters)}: rc {p.returncode}, {len(files)} counter files; {p.stderr[-300:]}"
        for f in files:
            for r in csv.DictReader(open(f)):
                c = r["Counter_Name"]
                if c not in counters: continue
                e = tot.setdefault(c, {}).setdefault(r["Kernel_Name"].split("(")[0], [0.0, set()])
                e[0] += float(r["Counter_Value"]); e[1].add(r["Dispatch_Id"])
        shutil.rmtree(d, ignore_errors=True)
    kernels = {}
    for k, (f, nf) in tot.get("FETCH_SIZE", {}).items():
        w, nw = tot.get("WRITE_SIZE", {}).get(k, (0.0, {0}))
        e = {"launches": len(nf), "fetch_KiB_per_launch": f / len(nf), "write_KiB_per_launch": w / max(1, len(nw))}
        h, mi = tot.get("TCC_HIT_sum", {}).get(k, (0.0, 0))[0], tot.get("TCC_MISS_sum", {}).get(k, (0.0, 0))[0]
        if h + mi > 0: e["l2_hit_rate"] = h / (h + mi)
        dr = tot.get("TCC_EA0_RDREQ_DRAM_32B_sum", {}).get(k)
        if dr is not None: e["dram_rd_32B_per_launch"] = dr[0] / max(1, len(dr[1]))
        # the texture-addresser / vector-L1 pipeline: busy cycles (averaged over the TAs) per cycle the GPU was active in this kernel's dispatches,
        # and the share of its L1-active cycles the L1 stalled on outstanding misses
        ta, ga = tot.get("GRBM_TA_BUSY", {}).get(k), tot.get("GRBM_GUI_ACTIVE", {}).get(k)
        if ta is not None and ga is not None and ga[0] > 0: e["ta_busy_frac"] = ta[0] / ga[0]
        ps, ge = tot.get("TCP_PENDING_STALL_CYCLES_sum", {}).get(k), tot.get("TCP_GATE_EN1_sum", {}).get(k)
        if ps is not None and ge is not None and ge[0] > 0: e["tcp_pending_stall_frac"] = ps[0] / ge[0]
        vi, vt = tot.get("SQ_INSTS_VALU", {}).get(k, (0.0, {0})), tot.get("SQ_THREAD_CYCLES_VALU", {}).get(k, (0.0, 0))[0]
        if vi[0] > 0:
            e["valu_insts_per_launch"] = vi[0] / max(1, len(vi[1])); e["valu_lanes_active"] = vt / vi[0]
        kernels[k] = e
    return (kernels, None) if kernels else (None, "no FETCH_SIZE rows in the counter files")


def merge_pmc(entries):
    """The counters of the kernels behind ONE timed slot as one entry.  The shading slot is k_shade_order + k_material + k_shade<.>
    (HIP events around the three launches): their per-launch totals add up -- each weighted by its own dispatch count, over the
    slot's count (the largest among them) --, lanes per vector instruction are averaged over the instructions."""
    if not entries:
        return None
    if len(entries) == 1:
        return entries[0]
    n = max(max(1, e.get("launches", 1)) for e in entries)
    out = {"launches": n}
    for f in ("fetch_KiB_per_launch", "write_KiB_per_launch", "hbm_bytes_per_launch", "valu_insts_per_launch", "dram_rd_32B_per_launch"):
        if any(f in e for e in entries):
            out[f] = sum(e.get(f, 0.0) * max(1, e.get("launches", 1)) for e in entries) / n
    insts = [(e["valu_insts_per_launch"] * max(1, e.get("launches", 1)), e["valu_lanes_active"]) for e in entries
             if e.get("valu_insts_per_launch") and e.get("valu_lanes_active") is not None]
    if insts:
        out["valu_lanes_active"] = sum(i * l for i, l in insts) / sum(i for i, _ in insts)
    return out


def kernel_rooflines(m, workload, live=None):
    """One roofline per hot kernel.  Algorithmic bytes:
      k_trace: 32 B per reference node fetch + 48 B per triangle test + 32 B per ray in + 16 B (closest) / 4 B (any) out
      k_shade: per path vertex 16 B ray direction + 16 B hit + 48 B state in + 48 B triangle + 48 B state out + 48 B pending
               direct-light terms, + 32 B per ray pushed (+ 16 B of MIS terms per MIS ray)
    Time: HIP events around every launch on the stream it runs on (pg_render), so each kernel is timed alone.
    Bound: the BVH + triangle working set against the 256 MiB Infinity Cache -- below it the node / triangle gathers are
    served on-die and the binding bandwidth is the L2's; above it they reach HBM."""
    cn, desc = m.cn, m.scene.desc
    n_nodes = max(desc.n_nodes, desc.n_nodes_all)
    n_prims = max(desc.n_tris, desc.n_prims_all)
    working_set = 64 * max(0, (n_nodes - 1) // 2) + 64 * n_prims  # child-pair records + triangle records, one 64-B line each
    bound = "l2" if working_set < INFINITY_CACHE_BYTES else "hbm"
    peak = L2_PEAK_GBS if bound == "l2" else HBM_PEAK_GBS
    # Memory-side traffic and L2 hit rates are NOT measured in this run: they are replayed from the committed PMC passes of the same
    # workload (tools/pmc_traffic.sh -> profiles/pmc_traffic.json; raw FETCH_SIZE / WRITE_SIZE in KiB) and every value that comes
    # from there carries "source".  FETCH_SIZE is scaled by the factor measured for this access pattern on a known byte count
    # (tools/pmc_calibrate.sh -> profiles/fetch_size_calibration.json); without that file the guide's streaming-read factor 2 is
    # used and labelled as uncalibrated.
    pmc_kernels, pmc_src, factor, factor_src = {}, None, 2.0, "MI355X_MICROARCH.md: x2 for 16 B/lane streaming reads (uncalibrated for gathers)"
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc):
        try:
            allw = json.load(open(pmc)).get("workloads", {})
            w = allw.get(workload, {})
            pmc_kernels, pmc_src = w.get("kernels", {}), "profiles/pmc_traffic.json" + (f"@{w['round']}" if "round" in w else "")
            if not pmc_kernels and " @ " in workload:
                # a pass of the same scene at another sample count (the divergent stand-ins' passes run at a quarter of the spp): hit rates
                # and lanes per instruction carry over, per-launch totals scale with the launches' size, i.e. with spp
                base, spp_now = workload.rsplit(" @ ", 1)
                for k, v in allw.items():
                    if k.rsplit(" @ ", 1)[0] == base:
                        scale = float(spp_now.split()[0]) / float(k.rsplit(" @ ", 1)[1].split()[0])
                        pmc_kernels = {kn: {f: (x * scale if f.endswith("_per_launch") else x) for f, x in kv.items()} for kn, kv in v.get("kernels", {}).items()}
                        pmc_src = f"profiles/pmc_traffic.json ({k.rsplit(' @ ', 1)[1]} pass, per-launch totals x {scale:g})"
                        break
        except Exception:
            pass
    replayed = True
    if live:  # counters collected by this run's own rocprofv3 --pmc passes (live_pmc): nothing replayed
        pmc_kernels, pmc_src, replayed = live, "live: rocprofv3 --pmc passes over one frame of this workload, run by this bench.py", False
    cal = os.path.join(ROOT, "profiles", "fetch_size_calibration.json")
    if os.path.exists(cal):
        try:
            c = json.load(open(cal))
            factor, factor_src = float(c["gather_factor"]), "profiles/fetch_size_calibration.json: " + c.get("note", "")
        except Exception:
            pass

    def one(name, tag, alg_bytes, ms, launches, units, unit_name, gather_pattern, bound=bound, peak=peak):
        launches = max(1, launches)
        achieved = alg_bytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        pk = merge_pmc([v for k, v in pmc_kernels.items() if k.startswith(tag)])  # tag: a tuple of accepted name prefixes
        traffic = None
        if pk is not None:
            f = factor if gather_pattern else 2.0
            traffic = (f * pk["fetch_KiB_per_launch"] + pk["write_KiB_per_launch"]) * 1024 if "fetch_KiB_per_launch" in pk else pk.get("hbm_bytes_per_launch")
        served_on_die = bound == "hbm" and achieved >= peak
        if served_on_die:  # more algorithmic bytes per second than HBM can deliver: the caches serve part of this kernel's gathers -- their roofline, then
            bound, peak = "l2", L2_PEAK_GBS
        r = {"kernel": name, "bound": bound, "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
             "traffic": traffic, "algorithmic_bytes_per_launch": alg_bytes / launches, "avg_launch_ms": ms / launches,
             "launches": launches, "total_ms": ms, f"bytes_per_{unit_name}": alg_bytes / max(1, units)}
        if pk is not None and pk.get("valu_insts_per_launch") and ms > 0:
            # what the SIMDs issued: wave-wide vector instructions per second against 1024 SIMD-32 units x 2.4 GHz / 2 cycles per 64-wide instruction
            rate = pk["valu_insts_per_launch"] * launches / (ms * 1e-3)
            r["vector_issue"] = {"insts_per_launch": pk["valu_insts_per_launch"], "achieved": rate, "peak": VALU_ISSUE_PEAK, "unit": "wave instructions/s",
                                 "frac": rate / VALU_ISSUE_PEAK, "lanes_active_of_64": pk.get("valu_lanes_active"), "source": pmc_src,
                                 "note": ("instruction count replayed from a committed PMC pass of this workload" if replayed else "instruction count of this run's own PMC pass") + " (SQ_INSTS_VALU), divided by this run's kernel time"}
        if pk is not None and pk.get("ta_busy_frac") is not None:
            r["load_path"] = {"ta_busy_frac": pk["ta_busy_frac"], "tcp_pending_stall_frac": pk.get("tcp_pending_stall_frac"), "source": pmc_src,
                              "note": "GRBM_TA_BUSY / GRBM_GUI_ACTIVE: the share of the kernel's time in which a texture-addresser (vector-memory address) unit was busy; "
                                      "TCP_PENDING_STALL_CYCLES / TCP_GATE_EN1: the share of the vector L1s' active cycles stalled on outstanding misses"}
        if served_on_die:
            r["note"] = (f"algorithmic rate above the HBM peak ({HBM_PEAK_GBS:.0f} GB/s) although the scene exceeds the Infinity Cache: the top of the tree is "
                         "served on-die (short traversals), so this kernel is priced against the L2")
        if traffic is not None:
            r["traffic_source"] = {"source": pmc_src, "fetch_size_factor": factor if gather_pattern else 2.0,
                                   "fetch_size_factor_source": factor_src if gather_pattern else "MI355X_MICROARCH.md (streaming reads)",
                                   "note": "replayed from a committed PMC pass of this workload, not measured in this run" if replayed else
                                           "FETCH_SIZE / WRITE_SIZE (KiB) per launch from this run's own rocprofv3 --pmc passes (separate passes, one frame)"}
            if ms > 0:
                # What the L2s' memory side (their fabric ports, TCC_EA0_RDREQ / WRREQ) moved.  Infinity-Cache hits are counted here: this
                # is NOT DRAM traffic (MI355X_MICROARCH.md "Infinity Cache").  The 8 TB/s beside it is the HBM peak, as a yardstick only.
                hb = traffic * launches / (ms * 1e-3) / 1e9
                r["l2_memory_side"] = {"achieved": hb, "compared_with": HBM_PEAK_GBS, "unit": "GB/s", "frac_of_hbm_peak": hb / HBM_PEAK_GBS, "source": pmc_src,
                                       "note": "requests of the L2s to the fabric; Infinity-Cache hits included, so an upper bound of the DRAM traffic"}
                if pk.get("dram_rd_32B_per_launch") is not None:
                    # TCC_EA0_RDREQ_DRAM_32B: the L2s' read requests whose target is DRAM-backed memory, in 32-B units ("1 64-byte request
                    # will be counted to 2, 128-byte as 4").  Still counted at the L2 port: it separates DRAM-backed from other targets,
                    # not Infinity-Cache hits from misses -- rocprofv3 exposes no memory-controller counter on this image.
                    db = pk["dram_rd_32B_per_launch"] * 32.0
                    r["l2_memory_side"]["dram_targeted_read_bytes_per_launch"] = db
                    r["l2_memory_side"]["dram_targeted_read_GBs"] = db * launches / (ms * 1e-3) / 1e9
        return r

    n_close, n_shadow, n_mis, n_items = cn["closest_rays"], cn["shadow_rays"], cn["mis_rays"], cn["shade_items"]
    n_next = max(0, n_close - cn["camera_rays"] - n_mis)
    kernels = [
        one("k_trace<0> (closest hit: BVHAccel::Intersect + Triangle::Intersect)", ("void k_trace<0,", "void k_trace<false"),
            32 * cn["closest_node_visits"] + 48 * cn["closest_tri_tests"] + 48 * n_close, cn["closest_ms"], cn["closest_launches"], n_close, "ray", True),
        one(("k_trace<1> (any hit, reference order" if os.environ.get("PG_ANYHIT_ORDER") == "reference" else "k_trace<2> (any hit, free order") + ": BVHAccel::IntersectP + Triangle::IntersectP)", ("void k_trace<2,", "void k_trace<1,", "void k_trace<true"),
            32 * cn["shadow_node_visits"] + 48 * cn["shadow_tri_tests"] + 36 * n_shadow, cn["shadow_ms"], cn["shadow_launches"], n_shadow, "ray", True),
        one("k_shade (PathIntegrator::Li loop body + EstimateDirect set-up; with k_shade_order / k_material where they run)", ("void k_shade<", "void k_shade_order<", "void k_material<"),
            224 * n_items + 32 * (n_next + n_shadow + n_mis) + 16 * n_mis, cn["shade_ms"], cn["shade_launches"], n_items, "vertex", False,
            # its streams are the queues and the path state (hundreds of bytes per vertex of a 10^8-vertex launch), not the scene: HBM at any scene size
            bound="hbm", peak=HBM_PEAK_GBS),
    ]
    kernels = [k for k in kernels if k["total_ms"] > 0]
    kernels.sort(key=lambda k: -k["total_ms"])  # dominant = the most time, each kernel timed alone
    return kernels, working_set, bound, peak, pmc_kernels, pmc_src


def gather_ceiling(m, working_set, pmc_kernels, pmc_src):
    """The traversal kernels are gathers of 64-B child-pair records and 48-B triangle records, one per lane and step.  What bounds
    them is the chip's rate of random record fetches at this working-set size, measured live by tools/ubench_gather.hip
    (pbrt-v3_amd/ubench_gather: every lane chases its own chain of records through a table of that size): DESIGN.md section 5.
    Closest hit: an interior step counts two reference node visits (both children), the root one per ray, so
    record fetches = (node visits - rays) / 2 + triangle tests."""
    cn = m.cn
    ub = os.path.join(ROOT, "pbrt-v3_amd", "ubench_gather")
    if not os.path.exists(ub) or cn["closest_ms"] <= 0 or EMULATED:
        return None
    try:
        ws_mb = max(3, int(round(working_set / 2**20)))
        ceil = json.loads(subprocess.run([ub, "--json", "2", str(ws_mb)], capture_output=True, text=True, timeout=120).stdout)
        c_l2, c_ws = ceil["2"]["together"], ceil[str(ws_mb)]["together"]
        fetches = max(0, cn["closest_node_visits"] - cn["closest_rays"]) / 2 + cn["closest_tri_tests"]
        rate = fetches / (cn["closest_ms"] * 1e-3)
        # A traversal is not a uniformly random walk: the top of the tree stays in the L2s.  With the kernel's L2 hit rate h (committed
        # PMC pass of this workload) the ceiling is the harmonic blend of the L2-resident rate and the rate at the working set's
        # size; without h only the L2-resident rate is a safe upper bound.
        h = next((v.get("l2_hit_rate") for k, v in pmc_kernels.items() if k.startswith(("void k_trace<0,", "void k_trace<false"))), None)
        ceiling = 1.0 / (h / c_l2 + (1.0 - h) / c_ws) if h is not None else c_l2
        return {"kernel": "k_trace<0>", "record_fetches_per_s": rate, "ceiling_records_per_s": ceiling, "frac": rate / ceiling,
                "ceiling_kind": ("1 / (h / C(2 MiB) + (1 - h) / C(working set)), h = the kernel's L2 hit rate" if h is not None
                                 else "C(2 MiB): L2-resident table (no PMC pass of this workload committed: upper bound)"),
                "l2_hit_rate": h, "l2_hit_rate_source": pmc_src if h is not None else None,
                "table_MiB": ws_mb, "ceiling_l2_resident": c_l2, "ceiling_at_working_set": c_ws,
                "note": "C(x) = random 64-B record fetches per second through the vector L1, measured live by pbrt-v3_amd/ubench_gather: one "
                        "chain per lane, table of x MiB; fetches the kernel serves from its LDS-resident tree top count in its rate, so a value above 1 is possible"}
    except Exception as e:  # the measurement tool is optional; the bench line is not
        return {"error": str(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="synthetic", choices=["synthetic", "synthetic-vol", "cornell", "divergent", "divergent-vol", "config0"],
                    help="synthetic: BASELINE config 3 (with --grid 1582: the 5 M-triangle HBM-regime stand-in); synthetic-vol (--grid 2237 "
                         "--spp 128): 10 M triangles in fog under volpath; cornell: config 2; divergent / divergent-vol (--tris 5000000 / "
                         "10000000): configs 4 / 5 as instanced PLY meshes with textures, alpha masks and a material palette")
    ap.add_argument("--grid", type=int, default=708, help="heightfield vertices per side (708 -> 999 698 triangles)")
    ap.add_argument("--tris", type=int, default=5000000, help="divergent workloads: instanced triangle count to reach")
    ap.add_argument("--xres", type=int, default=None, help="default 1920 (config0: 400)")
    ap.add_argument("--yres", type=int, default=None, help="default 1080 (config0: 400)")
    ap.add_argument("--spp", type=int, default=None, help="default 64 (config0: 8)")
    ap.add_argument("--filter", default="box", help='PixelFilter of the scene (BASELINE config: "box")')
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-full", action="store_true", help="cpu_baseline renders every sample of the frame (minutes)")
    ap.add_argument("--no-hbm-regime", action="store_true", help="skip the 3 extra frames of the 5 M-triangle workload behind roofline.hbm_regime")
    ap.add_argument("--no-live-pmc", action="store_true", help="do not run the rocprofv3 --pmc child passes; replay profiles/pmc_traffic.json (labelled) instead")
    ap.add_argument("--no-overlap", action="store_true", help="one GPU: do not run any-hit launches beside the next closest-hit launch in the timed frames "
                                                             "(every kernel alone on the chip, as in the serialised frame the rooflines are taken from; what rocprofv3 --stats should profile)")
    ap.add_argument("--out", default=None, help="write the rendered image (PFM) here")
    args = ap.parse_args()
    dx, dy, dspp = (400, 400, 8) if args.workload == "config0" else (1920, 1080, 64)
    args.xres, args.yres, args.spp = args.xres or dx, args.yres or dy, args.spp or dspp

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    ctx = Context(args)
    # The any-hit launch of a bounce runs on a second stream beside the next bounce's closest-hit launch: a shard's launches are small
    # and their tails weigh more (one rank of 8: 48.7 instead of 52.9 ms, profiles/r03j_shard_timing.json), and the whole frame on one
    # GPU gains 1.2 % (profiles/r04b_bench_overlap*.json).  Per-kernel times then overlap, so a one-GPU run takes them -- and its
    # rooflines -- from one more frame, serialised, after the timed region (run_workload, serial_frame).
    overlap = not args.no_overlap and os.environ.get("PG_OVERLAP_SHADOW", "1") != "0"
    args.overlap = overlap  # (handed to every scene through pg_scene_set_option: the library reads the environment only at scene creation)
    pkg = load_package()
    one_gpu = ctx.world == 1 and not ctx.multi
    m = run_workload(ctx, args, args.steps, args.warmup, keep_image=bool(args.out), serial_frame=overlap and one_gpu)
    # North star: ">= 40 % of the HBM roofline in the BVH-traversal kernel" is a statement about the regime where the BVH does not
    # fit the 256 MiB Infinity Cache, which config 3 (106 MiB) is not in.  After the headline steps, rank 0 of a 1-GPU run times 3
    # frames of the same scene at 5 M triangles (535 MiB of records) at the headline's resolution and spp: roofline.hbm_regime.
    hbm = None
    if ctx.world == 1 and not ctx.multi and not args.no_hbm_regime and not EMULATED and args.workload == "synthetic" and args.grid == 708:
        a5 = argparse.Namespace(**vars(args))
        a5.grid = 1582
        m.gs.close()
        try:
            hbm = run_workload(ctx, a5, 3, 1, serial_frame=overlap)
            hbm.args = a5
        except Exception as e:
            sys.stderr.write(f"bench: hbm-regime workload failed: {e}\n")

    if ctx.rank == 0:
        if args.out and m.image is not None:
            pkg.write_pfm(args.out, m.image)
        workload = describe(args, m.scene)
        # memory-side traffic, L2 hit rates and vector-instruction counts of the hot kernels: measured by this run (child rocprofv3 --pmc
        # passes over one frame each, after the timed region) unless --no-live-pmc / N > 1 / an emulated device
        # (this process's device scenes go first: a child pass renders the same workload in a process of its own, and with 100+ GB of
        # buffers still held here pg_render's k_material lists would not fit there -- it would measure the k_shade<2> fallback instead)
        m.gs.close()
        if hbm is not None: hbm.gs.close()
        if not EMULATED: torch.cuda.empty_cache()
        live, live_why, live_hbm = None, "--no-live-pmc", None
        wl_args = ["--workload", args.workload, "--grid", str(args.grid), "--tris", str(args.tris), "--xres", str(args.xres), "--yres", str(args.yres),
                   "--spp", str(args.spp), "--filter", args.filter]
        if ctx.world == 1 and not ctx.multi and not EMULATED and not args.no_live_pmc:
            live, live_why = live_pmc(wl_args, (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_HIT_sum", "TCC_MISS_sum"], ["SQ_INSTS_VALU", "SQ_THREAD_CYCLES_VALU"],
                                                ["GRBM_TA_BUSY", "GRBM_GUI_ACTIVE"], ["TCP_PENDING_STALL_CYCLES_sum", "TCP_GATE_EN1_sum"]))
            if live is None: sys.stderr.write(f"bench: live PMC passes failed ({live_why}); replaying profiles/pmc_traffic.json\n")
            if hbm is not None and live is not None:
                a5l = ["--workload", "synthetic", "--grid", str(hbm.args.grid), "--xres", str(args.xres), "--yres", str(args.yres), "--spp", str(args.spp), "--filter", args.filter]
                live_hbm, why5 = live_pmc(a5l, (["FETCH_SIZE"], ["WRITE_SIZE"], ["TCC_EA0_RDREQ_DRAM_32B_sum"]))
                if live_hbm is None: sys.stderr.write(f"bench: live PMC passes of the hbm-regime workload failed ({why5})\n")
        # the frame the per-kernel figures come from: the serialised one where the timed frames overlapped kernels
        km = m if m.serial is None else types.SimpleNamespace(cn=m.serial.cn, scene=m.scene)
        ksteps = args.steps if m.serial is None else 1
        kernels, working_set, bound, peak, pmc_kernels, pmc_src = kernel_rooflines(km, workload, live)
        gather = gather_ceiling(km, working_set, pmc_kernels, pmc_src)
        roofline = dict(kernels[0]) if kernels else {"kernel": None, "bound": bound, "achieved": 0.0, "peak": peak, "unit": "GB/s", "frac": 0.0, "traffic": None}
        roofline["working_set_bytes"] = working_set
        # every reading of the dominant kernel's rate at the top level: against the L2s' aggregate bandwidth, against the HBM peak (north star's
        # yardstick; above 1 = the caches serve the gathers), and what the L2s' memory side moved against the HBM peak
        if kernels:
            roofline["frac_of_l2"] = roofline["achieved"] / L2_PEAK_GBS
            roofline["frac_of_hbm_algorithmic"] = roofline["achieved"] / HBM_PEAK_GBS
            roofline["cache_served"] = bool(roofline["bound"] == "l2")
            if roofline.get("l2_memory_side"): roofline["l2_memory_side_frac_of_hbm"] = roofline["l2_memory_side"]["frac_of_hbm_peak"]
        if gather is not None:
            roofline["gather"] = gather
        roofline["bound_reason"] = (f"BVH + triangle records {working_set / 2**20:.0f} MiB " +
                                    ("fit the 256 MiB Infinity Cache: gathers are served on-die, L2 bandwidth is the ceiling"
                                     if bound == "l2" else "exceed the 256 MiB Infinity Cache: gathers reach HBM"))
        if hbm is not None:
            hk, hws, hbound, _, _, _ = kernel_rooflines(hbm if hbm.serial is None else types.SimpleNamespace(cn=hbm.serial.cn, scene=hbm.scene), describe(hbm.args, hbm.scene), live_hbm)
            t = next((k for k in hk if k["kernel"].startswith("k_trace<0>")), None)
            if t is not None and hbound == "hbm":
                roofline["hbm_regime"] = {
                    "kernel": t["kernel"], "workload": describe(hbm.args, hbm.scene), "working_set_bytes": hws, "bound": "hbm",
                    "achieved": t["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": t["achieved"] / HBM_PEAK_GBS,
                    "avg_launch_ms": t["avg_launch_ms"], "launches": t["launches"], "algorithmic_bytes_per_launch": t["algorithmic_bytes_per_launch"],
                    "steps": hbm.steps, "ms_per_step": hbm.elapsed / hbm.steps * 1e3, "Mrays_per_s": hbm.rays / hbm.elapsed / 1e6,
                    "note": "timed in THIS run (HIP events per launch); algorithmic bytes (SURVEY 8d) / kernel time / 8 TB/s. Algorithmic bytes count "
                            "every reference node fetch, also those the L2s serve, so this is the north star's roofline fraction, NOT memory-side "
                            "traffic: `l2_memory_side` is what the L2s' fabric ports moved for this kernel (FETCH_SIZE x factor + WRITE_SIZE; Infinity-Cache hits included), beside the same 8 TB/s"}
                for f in ("traffic", "l2_memory_side", "traffic_source"):  # the memory-side companion (weak point of round 3: 0.97 alone reads as HBM utilisation)
                    if t.get(f) is not None: roofline["hbm_regime"][f] = t[f]
        other_ms = {k: km.cn[k] for k in ("resolve_ms", "generate_ms", "film_ms")}
        if ctx.world == 1:
            sharding = ("one GPU renders every 16x16 film tile; " +
                        ("ONE-RANK RCCL run (PBRT_BENCH_FORCE_DIST=1): the packed gather of the frame runs over RCCL with one rank" if ctx.multi else "no gather") +
                        ("; any-hit launches beside the next closest-hit launches" if overlap else ""))
        else:
            sharding = (f"16x16 film tiles round-robin over {ctx.world} GPUs, one process per GPU; one packed gather per frame to rank 0 (" +
                        ("RCCL over xGMI" if ctx.backend == "nccl" else "PRE-FLIGHT: gloo on host copies") + "), overlapped with the next frame" +
                        ("; any-hit launches beside the closest-hit launches (per-kernel times overlap)" if overlap else ""))
        result = {
            "metric": "Mrays/s", "value": m.rays / m.elapsed / 1e6, "unit": "Mrays/s", "n_gpus": ctx.world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": m.elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "samples_per_s": m.samples / m.elapsed,
            "ranks_seen": dist.get_world_size() if ctx.multi else 1, "per_rank_ms": m.per_rank_ms,
            **({"per_rank_kernel_ms_per_step": m.per_rank_kernel_ms} if ctx.multi else {}),
            "config": {"workload": workload, "sharding": sharding, "shading_mode": shading_mode(m.cn),
                       "rays_per_sample": m.rays / max(1.0, m.samples), "host_parse_and_bvh_s": m.t_parse},
            "roofline": roofline,
            "roofline_kernels": kernels,
            "kernel_ms_per_step": {**{k["kernel"].split(" ")[0]: k["total_ms"] / ksteps for k in kernels},
                                   **{k[:-3]: v / ksteps for k, v in other_ms.items()}},
        }
        if roofline.get("hbm_regime"):  # also at the top level of the line (a parser that keeps only known keys of `roofline` still sees it)
            result["hbm_regime"] = roofline["hbm_regime"]
        if live:
            # the counters of this run's own PMC passes, kernel by kernel (the shading slot's roofline merges k_shade_order + k_material + k_shade<.>)
            result["pmc_by_kernel"] = {k.replace("void ", "")[:48]: {f: (round(v, 4) if isinstance(v, float) else v) for f, v in e.items()} for k, e in sorted(live.items())
                                       if e.get("fetch_KiB_per_launch", 0) * e.get("launches", 1) > 1e5}
        if m.serial is not None:
            result["kernel_times"] = {
                "from": "ONE extra frame after the timed region with every kernel alone on the chip (pg_scene_set_option PG_OPT_OVERLAP_SHADOW 0): kernel_ms_per_step, roofline and "
                        "roofline_kernels are that frame's HIP-event times; the timed frames (ms_per_step, value) run each any-hit launch beside the next "
                        "closest-hit launch, so they are shorter than the sum of the kernels",
                "serialized_frame_ms": m.serial.ms, "sum_of_kernels_ms": sum(result["kernel_ms_per_step"].values()), "overlapped_frame_ms": result["ms_per_step"]}
        if EMULATED:
            result["data"] = "synthetic (EMULATED DEVICE: functional check, not a measurement)"
        if ctx.world == 1 and not args.no_cpu_baseline and not EMULATED:
            result["cpu_baseline"] = cpu_baseline(m.workdir, args)
        print(json.dumps(result), flush=True)
    if ctx.multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
