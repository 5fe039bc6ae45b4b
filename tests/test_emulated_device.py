"""The `-m gpu` tests without a GPU: the product's device translation units compiled for the host under the SIMT emulator of tests/emu
(kernels, queues, wave-level code and host orchestration included; tests/emu/README.md), loaded through PBRT_GPU_LIB, and the GPU test
files run unchanged in a child pytest.  A subset runs in the normal CPU suite; PBRT_EMULATE_ALL=1 runs everything that is feasible
under emulation (about a quarter of an hour)."""
import os
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# not in the PBRT_EMULATE_ALL run: torch device buffers, the tile-serial samplers (they pass, but take
# minutes each: hundreds of thousands of tiny launches), full-size frames
SKIP = "not device_buffers and not sampler_ and not 02sequence and not full_size"


@pytest.fixture(scope="module")
def emulated(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not available")
    out = str(tmp_path_factory.mktemp("emulated"))
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tests", "emu", "build_emulated.py"), out], stdout=subprocess.DEVNULL)
    return os.path.join(out, "libpbrt_gpu_emulated.so")


def run_gpu_tests(lib, files, select, timeout):
    env = dict(os.environ, PBRT_GPU_LIB=lib, PBRT_EMULATED_DEVICE="1")
    # (several tests selected: three worker processes -- the emulated device is host code, the tests are independent; PBRT_EMULATE_WORKERS=0: serial)
    workers = os.environ.get("PBRT_EMULATE_WORKERS", "3")
    par = ["-n", workers] if workers != "0" and not any("::" in f for f in files) else []
    p = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", *par, "-k", select], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=timeout)
    tail = p.stdout[-3000:] + p.stderr[-1500:]
    assert p.returncode == 0, tail
    return tail


def test_ray_queries_on_the_emulated_device(emulated):
    """Closest-hit and any-hit kernels: Triangle / quadric .Reintersect at extreme magnitudes, Watertight + degenerate triangles, rays
    through instanced objects -- hits, t, barycentrics and the reference's node / triangle counters equal the oracle's."""
    out = run_gpu_tests(emulated, ["tests/test_gpu_parity.py"], "reintersect_on_device or watertight or instance_rays_bit_exact and instance_boxes", 900)
    assert " passed" in out and "failed" not in out


def test_hlbvh_build_on_the_emulated_device(emulated):
    """pg_hlbvh_build -- Morton codes, the hand-written radix sort (k_radix_*: LDS histograms, the one-block scan, eight ballots per
    scatter step), treelets, emitLBVH, the upper SAH tree -- equals the host front end's HLBVHBuild node for node."""
    out = run_gpu_tests(emulated, ["tests/test_hlbvh_build.py"], "not golden", 900)
    assert " passed" in out and "failed" not in out


def test_renders_on_the_emulated_device(emulated):
    """Whole renders through generate / trace / shade / resolve / film: films bit-identical to the correctly-rounded oracle (path and
    volpath, textures, spheres, instances, a general-filter film) and within tolerance of the reference's goldens."""
    select = ("(test_film_bit_identical_to_correctly_rounded_oracle or test_golden_images) and "
              "(cornell_32 or vol_fog_halfspace or sphere_light or instance_boxes or filter_gaussian)")
    out = run_gpu_tests(emulated, ["tests/test_gpu_parity.py"], select, 1200)
    assert " passed" in out and "failed" not in out


def test_round3_kernels_on_the_emulated_device(emulated):
    """What round 3 added to the device, each on one small scene: subsurface scattering (the probe chain walked twice through k_trace, the exit
    vertex), a GridDensityMedium (two shading phases around the transmittance rays), the divergent stand-in (instances, alpha masks,
    shading by material class), a PixelSampler whose dimensions cover a path (all pixels' arrays ahead, one wavefront)."""
    select = "test_golden_images and (sss_preset or grid_puff_sobol or divergent_small_vol or sampler_stratified_dims_tex)"
    out = run_gpu_tests(emulated, ["tests/test_gpu_parity.py"], select, 1500)
    assert " passed" in out and "failed" not in out
    out = run_gpu_tests(emulated, ["tests/test_gpu_shade_order.py"], "golden_scene and (divergent_small or tex_image)", 900)
    assert " passed" in out and "failed" not in out


def test_round5_kernels_on_the_emulated_device(emulated):
    """What round 5 changed on the device, each on small scenes: volpath with the path state, the ray's medium and the pending terms in
    queue order (k_shade<., VOL> QSTATE, k_through carrying queue positions, k_resolve_vol<true>) -- a fog scene, smoke behind null
    surfaces --, and a moving camera (AnimatedTransform interpolated per camera ray: rotation + translation + scale
    inside the shutter interval; image textures filtered through the differentials of the same interpolated transform)."""
    select = "test_golden_images and (vol_fog or vol_smoke or vol_path_none_glass or camanim_times_scale or camanim_lens_tex)"  # (the instanced volpath stand-in: round 3's test)
    out = run_gpu_tests(emulated, ["tests/test_gpu_parity.py"], select, 1500)
    assert " passed" in out and "failed" not in out


def test_round6_kernels_on_the_emulated_device(emulated):
    """What round 6 changed on the device: a GridDensityMedium and BSSRDF materials in ONE scene (k_shade<., VOL, SSS, GRID>, k_sss_exit in two phases around the
    exit vertex's transmittance rays), the integrator statistics of ABI 28 (test_golden_images compares them with the reference binary's printed ones), k_material's
    lists by position in the shading order and the triangles' attribute records (the textured stand-in)."""
    select = "test_golden_images and (grid_sss_sobol or grid_sss_motion or divergent_small or tex_image)"
    out = run_gpu_tests(emulated, ["tests/test_gpu_parity.py"], select, 1500)
    assert " passed" in out and "failed" not in out
    out = run_gpu_tests(emulated, ["tests/test_gpu_parity.py"], "test_integrator_statistics_run_on", 600)
    assert " passed" in out and "failed" not in out


def test_moving_shapes_and_instances_on_the_emulated_device(emulated):
    """Round 5, last: TransformedPrimitive over an AnimatedTransform on the device -- the queues carry the rays' times, k_trace<., XP_ANIM>
    interpolates a moving instance's transform (and inverts it: Gauss-Jordan of the blended scale) at the ray's time when it enters the
    instance and leaves the accepted hit's matrices for the shading kernels (pg_motion.h).  Moving meshes (a BVH of their own under one
    instance), TransformTimes inside the shutter, moving instances of a BVH object / a lone sphere / a lone triangle beside still and mirrored
    ones, a moving quadric, volpath, a moving camera on top, moving boxes of a subsurface material under volpath; and one random scene of the fuzz
    generator.  (Tile-serial samplers and grid media with motion -- motion_random, grid_puff_motion* -- take minutes each under emulation: GPU suite.)"""
    select = "test_golden_images and (motion_boxes_times or motion_instances_shutter or motion_vol or motion_camera_too or sss_motion_volpath or motion_rotate_big_times)"
    out = run_gpu_tests(emulated, ["tests/test_gpu_parity.py"], select, 1500)
    assert " passed" in out and "failed" not in out
    out = run_gpu_tests(emulated, ["tests/test_gpu_fuzz.py::test_random_scene_with_moving_shapes_and_instances[1]"], "moving_shapes", 1500)
    assert " passed" in out and "failed" not in out
    # motions that ROTATE (hasRotation: the slerp branch of instance_matrices_at, MotionBounds' boxes in the top-level BVH): one golden above, one random
    # scene (volpath) here; all 6 goldens and 178 random scenes went through the emulated device once (profiles/r05q_rotating_motion.txt)
    out = run_gpu_tests(emulated, ["tests/test_gpu_fuzz.py::test_random_scene_with_rotating_shapes_and_instances[1]"], "rotating_shapes", 1500)
    assert "1 passed" in out and "failed" not in out


def test_moving_shapes_inside_object_definitions_on_the_emulated_device(emulated):
    """Round 6, ABI 29: a TransformedPrimitive among an object definition's primitives (pbrtShape under an animated transformation between ObjectBegin and
    ObjectEnd, api.cpp:1386-1419) -- k_trace<., XP_NEST> keeps a second saved context and derives the instance's ray again when it leaves the inner
    object, the hit's instance word carries (outer, inner), k_shade<2> applies InterpolatedPrimToWorld of the inner and then of the outer.  Two goldens
    from the reference binary (a BVH object, a lone sphere and a lone triangle with moving shapes inside, under moving instances; one subsurface material on
    still and moving parts under volpath) and one random scene; all eight goldens and 64 random scenes went through the emulated device once (profiles/r06s_nested_motion.txt)."""
    out = run_gpu_tests(emulated, ["tests/test_gpu_parity.py"], "test_golden_images and (nest_motion_moving_instances or sss_nest_motion_volpath)", 1500)  # (the second: a subsurface material on still and moving parts, volpath, Sobol')
    assert "2 passed" in out and "failed" not in out
    out = run_gpu_tests(emulated, ["tests/test_gpu_fuzz.py::test_random_scene_with_moving_shapes_inside_object_definitions[1]"], "inside_object", 1500)
    assert "1 passed" in out and "failed" not in out


def test_material_pass_on_the_emulated_device(emulated):
    """Round 4's k_material + k_shade<3, .> (materials with textured parameters evaluated ahead of the shading launch) against the
    evaluation inside the shading kernel (PG_MAT_PRE=0): same film, strays and counters -- bump maps, the divergent stand-ins (instances,
    every material kind, path and volpath), an 8-BxDF mix of two uber materials."""
    select = "longest_lists or bump_maps or divergent_small or tex_materials"
    out = run_gpu_tests(emulated, ["tests/test_gpu_material_prepass.py"], select, 1500)
    assert " passed" in out and "failed" not in out


@pytest.mark.skipif(os.environ.get("PBRT_EMULATE_ALL") != "1", reason="set PBRT_EMULATE_ALL=1 for the whole feasible GPU suite under emulation (~15 min)")
def test_everything_feasible_on_the_emulated_device(emulated):
    out = run_gpu_tests(emulated, ["tests/test_gpu_parity.py", "tests/test_gpu_fuzz.py"], SKIP, 7200)
    assert " passed" in out


def test_reference_side_binding_on_the_emulated_device(emulated):
    """The drop-in end to end without a GPU: the unmodified reference (parser, scene construction, BVHAccel, Film) with the compiled binding
    of INTEGRATION.md section 2 (oracle/_ref/pbrt_gpubind) driving the device kernels under emulation -- tests/test_gpu_binding.py unchanged."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "pbrt_gpubind")):
        pytest.skip("oracle/_ref/pbrt_gpubind is built only where /root/reference exists")
    out = run_gpu_tests(emulated, ["tests/test_gpu_binding.py"], "cornell_32 or vol_fog", 900)
    assert " passed" in out and "failed" not in out


def test_native_multi_device_render_on_emulated_devices(emulated, pkg, tmp_path):
    """`pbrt_amd --gpus N` (one host thread per device, tiles t = r (mod N), shards gathered peer to peer on the first device, merged by
    Film::MergeShard) with HIP_EMU_DEVICES pretending to N devices: the image equals the single-device one bit for bit -- the native
    multi-GPU path executed end to end, which no GPU box with more than one device has been available for."""
    import numpy as np
    cli = os.path.join(ROOT, "pbrt-v3_amd", "pbrt_amd")
    if not os.path.exists(cli):
        pytest.skip("pbrt_amd not built")
    scene = os.path.join(ROOT, "tests", "golden", "cornell_40x24.pbrt")
    images = {}
    for n in (1, 3):
        out = str(tmp_path / f"n{n}.pfm")
        p = subprocess.run([cli, "--gpus", str(n), "--outfile", out, scene], capture_output=True, text=True, timeout=900,
                           env=dict(os.environ, PBRT_GPU_LIB=emulated, HIP_EMU_DEVICES=str(n)))
        assert p.returncode == 0 and f"on {n} GPU(s)" in p.stdout + p.stderr, (p.stdout + p.stderr)[-1500:]
        images[n] = pkg.read_pfm(out)
    assert np.array_equal(images[1], images[3])
    assert np.array_equal(images[1], pkg.read_pfm(scene[:-5] + ".pfm")) or np.abs(images[1] - pkg.read_pfm(scene[:-5] + ".pfm")).max() < 1e-4


def test_a_failing_rank_fails_the_sharded_render_instead_of_hanging_it(emulated, pkg, tmp_path):
    """pg_render_sharded: a rank whose render fails (here PG_TEST_FAIL_RANK; on hardware an out-of-memory device) must not leave the
    other ranks' threads waiting in the collective for it.  All threads meet at a host barrier between render and gather; with a failed
    rank none enters the gather and the call returns that rank's error."""
    cli = os.path.join(ROOT, "pbrt-v3_amd", "pbrt_amd")
    if not os.path.exists(cli):
        pytest.skip("pbrt_amd not built")
    scene = os.path.join(ROOT, "tests", "golden", "cornell_40x24.pbrt")
    for bad in (0, 2):
        p = subprocess.run([cli, "--gpus", "3", "--outfile", str(tmp_path / "x.pfm"), scene], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, PBRT_GPU_LIB=emulated, HIP_EMU_DEVICES="3", PG_TEST_FAIL_RANK=str(bad)))
        assert p.returncode != 0 and f"rank {bad}: PG_TEST_FAIL_RANK" in p.stdout + p.stderr, (p.stdout + p.stderr)[-1500:]


def _rank(rank, world, port, lib, name, out):
    """One rank of bench.py's step(): render this rank's tiles with pg_render into its own buffers (GpuScene.render_device -- the call
    bench.py makes), gather on rank 0 with the product's pbrt_v3_amd.distributed (gloo here, RCCL there), merge."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), PBRT_GPU_LIB=lib, PBRT_EMULATED_DEVICE="1")
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch.distributed as dist
    from __graft_entry__ import load_package
    pkg = load_package()
    from pbrt_v3_amd import distributed as pdist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    scene = pkg.HostScene(os.path.join(ROOT, "tests", "golden", name + ".pbrt"))
    gs = pkg.GpuScene(scene.desc)
    rd = scene.render_desc(tile_first=rank, tile_step=world)
    film, strays, nstrays, max_strays = pdist.shard_buffers(gs.tile_count(scene.render_desc(0, world)), "cpu", rd.tile_pixels)
    gs.render_device(rd, film.data_ptr(), strays.data_ptr(), max_strays, nstrays.data_ptr(), stream=None)
    lists = pdist.gather_film(film, strays, nstrays, dst=0)
    if rank == 0:
        shards = [(lists[0][r], lists[1][r], int(lists[2][r].item())) for r in range(world)]
        np.save(out, pdist.merge_shards(pkg, scene, gs.tile_count, shards))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_bench_step_on_emulated_devices(emulated, pkg, tmp_path):
    """bench.py's N > 1 step with everything but the transport real: two processes, each with its own (emulated) device rendering the
    tiles t = rank (mod 2) through pg_render into its shard buffers, torch.distributed gather to rank 0, Film merge -- the image is
    the reference's golden within the device tolerance and equals the single-process render bit for bit."""
    import numpy as np
    import torch.multiprocessing as mp
    name, out = "cornell_40x24", str(tmp_path / "two_ranks.npy")
    mp.spawn(_rank, args=(2, 29561, emulated, name, out), nprocs=2, join=True)
    img = np.load(out)
    ref = pkg.read_pfm(os.path.join(ROOT, "tests", "golden", name + ".pfm"))
    assert img.shape == ref.shape and (np.abs(img - ref) / np.maximum(1, np.abs(ref))).max() <= 1e-4
    single = str(tmp_path / "one_rank.npy")
    mp.spawn(_rank, args=(1, 29562, emulated, name, single), nprocs=1, join=True)
    assert np.array_equal(img, np.load(single))


def test_bench_launches_its_own_ranks_on_emulated_devices(emulated, pkg, tmp_path):
    """`python bench.py --gpus 2` exactly as the driver calls it -- no torchrun around it, WORLD_SIZE unset: the script starts its two
    ranks itself, each renders its tile shard on its own (emulated) device, ONE packed gather per frame (double-buffered, overlapping
    the next frame) brings the shards to rank 0, and the JSON line reports both ranks.  The image equals the one-rank run bit for bit."""
    import json
    import numpy as np
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PBRT_GPU_LIB=emulated, PBRT_EMULATED_DEVICE="1")
    images = {}
    for n in (2, 1):
        out = str(tmp_path / f"bench{n}.pfm")
        p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--workload", "cornell",
                            "--xres", "40", "--yres", "24", "--spp", "4", "--out", out], cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
        line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
        assert line["n_gpus"] == n and line["ranks_seen"] == n and len(line["per_rank_ms"]) == n and line["steps"] == 2
        assert ("gather" in line["config"]["sharding"]) == (n > 1) or "no gather" in line["config"]["sharding"]
        images[n] = pkg.read_pfm(out)
    assert np.array_equal(images[1], images[2])
    ref = pkg.read_pfm(os.path.join(ROOT, "tests", "golden", "cornell_40x24.pfm"))
    assert images[1].shape == ref.shape
