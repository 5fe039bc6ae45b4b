"""The drop-in, demonstrated: oracle/_ref/pbrt_gpubind is the UNMODIFIED reference (its parser, api.cpp state machine, shape /
material / light construction, BVHAccel build, Film and image output) with `Integrator "path"` / `"volpath"` bound to the
C ABI of include/pbrt_gpu.h by oracle/gpupath_binding.cpp.  Nothing of this repository's own front
end is involved in these renders: reference parser + reference BVH + device kernels must give the reference's image, and --
since both front ends hand the device the same scene -- the very image this repository's own front end produces."""
import os
import subprocess

import numpy as np
import pytest

import json

from conftest import GOLD, ROOT, parse_reference_stats

pytestmark = pytest.mark.gpu  # module-level constants are shared with tests/test_binding_cpu.py
BINDING = os.path.join(ROOT, "oracle", "_ref", "pbrt_gpubind")
SCENES = ["cornell_32", "cornell_crop", "cornell_lens", "cornell_plastic", "cornell_normals", "cornell_tangents", "cornell_ply",
          "cornell_loopsubdiv", "cornell_point", "cornell_spot_power", "cornell_power", "cornell_uniform", "cornell_mirror_glass",
          "cornell_orennayar", "cornell_ortho_lens", "cornell_twosided", "cornell_reverse", "cornell_xform", "cornell_filmopts",
          "filter_gaussian", "filter_mitchell_crop", "filter_widebox", "mat_uber", "mat_metal", "mat_substrate", "mat_translucent", "mat_mix",
          "mat_roughglass", "sphere_light", "sphere_partial", "quadric_lights", "hlbvh_synthetic", "synthetic_n40", "sobol_cornell",
          "sobol_round_crop", "vol_fog", "vol_smoke", "vol_path_none_glass", "sobol_vol_smoke", "sampler_random", "sampler_stratified",
          "sampler_stratified_dims", "filter_02sequence_lens", "sampler_maxmindist", "sampler_lowdisc_vol", "many_lights",
          # a moving camera: the binding hands over the REFERENCE's own decomposition of the two camera transforms (AnimatedTransform's T / R / S)
          "camanim_translate", "camanim_rotate", "camanim_small_rotate", "camanim_times_scale", "camanim_ortho", "camanim_vol",
          # TransformedPrimitives: object instances (a BVHAccel's nodes / primitives appended, a lone primitive), and MOVING shapes / instances --
          # the reference's own AnimatedTransform (both ends, T / R / S, the times) handed over, interpolated per ray on the device
          "instance_boxes", "instance_accel", "motion_boxes", "motion_boxes_times", "motion_small_rotation", "motion_instances",
          "motion_instances_shutter", "motion_sobol", "motion_random", "motion_stratified", "motion_vol", "motion_camera_too",
          # motions that rotate (hasRotation): the binding keeps the reference's own BVH, this repository's front end computes MotionBounds itself
          "motion_rotate_boxes", "motion_rotate_big_times", "motion_rotate_instances", "motion_rotate_distant_spatial", "motion_rotate_vol",
          "motion_rotate_camera_too",
          # a GridDensityMedium, BSSRDF materials, and both in one scene (round 6: the device refused the pair before)
          "grid_puff", "sss_subsurface", "grid_sss_puff", "grid_sss_random",
          # moving shapes inside object definitions (ABI 29): the reference's TransformedPrimitive under a TransformedPrimitive, flattened
          "nest_motion", "nest_motion_moving_instances", "nest_motion_rotate", "nest_motion_vol", "nest_motion_random", "sss_nest_motion", "sss_nest_motion_volpath"]


def run_binding(pkg, scene_file, out):
    if not os.path.exists(BINDING):
        pytest.fail("oracle/_ref/pbrt_gpubind missing: __graft_entry__.build() makes it where /root/reference exists, and it travels with the tree")
    env = dict(os.environ, PBRT_GPU_LIB=pkg.GPU_LIB_PATH)
    p = subprocess.run([BINDING, "--outfile", out, scene_file], capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    run_binding.stdout = p.stdout
    return pkg.read_pfm(out)


@pytest.mark.parametrize("name", SCENES)
def test_reference_front_end_plus_device_equals_reference_image(gpu, name, tmp_path):
    img = run_binding(gpu, os.path.join(GOLD, name + ".pbrt"), str(tmp_path / "bound.pfm"))
    ref = gpu.read_pfm(os.path.join(GOLD, name + ".pfm"))
    assert img.shape == ref.shape
    # the unmodified reference's parser, scene construction, BVHAccel and Film around the device kernels: the reference's own image, bit for bit
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), f"{(img != ref).any(axis=2).sum()} pixels differ from the reference binary's image"
    # and the reference's printed statistics (rays, zero-radiance paths, path length, volume / surface interactions): the device's counters under the
    # reference's own titles (gpupath_binding.cpp ReportDeviceStats) = what the CPU integrator printed for the golden
    printed, want = parse_reference_stats(run_binding.stdout), json.load(open(os.path.join(GOLD, name + ".json")))
    for k in want:
        if k != "tri_tests": assert printed.get(k, 0 if isinstance(want[k], int) else None) == want[k], (k, printed.get(k), want[k])
    # the same scene through this repository's own front end: identical film, bit for bit (same nodes, same primitive order,
    # same BxDF lists -- the specialised matte / plastic / mirror / glass kernels equal the BxDF-list kernels exactly)
    own, _ = gpu.render_scene(gpu.HostScene(os.path.join(GOLD, name + ".pbrt")))
    assert np.array_equal(own, img), f"{(own != img).any(axis=2).sum()} pixels differ between the two front ends"

