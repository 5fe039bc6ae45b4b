"""The compiled reference-side binding (oracle/_ref/pbrt_gpubind = the unmodified reference + oracle/gpupath_binding.cpp) checked
WITHOUT a GPU: PBRT_GPU_LIB points it at oracle/liboracle_abi_shim.so, which answers the C ABI of include/pbrt_gpu.h with
the CPU restatement.  The reference's own parser, scene construction and BVHAccel, flattened by the binding into
PgSceneDesc / PgRenderDesc, rendered by the oracle and merged by the reference's own Film, must reproduce the reference's
golden image bit for bit -- so the flattening the drop-in relies on is exact.  (tests/test_gpu_binding.py runs the same binary
against libpbrt_gpu.so on the GPU box.)"""
import os
import subprocess

import numpy as np
import pytest

import json

from conftest import GOLD, ROOT, parse_reference_stats
from test_gpu_binding import BINDING, SCENES

SHIM = os.path.join(ROOT, "oracle", "liboracle_abi_shim.so")


@pytest.fixture(scope="module")
def shim():
    if os.path.isdir("/root/reference/src"):  # keep the binding in step with include/pbrt_gpu.h (a no-op when it is up to date)
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "-f", "Makefile.ref", "-j8", "_ref/pbrt_gpubind"])
    if not os.path.exists(BINDING):
        pytest.skip("oracle/_ref/pbrt_gpubind is built only where /root/reference exists")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle_abi_shim.so"])
    return SHIM


@pytest.mark.parametrize("name", SCENES)
def test_binding_flattening_is_exact(pkg, shim, name, tmp_path):
    out = str(tmp_path / "bound.pfm")
    p = subprocess.run([BINDING, "--outfile", out, os.path.join(GOLD, name + ".pbrt")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PBRT_GPU_LIB=shim))
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    img, ref = pkg.read_pfm(out), pkg.read_pfm(os.path.join(GOLD, name + ".pfm"))
    assert img.shape == ref.shape and np.array_equal(img, ref), f"{(img != ref).any(axis=2).sum()} pixels differ, max |d| {np.abs(img - ref).max():.3e}"
    # the statistics the reference prints after the render are the CPU integrator's: the binding hands PgCounters to the reference's own
    # StatsAccumulator under the reference's titles (gpupath_binding.cpp ReportDeviceStats).  (Not the ray-triangle line: a hit count is not kept.)
    printed, want = parse_reference_stats(p.stdout), json.load(open(os.path.join(GOLD, name + ".json")))
    for k in want:
        if k != "tri_tests": assert printed.get(k, 0 if isinstance(want[k], int) else None) == want[k], (k, printed.get(k), want[k])


def test_binding_reports_what_it_cannot_flatten(shim, tmp_path):
    """Outside its closed set the binding stops with an Error, it does not fall back to the reference's CPU integrator."""
    out = str(tmp_path / "x.pfm")
    p = subprocess.run([BINDING, "--outfile", out, os.path.join(GOLD, "tex_checker.pbrt")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PBRT_GPU_LIB=shim))
    assert p.returncode != 0 and "outside the device path's closed set" in (p.stdout + p.stderr) and not os.path.exists(out)
