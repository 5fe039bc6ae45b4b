import glob
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scenes"))
GOLD = os.path.join(ROOT, "tests", "golden")
# The parity suite compares the reference's statistics too ("Ray-triangle intersection tests", and this library's node visits): it
# runs shadow rays in the reference's visiting order.  The product's default order for them is free (nearer child first: same
# occlusion answers, fewer nodes); tests/test_gpu_anyhit_order.py renders every golden and random scene in that order and requires
# films, stray samples and ray counts bit-identical to the reference-order render.
os.environ.setdefault("PG_ANYHIT_ORDER", "reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of host CPU time (the reference binary at full size); PBRT_SKIP_SLOW=1 skips")


@pytest.fixture(scope="session")
def pkg():
    """The product package; host library built on demand (g++, seconds)."""
    from __graft_entry__ import PKG_DIR, load_package
    if not os.path.exists(os.path.join(PKG_DIR, "libpbrt_host.so")):
        subprocess.check_call(["make", "-C", PKG_DIR, "host"])
    return load_package()


@pytest.fixture(scope="session")
def oracle(pkg):
    """The CPU restatement (test infrastructure)."""
    from oracle import oracle as o
    # keep the restatement in step with include/pbrt_gpu.h (a no-op when they are up to date)
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    o.lib()
    # liboracle.so (and the reference binary) compute with THIS host's libm; the goldens and the device with glibc 2.35's FMA variants
    # (csrc/pg_libm.h).  On a host whose libm differs, a live oracle differs from the device in last bits although nothing is wrong:
    # those comparisons are skipped there, the device-vs-golden-file tests (host-independent) still run.  INTEGRATION.md section 5.
    if not o.host_libm_matches() and os.environ.get("PBRT_IGNORE_LIBM_MISMATCH") != "1":
        pytest.skip("this host's libm is not the one csrc/pg_libm.h restates (another glibc, or a CPU without FMA3): live-oracle comparisons skipped")
    return o


def check_integrator_stats(cn, stats):
    """PgCounters' integrator statistics (ABI 28) against what the reference binary printed for a golden ("Integrator/Zero-radiance paths", "Path length",
    "Volume interactions", "Surface interactions": path.cpp:45-46, volpath.cpp:45-47; oracle/make_golden.py parse_stats): exactly, the average as printed."""
    if "paths_total" not in stats: return
    for k in ("paths_total", "paths_zero_radiance", "volume_interactions", "surface_interactions"):
        assert cn[k] == stats[k], (k, cn[k], stats[k])
    if "path_length_avg" in stats:
        assert cn["path_length_count"] > 0
        got = "%.3f" % (cn["path_length_sum"] / cn["path_length_count"])  # stats.cpp:141-147
        assert (got, cn["path_length_min"], cn["path_length_max"]) == (stats["path_length_avg"], stats["path_length_min"], stats["path_length_max"]), \
            ("path length", got, cn["path_length_min"], cn["path_length_max"], stats)
    else: assert cn["path_length_count"] == 0


def parse_reference_stats(txt):
    """What a run of the reference binary (or of the binding built on it) printed, in a golden's keys: oracle/make_golden.py parse_stats."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "oracle", "make_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.parse_stats(txt)


INTEGRATOR_STATS = ("paths_total", "paths_zero_radiance", "path_length_sum", "path_length_count", "path_length_min", "path_length_max", "volume_interactions",
                    "surface_interactions")  # PgCounters, ABI 28: device == CPU restatement, field for field


def golden_names():
    return sorted(os.path.basename(p)[:-5] for p in glob.glob(os.path.join(GOLD, "*.json")))


@pytest.fixture(scope="session")
def gpu(pkg):
    if os.environ.get("PBRT_EMULATED_DEVICE") == "1":  # tests/test_emulated_device.py: PBRT_GPU_LIB points at the device sources compiled
        pkg.gpu_lib()                                  # for the host under tests/emu/hip_emu.h -- the same tests, no GPU
        return pkg
    import torch
    if not torch.cuda.is_available():
        pytest.fail("no GPU visible: -m gpu tests must run on the MI355X box")
    pkg.gpu_lib()  # raises if libpbrt_gpu.so is missing -- never a silent fallback
    return pkg
