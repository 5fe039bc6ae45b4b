"""ctypes wrapper of liboracle.so (the CPU restatement, pbrt_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg -- never by the product package.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.path.join(_HERE, "liboracle.so")
REF_BINARY = os.path.join(_HERE, "_ref", "pbrt_oracle")
_lib = None
_libs = {}


def _pkg():
    sys.path.insert(0, _ROOT) if _ROOT not in sys.path else None
    from __graft_entry__ import load_package
    return load_package()


_libm_ok = None


def host_libm_matches():
    """Does THIS host's libm compute sinf / cosf / logf / expf / acosf / atanf / atan2f as csrc/pg_libm.h -- i.e. as the glibc 2.35 FMA
    variants the goldens were rendered with?  The device never depends on the host's libm; liboracle.so and the reference binary do.  On a
    host where this is False (another glibc, a CPU without FMA3) their images differ from the device's in last bits although nothing is
    wrong, so bit-exact comparisons against them say nothing there: smoke() then compares within BASELINE.json's 1e-4, the `oracle` test
    fixture skips, and the device-vs-golden tests (files, host-independent) remain the check.  A strided sample of 2^22 arguments per
    function (tests/test_libm_restated.py sweeps all 2^32); compiled on first use, cached for the process."""
    global _libm_ok
    if _libm_ok is None:
        import tempfile
        try:
            so = os.path.join(tempfile.mkdtemp(prefix="pbrt_libm_probe_"), "libm_pin.so")
            flags = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []  # (without the instruction __builtin_fma is a libm call: correct, slower)
            subprocess.check_call(["g++", "-O2", "-ffp-contract=off", *flags, "-fopenmp", "-fPIC", "-shared", os.path.join(_ROOT, "tests", "libm_pin.cpp"), "-o", so, "-lm"],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            L = C.CDLL(so)
            L.pin_unary.restype = C.c_longlong
            L.pin_unary.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_longlong, C.POINTER(C.c_uint32)]
            L.pin_atan2f.restype = C.c_longlong
            L.pin_atan2f.argtypes = [C.c_uint64, C.c_longlong, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
            bad = sum(L.pin_unary(fn, 12345, 1021, 1 << 22, None) for fn in range(7)) + L.pin_atan2f(7, 1 << 22, 0, None, None)
            _libm_ok = bad == 0
        except Exception as e:  # no compiler here: assume the image's own glibc (the one the goldens come from)
            sys.stderr.write(f"oracle: host libm probe could not run ({e}); assuming it matches csrc/pg_libm.h\n")
            _libm_ok = True
    return _libm_ok


def lib():
    """liboracle.so (built on demand)."""
    global _lib
    path = LIB_PATH
    if path in _libs:
        return _libs[path]
    if True:
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", _HERE, os.path.basename(path)])
        pkg = _pkg()
        abi = pkg.abi
        L = C.CDLL(path)
        L.oracle_render_tile_count.restype = C.c_int
        L.oracle_render_tile_count.argtypes = [C.POINTER(abi.PgRenderDesc)]
        L.oracle_render.restype = C.c_int
        L.oracle_render.argtypes = [C.POINTER(abi.PgSceneDesc), C.POINTER(abi.PgRenderDesc), C.c_void_p, C.c_void_p,
                                    C.c_int32, C.POINTER(C.c_int32), C.POINTER(abi.PgCounters)]
        L.oracle_intersect.restype = C.c_int
        L.oracle_intersect.argtypes = [C.POINTER(abi.PgSceneDesc), C.c_int32] + [C.c_void_p] * 6 + [C.POINTER(abi.PgCounters)]
        L.oracle_intersect_p.restype = C.c_int
        L.oracle_intersect_p.argtypes = [C.POINTER(abi.PgSceneDesc), C.c_int32] + [C.c_void_p] * 4 + [C.POINTER(abi.PgCounters)]
        L.oracle_radical_inverse.restype = C.c_float
        L.oracle_radical_inverse.argtypes = [C.c_int, C.c_uint64]
        L.oracle_scrambled_radical_inverse.restype = C.c_float
        L.oracle_scrambled_radical_inverse.argtypes = [C.c_int, C.c_uint64, C.c_void_p]
        L.oracle_halton_index.restype = C.c_int64
        L.oracle_halton_index.argtypes = [C.POINTER(abi.PgRenderDesc), C.c_int, C.c_int, C.c_int64]
        L.oracle_sobol_sample.restype = C.c_float
        L.oracle_sobol_sample.argtypes = [C.POINTER(abi.PgSceneDesc), C.c_int64, C.c_int]
        L.oracle_sampler_dimension.restype = C.c_float
        L.oracle_sampler_dimension.argtypes = [C.POINTER(abi.PgSceneDesc), C.POINTER(abi.PgRenderDesc), C.c_int, C.c_int, C.c_int64, C.c_int]
        L.oracle_halton_sample.restype = C.c_float
        L.oracle_halton_sample.argtypes = [C.POINTER(abi.PgSceneDesc), C.POINTER(abi.PgRenderDesc), C.c_int64, C.c_int]
        L.oracle_triangle_intersect.restype = C.c_int
        L.oracle_triangle_intersect.argtypes = [C.c_void_p] * 5 + [C.c_float, C.POINTER(C.c_float), C.c_void_p]
        L.oracle_phase_hg.restype = C.c_float
        L.oracle_phase_hg.argtypes = [C.c_float, C.c_float]
        L.oracle_hg_sample_p.restype = C.c_float
        L.oracle_hg_sample_p.argtypes = [C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_intersect_interaction.restype = C.c_int
        L.oracle_intersect_interaction.argtypes = [C.POINTER(abi.PgSceneDesc), C.c_void_p, C.c_void_p, C.c_float, C.POINTER(C.c_float), C.c_void_p]
        L.oracle_light_sample_pdf.restype = C.c_float
        L.oracle_light_sample_pdf.argtypes = [C.POINTER(abi.PgSceneDesc), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.oracle_spawn_ray_origin.restype = None
        L.oracle_spawn_ray_origin.argtypes = [C.c_void_p] * 5
        L.oracle_generate_ray.restype = None
        L.oracle_generate_ray.argtypes = [C.c_void_p] + [C.c_float] * 4 + [C.c_void_p]
        L.oracle_grid_density.restype = C.c_float
        L.oracle_grid_density.argtypes = [C.c_void_p] * 3
        L.oracle_grid_tr.restype = C.c_float
        L.oracle_grid_tr.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
        L.oracle_grid_sample.restype = C.c_int
        L.oracle_grid_sample.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_void_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]
        L.oracle_bssrdf_pdf_sp.restype = C.c_float
        L.oracle_bssrdf_pdf_sp.argtypes = [C.c_void_p] * 6
        L.oracle_bssrdf_probe_segment.restype = C.c_int
        L.oracle_bssrdf_probe_segment.argtypes = [C.c_void_p] * 4 + [C.c_float] * 3 + [C.c_void_p]
        L.oracle_bssrdf_radial.restype = None
        L.oracle_bssrdf_radial.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.oracle_lobe_f_pdf.restype = None
        L.oracle_lobe_f_pdf.argtypes = [C.c_void_p] * 4
        L.oracle_lobe_sample_f.restype = C.c_int
        L.oracle_lobe_sample_f.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p]
        L.oracle_find_interval_le.restype = C.c_int
        L.oracle_find_interval_le.argtypes = [C.c_int, C.c_void_p, C.c_float]
        L.oracle_catmull_rom_weights.restype = C.c_int
        L.oracle_catmull_rom_weights.argtypes = [C.c_int, C.c_void_p, C.c_float, C.POINTER(C.c_int), C.c_void_p]
        L.oracle_sample_catmull_rom_2d.restype = C.c_float
        L.oracle_sample_catmull_rom_2d.argtypes = [C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_float, C.c_float]
        L.oracle_invert_catmull_rom.restype = C.c_float
        L.oracle_invert_catmull_rom.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_float]
        L.oracle_fresnel_moment1.restype = C.c_float
        L.oracle_fresnel_moment1.argtypes = [C.c_float]
        _libs[path] = L
        _lib = L
    return _libs[path]


def render(desc, rd, max_strays=None):
    """oracle_render: same outputs as pg_render (film, strays) plus the reference's counters."""
    pkg = _pkg()
    L = lib()
    n = L.oracle_render_tile_count(C.byref(rd))
    if max_strays is None:
        max_strays = pkg.default_max_strays(rd, n)
    film = np.zeros(n * rd.tile_pixels, pkg.FILM_PIXEL_DTYPE)
    strays = np.zeros(max_strays, pkg.STRAY_DTYPE)
    ns = C.c_int32(0)
    cn = pkg.abi.PgCounters()
    st = L.oracle_render(C.byref(desc), C.byref(rd), film.ctypes.data, strays.ctypes.data, max_strays, C.byref(ns), C.byref(cn))
    if st != 0:
        raise RuntimeError(f"oracle_render failed: {st}")
    return film, strays[:ns.value], cn.as_dict()


def render_image(scene):
    """Full-frame oracle render of a HostScene, merged by the host Film: (h, w, 3) image + counters."""
    rd = scene.render_desc()
    film, strays, cn = render(scene.desc, rd)
    scene.film_clear()
    scene.film_merge(rd, film, strays)
    return scene.film_image(), cn


def intersect(desc, o, d, tmax):
    pkg = _pkg()
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); tmax = np.ascontiguousarray(tmax, np.float32)
    n = len(tmax)
    prim = np.empty(n, np.int32); t = np.empty(n, np.float32); bary = np.empty((n, 3), np.float32)
    cn = pkg.abi.PgCounters()
    lib().oracle_intersect(C.byref(desc), n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, prim.ctypes.data, t.ctypes.data,
                           bary.ctypes.data, C.byref(cn))
    return prim, t, bary, cn.as_dict()


def intersect_p(desc, o, d, tmax):
    pkg = _pkg()
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); tmax = np.ascontiguousarray(tmax, np.float32)
    n = len(tmax)
    occ = np.empty(n, np.uint8)
    cn = pkg.abi.PgCounters()
    lib().oracle_intersect_p(C.byref(desc), n, o.ctypes.data, d.ctypes.data, tmax.ctypes.data, occ.ctypes.data, C.byref(cn))
    return occ, cn.as_dict()


def run_reference(scene_file, out_pfm, nthreads=None, quiet=True, timeout=3600):
    """Render scene_file with the UNMODIFIED reference binary (oracle/_ref/pbrt_oracle). Returns its stdout.  (`timeout`: the reference does not
    terminate on some degenerate inputs -- a mirrored motion, for one -- and a sweep must not wait for it for ever.)"""
    if not os.path.exists(REF_BINARY):
        raise FileNotFoundError(f"{REF_BINARY} not built (make -C oracle ref; needs /root/reference)")
    cmd = [REF_BINARY, "--outfile", out_pfm]
    if nthreads:
        cmd += ["--nthreads", str(nthreads)]
    cmd.append(scene_file)
    return subprocess.run(cmd, check=True, capture_output=True, text=True, timeout=timeout).stdout
