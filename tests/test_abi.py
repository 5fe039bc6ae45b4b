"""The C-ABI boundary: struct layouts match include/*.h and both libraries export every declared symbol.
No compute is launched here (runs without a GPU)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


def declared_functions(header, prefix):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(%s\w+)\s*\(" % prefix, txt)))


def test_struct_layouts_match_header(pkg, tmp_path):
    src = tmp_path / "probe.c"
    names = ["PgBVHNode", "PgMaterial", "PgLight", "PgSphere", "PgTexRef", "PgTexture", "PgImage", "PgMedium", "PgDensityGrid", "PgBSSRDF", "PgAlphaMask", "PgTexturedMaterial", "PgBxDF", "PgObject", "PgInstance", "PgSceneDesc", "PgRenderDesc", "PgFilmPixel", "PgStraySample", "PgCounters"]
    body = "\n".join(f'size_t size_{n}(void) {{ return sizeof({n}); }}' for n in names)
    offs = [("PgSceneDesc", "perm_sums"), ("PgSceneDesc", "grid_density"), ("PgSceneDesc", "bssrdf_tables"), ("PgBSSRDF", "table"), ("PgDensityGrid", "world_to_medium"), ("PgRenderDesc", "tile_step"), ("PgRenderDesc", "pixel_bounds"), ("PgCounters", "render_ms"),
            ("PgBVHNode", "axis"), ("PgStraySample", "weight")]
    body += "\n" + "\n".join(f'size_t off_{s}_{f}(void) {{ return offsetof({s}, {f}); }}' for s, f in offs)
    src.write_text(f'#include <stddef.h>\n#include "{ROOT}/include/pbrt_gpu.h"\n{body}\n')
    so = tmp_path / "probe.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)])
    lib = C.CDLL(str(so))
    for n in names:
        f = getattr(lib, "size_" + n); f.restype = C.c_size_t
        assert f() == C.sizeof(getattr(pkg.abi, n)), n
    for s, fld in offs:
        f = getattr(lib, f"off_{s}_{fld}"); f.restype = C.c_size_t
        assert f() == getattr(getattr(pkg.abi, s), fld).offset, (s, fld)
    assert C.sizeof(pkg.abi.PgBVHNode) == 32  # == pbrt's LinearBVHNode (bvh.cpp:95-104)
    assert pkg.FILM_PIXEL_DTYPE.itemsize == C.sizeof(pkg.abi.PgFilmPixel) == 16
    assert pkg.STRAY_DTYPE.itemsize == C.sizeof(pkg.abi.PgStraySample) == 32


def test_python_tables_cover_every_declared_symbol(pkg):
    assert declared_functions("pbrt_gpu.h", "pg_") == sorted(pkg.abi.GPU_SYMBOLS)
    assert declared_functions("pbrt_host.h", "pbrt_host_") == sorted(pkg.abi.HOST_SYMBOLS)


def test_host_library_exports_all(pkg):
    lib = pkg.host_lib()
    for name in declared_functions("pbrt_host.h", "pbrt_host_"):
        assert hasattr(lib, name), name


def test_gpu_library_loads_and_exports_all(pkg):
    """libpbrt_gpu.so must exist (built by __graft_entry__.build()) and export the whole ABI."""
    if not os.path.exists(pkg.GPU_LIB_PATH):
        pytest.fail(f"{pkg.GPU_LIB_PATH} missing: run python __graft_entry__.py (hipcc cross-compiles gfx950 without a GPU)")
    lib = pkg.gpu_lib()
    for name in declared_functions("pbrt_gpu.h", "pg_"):
        assert hasattr(lib, name), name
    # error reporting works without a device: a null scene is rejected, not crashed on
    assert lib.pg_render_tile_count(None) < 0
    assert b"null" in lib.pg_last_error()


def test_missing_extension_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg, "_gpu", None)
    monkeypatch.setattr(pkg, "GPU_LIB_PATH", "/nonexistent/libpbrt_gpu.so")
    with pytest.raises(pkg.PbrtGpuError):
        pkg.gpu_lib()
