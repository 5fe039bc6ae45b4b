"""csrc/pg_libm.h -- the device's sinf / cosf / sincosf / logf / expf / acosf / atanf / atan2f -- against the system's libm, the one the
reference binary links (glibc 2.35, which picks its FMA variants at run time on this box and on the GPU box's host): compiled for the
host and compared over ALL 2^32 arguments of every unary function and 2^32 pairs (uniform bit patterns, equal and near exponents) plus
the edge-value grid for atan2f.  Zero differences (any NaN equals any NaN).  This is what lets tests/test_gpu_parity.py demand images
IDENTICAL to the reference binary's."""
import ctypes as C
import os
import subprocess

import pytest

from conftest import ROOT

NAMES = ["sinf", "cosf", "sincosf", "logf", "expf", "acosf", "atanf"]


@pytest.fixture(scope="module")
def pin(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("libm") / "libm_pin.so")
    # -ffp-contract=off: nothing but the header's explicit fma calls is fused, as in the device build; -mfma: they compile to the instruction
    subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-mfma", "-fopenmp", "-fPIC", "-shared", os.path.join(ROOT, "tests", "libm_pin.cpp"), "-o", so, "-lm"])
    lib = C.CDLL(so)
    lib.pin_unary.restype = C.c_longlong
    lib.pin_unary.argtypes = [C.c_int, C.c_uint32, C.c_uint32, C.c_longlong, C.POINTER(C.c_uint32)]
    lib.pin_atan2f.restype = C.c_longlong
    lib.pin_atan2f.argtypes = [C.c_uint64, C.c_longlong, C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    return lib


def has_fma():
    try:
        return " fma " in open("/proc/cpuinfo").read()
    except OSError:
        return False


@pytest.mark.skipif(not has_fma(), reason="glibc selects its non-FMA variants on this CPU; the goldens and pg_libm.h are the FMA ones")
@pytest.mark.parametrize("fn", range(len(NAMES)), ids=NAMES)
def test_every_float_argument(pin, fn):
    bad = C.c_uint32(0)
    n = pin.pin_unary(fn, 0, 1, 1 << 32, C.byref(bad))
    assert n == 0, f"{NAMES[fn]}: {n} of 2^32 arguments differ from the system libm, e.g. bits {bad.value:#010x}"


def test_atan2f_pairs(pin):
    by, bx = C.c_uint32(0), C.c_uint32(0)
    n = pin.pin_atan2f(1, 1 << 32, 0, C.byref(by), C.byref(bx))
    assert n == 0, f"atan2f: {n} of 2^32 pairs differ, e.g. y {by.value:#010x} x {bx.value:#010x}"
    n = pin.pin_atan2f(1, 22 * 22, 1, C.byref(by), C.byref(bx))
    assert n == 0, f"atan2f: edge pair y {by.value:#010x} x {bx.value:#010x} differs"
