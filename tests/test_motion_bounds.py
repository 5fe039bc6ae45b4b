"""AnimatedTransform::MotionBounds (core/transform.cpp:1215-1247) on the host front end (host/motion_bounds.cpp): the box of a moving shape or
instance whose motion ROTATES -- what the top-level BVH, the scene's world bound and through it the distant / infinite lights and the spatial
light grid are built from.  Bit for bit against the unmodified reference: 406 committed known answers (oracle/make_motion_kat.py ->
tests/golden/motion_bounds_kat.txt), fresh cases live where the reference is built, and the property the reference's own test checks
(src/tests/animatedtransform.cpp: the box holds the moving points)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KAT = os.path.join(ROOT, "tests", "golden", "motion_bounds_kat.txt")
PROBE = os.path.join(ROOT, "oracle", "_ref", "ref_probe")


def kat_module():
    """oracle/make_motion_kat.py (the case generator and the reference probe's caller), loaded by path: oracle/ itself stays off sys.path"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_motion_kat", os.path.join(ROOT, "oracle", "make_motion_kat.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def host_bounds(pkg, case):
    c = np.ascontiguousarray(case, dtype=np.float32)
    out = np.zeros(6, np.float32)
    rot = pkg.host_lib().pbrt_host_motion_bounds(c[0:16].ctypes.data, c[16:32].ctypes.data, C.c_float(c[32]), C.c_float(c[33]), c[34:40].ctypes.data, out.ctypes.data)
    return rot, out


def test_known_answers_from_the_reference(pkg):
    n = more = 0
    for line in open(KAT):
        ins, outs = line.split("|")
        case = np.array([int(x, 16) for x in ins.split()], dtype=np.uint32).view(np.float32)
        want = outs.split()
        rot, got = host_bounds(pkg, case)
        assert ["%08x" % u for u in got.view(np.uint32)] == want[1:], f"case {n}: {got} vs {np.array([int(x, 16) for x in want[1:]], dtype=np.uint32).view(np.float32)}"
        if want[0] == "1":  # a box that is more than the union of the ends' boxes is a rotation's
            assert rot == 1
            more += 1
        n += 1
    assert n == 406 and more > 150  # 400 random motions + oracle/make_motion_kat.py's edge cases (a singular end, extreme scales, half a turn)


def test_fresh_cases_against_the_reference_live(pkg):
    if not os.path.exists(PROBE):
        pytest.skip("oracle/_ref/ref_probe is built only where /root/reference exists")
    mk = kat_module()
    cs = mk.cases(7, 1500) + mk.composite_cases(5, 500)  # (the second kind: the motions of the reference's own tests/animatedtransform.cpp)
    for k, (c, a) in enumerate(zip(cs, mk.ask(cs))):
        _, got = host_bounds(pkg, c)
        assert ["%08x" % u for u in got.view(np.uint32)] == a.split()[1:], f"case {k}"


def test_the_box_holds_the_moving_corners(pkg):
    """The reference's own test (tests/animatedtransform.cpp:38-60): points of the object box, moved to random times, lie inside MotionBounds
    (up to the float error of the interpolation, for which that test grows the box by 1e-4 of its diagonal)."""
    mk = kat_module()
    rng = np.random.default_rng(3)
    checked = 0
    for c in mk.cases(11, 60) + mk.composite_cases(13, 40):
        rot, box = host_bounds(pkg, c)
        if not rot:
            continue
        diag = np.linalg.norm(box[3:] - box[:3])
        lo, hi = c[34:37].astype(np.float64), c[37:40].astype(np.float64)
        # the motion itself, restated in float64 from the decomposition the test derives independently (polar decomposition by SVD)
        def dec(m):
            M = m.reshape(4, 4).astype(np.float64)
            U, s, Vt = np.linalg.svd(M[:3, :3])
            R = U @ Vt
            return M[:3, 3], R, R.T @ M[:3, :3]
        def quat(R):
            w = np.sqrt(max(0, 1 + R[0, 0] + R[1, 1] + R[2, 2])) / 2
            if w < 1e-6:
                return None
            return np.array([(R[2, 1] - R[1, 2]) / (4 * w), (R[0, 2] - R[2, 0]) / (4 * w), (R[1, 0] - R[0, 1]) / (4 * w), w])
        T0, R0, S0 = dec(c[0:16])
        T1, R1, S1 = dec(c[16:32])
        if np.linalg.det(c[0:16].reshape(4, 4)[:3, :3]) < 0 or np.linalg.det(c[16:32].reshape(4, 4)[:3, :3]) < 0:
            continue  # the reference's iteration and the SVD choose different factors for a reflection
        q0, q1 = quat(R0), quat(R1)
        if q0 is None or q1 is None:
            continue
        if q0 @ q1 < 0:
            q1 = -q1
        th = np.arccos(np.clip(q0 @ q1, -1, 1))
        checked += 1
        for _ in range(40):
            dt = rng.uniform(0, 1)
            p = lo + rng.integers(0, 2, 3) * (hi - lo)
            qp = q1 - q0 * (q0 @ q1)
            qp /= np.linalg.norm(qp)
            q = q0 * np.cos(th * dt) + qp * np.sin(th * dt)
            x, y, z, w = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            pw = (1 - dt) * T0 + dt * T1 + R @ (((1 - dt) * S0 + dt * S1) @ p)
            assert np.all(pw >= box[:3] - 1e-4 * diag) and np.all(pw <= box[3:] + 1e-4 * diag), (pw, box)
    assert checked >= 40
