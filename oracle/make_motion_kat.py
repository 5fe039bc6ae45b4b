#!/usr/bin/env python3
"""Known answers for AnimatedTransform::MotionBounds (core/transform.cpp:1215-1247) from the UNMODIFIED reference (oracle/_ref/ref_probe
`motionbounds`, built by Makefile.ref): tests/golden/motion_bounds_kat.txt, one case per line -- 40 input bit patterns (start matrix, end
matrix, startTime, endTime, a Bounds3f), then `|`, whether the box differs from the union of the ends' boxes, and the box's six bit patterns.
Runs in the build container only; tests/test_motion_bounds.py replays the file anywhere and asks the probe about fresh cases where it exists."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PROBE = os.path.join(HERE, "_ref", "ref_probe")
OUT = os.path.join(HERE, "..", "tests", "golden", "motion_bounds_kat.txt")


def rot(axis, deg):
    a = np.asarray(axis, np.float64)
    a = a / np.linalg.norm(a)
    c, s = np.cos(np.radians(deg)), np.sin(np.radians(deg))
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    m = np.eye(4)
    m[:3, :3] = c * np.eye(3) + s * K + (1 - c) * np.outer(a, a)
    return m


def trs(t, axis, deg, sc):
    m = np.eye(4)
    m[:3, 3] = t
    return m @ rot(axis, deg) @ np.diag(list(sc) + [1.0])


def cases(seed, n):
    """Motions of every kind the front end meets: translation / scale only, rotations from a fraction of a degree (around the hasRotation
    threshold, 0.9995 = cos of 1.8 degrees of quaternion angle) to nearly a full half turn of the quaternion, shear (a rotation after a
    non-uniform scale), reflections, ends that are equal, unit and odd time ranges, boxes of every size including degenerate ones."""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        kind = i % 8
        t0, t1 = rng.uniform(-300, 300, 3), rng.uniform(-300, 300, 3)
        a0, a1 = rng.normal(size=3), rng.normal(size=3)
        d0 = rng.uniform(-180, 180)
        d1 = {0: rng.uniform(-180, 180), 1: d0 + rng.uniform(-8, 8), 2: d0 + rng.uniform(150, 359), 3: d0, 4: rng.uniform(-90, 90),
              5: d0 + rng.uniform(-4, 4), 6: rng.uniform(-180, 180), 7: rng.uniform(-180, 180)}[kind]
        s0 = rng.uniform(0.3, 2.5, 3)
        s1 = s0 if kind in (1, 5) else rng.uniform(0.3, 2.5, 3)
        if kind in (1, 2, 5): a1 = a0
        if kind == 6: s1 = s1 * np.array([1, -1, 1])
        m0, m1 = trs(t0, a0, d0, s0), trs(t1, a1, d1, s1)
        if kind == 3: m1 = m0.copy() if i % 16 == 3 else trs(t1, a0, d0, s0)
        if kind == 7: m1 = m1 @ rot(rng.normal(size=3), rng.uniform(-60, 60)) @ np.diag([1.0, rng.uniform(0.5, 2), 1, 1])  # rotation after a scale: shear
        times = (0.0, 1.0) if i % 3 else tuple(sorted(rng.uniform(-1, 2, 2)))
        lo = rng.uniform(-200, 200, 3)
        ext = rng.uniform(0, 150, 3) * (0 if i % 11 == 10 else 1)
        out.append(np.concatenate([m0.ravel(), m1.ravel(), times, lo, lo + ext]).astype(np.float32))
    return out


def composite_cases(seed, n):
    """The motions of the reference's own test (src/tests/animatedtransform.cpp:9-28, RandomTransform): each end the product of ten random factors --
    a scale by up to 10, a translation, a rotation by up to 200 degrees about a random axis -- so that the decompositions see shear; a random box."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        ends = []
        for _e in range(2):
            m = np.eye(4)
            for _f in range(10):
                k = rng.integers(0, 3)
                r = lambda: rng.uniform(-10, 10)
                if k == 0: m = m @ np.diag([abs(r()), abs(r()), abs(r()), 1.0])
                elif k == 1: m = m @ trs((r(), r(), r()), (0, 1, 0), 0, (1, 1, 1))
                else: m = m @ rot(rng.normal(size=3), r() * 20)
            ends.append(m)
        a, b = rng.uniform(-10, 10, 3), rng.uniform(-10, 10, 3)
        out.append(np.concatenate([ends[0].ravel(), ends[1].ravel(), (0.0, 1.0), np.minimum(a, b), np.maximum(a, b)]).astype(np.float32))
    return out


def edge_cases():
    """A singular end (Scale 0: the reference reports "Singular matrix in MatrixInvert" and carries on with what the elimination left), scales 40
    orders of magnitude apart, half a turn and a hair less, a rotation of a thousandth of a degree."""
    out = []
    for sc0, sc1, d in (((0, 1, 1), (1, 1, 1), 40), ((1, 1, 1), (1, 0, 1), 90), ((1e-20, 1, 1), (1, 1, 1e20), 120), ((1, 1, 1), (1, 1, 1), 180),
                        ((1, 1, 1), (1, 1, 1), 179.99), ((2, 2, 2), (2, 2, 2), 0.001)):
        m0, m1 = trs((1, 2, 3), (0, 1, 0), 0, sc0), trs((4, 5, 6), (0.2, 1, 0.1), d, sc1)
        out.append(np.concatenate([m0.ravel(), m1.ravel(), (0.0, 1.0), (-1, -1, -1), (1, 2, 3)]).astype(np.float32))
    return out


def ask(cs):
    text = "\n".join(" ".join("%08x" % u for u in c.view(np.uint32)) for c in cs) + "\n"
    res = subprocess.run([PROBE, "motionbounds"], input=text, capture_output=True, text=True, check=True).stdout.split("\n")
    return [l for l in res if l.strip()]


def main():
    if not os.path.exists(PROBE):
        sys.exit("build oracle/_ref first: make -C oracle -f Makefile.ref")
    cs = cases(20260925, 400) + edge_cases()
    ans = ask(cs)
    assert len(ans) == len(cs)
    with open(OUT, "w") as f:
        for c, a in zip(cs, ans):
            f.write(" ".join("%08x" % u for u in c.view(np.uint32)) + " | " + a.strip() + "\n")
    print(OUT, len(cs), "cases,", sum(a.startswith("1") for a in ans), "whose box is more than the ends' boxes")


if __name__ == "__main__":
    main()
