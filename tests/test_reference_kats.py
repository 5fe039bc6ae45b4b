"""The reference's own known-answer / property tests for this path, re-run against
the CPU oracle (and, under -m gpu, against the HIP kernels in test_gpu_parity.py):
  Triangle.BadCases        src/tests/shapes.cpp:544-559   literal vector, must NOT hit
  Triangle.Watertight      src/tests/shapes.cpp:28-129    seeded jittered sphere, every ray hits
  Triangle.Reintersect     src/tests/shapes.cpp:154-205   spawned rays never re-hit their triangle
  LowDiscrepancy.RadicalInverse / ScrambledRadicalInverse  src/tests/sampling.cpp:15-74
  FloatingPoint Next{Up,Down}  src/tests/fp_tests.cpp (via OffsetRayOrigin rounding)
"""
import ctypes as C

import numpy as np
import pytest

from kat_util import PCG32, jittered_sphere, uniform_sample_sphere


def tri_hit(oracle, p0, p1, p2, o, d, tmax=np.inf):
    t = C.c_float()
    b = (C.c_float * 3)()
    arr = [np.ascontiguousarray(x, np.float32) for x in (p0, p1, p2, o, d)]
    r = oracle.lib().oracle_triangle_intersect(*[a.ctypes.data for a in arr], C.c_float(tmax), C.byref(t), b)
    return bool(r), t.value, tuple(b)


def test_triangle_bad_cases(oracle):
    p = [(-1113.45459, -79.049614, -56.2431908), (-1113.45459, -87.0922699, -56.2431908), (-1113.45459, -79.2090149, -56.2431908)]
    hit, _, _ = tri_hit(oracle, *p, (-1081.47925, 99.9999542, 87.7701111), (-32.1072998, -183.355865, -144.607635), 0.9999)
    assert not hit


def test_triangle_watertight(pkg, oracle):
    """Every ray from inside the closed jittered sphere hits (also when aimed exactly at a vertex)."""
    verts, idx = jittered_sphere()
    scene = pkg.HostScene(text=sphere_scene(verts, idx))
    o, d = watertight_rays(verts, 4000)
    prim, t, bary, _ = oracle.intersect(scene.desc, o, d, np.full(len(o), np.inf, np.float32))
    assert (prim >= 0).all()


def sphere_scene(verts, idx):
    P = " ".join("%.9g" % v for v in np.asarray(verts, np.float32).reshape(-1))
    I = " ".join(str(i) for i in idx)
    return ('Camera "perspective"\nFilm "image" "integer xresolution" [16] "integer yresolution" [16] "string filename" "x.pfm"\n'
            'WorldBegin\nShape "trianglemesh" "integer indices" [ %s ] "point P" [ %s ]\nWorldEnd\n' % (I, P))


def watertight_rays(verts, n):
    os_, ds = [], []
    for i in range(n // 2):
        rng = PCG32(i)
        u = (rng.uniform_float(), rng.uniform_float())
        p = np.float32(0.5) * uniform_sample_sphere(u)
        u = (rng.uniform_float(), rng.uniform_float())
        os_.append(p); ds.append(uniform_sample_sphere(u))
        v = verts[rng.uniform_uint32_bounded(len(verts))]
        os_.append(p); ds.append((np.asarray(v, np.float32) - p).astype(np.float32))
    return np.asarray(os_, np.float32), np.asarray(ds, np.float32)


def test_triangle_reintersect(oracle):
    """SpawnRay / SpawnRayTo from a hit point never re-intersect the same triangle (pins pError,
    OffsetRayOrigin and the conservative t > deltaT test)."""
    lib = oracle.lib()
    checked = 0
    for i in range(60):
        rng = PCG32(i)
        def pexp():
            logu = np.float32((1 - (u := rng.uniform_float())) * -8.0 + u * 8)  # Lerp(u, -8, 8)
            sign = -1.0 if rng.uniform_float() < 0.5 else 1.0
            return np.float32(sign * 10.0 ** float(logu))
        v = np.array([[pexp() for _ in range(3)] for _ in range(3)], np.float32)
        if np.sum(np.cross((v[1] - v[0]).astype(np.float64), (v[2] - v[0]).astype(np.float64)) ** 2) < 1e-20:
            continue
        u0, u1 = rng.uniform_float(), rng.uniform_float()
        su0 = np.float32(np.sqrt(np.float32(u0)))
        b0, b1 = np.float32(1) - su0, np.float32(u1) * su0
        ptri = (b0 * v[0] + b1 * v[1] + (np.float32(1) - b0 - b1) * v[2]).astype(np.float32)
        o = np.array([pexp() for _ in range(3)], np.float32)
        hit, t, b = tri_hit(oracle, v[0], v[1], v[2], o, (ptri - o).astype(np.float32))
        if not hit:
            continue
        b = np.asarray(b, np.float32)
        phit = (b[0] * v[0] + b[1] * v[1] + b[2] * v[2]).astype(np.float32)
        gamma7 = np.float32(7 * 2.0 ** -24) / np.float32(1 - 7 * 2.0 ** -24)
        perr = (gamma7 * (np.abs(b[0] * v[0]) + np.abs(b[1] * v[1]) + np.abs(b[2] * v[2]))).astype(np.float32)
        n = np.cross((v[0] - v[2]).astype(np.float64), (v[1] - v[2]).astype(np.float64))
        n = (n / np.linalg.norm(n)).astype(np.float32)
        out = np.zeros(3, np.float32)
        for j in range(200):
            w = uniform_sample_sphere((rng.uniform_float(), rng.uniform_float()))
            lib.oracle_spawn_ray_origin(phit.ctypes.data, perr.ctypes.data, n.ctypes.data, w.ctypes.data, out.ctypes.data)
            assert not tri_hit(oracle, v[0], v[1], v[2], out.copy(), w)[0]
            p2 = np.array([pexp() for _ in range(3)], np.float32)
            dvec = (p2 - phit).astype(np.float32)
            lib.oracle_spawn_ray_origin(phit.ctypes.data, perr.ctypes.data, n.ctypes.data, dvec.ctypes.data, out.ctypes.data)
            assert not tri_hit(oracle, v[0], v[1], v[2], out.copy(), dvec, tmax=1 - 0.0001)[0]
            checked += 1
    assert checked > 2000


def reintersect_cases(oracle, n_tris=24, n_rays=120):
    """Triangle.Reintersect (src/tests/shapes.cpp:154-205) as scene-level queries: for each random triangle (coordinates from 1e-8 to
    1e8) a one-triangle scene and the rays the test spawns from a hit point -- SpawnRay in random directions (tMax = inf) and SpawnRayTo
    random points (tMax = 1 - ShadowEpsilon), origins offset by OffsetRayOrigin.  Yields (scene text, o, d, tmax); no ray may hit."""
    lib = oracle.lib()
    made = 0
    for i in range(200):
        if made == n_tris:
            break
        rng = PCG32(1000 + i)
        def _pexp(rng):  # shapes.cpp's pExp: +-10^Lerp(u, -8, 8)
            u = rng.uniform_float()
            logu = np.float32((1 - u) * -8.0 + u * 8)
            sign = -1.0 if rng.uniform_float() < 0.5 else 1.0
            return np.float32(sign * 10.0 ** float(logu))
        v = np.array([[_pexp(rng) for _ in range(3)] for _ in range(3)], np.float32)
        if np.sum(np.cross((v[1] - v[0]).astype(np.float64), (v[2] - v[0]).astype(np.float64)) ** 2) < 1e-20:
            continue
        u0, u1 = rng.uniform_float(), rng.uniform_float()
        su0 = np.float32(np.sqrt(np.float32(u0)))
        b0, b1 = np.float32(1) - su0, np.float32(u1) * su0
        ptri = (b0 * v[0] + b1 * v[1] + (np.float32(1) - b0 - b1) * v[2]).astype(np.float32)
        o = np.array([_pexp(rng) for _ in range(3)], np.float32)
        hit, t, b = tri_hit(oracle, v[0], v[1], v[2], o, (ptri - o).astype(np.float32))
        if not hit:
            continue
        b = np.asarray(b, np.float32)
        phit = (b[0] * v[0] + b[1] * v[1] + b[2] * v[2]).astype(np.float32)
        gamma7 = np.float32(7 * 2.0 ** -24) / np.float32(1 - 7 * 2.0 ** -24)
        perr = (gamma7 * (np.abs(b[0] * v[0]) + np.abs(b[1] * v[1]) + np.abs(b[2] * v[2]))).astype(np.float32)
        n = np.cross((v[0] - v[2]).astype(np.float64), (v[1] - v[2]).astype(np.float64))
        n = (n / np.linalg.norm(n)).astype(np.float32)
        os_, ds, tm = [], [], []
        out = np.zeros(3, np.float32)
        for j in range(n_rays):
            w = uniform_sample_sphere((rng.uniform_float(), rng.uniform_float()))
            lib.oracle_spawn_ray_origin(phit.ctypes.data, perr.ctypes.data, n.ctypes.data, w.ctypes.data, out.ctypes.data)
            os_.append(out.copy()); ds.append(w); tm.append(np.inf)
            p2 = np.array([_pexp(rng) for _ in range(3)], np.float32)
            dvec = (p2 - phit).astype(np.float32)
            lib.oracle_spawn_ray_origin(phit.ctypes.data, perr.ctypes.data, n.ctypes.data, dvec.ctypes.data, out.ctypes.data)
            os_.append(out.copy()); ds.append(dvec); tm.append(np.float32(1 - 0.0001))
        text = ('Film "image" "integer xresolution" [8] "integer yresolution" [8] "string filename" "r.pfm"\nWorldBegin\n'
                'Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ %s ]\nWorldEnd\n' % " ".join("%.9g" % x for x in v.ravel()))
        made += 1
        yield text, np.asarray(os_, np.float32), np.asarray(ds, np.float32), np.asarray(tm, np.float32)


def test_triangle_reintersect_through_the_scene(pkg, oracle):
    """The same property through Scene::Intersect / IntersectP over the flattened one-triangle scene (front end, BVH leaf, the oracle's
    scene-level entry points) -- the form tests/test_gpu_parity.py::test_triangle_reintersect_on_device runs on the HIP kernels."""
    n = 0
    for text, o, d, tmax in reintersect_cases(oracle):
        scene = pkg.HostScene(text=text)
        assert scene.desc.n_tris == 1
        prim, _, _, _ = oracle.intersect(scene.desc, o, d, tmax)
        occ, _ = oracle.intersect_p(scene.desc, o, d, tmax)
        assert (prim < 0).all() and not occ.any()
        n += len(tmax)
    assert n >= 24 * 240


def test_radical_inverse_base2_is_bit_reversal(oracle):
    lib = oracle.lib()
    for a in range(1024):
        rev = int(f"{a:032b}"[::-1], 2)
        assert lib.oracle_radical_inverse(0, a) == np.float32(np.float32(rev) * np.float32(2.3283064365386963e-10))


def test_scrambled_radical_inverse_vs_naive(oracle):
    lib = oracle.lib()
    primes = [p for p in range(2, 800) if all(p % q for q in range(2, int(p ** 0.5) + 1))][:128]
    for dim, base in enumerate(primes):
        rng = PCG32(dim)
        perm = list(range(base - 1, -1, -1))
        for i in range(base):  # Shuffle, sampling.h:151-157
            other = i + rng.uniform_uint32_bounded(base - i)
            perm[i], perm[other] = perm[other], perm[i]
        parr = np.asarray(perm, np.uint16)
        for index in (0, 1, 2, 1151, 32351, 4363211, 681122):
            val, inv_base, a = 0.0, 1.0 / base, index
            inv_bi = inv_base
            for _ in range(32):
                val += perm[a % base] * inv_bi
                a //= base
                inv_bi *= inv_base
            got = lib.oracle_scrambled_radical_inverse(dim, index, parr.ctypes.data)
            assert abs(val - got) < 1e-5


def test_radical_inverse_unscrambled_matches_identity_permutation(oracle):
    """ScrambledRadicalInverse with the identity permutation == RadicalInverse (generator identity)."""
    lib = oracle.lib()
    for dim, base in ((1, 3), (2, 5), (5, 13)):
        ident = np.arange(base, dtype=np.uint16)
        for a in (0, 1, 7, 1000, 123456, 2 ** 31 + 5, 2 ** 40 + 12345):
            assert lib.oracle_radical_inverse(dim, a) == lib.oracle_scrambled_radical_inverse(dim, a, ident.ctypes.data)


SOBOL_SCENE = """LookAt 0 0 5  0 0 0  0 1 0
Camera "perspective"
Film "image" "integer xresolution" [ 10 ] "integer yresolution" [ 10 ] "string filename" "s.pfm"
Sampler "sobol" "integer pixelsamples" [ %d ]
PixelFilter "box"
Integrator "path"
WorldBegin
Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point P" [ -1 -1 0  1 -1 0  0 1 0 ]
WorldEnd
"""


def test_sobol_first_dimension_is_bit_reversal(pkg, oracle):
    """LowDiscrepancy.Sobol, tests/sampling.cpp:127-131: dimension 0 of SobolSampleFloat is the base-2 radical inverse."""
    scene = pkg.HostScene(text=SOBOL_SCENE % 4)
    lib = oracle.lib()
    for i in range(8192):
        rev = int(f"{i:032b}"[::-1], 2)
        assert lib.oracle_sobol_sample(scene.desc, i, 0) == np.float32(np.float32(rev) * np.float32(2.3283064365386963e-10))


@pytest.mark.parametrize("log_samples", range(2, 11))
def test_sobol_elementary_intervals(pkg, oracle, log_samples):
    """LowDiscrepancy.ElementaryIntervals for the SobolSampler, tests/sampling.cpp:136-188: SobolSampler(2^k, Bounds2i((0,0),
    (10,10))), pixel (0, 0): the 2^k samples' first Get2D() puts exactly one sample into every elementary interval of every
    shape (2^i x 2^(k-i)).  Exercises CreateSobolSampler (resolution 16), SobolIntervalToIndex and the pixel remapping."""
    scene = pkg.HostScene(text=SOBOL_SCENE % (1 << log_samples))
    rd = scene.render_desc()
    assert rd.sampler == 1 and rd.spp == 1 << log_samples and rd.sobol_resolution == 16 and rd.sobol_log2_resolution == 4
    lib = oracle.lib()
    pts = np.array([[lib.oracle_sampler_dimension(scene.desc, rd, 0, 0, k, d) for d in (0, 1)] for k in range(rd.spp)], np.float32)
    assert (pts >= 0).all() and (pts < 1).all()
    for i in range(log_samples + 1):
        nx, ny = 1 << i, 1 << (log_samples - i)
        idx = np.floor(np.float32(ny) * pts[:, 1]).astype(int) * nx + np.floor(np.float32(nx) * pts[:, 0]).astype(int)
        assert len(set(idx.tolist())) == rd.spp


# ---- HenyeyGreenstein.* (tests/hg.cpp) ---------------------------------------------------------------------------------
def test_hg_sampling_match(oracle):
    """HenyeyGreenstein.SamplingMatch: the value Sample_p returns is p(wo, wi) of the direction it produced."""
    lib, rng = oracle.lib(), PCG32()
    for g in np.arange(-0.75, 0.76, 0.25, dtype=np.float32):
        for _ in range(100):
            wo = uniform_sample_sphere((rng.uniform_float(), rng.uniform_float()))
            u = np.array([rng.uniform_float(), rng.uniform_float()], np.float32)
            wi = np.zeros(3, np.float32)
            p0 = lib.oracle_hg_sample_p(float(g), wo.ctypes.data, u.ctypes.data, wi.ctypes.data)
            assert abs(p0 - lib.oracle_phase_hg(float(np.dot(wo, wi)), float(g))) < 1e-4, g


@pytest.mark.parametrize("g", [0.95, -0.95])
def test_hg_sampling_orientation(oracle, g):
    """HenyeyGreenstein.SamplingOrientationForward / Backward: with |g| = 0.95 nearly all samples continue / reverse."""
    lib, rng = oracle.lib(), PCG32()
    wo = np.array([-1, 0, 0], np.float32)
    fwd = 0
    for _ in range(100):
        u = np.array([rng.uniform_float(), rng.uniform_float()], np.float32)
        wi = np.zeros(3, np.float32)
        lib.oracle_hg_sample_p(g, wo.ctypes.data, u.ctypes.data, wi.ctypes.data)
        fwd += wi[0] > 0
    assert (fwd >= 10 * (100 - fwd)) if g > 0 else ((100 - fwd) >= 10 * fwd)


def test_hg_normalized(oracle):
    """HenyeyGreenstein.Normalized: the phase function integrates to 1 (its mean over the sphere is 1 / 4 pi)."""
    lib = oracle.lib()
    rng = np.random.default_rng(1)
    z = 1 - 2 * rng.random(100000)
    for g in (-0.75, -0.5, -0.25, 0.0, 0.25, 0.5, 0.75):
        mean = np.mean([lib.oracle_phase_hg(float(c), g) for c in z[:20000]])
        assert abs(mean - 1 / (4 * np.pi)) < 1e-3, g


# ---- FullSphere / PartialSphere / Cylinder .Reintersect and PartialSphere.Normal (tests/shapes.cpp:372-513) -----------------
QUADRIC_SCENE = """LookAt 0 0 5  0 0 0  0 1 0
Camera "perspective"
Film "image" "integer xresolution" [ 4 ] "integer yresolution" [ 4 ] "string filename" "q.pfm"
WorldBegin
%s
WorldEnd
"""


def _pexp(rng, e=8.0):
    u = rng.uniform_float()
    return np.float32(10.0 ** float(np.float32((1 - u) * -e + u * e)))


def _reintersect_convex(pkg, oracle, shape_text, rng, n_out=150, device=None):
    """TestReintersectConvex: no ray leaving a hit point into the normal's hemisphere (SpawnRay / SpawnRayTo) hits the shape again.
    The spawned rays are traced by the oracle (also by its correctly-rounded-libm build, whose arithmetic is the device's) and, with
    `device` = the package on a GPU box, by the HIP kernels."""
    import ctypes as C
    scene = pkg.HostScene(text=QUADRIC_SCENE % shape_text)
    lib = oracle.lib()
    o = np.array([_pexp(rng) for _ in range(3)], np.float32)
    nodes = scene.nodes()
    lo, hi = nodes["bmin"][0], nodes["bmax"][0]
    tt = np.array([rng.uniform_float() for _ in range(3)], np.float32)
    p2 = ((1 - tt) * lo + tt * hi).astype(np.float32)  # bbox.Lerp(t)
    d = (p2 - o).astype(np.float32)
    if rng.uniform_float() < .5: d = (d / np.float32(np.sqrt(np.sum(d * d, dtype=np.float32)))).astype(np.float32)
    t, out = C.c_float(0), np.zeros(9, np.float32)
    if not lib.oracle_intersect_interaction(scene.desc, o.ctypes.data, d.ctypes.data, np.inf, C.byref(t), out.ctypes.data): return 0, None
    p, perr, n = out[:3].copy(), out[3:6].copy(), out[6:].copy()
    orig = np.zeros(3, np.float32)
    os_, ds, tm = [], [], []
    for _ in range(n_out):
        w = uniform_sample_sphere((rng.uniform_float(), rng.uniform_float()))
        if np.dot(w, n) < 0: w = -w  # Faceforward(w, isect.n)
        lib.oracle_spawn_ray_origin(p.ctypes.data, perr.ctypes.data, n.ctypes.data, w.ctypes.data, orig.ctypes.data)
        os_.append(orig.copy()); ds.append(w); tm.append(np.inf)
        q = np.array([_pexp(rng) for _ in range(3)], np.float32)
        w = (q - p).astype(np.float32)
        if np.dot(w, n) < 0: w = -w
        lib.oracle_spawn_ray_origin(p.ctypes.data, perr.ctypes.data, n.ctypes.data, w.ctypes.data, orig.ctypes.data)
        os_.append(orig.copy()); ds.append(w); tm.append(1 - 0.0001)  # SpawnRayTo: tMax = 1 - ShadowEpsilon
    os_, ds, tm = np.asarray(os_, np.float32), np.asarray(ds, np.float32), np.asarray(tm, np.float32)
    for cr in (False,):  # (one oracle: the device computes libm as the reference does, pg_libm.h)
        prim, _, _, _ = oracle.intersect(scene.desc, os_, ds, tm)
        occ, _ = oracle.intersect_p(scene.desc, os_, ds, tm)
        assert (prim < 0).all() and not occ.any()
    if device is not None:
        gs = device.GpuScene(scene.desc)
        dprim, _, _ = gs.intersect(os_, ds, tm)
        docc = gs.intersect_p(os_, ds, tm)
        gs.close()
        assert (dprim < 0).all() and not np.asarray(docc).any()
    return len(os_), (p, n)


@pytest.mark.parametrize("kind", ["full_sphere", "partial_sphere", "cylinder"])
def test_quadric_reintersect(pkg, oracle, kind):
    quadric_reintersect_run(pkg, oracle, kind)


def quadric_reintersect_run(pkg, oracle, kind, device=None):
    checked = 0
    for i in range(40):
        rng = PCG32(i)
        radius = _pexp(rng, 4)
        if kind == "cylinder":
            zmin = _pexp(rng, 4) * (-1 if rng.uniform_float() < 0.5 else 1)
            zmax = _pexp(rng, 4) * (-1 if rng.uniform_float() < 0.5 else 1)
        elif kind == "partial_sphere":
            zmin = -radius if rng.uniform_float() < 0.5 else (lambda u: (1 - u) * -radius + u * radius)(np.float32(rng.uniform_float()))
            zmax = radius if rng.uniform_float() < 0.5 else (lambda u: (1 - u) * -radius + u * radius)(np.float32(rng.uniform_float()))
        else: zmin, zmax = -radius, radius
        phimax = 360.0 if (kind == "full_sphere" or rng.uniform_float() < 0.5) else rng.uniform_float() * 360.0
        shape = ('Shape "%s" "float radius" [ %.9g ] "float zmin" [ %.9g ] "float zmax" [ %.9g ] "float phimax" [ %.9g ]'
                 % ("cylinder" if kind == "cylinder" else "sphere", radius, zmin, zmax, phimax))
        n, hit = _reintersect_convex(pkg, oracle, shape, rng, device=device)
        checked += n
        if hit and kind == "partial_sphere":  # ParialSphere.Normal: the normal of an untransformed sphere is radial
            p, nrm = hit
            assert abs(1 - np.dot(nrm / np.linalg.norm(nrm), p / np.linalg.norm(p))) < 1e-5
    assert checked > 1500


def test_triangle_sampling_solid_angle(pkg, oracle):
    """Triangle.Sampling, tests/shapes.cpp:210-271: the solid angle a triangle subtends, estimated with Triangle::Sample(ref, u)
    (sum of 1 / pdf) and with uniform directions (fraction of rays that hit it), agree -- both over radical-inverse points."""
    lib = oracle.lib()
    count, n_pdf = 1 << 19, 1 << 13  # uniform directions need many more samples than the importance-sampled estimate
    js = np.arange(count, dtype=np.uint64)
    # RadicalInverse bases 2 and 3 (lowdiscrepancy.cpp:389-436), vectorised; the oracle's own function is checked elsewhere
    def radinv(base, a):
        inv, v, f = np.zeros(len(a)), a.copy(), 1.0 / base
        while v.any():
            inv += (v % base) * f
            v //= base
            f /= base
        return inv
    us_all = np.stack([radinv(2, js), radinv(3, js)], 1).astype(np.float32)
    z = 1 - 2 * us_all[:, 0]
    r = np.sqrt(np.maximum(0, 1 - z * z))
    phi = 2 * np.pi * us_all[:, 1]
    dirs = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(np.float32)
    us = us_all[:n_pdf]
    compared = 0
    for i in range(12):
        rng = PCG32(i)
        punif = lambda r=10.0: np.float32((1 - (u := rng.uniform_float())) * -r + u * r)
        v = np.array([[punif() for _ in range(3)] for _ in range(3)], np.float32)
        if np.sum(np.cross((v[1] - v[0]).astype(np.float64), (v[2] - v[0]).astype(np.float64)) ** 2) < 1e-20: continue
        pc = np.array([punif(), punif(), punif()], np.float32)
        pc[rng.uniform_uint32() % 3] = (-13.0 if rng.uniform_float() > .5 else 13.0)
        tri = ("AttributeBegin\nAreaLightSource \"diffuse\" \"bool twosided\" \"true\"\nShape \"trianglemesh\" \"integer indices\" [ 0 1 2 ] \"point P\" [ %s ]\nAttributeEnd"
               % " ".join("%.9g" % x for x in v.ravel()))
        scene = pkg.HostScene(text=QUADRIC_SCENE % tri)
        occ, _ = oracle.intersect_p(scene.desc, np.tile(pc, (count, 1)), dirs, np.full(count, np.inf, np.float32))
        unif = occ.sum() / (count * (1 / (4 * np.pi)))
        wi, ps = np.zeros(3, np.float32), np.zeros(3, np.float32)
        est = 0.0
        for u in us:
            pdf = lib.oracle_light_sample_pdf(scene.desc, 0, pc.ctypes.data, u.ctypes.data, wi.ctypes.data, ps.ctypes.data)
            assert pdf > 0
            est += 1.0 / (n_pdf * pdf)
        if est > 1e-3:
            err = abs(est - unif) if (abs(est) < 1e-4 or abs(unif) < 1e-4) else abs((est - unif) / unif)
            assert err < .1, (i, est, unif)
            compared += 1
    assert compared >= 5


@pytest.mark.parametrize("shape,p", [('Shape "sphere" "float radius" [ 1 ]', (-.25, -1, .8)), ('Shape "sphere" "float radius" [ 1 ]', (1, .9, -.8)),
                                     ('Shape "cylinder" "float radius" [ .25 ] "float zmin" [ -1 ] "float zmax" [ 1 ]', (.5, .25, .5)),
                                     ('Shape "disk" "float radius" [ 1.25 ]', (.5, -.8, .5))])
def test_quadric_sampling_solid_angle(pkg, oracle, shape, p):
    """Sphere / Cylinder / Disk .SolidAngle, tests/shapes.cpp:331-370: Shape::SolidAngle (the mean of 1 / pdf over Sample(ref, u))
    against the fraction of uniform directions that hit the shape, under Translate(1, .5, -.8) * RotateX(30); a point inside the
    sphere sees 4 pi."""
    lib = oracle.lib()
    text = 'AttributeBegin\nTranslate 1 .5 -.8\nRotate 30 1 0 0\nAreaLightSource "diffuse" "bool twosided" "true"\n%s\nAttributeEnd' % shape
    scene = pkg.HostScene(text=QUADRIC_SCENE % text)
    rng = np.random.default_rng(7)
    n = 1 << 18
    z = 1 - 2 * rng.random(n)
    r = np.sqrt(np.maximum(0, 1 - z * z))
    phi = 2 * np.pi * rng.random(n)
    dirs = np.stack([r * np.cos(phi), r * np.sin(phi), z], 1).astype(np.float32)
    pc = np.asarray(p, np.float32)
    occ, _ = oracle.intersect_p(scene.desc, np.tile(pc, (n, 1)), dirs, np.full(n, np.inf, np.float32))
    mc = occ.mean() * 4 * np.pi
    wi, ps = np.zeros(3, np.float32), np.zeros(3, np.float32)
    m = 1 << 16
    pdfs, targets = [], []
    jj = np.arange(m, dtype=np.uint64)
    def radinv(base, a):
        inv, v, f = np.zeros(len(a)), a.copy(), 1.0 / base
        while v.any():
            inv += (v % base) * f
            v //= base
            f /= base
        return inv
    for u in np.stack([radinv(2, jj), radinv(3, jj)], 1).astype(np.float32):
        pdfs.append(lib.oracle_light_sample_pdf(scene.desc, 0, pc.ctypes.data, u.ctypes.data, wi.ctypes.data, ps.ctypes.data))
        targets.append(ps.copy())
    pdfs, targets = np.asarray(pdfs), np.asarray(targets, np.float32)
    # Shape::SolidAngle (shape.cpp:89-103) leaves out samples the shape itself hides: IntersectP(Ray(p, pShape.p - p, .999f))
    hidden, _ = oracle.intersect_p(scene.desc, np.tile(pc, (m, 1)), (targets - pc).astype(np.float32), np.full(m, .999, np.float32))
    ok = (pdfs > 0) & ~hidden.astype(bool)
    est = float(np.sum(1.0 / pdfs[ok]) / m)
    assert abs(est - mc) < 0.03 * max(1.0, mc), (est, mc)
