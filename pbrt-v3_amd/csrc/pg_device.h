// pg_device.h -- device-side arithmetic of the MI355X path tracer (gfx950).
//
// Every function keeps the reference's float operation order (citations are to
// /root/reference/src) and the translation unit is compiled with
// -ffp-contract=off and correctly rounded fp32 divide/sqrt, so a lane computes
// the same IEEE results as the CPU reference, which emits no FMA either.
// std::min/std::max semantics ((b<a)?b:a) are spelled out instead of fminf/fmaxf.
#ifndef PG_DEVICE_H
#define PG_DEVICE_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "pg_libm.h"  // sinf / cosf / acosf / atan2f / logf / expf as the reference's libm computes them

#define PG_DEV __device__ __forceinline__
#define PG_HD __host__ __device__ inline
// nothing is scheduled across this point: keeps independent long sequences (three IEEE divisions, say) from being interleaved where their
// temporaries together would set the kernel's register peak (a no-op for values, and for the host builds of tests/)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(HIP_EMU_H)
#define PG_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
#else
#define PG_SCHED_BARRIER() ((void)0)
#endif

struct V3 { float x, y, z; };
struct Spec { float r, g, b; };

// core/pbrt.h:196-215, rng.h:52
#define PG_INF __builtin_huge_valf()
#define PG_MACH_EPS 5.9604644775390625e-08f
#define PG_SHADOW_EPS 0.0001f
#define PG_PI 3.14159265358979323846f
#define PG_INVPI 0.31830988618379067154f
#define PG_PIOVER2 1.57079632679489661923f
#define PG_PIOVER4 0.78539816339744830961f
#define PG_ONE_MINUS_EPS 0x1.fffffep-1f

PG_DEV float pgamma(int n) { return (n * PG_MACH_EPS) / (1 - n * PG_MACH_EPS); }  // pbrt.h:289-291
PG_DEV float pmin(float a, float b) { return (b < a) ? b : a; }
PG_DEV float pmax(float a, float b) { return (a < b) ? b : a; }
PG_DEV float plerp(float t, float v1, float v2) { return (1 - t) * v1 + t * v2; }  // pbrt.h:417

PG_HD V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PG_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
PG_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
PG_HD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
PG_HD V3 operator*(V3 a, float s) { return mk(s * a.x, s * a.y, s * a.z); }  // geometry.h:232-234
PG_HD V3 vdiv(V3 a, float f) { float inv = 1.f / f; return mk(a.x * inv, a.y * inv, a.z * inv); }  // geometry.h:243-248
PG_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PG_DEV float absdot(V3 a, V3 b) { return fabsf(dot(a, b)); }
PG_HD float lensq(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
PG_HD V3 normalize(V3 a) { return vdiv(a, sqrtf(lensq(a))); }  // geometry.h:984-986
PG_HD V3 cross(V3 a, V3 b) {  // geometry.h:957-963, evaluated in double
    double ax = a.x, ay = a.y, az = a.z, bx = b.x, by = b.y, bz = b.z;
    return mk((float)((ay * bz) - (az * by)), (float)((az * bx) - (ax * bz)), (float)((ax * by) - (ay * bx)));
}
PG_DEV V3 vabs(V3 a) { return mk(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
PG_DEV float maxcomp(V3 v) { return pmax(v.x, pmax(v.y, v.z)); }  // geometry.h:993-995
PG_HD void coordinate_system(V3 v1, V3 &v2, V3 &v3) {  // geometry.h:1020-1028
    if (fabsf(v1.x) > fabsf(v1.y)) v2 = vdiv(mk(-v1.z, 0, v1.x), sqrtf(v1.x * v1.x + v1.z * v1.z));
    else v2 = vdiv(mk(0, v1.z, -v1.y), sqrtf(v1.y * v1.y + v1.z * v1.z));
    v3 = cross(v1, v2);
}
PG_DEV float next_float_up(float v) {  // pbrt.h:241-253
    if (isinf(v) && v > 0.f) return v;
    if (v == -0.f) v = 0.f;
    uint32_t ui = __float_as_uint(v);
    if (v >= 0) ++ui; else --ui;
    return __uint_as_float(ui);
}
PG_DEV float next_float_down(float v) {  // pbrt.h:255-265
    if (isinf(v) && v < 0.f) return v;
    if (v == 0.f) v = -0.f;
    uint32_t ui = __float_as_uint(v);
    if (v > 0) --ui; else ++ui;
    return __uint_as_float(ui);
}
PG_DEV float round_away(float po, float off) {
    if (off > 0) return next_float_up(po);
    else if (off < 0) return next_float_down(po);
    return po;
}
PG_DEV V3 offset_ray_origin(V3 p, V3 pError, V3 n, V3 w) {  // geometry.h:1440-1454
    float d = dot(vabs(n), pError);
    V3 offset = n * d;
    if (dot(w, n) < 0) offset = -offset;
    V3 po = p + offset;
    po.x = round_away(po.x, offset.x);
    po.y = round_away(po.y, offset.y);
    po.z = round_away(po.z, offset.z);
    return po;
}

// spectrum.h:100-330 (RGBSpectrum)
PG_DEV Spec sp(float v) { Spec s; s.r = s.g = s.b = v; return s; }
PG_DEV Spec sp3(float r, float g, float b) { Spec s; s.r = r; s.g = g; s.b = b; return s; }
PG_DEV Spec operator*(Spec a, Spec b) { return sp3(a.r * b.r, a.g * b.g, a.b * b.b); }
PG_DEV Spec operator*(Spec a, float f) { return sp3(a.r * f, a.g * f, a.b * f); }
PG_DEV Spec operator/(Spec a, float f) { return sp3(a.r / f, a.g / f, a.b / f); }
PG_DEV Spec operator+(Spec a, Spec b) { return sp3(a.r + b.r, a.g + b.g, a.b + b.b); }
PG_DEV bool is_black(Spec a) { return a.r == 0.f && a.g == 0.f && a.b == 0.f; }
PG_DEV float lum(Spec a) { return 0.212671f * a.r + 0.715160f * a.g + 0.072169f * a.b; }  // spectrum.h:462-465
PG_DEV float max_component(Spec a) { float m = a.r; m = pmax(m, a.g); m = pmax(m, a.b); return m; }

// ---------------------------------------------------------------------------
// Triangle::Intersect / IntersectP arithmetic up to the accepted (t, b0, b1, b2),
// shapes/triangle.cpp:200-291 (== :438-517).  The degenerate-triangle rejection
// of :309-317 is a property of the triangle alone and is carried as a
// precomputed per-triangle flag by the caller.
// ---------------------------------------------------------------------------
// The ray-only part of the test (:205-220): the permutation that makes the largest |d| component z, and the
// shear constants.  Computed once per ray by the traversal kernel, per call elsewhere.
struct TriRay { int kz; float Sx, Sy, Sz; };
PG_DEV TriRay tri_ray_setup(V3 dir) {
    float ax = fabsf(dir.x), ay = fabsf(dir.y), az = fabsf(dir.z);
    TriRay r;
    r.kz = (ax > ay) ? ((ax > az) ? 0 : 2) : ((ay > az) ? 1 : 2);  // MaxDimension(Abs(d)), geometry.h:998-1000
    V3 d = r.kz == 0 ? mk(dir.y, dir.z, dir.x) : (r.kz == 1 ? mk(dir.z, dir.x, dir.y) : dir);  // kx = kz+1, ky = kx+1 (mod 3)
    r.Sx = -d.x / d.z; r.Sy = -d.y / d.z; r.Sz = 1.f / d.z;
    return r;
}
PG_DEV V3 tri_permute(V3 v, int kz) {
    return mk(kz == 0 ? v.y : (kz == 1 ? v.z : v.x), kz == 0 ? v.z : (kz == 1 ? v.x : v.y), kz == 0 ? v.x : (kz == 1 ? v.y : v.z));
}
PG_DEV bool tri_test_pre(V3 p0, V3 p1, V3 p2, V3 o, TriRay tr, float tMax, float &t, float &b0, float &b1, float &b2) {
    V3 p0t = tri_permute(p0 - o, tr.kz), p1t = tri_permute(p1 - o, tr.kz), p2t = tri_permute(p2 - o, tr.kz);
    const float Sx = tr.Sx, Sy = tr.Sy, Sz = tr.Sz;
    p0t.x += Sx * p0t.z; p0t.y += Sy * p0t.z;
    p1t.x += Sx * p1t.z; p1t.y += Sy * p1t.z;
    p2t.x += Sx * p2t.z; p2t.y += Sy * p2t.z;
    float e0 = p1t.x * p2t.y - p1t.y * p2t.x;
    float e1 = p2t.x * p0t.y - p2t.y * p0t.x;
    float e2 = p0t.x * p1t.y - p0t.y * p1t.x;
    if (e0 == 0.0f || e1 == 0.0f || e2 == 0.0f) {  // :232-245 FP64 fallback on edges
        double p2txp1ty = (double)p2t.x * (double)p1t.y;
        double p2typ1tx = (double)p2t.y * (double)p1t.x;
        e0 = (float)(p2typ1tx - p2txp1ty);
        double p0txp2ty = (double)p0t.x * (double)p2t.y;
        double p0typ2tx = (double)p0t.y * (double)p2t.x;
        e1 = (float)(p0typ2tx - p0txp2ty);
        double p1txp0ty = (double)p1t.x * (double)p0t.y;
        double p1typ0tx = (double)p1t.y * (double)p0t.x;
        e2 = (float)(p1typ0tx - p1txp0ty);
    }
    if ((e0 < 0 || e1 < 0 || e2 < 0) && (e0 > 0 || e1 > 0 || e2 > 0)) return false;
    float det = e0 + e1 + e2;
    if (det == 0) return false;
    p0t.z *= Sz; p1t.z *= Sz; p2t.z *= Sz;
    float tScaled = e0 * p0t.z + e1 * p1t.z + e2 * p2t.z;
    if (det < 0 && (tScaled >= 0 || tScaled < tMax * det)) return false;
    else if (det > 0 && (tScaled <= 0 || tScaled > tMax * det)) return false;
    float invDet = 1 / det;
    b0 = e0 * invDet; b1 = e1 * invDet; b2 = e2 * invDet;
    t = tScaled * invDet;
    float maxZt = maxcomp(vabs(mk(p0t.z, p1t.z, p2t.z)));
    float deltaZ = pgamma(3) * maxZt;
    float maxXt = maxcomp(vabs(mk(p0t.x, p1t.x, p2t.x)));
    float maxYt = maxcomp(vabs(mk(p0t.y, p1t.y, p2t.y)));
    float deltaX = pgamma(5) * (maxXt + maxZt);
    float deltaY = pgamma(5) * (maxYt + maxZt);
    float deltaE = 2 * (pgamma(2) * maxXt * maxYt + deltaY * maxXt + deltaX * maxYt);
    float maxE = maxcomp(vabs(mk(e0, e1, e2)));
    float deltaT = 3 * (pgamma(3) * maxE * maxZt + deltaE * maxZt + deltaZ * maxE) * fabsf(invDet);
    if (t <= deltaT) return false;
    return true;
}
PG_DEV bool tri_test(V3 p0, V3 p1, V3 p2, V3 o, V3 dir, float tMax, float &t, float &b0, float &b1, float &b2) {
    return tri_test_pre(p0, p1, p2, o, tri_ray_setup(dir), tMax, t, b0, b1, b2);
}

// Whether Triangle::Intersect would reject every hit on this triangle as
// "bogus" (shapes/triangle.cpp:293-317): degenerate uv parameterisation (or
// zero dpdu x dpdv) AND zero geometric normal.  Also yields dpdu for shading.
PG_HD bool tri_dpdu_dpdv(V3 p0, V3 p1, V3 p2, const float uv[6], V3 &dpdu, V3 &dpdv) {
    dpdv = mk(0, 0, 0);
    dpdu = mk(0, 0, 0);
    float duv02x = uv[0] - uv[4], duv02y = uv[1] - uv[5];
    float duv12x = uv[2] - uv[4], duv12y = uv[3] - uv[5];
    V3 dp02 = p0 - p2, dp12 = p1 - p2;
    float determinant = duv02x * duv12y - duv02y * duv12x;
    bool degenerateUV = (double)fabsf(determinant) < 1e-8;
    if (!degenerateUV) {
        float invdet = 1 / determinant;
        dpdu = (dp02 * duv12y - dp12 * duv02y) * invdet;
        dpdv = (dp02 * (-duv12x) + dp12 * duv02x) * invdet;
    }
    if (degenerateUV || lensq(cross(dpdu, dpdv)) == 0) {
        V3 ng = cross(p2 - p0, p1 - p0);
        if (lensq(ng) == 0) return false;
        coordinate_system(normalize(ng), dpdu, dpdv);
    }
    return true;
}
PG_HD bool tri_dpdu(V3 p0, V3 p1, V3 p2, const float uv[6], V3 &dpdu) { V3 dpdv; return tri_dpdu_dpdv(p0, p1, p2, uv, dpdu, dpdv); }

// Bounds3::IntersectP(ray, invDir, dirIsNeg), geometry.h:1412-1438.
PG_DEV bool slab_test(float lox, float loy, float loz, float hix, float hiy, float hiz, V3 o, V3 invDir, bool nx, bool ny,
                      bool nz, float rayTMax) {
    float tMin = ((nx ? hix : lox) - o.x) * invDir.x;
    float tMax = ((nx ? lox : hix) - o.x) * invDir.x;
    float tyMin = ((ny ? hiy : loy) - o.y) * invDir.y;
    float tyMax = ((ny ? loy : hiy) - o.y) * invDir.y;
    const float widen = 1 + 2 * pgamma(3);
    tMax *= widen;
    tyMax *= widen;
    if (tMin > tyMax || tyMin > tMax) return false;
    if (tyMin > tMin) tMin = tyMin;
    if (tyMax < tMax) tMax = tyMax;
    float tzMin = ((nz ? hiz : loz) - o.z) * invDir.z;
    float tzMax = ((nz ? loz : hiz) - o.z) * invDir.z;
    tzMax *= widen;
    if (tMin > tzMax || tzMin > tMax) return false;
    if (tzMin > tMin) tMin = tzMin;
    if (tzMax < tMax) tMax = tzMax;
    return (tMin < rayTMax) && (tMax > 0);
}

// ---------------------------------------------------------------------------
// Low-discrepancy sampling: lowdiscrepancy.{h,cpp}, samplers/halton.cpp
// ---------------------------------------------------------------------------
struct HaltonTables {
    const uint16_t *perms;   // concatenated digit permutations
    const int32_t *permSums; // offset of base d
    const int32_t *primes;   // primes[d]
    int nDims;
};

PG_DEV float radical_inverse_base2(uint64_t a) {  // lowdiscrepancy.cpp:427-435
    uint64_t r = __brevll(a);
    return (float)((double)r * 0x1p-64);
}
// RadicalInverseSpecialized<base>, lowdiscrepancy.cpp:389-403.  Integer digit
// extraction is exact in any word size; 32-bit math is used when a fits.
PG_DEV float radical_inverse(uint32_t base, uint64_t a) {
    const float invBase = 1.f / (float)base;
    uint64_t reversedDigits = 0;
    float invBaseN = 1;
    if ((a >> 32) == 0) {
        uint32_t a32 = (uint32_t)a;
        while (a32) {
            uint32_t next = a32 / base;
            uint32_t digit = a32 - next * base;
            reversedDigits = reversedDigits * base + digit;
            invBaseN *= invBase;
            a32 = next;
        }
    } else {
        while (a) {
            uint64_t next = a / base;
            uint64_t digit = a - next * base;
            reversedDigits = reversedDigits * base + digit;
            invBaseN *= invBase;
            a = next;
        }
    }
    return pmin((float)reversedDigits * invBaseN, PG_ONE_MINUS_EPS);
}
// ScrambledRadicalInverseSpecialized<base>, lowdiscrepancy.cpp:405-424.
PG_DEV float scrambled_radical_inverse(uint32_t base, const uint16_t *perm, uint64_t a) {
    const float invBase = 1.f / (float)base;
    uint64_t reversedDigits = 0;
    float invBaseN = 1;
    if ((a >> 32) == 0) {
        uint32_t a32 = (uint32_t)a;
        while (a32) {
            uint32_t next = a32 / base;
            uint32_t digit = a32 - next * base;
            reversedDigits = reversedDigits * base + perm[digit];
            invBaseN *= invBase;
            a32 = next;
        }
    } else {
        while (a) {
            uint64_t next = a / base;
            uint64_t digit = a - next * base;
            reversedDigits = reversedDigits * base + perm[digit];
            invBaseN *= invBase;
            a = next;
        }
    }
    return pmin(invBaseN * ((float)reversedDigits + invBase * (float)perm[0] / (1 - invBase)), PG_ONE_MINUS_EPS);
}

// ConcentricSampleDisk, sampling.cpp:113-130.  The reference calls libm's
// cosf / sinf: pg_libm.h returns their bits.
PG_DEV void concentric_sample_disk(float u0, float u1, float &dx, float &dy) {
    float ox = 2.f * u0 - 1, oy = 2.f * u1 - 1;
    if (ox == 0 && oy == 0) { dx = 0; dy = 0; return; }
    float theta, r;
    if (fabsf(ox) > fabsf(oy)) { r = ox; theta = PG_PIOVER4 * (oy / ox); }
    else { r = oy; theta = PG_PIOVER2 - PG_PIOVER4 * (ox / oy); }
    float s, c;
    pg_sincosf(theta, &s, &c);
    dx = r * (float)c;
    dy = r * (float)s;
}
PG_DEV V3 cosine_sample_hemisphere(float u0, float u1) {  // sampling.h:159-163
    float dx, dy;
    concentric_sample_disk(u0, u1, dx, dy);
    float z = sqrtf(pmax(0.f, 1 - dx * dx - dy * dy));
    return mk(dx, dy, z);
}
PG_DEV float power_heuristic(int nf, float fPdf, int ng, float gPdf) {  // sampling.h:171-174
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
#endif
