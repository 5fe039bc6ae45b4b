// pg_sphere.h -- Shape "sphere" on the device: the quadric's error-bounded root search (shapes/sphere.cpp:48-106 over
// core/efloat.h), the with-error Transform operators it needs (core/transform.h:219-384) and the surface interaction at an
// accepted root (sphere.cpp:108-160, interaction.cpp:44-71, transform.cpp:262-297).
//
// A sphere stays in object space as in the reference (ObjectToWorld / WorldToObject are full 4x4 matrices), so every ray is
// carried into object space with the reference's running error bounds and the intersection is decided by interval
// arithmetic: same float operations in the same order, hence the same accept / reject decision and the same tHit.
// The traversal kernel stores only tHit for a sphere hit; shading recomputes the hit point from the ray and tHit, which is
// the arithmetic Sphere::Intersect itself performs after the root is chosen.
#ifndef PG_SPHERE_H
#define PG_SPHERE_H
#include "pg_device.h"
#include "../../include/pbrt_gpu.h"

struct EFloat { float v, low, high; };  // efloat.h:45-233 in an NDEBUG build: value + conservative interval
PG_DEV EFloat ef_make(float v, float err) {  // efloat.h:50-63
    EFloat r; r.v = v;
    if (err == 0.f) r.low = r.high = v;
    else { r.low = next_float_down(v - err); r.high = next_float_up(v + err); }
    return r;
}
PG_DEV EFloat ef_add(EFloat a, EFloat b) { EFloat r; r.v = a.v + b.v; r.low = next_float_down(a.low + b.low); r.high = next_float_up(a.high + b.high); return r; }
PG_DEV EFloat ef_sub(EFloat a, EFloat b) { EFloat r; r.v = a.v - b.v; r.low = next_float_down(a.low - b.high); r.high = next_float_up(a.high - b.low); return r; }
PG_DEV EFloat ef_mul(EFloat a, EFloat b) {  // efloat.h:111-127
    EFloat r; r.v = a.v * b.v;
    const float p0 = a.low * b.low, p1 = a.high * b.low, p2 = a.low * b.high, p3 = a.high * b.high;
    r.low = next_float_down(pmin(pmin(p0, p1), pmin(p2, p3)));
    r.high = next_float_up(pmax(pmax(p0, p1), pmax(p2, p3)));
    return r;
}
PG_DEV EFloat ef_div(EFloat a, EFloat b) {  // efloat.h:128-151
    EFloat r; r.v = a.v / b.v;
    if (b.low < 0 && b.high > 0) { r.low = -PG_INF; r.high = PG_INF; }
    else {
        const float d0 = a.low / b.low, d1 = a.high / b.low, d2 = a.low / b.high, d3 = a.high / b.high;
        r.low = next_float_down(pmin(pmin(d0, d1), pmin(d2, d3)));
        r.high = next_float_up(pmax(pmax(d0, d1), pmax(d2, d3)));
    }
    return r;
}
PG_DEV bool ef_quadratic(EFloat A, EFloat B, EFloat C, EFloat &t0, EFloat &t1) {  // efloat.h:268-288
    const double discrim = (double)B.v * (double)B.v - 4. * (double)A.v * (double)C.v;
    if (discrim < 0.) return false;
    const double rootDiscrim = sqrt(discrim);
    const EFloat frd = ef_make((float)rootDiscrim, (float)((double)PG_MACH_EPS * rootDiscrim));
    EFloat q;
    if (B.v < 0) q = ef_mul(ef_make(-.5f, 0), ef_sub(B, frd));
    else q = ef_mul(ef_make(-.5f, 0), ef_add(B, frd));
    t0 = ef_div(q, A);
    t1 = ef_div(C, q);
    if (t0.v > t1.v) { const EFloat t = t0; t0 = t1; t1 = t; }
    return true;
}

// Transform::operator() flavours, row-major 4x4 (m = the transform's matrix, mInv = its inverse)
PG_DEV V3 m4_point(const float *m, V3 p) {  // transform.h:219-231
    const float x = p.x, y = p.y, z = p.z;
    const float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    const float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    const float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    const float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1) return mk(xp, yp, zp);
    return vdiv(mk(xp, yp, zp), wp);
}
PG_DEV V3 m4_vec(const float *m, V3 v) {  // transform.h:233-239
    return mk(m[0] * v.x + m[1] * v.y + m[2] * v.z, m[4] * v.x + m[5] * v.y + m[6] * v.z, m[8] * v.x + m[9] * v.y + m[10] * v.z);
}
PG_DEV V3 m4_normal(const float *mInv, V3 n) {  // transform.h:241-247
    return mk(mInv[0] * n.x + mInv[4] * n.y + mInv[8] * n.z, mInv[1] * n.x + mInv[5] * n.y + mInv[9] * n.z,
              mInv[2] * n.x + mInv[6] * n.y + mInv[10] * n.z);
}
PG_DEV V3 m4_point_err(const float *m, V3 p, V3 &pError) {  // transform.h:277-300
    const float x = p.x, y = p.y, z = p.z;
    const float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    const float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    const float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    const float wp = (m[12] * x + m[13] * y) + (m[14] * z + m[15]);
    const float xAbsSum = (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
    const float yAbsSum = (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
    const float zAbsSum = (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    pError = mk(xAbsSum, yAbsSum, zAbsSum) * pgamma(3);
    if (wp == 1) return mk(xp, yp, zp);
    return vdiv(mk(xp, yp, zp), wp);
}
// the same with the homogeneous weight wp = (m[12] x + m[13] y) + (m[14] z + m[15]) formed by the caller (k_trace's instance entry: from row 3's
// constants where the row is (0, 0, 0, 1)); m: rows 0 - 2
PG_DEV V3 m4_point_err_w(const float *m, V3 p, float wp, V3 &pError) {
    const float x = p.x, y = p.y, z = p.z;
    const float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    const float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    const float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    const float xAbsSum = (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
    const float yAbsSum = (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
    const float zAbsSum = (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    pError = mk(xAbsSum, yAbsSum, zAbsSum) * pgamma(3);
    if (wp == 1) return mk(xp, yp, zp);
    V3 r;  // a projective matrix (rare): one division after the other
    r.x = xp / wp; PG_SCHED_BARRIER(); r.y = yp / wp; PG_SCHED_BARRIER(); r.z = zp / wp;
    return r;
}
PG_DEV V3 m4_point_err2(const float *m, V3 pt, V3 ptError, V3 &absError) {  // transform.h:302-332
    const float x = pt.x, y = pt.y, z = pt.z;
    const float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    const float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    const float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    const float wp = (m[12] * x + m[13] * y) + (m[14] * z + m[15]);
    const float g3 = pgamma(3);
    absError.x = (g3 + 1.f) * (fabsf(m[0]) * ptError.x + fabsf(m[1]) * ptError.y + fabsf(m[2]) * ptError.z) +
                 g3 * (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
    absError.y = (g3 + 1.f) * (fabsf(m[4]) * ptError.x + fabsf(m[5]) * ptError.y + fabsf(m[6]) * ptError.z) +
                 g3 * (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
    absError.z = (g3 + 1.f) * (fabsf(m[8]) * ptError.x + fabsf(m[9]) * ptError.y + fabsf(m[10]) * ptError.z) +
                 g3 * (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    if (wp == 1.f) return mk(xp, yp, zp);
    return vdiv(mk(xp, yp, zp), wp);
}
PG_DEV V3 m4_vec_err(const float *m, V3 v, V3 &absError) {  // transform.h:334-351
    const float g3 = pgamma(3);
    absError.x = g3 * (fabsf(m[0] * v.x) + fabsf(m[1] * v.y) + fabsf(m[2] * v.z));
    absError.y = g3 * (fabsf(m[4] * v.x) + fabsf(m[5] * v.y) + fabsf(m[6] * v.z));
    absError.z = g3 * (fabsf(m[8] * v.x) + fabsf(m[9] * v.y) + fabsf(m[10] * v.z));
    return m4_vec(m, v);
}

// Transform::operator()(const Ray &), transform.h:249-262, as TransformedPrimitive applies it with WorldToInstance
// (primitive.cpp:80): origin moved to the edge of its error box; dt is what the caller takes off tMax
PG_DEV void instance_ray(const float *w2i, V3 ro, V3 rd, V3 &o, V3 &d, float &dt) {
    V3 oErr;
    o = m4_point_err(w2i, ro, oErr);
    d = m4_vec(w2i, rd);
    const float lengthSquared = lensq(d);
    dt = 0;
    if (lengthSquared > 0) {
        dt = dot(vabs(d), oErr) / lengthSquared;
        o = o + d * dt;
    }
}
// (*WorldToObject)(ray, &oErr, &dErr), transform.h:372-384: the object-space ray, its origin moved to the edge of its error box
PG_DEV void sphere_object_ray(const PgSphere &sp, V3 ro, V3 rd, V3 &o, V3 &d, V3 &oErr, V3 &dErr) {
    o = m4_point_err(sp.w2o, ro, oErr);
    d = m4_vec_err(sp.w2o, rd, dErr);
    const float lengthSquared = lensq(d);
    if (lengthSquared > 0) {
        const float dt = dot(vabs(d), oErr) / lengthSquared;
        o = o + d * dt;
    }
}
// Object-space hit point of parameter t, re-projected onto the sphere, and its azimuth (sphere.cpp:80-87)
PG_DEV V3 sphere_hit_point(const PgSphere &sp, V3 o, V3 d, float t, float &phi) {
    V3 pHit = o + d * t;
    pHit = pHit * (sp.radius / sqrtf(lensq(pHit)));
    if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * sp.radius;
    phi = pg_atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * PG_PI;
    return pHit;
}
PG_DEV bool sphere_clipped(const PgSphere &sp, V3 pHit, float phi) {  // sphere.cpp:90-91
    return (sp.z_min > -sp.radius && pHit.z < sp.z_min) || (sp.z_max < sp.radius && pHit.z > sp.z_max) || phi > sp.phi_max;
}
// Sphere::Intersect's root search = Sphere::IntersectP (sphere.cpp:48-106, :165-200).  True: tHit is the accepted root.
PG_DEV bool sphere_test_s(const PgSphere &sp, V3 ro, V3 rd, float tMax, float &tHit) {
    V3 o, d, oErr, dErr;
    sphere_object_ray(sp, ro, rd, o, d, oErr, dErr);
    const EFloat ox = ef_make(o.x, oErr.x), oy = ef_make(o.y, oErr.y), oz = ef_make(o.z, oErr.z);
    const EFloat dx = ef_make(d.x, dErr.x), dy = ef_make(d.y, dErr.y), dz = ef_make(d.z, dErr.z);
    const EFloat a = ef_add(ef_add(ef_mul(dx, dx), ef_mul(dy, dy)), ef_mul(dz, dz));
    const EFloat b = ef_mul(ef_make(2, 0), ef_add(ef_add(ef_mul(dx, ox), ef_mul(dy, oy)), ef_mul(dz, oz)));
    const EFloat rad = ef_make(sp.radius, 0);
    const EFloat c = ef_sub(ef_add(ef_add(ef_mul(ox, ox), ef_mul(oy, oy)), ef_mul(oz, oz)), ef_mul(rad, rad));
    EFloat t0, t1;
    if (!ef_quadratic(a, b, c, t0, t1)) return false;
    if (t0.high > tMax || t1.low <= 0) return false;
    EFloat tShapeHit = t0;
    if (tShapeHit.low <= 0) {
        tShapeHit = t1;
        if (tShapeHit.high > tMax) return false;
    }
    float phi;
    V3 pHit = sphere_hit_point(sp, o, d, tShapeHit.v, phi);
    if (sphere_clipped(sp, pHit, phi)) {
        if (tShapeHit.v == t1.v) return false;
        if (t1.high > tMax) return false;
        tShapeHit = t1;
        pHit = sphere_hit_point(sp, o, d, tShapeHit.v, phi);
        if (sphere_clipped(sp, pHit, phi)) return false;
    }
    tHit = tShapeHit.v;
    return true;
}

// What the path needs of the SurfaceInteraction a quadric's Intersect builds for the root tHit of world ray (ro, rd).
struct SphereHit { V3 p, pError, wo, n, dpdu, dpdv, dndu, dndv; float u, v; };
// the tail the quadrics share: SurfaceInteraction ctor in object space (interaction.cpp:44-71), then
// (*ObjectToWorld)(SurfaceInteraction) (transform.cpp:262-297); shading.n == n for a quadric
PG_DEV SphereHit quadric_finish(const PgSphere &sp, V3 d, V3 pHit, V3 pError, V3 dpdu, V3 dpdv, V3 d2Pduu, V3 d2Pduv, V3 d2Pdvv, float u, float v) {
    V3 n = normalize(cross(dpdu, dpdv));
    if (sp.reverse_orientation ^ sp.swaps_handedness) n = n * -1.f;
    const V3 wo = normalize(-d);  // Interaction ctor, interaction.h:60
    SphereHit h;
    h.p = m4_point_err2(sp.o2w, pHit, pError, h.pError);
    h.n = normalize(m4_normal(sp.w2o, n));
    h.wo = normalize(m4_vec(sp.o2w, wo));
    h.dpdu = m4_vec(sp.o2w, dpdu);
    h.dpdv = m4_vec(sp.o2w, dpdv);
    // dndu, dndv from the fundamental forms (sphere.cpp:131-145, cylinder.cpp:111-130); only bump mapping reads them
    V3 dndu = mk(0, 0, 0), dndv = mk(0, 0, 0);
    if (sp.shape != PG_SHAPE_DISK) {
        const float E = dot(dpdu, dpdu), F = dot(dpdu, dpdv), G = dot(dpdv, dpdv);
        const V3 N = normalize(cross(dpdu, dpdv));
        const float e = dot(N, d2Pduu), f = dot(N, d2Pduv), g = dot(N, d2Pdvv);
        const float invEGF2 = 1 / (E * G - F * F);
        dndu = dpdu * ((f * F - e * G) * invEGF2) + dpdv * ((e * F - f * E) * invEGF2);
        dndv = dpdu * ((g * F - f * G) * invEGF2) + dpdv * ((f * F - g * E) * invEGF2);
    }
    h.dndu = m4_normal(sp.w2o, dndu); h.dndv = m4_normal(sp.w2o, dndv);
    h.u = u; h.v = v;
    return h;
}
PG_DEV SphereHit sphere_interaction_s(const PgSphere &sp, V3 ro, V3 rd, float tHit) {
    V3 o, d, oErr, dErr;
    sphere_object_ray(sp, ro, rd, o, d, oErr, dErr);
    float phi;
    const V3 pHit = sphere_hit_point(sp, o, d, tHit, phi);
    // sphere.cpp:108-124 (u, v, dndu, dndv feed only textures and ray differentials)
    const float cz = pHit.z / sp.radius;
    const float theta = pg_acosf((cz < -1 ? -1.f : (cz > 1 ? 1.f : cz)));  // std::acos(Clamp(pHit.z / radius, -1, 1))
    const float zRadius = sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    const float invZRadius = 1 / zRadius;
    const float cosPhi = pHit.x * invZRadius;
    const float sinPhi = pHit.y * invZRadius;
    const V3 dpdu = mk(-sp.phi_max * pHit.y, sp.phi_max * pHit.x, 0);
    const V3 dpdv = mk(pHit.z * cosPhi, pHit.z * sinPhi, -sp.radius * pg_sinf(theta)) * (sp.theta_max - sp.theta_min);
    const V3 pError = vabs(pHit) * pgamma(5);  // sphere.cpp:148
    const V3 d2Pduu = mk(pHit.x, pHit.y, 0) * (-sp.phi_max * sp.phi_max);
    const V3 d2Pduv = mk(-sinPhi, cosPhi, 0.f) * ((sp.theta_max - sp.theta_min) * pHit.z * sp.phi_max);
    const V3 d2Pdvv = mk(pHit.x, pHit.y, pHit.z) * (-(sp.theta_max - sp.theta_min) * (sp.theta_max - sp.theta_min));
    return quadric_finish(sp, d, pHit, pError, dpdu, dpdv, d2Pduu, d2Pduv, d2Pdvv, phi / sp.phi_max, (theta - sp.theta_min) / (sp.theta_max - sp.theta_min));
}

// ---- Cylinder (shapes/cylinder.cpp:48-198) and Disk (shapes/disk.cpp:48-122): same record, same conventions ----------
PG_DEV V3 cylinder_hit_point(const PgSphere &sp, V3 o, V3 d, float t, float &phi) {  // cylinder.cpp:79-87
    V3 pHit = o + d * t;
    const float hitRad = sqrtf(pHit.x * pHit.x + pHit.y * pHit.y);
    pHit.x *= sp.radius / hitRad;
    pHit.y *= sp.radius / hitRad;
    phi = pg_atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * PG_PI;
    return pHit;
}
PG_DEV bool cylinder_test(const PgSphere &sp, V3 ro, V3 rd, float tMax, float &tHit) {
    V3 o, d, oErr, dErr;
    sphere_object_ray(sp, ro, rd, o, d, oErr, dErr);
    const EFloat ox = ef_make(o.x, oErr.x), oy = ef_make(o.y, oErr.y);
    const EFloat dx = ef_make(d.x, dErr.x), dy = ef_make(d.y, dErr.y);
    const EFloat a = ef_add(ef_mul(dx, dx), ef_mul(dy, dy));
    const EFloat b = ef_mul(ef_make(2, 0), ef_add(ef_mul(dx, ox), ef_mul(dy, oy)));
    const EFloat rad = ef_make(sp.radius, 0);
    const EFloat c = ef_sub(ef_add(ef_mul(ox, ox), ef_mul(oy, oy)), ef_mul(rad, rad));
    EFloat t0, t1;
    if (!ef_quadratic(a, b, c, t0, t1)) return false;
    if (t0.high > tMax || t1.low <= 0) return false;
    EFloat tShapeHit = t0;
    if (tShapeHit.low <= 0) {
        tShapeHit = t1;
        if (tShapeHit.high > tMax) return false;
    }
    float phi;
    V3 pHit = cylinder_hit_point(sp, o, d, tShapeHit.v, phi);
    if (pHit.z < sp.z_min || pHit.z > sp.z_max || phi > sp.phi_max) {
        if (tShapeHit.v == t1.v) return false;
        tShapeHit = t1;
        if (t1.high > tMax) return false;
        pHit = cylinder_hit_point(sp, o, d, tShapeHit.v, phi);
        if (pHit.z < sp.z_min || pHit.z > sp.z_max || phi > sp.phi_max) return false;
    }
    tHit = tShapeHit.v;
    return true;
}
PG_DEV SphereHit cylinder_interaction(const PgSphere &sp, V3 ro, V3 rd, float tHit) {
    V3 o, d, oErr, dErr;
    sphere_object_ray(sp, ro, rd, o, d, oErr, dErr);
    float phi;
    const V3 pHit = cylinder_hit_point(sp, o, d, tHit, phi);
    const V3 dpdu = mk(-sp.phi_max * pHit.y, sp.phi_max * pHit.x, 0);  // cylinder.cpp:103-104
    const V3 dpdv = mk(0, 0, sp.z_max - sp.z_min);
    const V3 pError = vabs(mk(pHit.x, pHit.y, 0)) * pgamma(3);  // :134
    return quadric_finish(sp, d, pHit, pError, dpdu, dpdv, mk(pHit.x, pHit.y, 0) * (-sp.phi_max * sp.phi_max), mk(0, 0, 0), mk(0, 0, 0), phi / sp.phi_max,
                          (pHit.z - sp.z_min) / (sp.z_max - sp.z_min));
}
PG_DEV bool disk_test(const PgSphere &sp, V3 ro, V3 rd, float tMax, float &tHit) {  // disk.cpp:48-70
    V3 o, d, oErr, dErr;
    sphere_object_ray(sp, ro, rd, o, d, oErr, dErr);
    if (d.z == 0) return false;
    const float tShapeHit = (sp.height - o.z) / d.z;
    if (tShapeHit <= 0 || tShapeHit >= tMax) return false;
    const V3 pHit = o + d * tShapeHit;
    const float dist2 = pHit.x * pHit.x + pHit.y * pHit.y;
    if (dist2 > sp.radius * sp.radius || dist2 < sp.inner_radius * sp.inner_radius) return false;
    float phi = pg_atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * PG_PI;
    if (phi > sp.phi_max) return false;
    tHit = tShapeHit;
    return true;
}
PG_DEV SphereHit disk_interaction(const PgSphere &sp, V3 ro, V3 rd, float tHit) {  // disk.cpp:72-95
    V3 o, d, oErr, dErr;
    sphere_object_ray(sp, ro, rd, o, d, oErr, dErr);
    V3 pHit = o + d * tHit;
    const float dist2 = pHit.x * pHit.x + pHit.y * pHit.y;
    const float rHit = sqrtf(dist2);
    const V3 dpdu = mk(-sp.phi_max * pHit.y, sp.phi_max * pHit.x, 0);
    const V3 dpdv = vdiv(mk(pHit.x, pHit.y, 0.f) * (sp.inner_radius - sp.radius), rHit);
    float phi = pg_atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * PG_PI;
    pHit.z = sp.height;
    return quadric_finish(sp, d, pHit, mk(0, 0, 0), dpdu, dpdv, mk(0, 0, 0), mk(0, 0, 0), mk(0, 0, 0), phi / sp.phi_max, (sp.radius - rHit) / (sp.radius - sp.inner_radius));
}
// ---- Cone (shapes/cone.cpp:56-198), Paraboloid (paraboloid.cpp:57-202) and Hyperboloid (hyperboloid.cpp:73-229): the three
// quadrics the reference only intersects (no Sample(): they cannot be area lights).  They share the root selection and differ
// in the implicit coefficients, the inverse mapping / clipping and the parametric derivatives.
PG_DEV float ef_abs_error(EFloat e) { return next_float_up(pmax(fabsf(e.high - e.v), fabsf(e.v - e.low))); }  // efloat.h:91-92
struct Q3Roots { V3 o, d; EFloat ox, oy, oz, dx, dy, dz, t0, t1; };
PG_DEV bool quadric3_roots(const PgSphere &sp, V3 ro, V3 rd, Q3Roots &q) {
    V3 oErr, dErr;
    sphere_object_ray(sp, ro, rd, q.o, q.d, oErr, dErr);
    q.ox = ef_make(q.o.x, oErr.x); q.oy = ef_make(q.o.y, oErr.y); q.oz = ef_make(q.o.z, oErr.z);
    q.dx = ef_make(q.d.x, dErr.x); q.dy = ef_make(q.d.y, dErr.y); q.dz = ef_make(q.d.z, dErr.z);
    const EFloat ox = q.ox, oy = q.oy, oz = q.oz, dx = q.dx, dy = q.dy, dz = q.dz, two = ef_make(2, 0);
    EFloat a, b, c;
    if (sp.shape == PG_SHAPE_CONE) {  // cone.cpp:66-73
        const EFloat h = ef_make(sp.height, 0);
        EFloat k = ef_div(ef_make(sp.radius, 0), h);
        k = ef_mul(k, k);
        a = ef_sub(ef_add(ef_mul(dx, dx), ef_mul(dy, dy)), ef_mul(ef_mul(k, dz), dz));
        b = ef_mul(two, ef_sub(ef_add(ef_mul(dx, ox), ef_mul(dy, oy)), ef_mul(ef_mul(k, dz), ef_sub(oz, h))));
        c = ef_sub(ef_add(ef_mul(ox, ox), ef_mul(oy, oy)), ef_mul(ef_mul(k, ef_sub(oz, h)), ef_sub(oz, h)));
    } else if (sp.shape == PG_SHAPE_PARABOLOID) {  // paraboloid.cpp:68-74
        const EFloat k = ef_div(ef_make(sp.z_max, 0), ef_mul(ef_make(sp.radius, 0), ef_make(sp.radius, 0)));
        a = ef_mul(k, ef_add(ef_mul(dx, dx), ef_mul(dy, dy)));
        b = ef_sub(ef_mul(ef_mul(two, k), ef_add(ef_mul(dx, ox), ef_mul(dy, oy))), dz);
        c = ef_sub(ef_mul(k, ef_add(ef_mul(ox, ox), ef_mul(oy, oy))), oz);
    } else {  // hyperboloid.cpp:85-91
        const EFloat ah = ef_make(sp.ah, 0), ch = ef_make(sp.ch, 0);
        a = ef_sub(ef_add(ef_mul(ef_mul(ah, dx), dx), ef_mul(ef_mul(ah, dy), dy)), ef_mul(ef_mul(ch, dz), dz));
        b = ef_mul(two, ef_sub(ef_add(ef_mul(ef_mul(ah, dx), ox), ef_mul(ef_mul(ah, dy), oy)), ef_mul(ef_mul(ch, dz), oz)));
        c = ef_sub(ef_sub(ef_add(ef_mul(ef_mul(ah, ox), ox), ef_mul(ef_mul(ah, oy), oy)), ef_mul(ef_mul(ch, oz), oz)), ef_make(1, 0));
    }
    return ef_quadratic(a, b, c, q.t0, q.t1);
}
// the hit point of root t, its phi (and the hyperboloid's v); false when it lies outside the clipping parameters
PG_DEV bool quadric3_map(const PgSphere &sp, V3 o, V3 d, float t, V3 &pHit, float &phi, float &v) {
    pHit = o + d * t;
    v = 0;
    if (sp.shape == PG_SHAPE_HYPERBOLOID) {  // hyperboloid.cpp:106-111
        const V3 p1 = mk(sp.p1[0], sp.p1[1], sp.p1[2]), p2 = mk(sp.p2[0], sp.p2[1], sp.p2[2]);
        v = (pHit.z - p1.z) / (p2.z - p1.z);
        const V3 pr = p1 * (1 - v) + p2 * v;
        phi = pg_atan2f((pr.x * pHit.y - pHit.x * pr.y), (pHit.x * pr.x + pHit.y * pr.y));
    } else phi = pg_atan2f(pHit.y, pHit.x);
    if (phi < 0) phi += 2 * PG_PI;
    const float zLo = sp.shape == PG_SHAPE_CONE ? 0.f : sp.z_min, zHi = sp.shape == PG_SHAPE_CONE ? sp.height : sp.z_max;
    return !(pHit.z < zLo || pHit.z > zHi || phi > sp.phi_max);
}
PG_DEV bool quadric3_test(const PgSphere &sp, V3 ro, V3 rd, float tMax, float &tHit) {
    Q3Roots q;
    if (!quadric3_roots(sp, ro, rd, q)) return false;
    if (q.t0.high > tMax || q.t1.low <= 0) return false;
    EFloat tShapeHit = q.t0;
    if (tShapeHit.low <= 0) {
        tShapeHit = q.t1;
        if (tShapeHit.high > tMax) return false;
    }
    V3 pHit;
    float phi, v;
    if (!quadric3_map(sp, q.o, q.d, tShapeHit.v, pHit, phi, v)) {
        if (tShapeHit.v == q.t1.v) return false;
        tShapeHit = q.t1;
        if (q.t1.high > tMax) return false;
        if (!quadric3_map(sp, q.o, q.d, tShapeHit.v, pHit, phi, v)) return false;
    }
    tHit = tShapeHit.v;
    return true;
}
PG_DEV SphereHit quadric3_interaction(const PgSphere &sp, V3 ro, V3 rd, float tHit) {
    Q3Roots q;
    quadric3_roots(sp, ro, rd, q);
    const EFloat tShapeHit = (q.t0.v == tHit) ? q.t0 : q.t1;  // the root quadric3_test accepted (equal values: the test kept t0)
    V3 pHit;
    float phi, v;
    quadric3_map(sp, q.o, q.d, tHit, pHit, phi, v);
    const float phiMax = sp.phi_max, u = phi / phiMax;
    const V3 dpdu = mk(-phiMax * pHit.y, phiMax * pHit.x, 0);
    const V3 d2Pduu = mk(pHit.x, pHit.y, 0) * (-phiMax * phiMax);
    V3 dpdv, d2Pduv, d2Pdvv = mk(0, 0, 0);
    if (sp.shape == PG_SHAPE_CONE) {  // cone.cpp:106-116
        v = pHit.z / sp.height;
        dpdv = mk(-pHit.x / (1.f - v), -pHit.y / (1.f - v), sp.height);
        d2Pduv = mk(pHit.y, -pHit.x, 0.f) * (phiMax / (1.f - v));
    } else if (sp.shape == PG_SHAPE_PARABOLOID) {  // paraboloid.cpp:107-124
        const float zMin = sp.z_min, zMax = sp.z_max;
        v = (pHit.z - zMin) / (zMax - zMin);
        dpdv = mk(pHit.x / (2 * pHit.z), pHit.y / (2 * pHit.z), 1.f) * (zMax - zMin);
        d2Pduv = mk(-pHit.y / (2 * pHit.z), pHit.x / (2 * pHit.z), 0) * ((zMax - zMin) * phiMax);
        d2Pdvv = mk(pHit.x / (4 * pHit.z * pHit.z), pHit.y / (4 * pHit.z * pHit.z), 0) * (-(zMax - zMin) * (zMax - zMin));
    } else {  // hyperboloid.cpp:128-140
        float sP, cP;
        pg_sincosf(phi, &sP, &cP);
        const float cosPhi = (float)cP, sinPhi = (float)sP;
        const float ex = sp.p2[0] - sp.p1[0], ey = sp.p2[1] - sp.p1[1];
        dpdv = mk(ex * cosPhi - ey * sinPhi, ex * sinPhi + ey * cosPhi, sp.p2[2] - sp.p1[2]);
        d2Pduv = mk(-dpdv.y, dpdv.x, 0.f) * phiMax;
    }
    // error bounds of the point computed with the ray equation (cone.cpp:135-140)
    const EFloat px = ef_add(q.ox, ef_mul(tShapeHit, q.dx)), py = ef_add(q.oy, ef_mul(tShapeHit, q.dy)), pz = ef_add(q.oz, ef_mul(tShapeHit, q.dz));
    const V3 pError = mk(ef_abs_error(px), ef_abs_error(py), ef_abs_error(pz));
    return quadric_finish(sp, q.d, pHit, pError, dpdu, dpdv, d2Pduu, d2Pduv, d2Pdvv, u, v);
}
// Shape::Intersect[P] of the quadric record, by shape
PG_DEV bool sphere_test(const PgSphere &sp, V3 ro, V3 rd, float tMax, float &tHit) {
    if (sp.shape == PG_SHAPE_CYLINDER) return cylinder_test(sp, ro, rd, tMax, tHit);
    if (sp.shape == PG_SHAPE_DISK) return disk_test(sp, ro, rd, tMax, tHit);
    if (sp.shape >= PG_SHAPE_CONE) return quadric3_test(sp, ro, rd, tMax, tHit);
    return sphere_test_s(sp, ro, rd, tMax, tHit);
}
PG_DEV SphereHit sphere_interaction(const PgSphere &sp, V3 ro, V3 rd, float tHit) {
    if (sp.shape == PG_SHAPE_CYLINDER) return cylinder_interaction(sp, ro, rd, tHit);
    if (sp.shape == PG_SHAPE_DISK) return disk_interaction(sp, ro, rd, tHit);
    if (sp.shape >= PG_SHAPE_CONE) return quadric3_interaction(sp, ro, rd, tHit);
    return sphere_interaction_s(sp, ro, rd, tHit);
}
#endif
