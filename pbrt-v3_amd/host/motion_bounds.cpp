// AnimatedTransform::MotionBounds (transform.cpp:1215-1247): the world bound of a TransformedPrimitive whose motion ROTATES
// (primitive.h:107-109).  Host only, once per moving shape or instance: the device and the oracle already interpolate the transform per ray
// (csrc/pg_motion.h); what the rotation adds is the box the top-level BVH and the scene's world bound (hence distant / infinite lights and
// the spatial light grid) are built from, which must be the reference's to the bit.
//
//   * every component of a point's path has its extrema at the ends or where  c1 + (c2 + c3 t) cos(2 theta t) + (c4 + c5 t) sin(2 theta t) = 0
//     (book section 2.9.4); c1 .. c5 are linear in the point (DerivativeTerm::Eval, transform.h:444-446) with coefficients that are
//     polynomials in the two decompositions.  Those 60 polynomials are data/motion_terms.bin (tools/extract_motion_terms.py: postfix programs
//     that keep the reference's association), run here on a float stack;
//   * the zeros: interval arithmetic over [0, 1] halved eight times, then four Newton steps (IntervalFindZeros, transform.cpp:354-394;
//     Interval / Sin / Cos :314-352), float throughout with the host's libm -- the same one the reference binary calls;
//   * the point at a zero: Interpolate (transform.cpp:1144-1169) at Lerp(zero, startTime, endTime), the same arithmetic as pg_motion.h.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <atomic>
#include "error.h"
#include "geometry.h"

#ifndef PG_MOTION_BIN
#error "PG_MOTION_BIN (path of data/motion_terms.bin) must be defined by the build"
#endif
__asm__(".section .rodata\n"
        ".balign 4\n"
        ".global pg_motion_blob\n"
        "pg_motion_blob:\n"
        ".incbin \"" PG_MOTION_BIN "\"\n"
        ".global pg_motion_blob_end\n"
        "pg_motion_blob_end:\n"
        ".previous\n");
extern "C" const unsigned char pg_motion_blob[], pg_motion_blob_end[];

namespace pbrt {
namespace {
// MotionBounds calls that could not give the reference's box: a derivative with more than 8 zeros (the reference's CHECK_LE aborts there),
// or the polynomial table unreadable.  The front end and pbrt_host_motion_bounds turn a rise of this count into an Error and a refused frame.
std::atomic<int> g_motionBoundsFailures{0};
enum { kVars = 33, kPrograms = 60 };
struct Programs {
    bool valid = true;
    float consts[64];
    const unsigned char *code[kPrograms];
    int len[kPrograms];
};
const Programs &MotionPrograms() {
    static Programs P = [] {
        Programs p;
        const unsigned char *b = pg_motion_blob, *end = pg_motion_blob_end;
        int32_t hdr[3];
        memcpy(hdr, b, sizeof(hdr));
        bool ok = hdr[0] == 0x4E544F4D && hdr[1] == kPrograms && hdr[2] >= 0 && hdr[2] <= 64;
        if (ok) {
            memcpy(p.consts, b + 12, 4 * hdr[2]);
            b += 12 + 4 * hdr[2];
            for (int i = 0; i < kPrograms && ok; ++i) {
                ok = b + 2 <= end;
                if (!ok) break;
                uint16_t n;
                memcpy(&n, b, 2);
                p.code[i] = b + 2; p.len[i] = n;
                b += 2 + n;
                ok = b <= end;
                int depth = 0;  // every program is checked once, here: operands exist, the stack stays within RunProgram's, one value is left
                for (int k = 0; k < n && ok; ++k) {
                    const unsigned op = p.code[i][k];
                    if (op < kVars || (op >= 64 && op < 64u + (unsigned)hdr[2])) ok = ++depth < 32;
                    else if (op == 128) ok = depth >= 1;
                    else if (op >= 129 && op <= 131) ok = --depth >= 1;
                    else ok = false;
                }
                ok = ok && depth == 1;
            }
            ok = ok && b == end;
        }
        // a corrupt table makes every rotating MotionBounds fail (MotionBoundsFailures: the frame is refused); it does not end the host process
        if (!ok) { Error("motion_terms.bin (the motion derivative's polynomials) is missing or corrupt: rotating motions cannot be bounded"); p.valid = false; }
        return p;
    }();
    return P;
}
// one polynomial: a postfix program over the 33 variables (validated when the blob was read); every operation rounds to float once, in the
// order the reference's expression has
Float RunProgram(const Programs &P, int i, const Float *vars) {
    Float stack[32];
    int sp = 0;
    const unsigned char *c = P.code[i];
    for (int k = 0; k < P.len[i]; ++k) {
        const unsigned op = c[k];
        if (op < kVars) stack[sp++] = vars[op];
        else if (op >= 64 && op < 128) stack[sp++] = P.consts[op - 64];
        else if (op == 128) stack[sp - 1] = -stack[sp - 1];
        else {
            const Float b = stack[--sp], a = stack[sp - 1];
            stack[sp - 1] = op == 129 ? a + b : (op == 130 ? a - b : a * b);
        }
    }
    return stack[0];
}

struct Interval {  // transform.cpp:314-333
    Float low, high;
    Interval(Float v) : low(v), high(v) {}
    Interval(Float a, Float b) : low(std::min(a, b)), high(std::max(a, b)) {}
};
Interval operator+(const Interval &a, const Interval &b) { return Interval(a.low + b.low, a.high + b.high); }
Interval operator*(const Interval &a, const Interval &b) {
    const Float ll = a.low * b.low, hl = a.high * b.low, lh = a.low * b.high, hh = a.high * b.high;
    return Interval(std::min(std::min(ll, hl), std::min(lh, hh)), std::max(std::max(ll, hl), std::max(lh, hh)));
}
// Pi is a Float constant (pbrt.h:203): Pi / 2 and (3.f / 2.f) * Pi are float products, the comparisons float comparisons
const Float kPi = 3.14159265358979323846f;
Interval SinI(const Interval &i) {
    Float lo = std::sin(i.low), hi = std::sin(i.high);
    if (lo > hi) std::swap(lo, hi);
    if (i.low < kPi / 2 && i.high > kPi / 2) hi = 1.;
    if (i.low < (3.f / 2.f) * kPi && i.high > (3.f / 2.f) * kPi) lo = -1.;
    return Interval(lo, hi);
}
Interval CosI(const Interval &i) {
    Float lo = std::cos(i.low), hi = std::cos(i.high);
    if (lo > hi) std::swap(lo, hi);
    if (i.low < kPi && i.high > kPi) lo = -1.;
    return Interval(lo, hi);
}
void FindZeros(Float c1, Float c2, Float c3, Float c4, Float c5, Float theta, Interval tI, Float *zeros, int *nZeros, int depth = 8) {
    const Interval arg = Interval(2 * theta) * tI;
    const Interval range = Interval(c1) + (Interval(c2) + Interval(c3) * tI) * CosI(arg) + (Interval(c4) + Interval(c5) * tI) * SinI(arg);
    if (range.low > 0. || range.high < 0. || range.low == range.high) return;
    if (depth > 0) {
        const Float mid = (tI.low + tI.high) * 0.5f;
        FindZeros(c1, c2, c3, c4, c5, theta, Interval(tI.low, mid), zeros, nZeros, depth - 1);
        FindZeros(c1, c2, c3, c4, c5, theta, Interval(mid, tI.high), zeros, nZeros, depth - 1);
        return;
    }
    Float t = (tI.low + tI.high) * 0.5f;
    for (int i = 0; i < 4; ++i) {
        const Float f = c1 + (c2 + c3 * t) * std::cos(2.f * theta * t) + (c4 + c5 * t) * std::sin(2.f * theta * t);
        const Float fPrime = (c3 + 2 * (c4 + c5 * t) * theta) * std::cos(2.f * t * theta) + (c5 - 2 * (c2 + c3 * t) * theta) * std::sin(2.f * t * theta);
        if (f == 0 || fPrime == 0) break;
        t = t - f / fPrime;
    }
    if (t >= tI.low - 1e-3f && t < tI.high + 1e-3f) {
        // the reference's CHECK_LE(*nZeros, 8) ends its process here; this library reports it and refuses the frame (MotionBoundsFailures)
        if (*nZeros >= 8) { g_motionBoundsFailures.fetch_add(1); return; }
        zeros[(*nZeros)++] = t;
    }
}

struct Motion {  // what AnimatedTransform's constructor keeps (transform.cpp:396-1100)
    Float T[2][3], R[2][4], S[2][9];
    Float theta;
    Float k[3][5][4];  // component, term c1 .. c5, (kc, kx, ky, kz)
};
Float QDot(const Float *a, const Float *b) { return (a[0] * b[0] + a[1] * b[1] + a[2] * b[2]) + a[3] * b[3]; }  // Dot(q1.v, q2.v) + q1.w * q2.w
void QPerp(const Float *q0, const Float *q1, Float cosTheta, Float *out) {  // Normalize(q1 - q0 * cosTheta): v scaled by 1 / len, w divided
    Float p[4];
    for (int i = 0; i < 4; ++i) p[i] = q1[i] - q0[i] * cosTheta;
    const Float len = std::sqrt(QDot(p, p)), inv = 1.f / len;
    out[0] = p[0] * inv; out[1] = p[1] * inv; out[2] = p[2] * inv; out[3] = p[3] / len;
}
// the point p at `time`: Interpolate, then Transform::operator()(Point3f)
Point3f PointAt(const Motion &M, const Transform &start, const Transform &end, Float startTime, Float endTime, Float time, const Point3f &p) {
    if (time <= startTime) return start.Pt(p);
    if (time >= endTime) return end.Pt(p);
    const Float dt = (time - startTime) / (endTime - startTime);
    const Vector3f trans((1 - dt) * M.T[0][0] + dt * M.T[1][0], (1 - dt) * M.T[0][1] + dt * M.T[1][1], (1 - dt) * M.T[0][2] + dt * M.T[1][2]);
    const Float *q1 = M.R[0], *q2 = M.R[1];
    const Float cosTheta = QDot(q1, q2);
    Float q[4];
    if (cosTheta > .9995f) {  // (cannot happen with hasRotation; kept so that this is Slerp)
        const Float a = 1 - dt;
        Float s[4];
        for (int i = 0; i < 4; ++i) s[i] = q1[i] * a + q2[i] * dt;
        const Float len = std::sqrt(QDot(s, s)), inv = 1.f / len;
        q[0] = s[0] * inv; q[1] = s[1] * inv; q[2] = s[2] * inv; q[3] = s[3] / len;
    } else {
        const Float theta = std::acos(cosTheta < -1 ? -1.f : (cosTheta > 1 ? 1.f : cosTheta));
        const Float thetap = theta * dt;
        Float u[4];
        QPerp(q1, q2, cosTheta, u);
        const Float cs = std::cos(thetap), sn = std::sin(thetap);
        for (int i = 0; i < 4; ++i) q[i] = q1[i] * cs + u[i] * sn;
    }
    const Float xx = q[0] * q[0], yy = q[1] * q[1], zz = q[2] * q[2], xy = q[0] * q[1], xz = q[0] * q[2], yz = q[1] * q[2];
    const Float wx = q[0] * q[3], wy = q[1] * q[3], wz = q[2] * q[3];
    // Quaternion::ToTransform (quaternion.cpp:41-59) hands back the transpose of the matrix it fills
    const Matrix4x4 rot(1 - 2 * (yy + zz), 2 * (xy - wz), 2 * (xz + wy), 0, 2 * (xy + wz), 1 - 2 * (xx + zz), 2 * (yz - wx), 0,
                        2 * (xz - wy), 2 * (yz + wx), 1 - 2 * (xx + yy), 0, 0, 0, 0, 1);
    Matrix4x4 scl;
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) scl.m[i][j] = Lerp(dt, M.S[0][3 * i + j], M.S[1][3 * i + j]);
    const Matrix4x4 m = Matrix4x4::Mul(Matrix4x4::Mul(Translate(trans).GetMatrix(), rot), scl);
    return Transform(m, Matrix4x4()).Pt(p);  // only the matrix acts on a point
}
}  // namespace

bool MotionHasRotation(const Transform &start, const Transform &end) {  // transform.cpp:406-411
    Float T[3], R0[4], R1[4], S[9];
    DecomposeTransform(start.GetMatrix(), T, R0, S);
    DecomposeTransform(end.GetMatrix(), T, R1, S);
    if (QDot(R0, R1) < 0) for (int i = 0; i < 4; ++i) R1[i] = -R1[i];
    return QDot(R0, R1) < 0.9995f;
}

Bounds3f MotionBounds(const Transform &start, Float startTime, const Transform &end, Float endTime, const Bounds3f &b) {
    auto ofEnds = [&](const Transform &t) {  // Transform::operator()(Bounds3f), transform.cpp:237-249
        Bounds3f r;
        for (int corner = 0; corner < 8; ++corner) {
            const Point3f w = t.Pt(Point3f((corner & 1) ? b.pMax.x : b.pMin.x, (corner & 2) ? b.pMax.y : b.pMin.y, (corner & 4) ? b.pMax.z : b.pMin.z));
            r = corner == 0 ? Bounds3f(w) : Union(r, w);
        }
        return r;
    };
    const bool actuallyAnimated = !(start.GetMatrix() == end.GetMatrix()) || !(start.GetInverseMatrix() == end.GetInverseMatrix());
    if (!actuallyAnimated) return ofEnds(start);
    Motion M;
    DecomposeTransform(start.GetMatrix(), M.T[0], M.R[0], M.S[0]);
    DecomposeTransform(end.GetMatrix(), M.T[1], M.R[1], M.S[1]);
    if (QDot(M.R[0], M.R[1]) < 0) for (int i = 0; i < 4; ++i) M.R[1][i] = -M.R[1][i];
    const Float cosTheta = QDot(M.R[0], M.R[1]);
    if (!(cosTheta < 0.9995f)) return Union(ofEnds(start), ofEnds(end));
    M.theta = std::acos(cosTheta < -1 ? -1.f : (cosTheta > 1 ? 1.f : cosTheta));
    {   // the constructor's variables, in the order tools/extract_motion_terms.py numbers them
        Float vars[kVars], qperp[4];
        QPerp(M.R[0], M.R[1], cosTheta, qperp);
        int n = 0;
        for (int e = 0; e < 2; ++e) for (int i = 0; i < 3; ++i) vars[n++] = M.T[e][i];
        for (int i = 0; i < 4; ++i) vars[n++] = M.R[0][i];
        for (int i = 0; i < 4; ++i) vars[n++] = qperp[i];
        for (int e = 0; e < 2; ++e) for (int i = 0; i < 9; ++i) vars[n++] = M.S[e][i];
        vars[n++] = M.theta;
        const Programs &P = MotionPrograms();
        if (!P.valid) { g_motionBoundsFailures.fetch_add(1); return Union(ofEnds(start), ofEnds(end)); }
        for (int c = 0; c < 3; ++c) for (int term = 0; term < 5; ++term) for (int j = 0; j < 4; ++j) M.k[c][term][j] = RunProgram(P, (c * 5 + term) * 4 + j, vars);
    }
    Bounds3f bounds;
    for (int corner = 0; corner < 8; ++corner) {  // BoundPointMotion of every corner (Bounds3::Corner, geometry.h:760-764)
        const Point3f p((corner & 1) ? b.pMax.x : b.pMin.x, (corner & 2) ? b.pMax.y : b.pMin.y, (corner & 4) ? b.pMax.z : b.pMin.z);
        Bounds3f pb(start.Pt(p), end.Pt(p));
        for (int c = 0; c < 3; ++c) {
            Float cc[5];
            for (int term = 0; term < 5; ++term) { const Float *k = M.k[c][term]; cc[term] = k[0] + k[1] * p.x + k[2] * p.y + k[3] * p.z; }
            Float zeros[8];
            int nZeros = 0;
            FindZeros(cc[0], cc[1], cc[2], cc[3], cc[4], M.theta, Interval(0., 1.), zeros, &nZeros);
            for (int i = 0; i < nZeros; ++i) pb = Union(pb, PointAt(M, start, end, startTime, endTime, Lerp(zeros[i], startTime, endTime), p));
        }
        bounds = Union(bounds, pb);
    }
    return bounds;
}
int MotionBoundsFailures() { return g_motionBoundsFailures.load(); }
}  // namespace pbrt
