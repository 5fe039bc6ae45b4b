// Host-side geometry/transform subset for the MI355X path-tracing front end.
// Float arithmetic follows pbrt-v3's operation order exactly so that matrices,
// bounds and BVH splits produced here are bit-identical to the reference's
// (src/core/geometry.h, src/core/transform.{h,cpp}); each function cites the
// reference lines it mirrors.  Float == float (pbrt.h:127-129).
#ifndef PBRT_AMD_HOST_GEOMETRY_H
#define PBRT_AMD_HOST_GEOMETRY_H
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace pbrt {
typedef float Float;
static constexpr Float Pi = 3.14159265358979323846;
static constexpr Float Infinity = std::numeric_limits<Float>::infinity();
static constexpr Float MachineEpsilon = std::numeric_limits<Float>::epsilon() * 0.5;
inline Float gamma(int n) { return (n * MachineEpsilon) / (1 - n * MachineEpsilon); }  // pbrt.h:289-291
inline Float Clamp(Float v, Float lo, Float hi) { return v < lo ? lo : (v > hi ? hi : v); }  // pbrt.h:305-311
inline Float Radians(Float deg) { return (Pi / 180) * deg; }                            // pbrt.h:324
inline Float Lerp(Float t, Float v1, Float v2) { return (1 - t) * v1 + t * v2; }        // pbrt.h:417

struct Vector3f {
    Float x = 0, y = 0, z = 0;
    Vector3f() {}
    Vector3f(Float x, Float y, Float z) : x(x), y(y), z(z) {}
    Float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    Float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    Vector3f operator+(const Vector3f &v) const { return Vector3f(x + v.x, y + v.y, z + v.z); }
    Vector3f operator-(const Vector3f &v) const { return Vector3f(x - v.x, y - v.y, z - v.z); }
    Vector3f operator*(Float s) const { return Vector3f(s * x, s * y, s * z); }          // geometry.h:232-234
    Vector3f operator/(Float f) const {                                                   // geometry.h:243-248
        Float inv = (Float)1 / f;
        return Vector3f(x * inv, y * inv, z * inv);
    }
    Vector3f operator-() const { return Vector3f(-x, -y, -z); }
    Float LengthSquared() const { return x * x + y * y + z * z; }
    Float Length() const { return std::sqrt(LengthSquared()); }
};
typedef Vector3f Point3f;   // host code keeps one 3-float type; the distinct
typedef Vector3f Normal3f;  // reference operators are spelled out where they differ.

inline Vector3f operator*(Float s, const Vector3f &v) { return v * s; }
inline Float Dot(const Vector3f &a, const Vector3f &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline Vector3f Cross(const Vector3f &v1, const Vector3f &v2) {  // geometry.h:957-963 (double)
    double v1x = v1.x, v1y = v1.y, v1z = v1.z;
    double v2x = v2.x, v2y = v2.y, v2z = v2.z;
    return Vector3f((v1y * v2z) - (v1z * v2y), (v1z * v2x) - (v1x * v2z), (v1x * v2y) - (v1y * v2x));
}
inline Vector3f Normalize(const Vector3f &v) { return v / v.Length(); }
// Point3::operator/ multiplies inv*x (geometry.h:499-503); identical value to x*inv.
inline Vector3f Min(const Vector3f &a, const Vector3f &b) {
    return Vector3f(std::min(a.x, b.x), std::min(a.y, b.y), std::min(a.z, b.z));
}
inline Vector3f Max(const Vector3f &a, const Vector3f &b) {
    return Vector3f(std::max(a.x, b.x), std::max(a.y, b.y), std::max(a.z, b.z));
}

struct Bounds3f {  // geometry.h:700-830
    Point3f pMin, pMax;
    Bounds3f() {
        Float minNum = std::numeric_limits<Float>::lowest();
        Float maxNum = std::numeric_limits<Float>::max();
        pMin = Point3f(maxNum, maxNum, maxNum);
        pMax = Point3f(minNum, minNum, minNum);
    }
    explicit Bounds3f(const Point3f &p) : pMin(p), pMax(p) {}
    Bounds3f(const Point3f &p1, const Point3f &p2) : pMin(Min(p1, p2)), pMax(Max(p1, p2)) {}
    Vector3f Diagonal() const { return pMax - pMin; }
    Float SurfaceArea() const {
        Vector3f d = Diagonal();
        return 2 * (d.x * d.y + d.x * d.z + d.y * d.z);
    }
    int MaximumExtent() const {
        Vector3f d = Diagonal();
        if (d.x > d.y && d.x > d.z) return 0;
        else if (d.y > d.z) return 1;
        else return 2;
    }
    Vector3f Offset(const Point3f &p) const {
        Vector3f o = p - pMin;
        if (pMax.x > pMin.x) o.x /= pMax.x - pMin.x;
        if (pMax.y > pMin.y) o.y /= pMax.y - pMin.y;
        if (pMax.z > pMin.z) o.z /= pMax.z - pMin.z;
        return o;
    }
};
inline Bounds3f Union(const Bounds3f &b, const Point3f &p) {
    Bounds3f r; r.pMin = Min(b.pMin, p); r.pMax = Max(b.pMax, p); return r;
}
inline Bounds3f Union(const Bounds3f &a, const Bounds3f &b) {
    Bounds3f r; r.pMin = Min(a.pMin, b.pMin); r.pMax = Max(a.pMax, b.pMax); return r;
}

struct Matrix4x4 {  // transform.h:58-110
    Float m[4][4];
    Matrix4x4() {
        m[0][0] = m[1][1] = m[2][2] = m[3][3] = 1.f;
        m[0][1] = m[0][2] = m[0][3] = m[1][0] = m[1][2] = m[1][3] = m[2][0] = m[2][1] = m[2][3] =
            m[3][0] = m[3][1] = m[3][2] = 0.f;
    }
    Matrix4x4(Float t00, Float t01, Float t02, Float t03, Float t10, Float t11, Float t12, Float t13,
              Float t20, Float t21, Float t22, Float t23, Float t30, Float t31, Float t32, Float t33) {
        m[0][0] = t00; m[0][1] = t01; m[0][2] = t02; m[0][3] = t03;
        m[1][0] = t10; m[1][1] = t11; m[1][2] = t12; m[1][3] = t13;
        m[2][0] = t20; m[2][1] = t21; m[2][2] = t22; m[2][3] = t23;
        m[3][0] = t30; m[3][1] = t31; m[3][2] = t32; m[3][3] = t33;
    }
    bool operator==(const Matrix4x4 &o) const {  // transform.h:77-82: float ==, so -0 equals 0
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) if (m[i][j] != o.m[i][j]) return false;
        return true;
    }
    static Matrix4x4 Mul(const Matrix4x4 &m1, const Matrix4x4 &m2) {  // transform.h:86-93
        Matrix4x4 r;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j)
                r.m[i][j] = m1.m[i][0] * m2.m[0][j] + m1.m[i][1] * m2.m[1][j] +
                            m1.m[i][2] * m2.m[2][j] + m1.m[i][3] * m2.m[3][j];
        return r;
    }
};
Matrix4x4 Transpose(const Matrix4x4 &m);
Matrix4x4 Inverse(const Matrix4x4 &m);

class Transform {  // transform.h:112-205
  public:
    Transform() {}
    explicit Transform(const Matrix4x4 &m) : m(m), mInv(Inverse(m)) {}
    Transform(const Matrix4x4 &m, const Matrix4x4 &mInv) : m(m), mInv(mInv) {}
    friend Transform Inverse(const Transform &t) { return Transform(t.mInv, t.m); }
    const Matrix4x4 &GetMatrix() const { return m; }
    const Matrix4x4 &GetInverseMatrix() const { return mInv; }
    bool IsIdentity() const { return m == Matrix4x4(); }
    Transform operator*(const Transform &t2) const {  // transform.cpp:251-253
        return Transform(Matrix4x4::Mul(m, t2.m), Matrix4x4::Mul(t2.mInv, mInv));
    }
    bool SwapsHandedness() const;           // transform.cpp:255-260
    Point3f Pt(const Point3f &p) const;     // transform.h:219-231  operator()(Point3)
    Vector3f Vec(const Vector3f &v) const;  // transform.h:233-239  operator()(Vector3)
    Normal3f Nrm(const Normal3f &n) const;  // transform.h:241-247  operator()(Normal3)
  private:
    Matrix4x4 m, mInv;
};
// AnimatedTransform::Decompose (transform.cpp:1103-1142) with Quaternion(const Transform &) (quaternion.cpp:61-92): m = T * R * S,
// R as (v.x, v.y, v.z, w), S's upper 3x3 row-major -- what AnimatedTransform::Interpolate blends per ray
void DecomposeTransform(const Matrix4x4 &m, Float T[3], Float R[4], Float S[9]);
// AnimatedTransform(start, startTime, end, endTime).MotionBounds(b) and its hasRotation (transform.cpp:1215-1247, :411): host/motion_bounds.cpp
Bounds3f MotionBounds(const Transform &start, Float startTime, const Transform &end, Float endTime, const Bounds3f &b);
// how many MotionBounds calls so far could not give the reference's box (where the reference aborts: more than 8 zeros of a motion derivative)
int MotionBoundsFailures();
bool MotionHasRotation(const Transform &start, const Transform &end);
Transform Translate(const Vector3f &delta);
Transform Scale(Float x, Float y, Float z);
Transform Rotate(Float theta, const Vector3f &axis);
Transform LookAt(const Point3f &pos, const Point3f &look, const Vector3f &up);
Transform Perspective(Float fov, Float znear, Float zfar);
}  // namespace pbrt
#endif
