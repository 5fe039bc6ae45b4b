// Matrix / Transform routines of the host front end.
//
// The camera, object and light matrices computed here are handed to the device and must equal the reference's bit for
// bit, so the ORDER OF FLOAT OPERATIONS of src/core/transform.cpp is a constraint; the code around it is this build's own.
// Order-constrained expressions, each marked [order] below:
//   * Inverse: Gauss-Jordan with full pivoting -- pivot choice (`>=`, so the LAST largest entry in row-major scan order
//     wins), reciprocal pivot formed in double and rounded once, row scaled before elimination, `row_j -= row_p * f`
//     element by element, column un-scrambling in reverse pivot order (transform.cpp:83-139);
//   * Rotate: the nine Rodrigues entries as a*b*(1-cos) +- c*sin and a*a + (1-a*a)*cos (transform.cpp:182-206);
//   * Perspective: f/(f-n), -f*n/(f-n), 1/tan(radians(fov)/2) (transform.cpp:304-312);
//   * point / vector / normal application: row sums left to right (transform.h:219-247).
#include "geometry.h"
#include "error.h"

#include <utility>

namespace pbrt {

Matrix4x4 Transpose(const Matrix4x4 &m) {
    Matrix4x4 t;
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) t.m[c][r] = m.m[r][c];
    return t;
}

Matrix4x4 Inverse(const Matrix4x4 &src) {
    Matrix4x4 work = src;
    Float(*a)[4] = work.m;
    bool pivoted[4] = {false, false, false, false};  // column c has been a pivot column (its pivot row now sits in row c)
    std::pair<int, int> swaps[4];                    // (row, column) of every pivot, in order
    for (int step = 0; step < 4; ++step) {
        // [order] largest |entry| among rows and columns not yet pivoted; ties go to the last one scanned
        int pr = -1, pc = -1;
        Float best = 0.f;
        for (int r = 0; r < 4; ++r) {
            if (pivoted[r]) continue;
            for (int c = 0; c < 4; ++c) {
                if (pivoted[c]) continue;
                const Float mag = std::abs(a[r][c]);
                if (mag >= best) { best = mag; pr = r; pc = c; }
            }
        }
        if (pr < 0) {  // only NaN entries left: nothing compares >= 0
            Error("Singular matrix in MatrixInvert");
            return Matrix4x4();
        }
        pivoted[pc] = true;
        if (pr != pc)
            for (int c = 0; c < 4; ++c) std::swap(a[pr][c], a[pc][c]);
        swaps[step] = {pr, pc};
        if (a[pc][pc] == 0.f) Error("Singular matrix in MatrixInvert");
        // [order] reciprocal in double, rounded to float once; the pivot entry becomes 1 BEFORE the row is scaled
        const Float scale = (Float)(1.0 / (double)a[pc][pc]);
        a[pc][pc] = 1.f;
        for (int c = 0; c < 4; ++c) a[pc][c] *= scale;
        // [order] eliminate the pivot column from the other rows: entry -= pivotRow[c] * factor, factor's slot zeroed first
        for (int r = 0; r < 4; ++r) {
            if (r == pc) continue;
            const Float factor = a[r][pc];
            a[r][pc] = 0.f;
            for (int c = 0; c < 4; ++c) a[r][c] -= a[pc][c] * factor;
        }
    }
    for (int step = 3; step >= 0; --step) {  // undo the row swaps as column swaps, last pivot first
        const int r = swaps[step].first, c = swaps[step].second;
        if (r == c) continue;
        for (int k = 0; k < 4; ++k) std::swap(a[k][r], a[k][c]);
    }
    return work;
}

bool Transform::SwapsHandedness() const {  // sign of the upper-left 3x3 determinant, cofactors along the first row
    const Float(*a)[4] = m.m;
    const Float c0 = a[1][1] * a[2][2] - a[1][2] * a[2][1];
    const Float c1 = a[1][0] * a[2][2] - a[1][2] * a[2][0];
    const Float c2 = a[1][0] * a[2][1] - a[1][1] * a[2][0];
    const Float det = a[0][0] * c0 - a[0][1] * c1 + a[0][2] * c2;
    return det < 0;
}

// [order] one row of M applied to (x, y, z[, 1]): products summed left to right
static inline Float rowDot3(const Float *row, Float x, Float y, Float z) { return row[0] * x + row[1] * y + row[2] * z; }
static inline Float rowDot4(const Float *row, Float x, Float y, Float z) { return row[0] * x + row[1] * y + row[2] * z + row[3]; }

Point3f Transform::Pt(const Point3f &p) const {
    const Float h[4] = {rowDot4(m.m[0], p.x, p.y, p.z), rowDot4(m.m[1], p.x, p.y, p.z), rowDot4(m.m[2], p.x, p.y, p.z),
                        rowDot4(m.m[3], p.x, p.y, p.z)};
    if (h[3] == 1) return Point3f(h[0], h[1], h[2]);
    const Float rcp = (Float)1 / h[3];  // the homogeneous divide multiplies by the reciprocal (geometry.h:499-503)
    return Point3f(rcp * h[0], rcp * h[1], rcp * h[2]);
}
Vector3f Transform::Vec(const Vector3f &v) const {
    return Vector3f(rowDot3(m.m[0], v.x, v.y, v.z), rowDot3(m.m[1], v.x, v.y, v.z), rowDot3(m.m[2], v.x, v.y, v.z));
}
Normal3f Transform::Nrm(const Normal3f &n) const {  // normals go through the inverse transpose: columns of mInv
    Float out[3];
    for (int c = 0; c < 3; ++c) out[c] = mInv.m[0][c] * n.x + mInv.m[1][c] * n.y + mInv.m[2][c] * n.z;
    return Normal3f(out[0], out[1], out[2]);
}

static Matrix4x4 diagonalWithOffset(Float sx, Float sy, Float sz, Float tx, Float ty, Float tz) {
    Matrix4x4 r;
    r.m[0][0] = sx; r.m[1][1] = sy; r.m[2][2] = sz;
    r.m[0][3] = tx; r.m[1][3] = ty; r.m[2][3] = tz;
    return r;
}
Transform Translate(const Vector3f &d) {
    return Transform(diagonalWithOffset(1, 1, 1, d.x, d.y, d.z), diagonalWithOffset(1, 1, 1, -d.x, -d.y, -d.z));
}
Transform Scale(Float x, Float y, Float z) {
    return Transform(diagonalWithOffset(x, y, z, 0, 0, 0), diagonalWithOffset(1 / x, 1 / y, 1 / z, 0, 0, 0));
}

Transform Rotate(Float theta, const Vector3f &axis) {
    const Vector3f a = Normalize(axis);
    const Float s = std::sin(Radians(theta)), c = std::cos(Radians(theta));
    // [order] Rodrigues' formula entry by entry; the inverse of a rotation is its transpose
    Matrix4x4 r;
    for (int i = 0; i < 3; ++i) {
        const int j = (i + 1) % 3, k = (i + 2) % 3;  // (i, j, k) cyclic: entry (i,j) carries -a_k sin, entry (i,k) carries +a_j sin
        r.m[i][i] = a[i] * a[i] + (1 - a[i] * a[i]) * c;
        // the products are written with the lower axis index first, as a.x*a.y, a.x*a.z, a.y*a.z
        const Float pij = i < j ? a[i] * a[j] : a[j] * a[i], pik = i < k ? a[i] * a[k] : a[k] * a[i];
        r.m[i][j] = pij * (1 - c) - a[k] * s;
        r.m[i][k] = pik * (1 - c) + a[j] * s;
    }
    return Transform(r, Transpose(r));
}

Transform LookAt(const Point3f &pos, const Point3f &look, const Vector3f &up) {
    const Vector3f dir = Normalize(look - pos);
    const Vector3f side = Cross(Normalize(up), dir);
    if (side.Length() == 0) {
        Error("\"up\" vector (%f, %f, %f) and viewing direction (%f, %f, %f) "
              "passed to LookAt are pointing in the same direction.  Using "
              "the identity transformation.", up.x, up.y, up.z, dir.x, dir.y, dir.z);
        return Transform();
    }
    const Vector3f right = Normalize(side), newUp = Cross(dir, right);
    // camera-to-world: columns right | newUp | dir | pos; the world-to-camera matrix the directive asks for is its inverse
    const Vector3f *col[4] = {&right, &newUp, &dir, &pos};
    Matrix4x4 c2w;
    for (int c = 0; c < 4; ++c) {
        c2w.m[0][c] = col[c]->x; c2w.m[1][c] = col[c]->y; c2w.m[2][c] = col[c]->z;
        c2w.m[3][c] = c == 3 ? 1.f : 0.f;
    }
    return Transform(Inverse(c2w), c2w);
}

Transform Perspective(Float fov, Float n, Float f) {
    Matrix4x4 persp;  // x, y pass through; z' = (f z - f n) / (f - n), w' = z
    persp.m[2][2] = f / (f - n);       // [order]
    persp.m[2][3] = -f * n / (f - n);  // [order]
    persp.m[3][2] = 1; persp.m[3][3] = 0;
    const Float cotHalf = 1 / std::tan(Radians(fov) / 2);  // [order]
    return Scale(cotHalf, cotHalf, 1) * Transform(persp);
}
// AnimatedTransform::Decompose, transform.cpp:1103-1142: translation off the last column, rotation by polar decomposition (R <- (R +
// (R^T)^-1) / 2 until the rows move by less than 1e-4, at most 100 times), scale = R^-1 M; the rotation as a quaternion the way
// Quaternion(const Transform &) builds it (quaternion.cpp:61-92).  Float arithmetic in the reference's order: the device blends these
// numbers per camera ray and must arrive at the reference's matrices.
void DecomposeTransform(const Matrix4x4 &m, Float T[3], Float Rq[4], Float S[9]) {
    T[0] = m.m[0][3]; T[1] = m.m[1][3]; T[2] = m.m[2][3];
    Matrix4x4 M = m;
    for (int i = 0; i < 3; ++i) M.m[i][3] = M.m[3][i] = 0.f;
    M.m[3][3] = 1.f;
    Float norm;
    int count = 0;
    Matrix4x4 R = M;
    do {
        Matrix4x4 Rnext;
        Matrix4x4 Rit = Inverse(Transpose(R));
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) Rnext.m[i][j] = 0.5f * (R.m[i][j] + Rit.m[i][j]);
        norm = 0;
        for (int i = 0; i < 3; ++i) {
            Float n = std::abs(R.m[i][0] - Rnext.m[i][0]) + std::abs(R.m[i][1] - Rnext.m[i][1]) + std::abs(R.m[i][2] - Rnext.m[i][2]);
            norm = std::max(norm, n);
        }
        R = Rnext;
    } while (++count < 100 && norm > .0001);
    {   // Quaternion(Transform(R)): from the trace, or from the largest diagonal entry
        const Float trace = R.m[0][0] + R.m[1][1] + R.m[2][2];
        if (trace > 0.f) {
            Float s = std::sqrt(trace + 1.0f);
            Rq[3] = s / 2.0f;
            s = 0.5f / s;
            Rq[0] = (R.m[2][1] - R.m[1][2]) * s;
            Rq[1] = (R.m[0][2] - R.m[2][0]) * s;
            Rq[2] = (R.m[1][0] - R.m[0][1]) * s;
        } else {
            const int nxt[3] = {1, 2, 0};
            Float q[3];
            int i = 0;
            if (R.m[1][1] > R.m[0][0]) i = 1;
            if (R.m[2][2] > R.m[i][i]) i = 2;
            const int j = nxt[i], k = nxt[j];
            Float s = std::sqrt((R.m[i][i] - (R.m[j][j] + R.m[k][k])) + 1.0f);
            q[i] = s * 0.5f;
            if (s != 0.f) s = 0.5f / s;
            Rq[3] = (R.m[k][j] - R.m[j][k]) * s;
            q[j] = (R.m[j][i] + R.m[i][j]) * s;
            q[k] = (R.m[k][i] + R.m[i][k]) * s;
            Rq[0] = q[0]; Rq[1] = q[1]; Rq[2] = q[2];
        }
    }
    const Matrix4x4 Sm = Matrix4x4::Mul(Inverse(R), M);
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) S[3 * i + j] = Sm.m[i][j];
}
}  // namespace pbrt
