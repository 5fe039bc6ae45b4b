// AnimatedTransform::Interpolate on the device (transform.cpp:1144-1169): what a moving camera (k_generate) and a moving shape or instance
// (k_trace's instance steps) apply to a ray of a given time.  The decompositions T / R / S of the two end transforms come from the host
// (Decompose, transform.cpp:1103-1142: include/pbrt_gpu.h); float arithmetic in the reference's order, libm as pg_libm.h restates it.
#ifndef PG_MOTION_H
#define PG_MOTION_H
#include "pg_device.h"
#include "../../include/pbrt_gpu.h"
#ifndef PG_XF_STRIDE
#define PG_XF_STRIDE 36  // floats per entry of DScene::animXf (pg_kernels.h)
#endif

// r = m1 m2, Matrix4x4::Mul (transform.h:86-93): every entry the four products summed left to right
PG_DEV void m4_mul(const float *m1, const float *m2, float *r) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            r[4 * i + j] = m1[4 * i] * m2[j] + m1[4 * i + 1] * m2[4 + j] + m1[4 * i + 2] * m2[8 + j] + m1[4 * i + 3] * m2[12 + j];
}
// Inverse(const Matrix4x4 &), transform.cpp:83-139: Gauss-Jordan with full pivoting.  The pivot is the LAST largest |entry| (>=) of a row-major
// scan over the rows and columns that have not been pivots; 1 / pivot is a double division rounded to float once; the pivot row is scaled after
// its pivot entry became 1; every other row loses row * factor entry by entry, the factor's own slot zeroed first; the row swaps come back as
// column swaps in reverse order.  (Rare: once per ray that enters a moving instance; the indexed 4 x 4 lives in scratch.)
static __device__ __noinline__ void m4_inverse_gj(const float *src, float *out) {
    float a[4][4];
    int swapR[4], swapC[4];
    unsigned used = 0;
    for (int i = 0; i < 16; ++i) a[i >> 2][i & 3] = src[i];
    for (int step = 0; step < 4; ++step) {
        int pr = 0, pc = 0;
        float best = 0.f;
        for (int r = 0; r < 4; ++r) {
            if (used >> r & 1) continue;
            for (int c = 0; c < 4; ++c)
                if (!(used >> c & 1) && fabsf(a[r][c]) >= best) { best = fabsf(a[r][c]); pr = r; pc = c; }
        }
        used |= 1u << pc;
        if (pr != pc) for (int c = 0; c < 4; ++c) { const float t = a[pr][c]; a[pr][c] = a[pc][c]; a[pc][c] = t; }
        swapR[step] = pr; swapC[step] = pc;
        const float pivinv = (float)(1. / (double)a[pc][pc]);
        a[pc][pc] = 1.f;
        for (int c = 0; c < 4; ++c) a[pc][c] *= pivinv;
        for (int r = 0; r < 4; ++r) {
            if (r == pc) continue;
            const float save = a[r][pc];
            a[r][pc] = 0;
            for (int c = 0; c < 4; ++c) a[r][c] -= a[pc][c] * save;
        }
    }
    for (int step = 3; step >= 0; --step)
        if (swapR[step] != swapC[step])
            for (int k = 0; k < 4; ++k) { const float t = a[k][swapR[step]]; a[k][swapR[step]] = a[k][swapC[step]]; a[k][swapC[step]] = t; }
    for (int i = 0; i < 16; ++i) out[i] = a[i >> 2][i & 3];
}
// Interpolate between the ends: Translate(lerp T) * Slerp(dt, R0, R1).ToTransform() * Transform(lerp S).  m = the product of the three matrices;
// INV: mInv = the product of their inverses in the opposite order (Transform::operator*, transform.cpp:251-254) -- -trans, the rotation's
// transpose (Quaternion::ToTransform returns Transform(Transpose(m), m), quaternion.cpp:57-58), Inverse(scale) by Gauss-Jordan (transform.h:123-124).
// Quaternion arithmetic as quaternion.h:52-99 spells it (v /= f multiplies by 1 / f, w /= f divides), Slerp quaternion.cpp:94-104, ToTransform :41-59.
template <bool INV> PG_DEV void interpolate_trs(const float T[2][3], const float R[2][4], const float S[2][9], float dt, float *m, float *mInv) {
    const float tx = (1 - dt) * T[0][0] + dt * T[1][0], ty = (1 - dt) * T[0][1] + dt * T[1][1], tz = (1 - dt) * T[0][2] + dt * T[1][2];
    const float *q1 = R[0], *q2 = R[1];
    const float cosTheta = (q1[0] * q2[0] + q1[1] * q2[1] + q1[2] * q2[2]) + q1[3] * q2[3];  // Dot(q1, q2) = Dot(q1.v, q2.v) + q1.w * q2.w
    float qx, qy, qz, qw;
    if (cosTheta > .9995f) {  // Normalize((1 - t) * q1 + t * q2)
        const float a = 1 - dt;
        const float sx = q1[0] * a + q2[0] * dt, sy = q1[1] * a + q2[1] * dt, sz = q1[2] * a + q2[2] * dt, sw = q1[3] * a + q2[3] * dt;
        const float len = sqrtf((sx * sx + sy * sy + sz * sz) + sw * sw), inv = 1.f / len;
        qx = sx * inv; qy = sy * inv; qz = sz * inv; qw = sw / len;
    } else {
        const float theta = pg_acosf(cosTheta < -1 ? -1.f : (cosTheta > 1 ? 1.f : cosTheta));
        const float thetap = theta * dt;
        // qperp = Normalize(q2 - q1 * cosTheta)
        const float px = q2[0] - q1[0] * cosTheta, py = q2[1] - q1[1] * cosTheta, pz = q2[2] - q1[2] * cosTheta, pw = q2[3] - q1[3] * cosTheta;
        const float len = sqrtf((px * px + py * py + pz * pz) + pw * pw), inv = 1.f / len;
        const float ux = px * inv, uy = py * inv, uz = pz * inv, uw = pw / len;
        float sn, cs;
        pg_sincosf(thetap, &sn, &cs);
        qx = q1[0] * cs + ux * sn; qy = q1[1] * cs + uy * sn; qz = q1[2] * cs + uz * sn; qw = q1[3] * cs + uw * sn;
    }
    const float xx = qx * qx, yy = qy * qy, zz = qz * qz, xy = qx * qy, xz = qx * qz, yz = qy * qz, wx = qx * qw, wy = qy * qw, wz = qz * qw;
    // rotate.ToTransform().m = Transpose(the matrix of quaternion.cpp:47-55)
    const float rot[16] = {1 - 2 * (yy + zz), 2 * (xy - wz), 2 * (xz + wy), 0, 2 * (xy + wz), 1 - 2 * (xx + zz), 2 * (yz - wx), 0,
                           2 * (xz - wy), 2 * (yz + wx), 1 - 2 * (xx + yy), 0, 0, 0, 0, 1};
    const float trn[16] = {1, 0, 0, tx, 0, 1, 0, ty, 0, 0, 1, tz, 0, 0, 0, 1};
    float scl[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) scl[4 * i + j] = (1 - dt) * S[0][3 * i + j] + dt * S[1][3 * i + j];  // Lerp, pbrt.h:417
    float tr[16];
    m4_mul(trn, rot, tr);
    m4_mul(tr, scl, m);
    if (INV) {
        const float trnInv[16] = {1, 0, 0, -tx, 0, 1, 0, -ty, 0, 0, 1, -tz, 0, 0, 0, 1};
        float rotInv[16], sclInv[16], trInv[16];
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) rotInv[4 * i + j] = rot[4 * j + i];
        m4_inverse_gj(scl, sclInv);
        m4_mul(rotInv, trnInv, trInv);  // (Translate * Rotate).mInv = Rotate.mInv * Translate.mInv
        m4_mul(sclInv, trInv, mInv);
    }
}
// PrimitiveToWorld.Interpolate(r.time, &InterpolatedPrimToWorld) of a moving instance (primitive.cpp:78-80, :99-101): xf[0..15] = its matrix,
// xf[16..31] = its inverse, xf[32] = 1.f when the matrix is the identity (Transform::IsIdentity, transform.h:157-164: primitive.cpp:86).
static __device__ __noinline__ void instance_matrices_at(const PgInstance &in, float time, float *xf) {
    if (time <= in.time[0]) { for (int k = 0; k < 16; ++k) { xf[k] = in.i2w[k]; xf[16 + k] = in.w2i[k]; } }
    else if (time >= in.time[1]) { for (int k = 0; k < 16; ++k) { xf[k] = in.i2w_end[k]; xf[16 + k] = in.w2i_end[k]; } }
    else interpolate_trs<true>(in.T, in.R, in.S, (time - in.time[0]) / (in.time[1] - in.time[0]), xf, xf + 16);
    bool ident = true;
    for (int k = 0; k < 16; ++k) ident = ident && xf[k] == ((k % 5) == 0 ? 1.f : 0.f);
    xf[32] = ident ? 1.f : 0.f;
}
#endif
