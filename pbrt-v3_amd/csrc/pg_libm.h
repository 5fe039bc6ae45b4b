// pg_libm.h -- the float libm functions the path calls, restated so that a lane returns the bits the CPU reference gets.
//
// The reference calls std::sin / cos / acos / atan2 / log / exp on `Float` = float, i.e. glibc's sinf, cosf (or sincosf, when
// gcc merges a sin / cos pair of one argument), acosf, atan2f, logf, expf.  None of them is correctly rounded (0.5 - 1 ulp
// error bounds: 1 - 15 % of the arguments get the float next to the correctly rounded one), so "evaluate in double, round
// once" -- what this library did until round 3 -- differs from the reference in the last bit of a per cent of the calls, and a
// few bounces later that tips a discrete event of a sample: config 0 (a sphere light: sin / cos / acos per light sample) had
// 1 809 of 160 000 pixels off, up to 8e-3 relative.  The functions below are the published algorithms of the glibc the
// reference links (2.35: sinf / cosf / sincosf / logf / expf are the ARM "optimized routines" -- double-precision polynomials
// on a table-driven reduction --, acosf / atanf / atan2f are Sun's fdlibm float code), written operation by operation; the
// x86-64 build selects its FMA variants of the first group at run time on every CPU with FMA3 (the build box, the GPU box's
// host), which is that same code with gcc's contraction of a * b + c, spelled out here as pg_fma.  The second group is
// plain float arithmetic, no contraction.
//
// Pinned: tests/test_libm_restated.py compiles this header for the host and compares every function with the system's
// libm over ALL 2^32 arguments (unary) / billions of pairs (atan2f): zero differences, NaN payloads aside.
// Cheaper too: a double sincos() on the device is ~250 instructions with a Payne-Hanek path, this sinf is 12 double
// operations.
#ifndef PG_LIBM_H
#define PG_LIBM_H
#include <stdint.h>
#include <string.h>
#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PG_LIBM_FN __host__ __device__ inline
#else
#define PG_LIBM_FN static inline
#endif

PG_LIBM_FN uint32_t pgm_asuint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PG_LIBM_FN float pgm_asfloat(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
PG_LIBM_FN uint64_t pgm_asuint64(double f) { uint64_t u; memcpy(&u, &f, 8); return u; }
PG_LIBM_FN double pgm_asdouble(uint64_t u) { double f; memcpy(&f, &u, 8); return f; }
PG_LIBM_FN double pg_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
PG_LIBM_FN uint32_t pgm_abstop12(float x) { return (pgm_asuint(x) >> 20) & 0x7ff; }

// ---- sinf / cosf / sincosf: glibc sysdeps/ieee754/flt-32/s_sincosf.h, s_sinf.c, s_cosf.c, s_sincosf.c (2.35) -------------
struct PgSincosTable { double sign[4], hpi_inv, hpi, c0, c1, c2, c3, c4, s1, s2, s3; };
// the polynomial of quadrant n (odd: cosine) on the reduced argument; x2 = x * x
PG_LIBM_FN float pgm_sinf_poly(double x, double x2, bool negate, int n) {
    // __sincosf_table[0] / [1]: [1] is [0] with the cosine coefficients negated
    const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10, c4 = 0x1.99343027bf8c3p-16;
    const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
    if ((n & 1) == 0) {
        const double x3 = x * x2;
        const double s1_ = pg_fma(x2, s3, s2);
        const double x7 = x3 * x2;
        const double s = pg_fma(x3, s1, x);
        return (float)pg_fma(x7, s1_, s);
    }
    const double sg = negate ? -1.0 : 1.0;
    const double x4 = x2 * x2;
    const double c2_ = pg_fma(x2, sg * c4, sg * c3);
    const double c1_ = pg_fma(x2, sg * c1, sg * c0);
    const double x6 = x4 * x2;
    const double c = pg_fma(x4, sg * c2, c1_);
    return (float)pg_fma(x6, c2_, c);
}
// reduce_fast: |x| < 120; quadrant in *np, the argument minus n * pi/2
PG_LIBM_FN double pgm_reduce_fast(double x, int *np) {
    const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
    const double r = x * hpi_inv;
    const int n = ((int32_t)r + 0x800000) >> 24;
    *np = n;
    return pg_fma(-(double)n, hpi, x);
}
// reduce_large: 120 <= |x| < inf, by the bits of 4 / pi
PG_LIBM_FN double pgm_reduce_large(uint32_t xi, int *np) {
    const uint32_t inv_pio4[24] = {0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27, 0x29fc2757, 0xfc2757d1,
                                   0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295, 0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041};
    const uint32_t *arr = &inv_pio4[(xi >> 26) & 15];
    const int shift = (xi >> 23) & 7;
    uint64_t n, res0, res1, res2;
    xi = (xi & 0xffffff) | 0x800000;
    xi <<= shift;
    res0 = xi * arr[0];
    res1 = (uint64_t)xi * arr[4];
    res2 = (uint64_t)xi * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    n = (res0 + (1ULL << 61)) >> 62;
    res0 -= n << 62;
    const double x = (double)(int64_t)res0;
    *np = (int)n;
    return x * 0x1.921FB54442D18p-62;
}
// both at once (what the two single functions return, argument by argument): sinf(y) -> *sp, cosf(y) -> *cp
PG_LIBM_FN void pg_sincosf(float y, float *sp, float *cp) {
    double x = y, s;
    int n;
    if (pgm_abstop12(y) < pgm_abstop12(0x1.921FB6p-1f)) {  // |y| < pi / 4
        s = x * x;
        if (pgm_abstop12(y) < pgm_abstop12(0x1p-12f)) { *sp = y; *cp = 1.0f; return; }
        *sp = pgm_sinf_poly(x, s, false, 0);
        *cp = pgm_sinf_poly(x, s, false, 1);
        return;
    }
    const double sign4[4] = {1.0, -1.0, -1.0, 1.0};
    bool negate;
    if (pgm_abstop12(y) < pgm_abstop12(120.0f)) {
        x = pgm_reduce_fast(x, &n);
        s = sign4[n & 3];
        negate = (n & 2) != 0;
    } else if (pgm_abstop12(y) < pgm_abstop12(__builtin_inff())) {
        const uint32_t xi = pgm_asuint(y);
        const int sign = xi >> 31;
        x = pgm_reduce_large(xi, &n);
        s = sign4[(n + sign) & 3];
        negate = ((n + sign) & 2) != 0;
    } else { *sp = *cp = (y - y) / (y - y); return; }  // __math_invalidf
    *sp = pgm_sinf_poly(x * s, x * x, negate, n);
    *cp = pgm_sinf_poly(x * s, x * x, negate, n ^ 1);
}
PG_LIBM_FN float pg_sinf(float y) { float s, c; pg_sincosf(y, &s, &c); return s; }
PG_LIBM_FN float pg_cosf(float y) { float s, c; pg_sincosf(y, &s, &c); return c; }

// ---- logf: glibc sysdeps/ieee754/flt-32/e_logf.c + e_logf_data.c (2.35) ----------------------------------------------------
PG_LIBM_FN float pg_logf(float x) {
    const double T[16][2] = {{0x1.661ec79f8f3bep+0, -0x1.57bf7808caadep-2}, {0x1.571ed4aaf883dp+0, -0x1.2bef0a7c06ddbp-2}, {0x1.49539f0f010bp+0, -0x1.01eae7f513a67p-2},
                             {0x1.3c995b0b80385p+0, -0x1.b31d8a68224e9p-3}, {0x1.30d190c8864a5p+0, -0x1.6574f0ac07758p-3}, {0x1.25e227b0b8eap+0, -0x1.1aa2bc79c81p-3},
                             {0x1.1bb4a4a1a343fp+0, -0x1.a4e76ce8c0e5ep-4}, {0x1.12358f08ae5bap+0, -0x1.1973c5a611cccp-4}, {0x1.0953f419900a7p+0, -0x1.252f438e10c1ep-5},
                             {0x1p+0, 0x0p+0}, {0x1.e608cfd9a47acp-1, 0x1.aa5aa5df25984p-5}, {0x1.ca4b31f026aap-1, 0x1.c5e53aa362eb4p-4},
                             {0x1.b2036576afce6p-1, 0x1.526e57720db08p-3}, {0x1.9c2d163a1aa2dp-1, 0x1.bc2860d22477p-3}, {0x1.886e6037841edp-1, 0x1.1058bc8a07ee1p-2},
                             {0x1.767dcf5534862p-1, 0x1.4043057b6ee09p-2}};
    const double Ln2 = 0x1.62e42fefa39efp-1, A0 = -0x1.00ea348b88334p-2, A1 = 0x1.5575b0be00b6ap-2, A2 = -0x1.ffffef20a4123p-2;
    uint32_t ix = pgm_asuint(x);
    if (ix == 0x3f800000) return 0;
    if (ix - 0x00800000 >= 0x7f800000 - 0x00800000) {  // x < 0x1p-126 or inf or nan
        if (ix * 2 == 0) return -__builtin_inff();
        if (ix == 0x7f800000) return x;
        if ((ix & 0x80000000) || ix * 2 >= 0xff000000) return (x - x) / (x - x);
        ix = pgm_asuint(x * 0x1p23f);  // subnormal: normalise
        ix -= 23 << 23;
    }
    const uint32_t tmp = ix - 0x3f330000;
    const int i = (tmp >> (23 - 4)) % 16;
    const int k = (int32_t)tmp >> 23;
    const uint32_t iz = ix - (tmp & 0x1ffu << 23);
    const double invc = T[i][0], logc = T[i][1];
    const double z = (double)pgm_asfloat(iz);
    const double r = pg_fma(z, invc, -1.0);
    const double y0 = pg_fma((double)k, Ln2, logc);
    const double r2 = r * r;
    double y = pg_fma(A1, r, A2);
    y = pg_fma(A0, r2, y);
    y = pg_fma(y, r2, y0 + r);
    return (float)y;
}

// ---- expf: glibc sysdeps/ieee754/flt-32/e_expf.c + e_exp2f_data.c (2.35) ---------------------------------------------------
PG_LIBM_FN float pg_expf(float x) {
    const uint64_t T[32] = {0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b, 0x3fef54873168b9aa, 0x3fef387a6e756238,
                            0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb, 0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82,
                            0x3feeab07dd485429, 0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13, 0x3feeace5422aa0db,
                            0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad, 0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069,
                            0x3fef5818dcfba487, 0x3fef7c97337b9b5f, 0x3fefa4afa2a490da, 0x3fefd0765b6e4540};
    const double InvLn2N = 0x1.71547652b82fep+5, SHIFT = 0x1.8p+52;
    const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
    const double xd = (double)x;
    const uint32_t abstop = (pgm_asuint(x) >> 20) & 0x7ff;
    if (abstop >= (pgm_asuint(88.0f) >> 20)) {  // |x| >= 88 or nan
        if (pgm_asuint(x) == pgm_asuint(-__builtin_inff())) return 0.0f;
        if (abstop >= (pgm_asuint(__builtin_inff()) >> 20)) return x + x;
        if (x > 0x1.62e42ep6f) return __builtin_inff();   // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;              // underflow
    }
    // z = InvLn2N * xd feeds two additions, and gcc contracts both: the product is never rounded on its own
    double kd = pg_fma(InvLn2N, xd, SHIFT);
    const uint64_t ki = pgm_asuint64(kd);
    kd -= SHIFT;
    const double r = pg_fma(InvLn2N, xd, -kd);
    double z;
    uint64_t t = T[ki % 32];
    t += ki << (52 - 5);
    const double s = pgm_asdouble(t);
    z = pg_fma(C0, r, C1);
    const double r2 = r * r;
    double y = pg_fma(C2, r, 1.0);
    y = pg_fma(z, r2, y);
    y = y * s;
    return (float)y;
}

// ---- acosf: glibc sysdeps/ieee754/flt-32/e_acosf.c (fdlibm; plain float arithmetic) ----------------------------------------
PG_LIBM_FN float pgm_sqrtf(float x) { return __builtin_sqrtf(x); }
PG_LIBM_FN float pg_acosf(float x) {
    const float one = 1.0f, pi = pgm_asfloat(0x40490fda), pio2_hi = pgm_asfloat(0x3fc90fda), pio2_lo = pgm_asfloat(0x33a22168);
    const float pS0 = pgm_asfloat(0x3e2aaaab), pS1 = pgm_asfloat(0xbea6b090), pS2 = pgm_asfloat(0x3e4e0aa8), pS3 = pgm_asfloat(0xbd241146),
                pS4 = pgm_asfloat(0x3a4f7f04), pS5 = pgm_asfloat(0x3811ef08), qS1 = pgm_asfloat(0xc019d139), qS2 = pgm_asfloat(0x4001572d),
                qS3 = pgm_asfloat(0xbf303361), qS4 = pgm_asfloat(0x3d9dc62e);
    float z, p, q, r, w, s, c, df;
    const int32_t hx = (int32_t)pgm_asuint(x), ix = hx & 0x7fffffff;
    if (ix == 0x3f800000) return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
    if (ix > 0x3f800000) return (x - x) / (x - x);
    if (ix < 0x3f000000) {  // |x| < 0.5
        if (ix <= 0x23000000) return pio2_hi + pio2_lo;
        z = x * x;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        return pio2_hi - (x - (pio2_lo - r * x));
    } else if (hx < 0) {  // x < -0.5
        z = (one + x) * 0.5f;
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        s = pgm_sqrtf(z);
        r = p / q;
        w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {  // x > 0.5
        z = (one - x) * 0.5f;
        s = pgm_sqrtf(z);
        df = pgm_asfloat(pgm_asuint(s) & 0xfffff000);
        c = (z - df * df) / (s + df);
        p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        r = p / q;
        w = r * s + c;
        return 2.0f * (df + w);
    }
}

// ---- atanf / atan2f: glibc sysdeps/ieee754/flt-32/s_atanf.c, e_atan2f.c (fdlibm; plain float arithmetic) -------------------
PG_LIBM_FN float pg_atanf(float x) {
    const uint32_t atanhi[4] = {0x3eed6338, 0x3f490fda, 0x3f7b985e, 0x3fc90fda}, atanlo[4] = {0x31ac3769, 0x33222168, 0x33140fb4, 0x33a22168};
    // (aT[0]: the decimal 3.3333334327e-01 of the source converts to 0x3eaaaaab; the hexadecimal in its comment says ...aa)
    const uint32_t aT[11] = {0x3eaaaaab, 0xbe4ccccd, 0x3e124925, 0xbde38e38, 0x3dba2e6e, 0xbd9d8795, 0x3d886b35, 0xbd6ef16b, 0x3d4bda59, 0xbd15a221, 0x3c8569d7};
#define PGM_AT(i) pgm_asfloat(aT[i])
    const float one = 1.0f;
    float w, s1, s2, z;
    const int32_t hx = (int32_t)pgm_asuint(x), ix = hx & 0x7fffffff;
    int id;
    if (ix >= 0x4c000000) {  // |x| >= 2^25
        if (ix > 0x7f800000) return x + x;
        return hx > 0 ? pgm_asfloat(atanhi[3]) + pgm_asfloat(atanlo[3]) : -pgm_asfloat(atanhi[3]) - pgm_asfloat(atanlo[3]);
    }
    if (ix < 0x3ee00000) {  // |x| < 0.4375
        if (ix < 0x31000000) return x;  // |x| < 2^-29
        id = -1;
    } else {
        x = __builtin_fabsf(x);
        if (ix < 0x3f980000) {  // |x| < 1.1875
            if (ix < 0x3f300000) { id = 0; x = (2.0f * x - one) / (2.0f + x); }
            else { id = 1; x = (x - one) / (x + one); }
        } else {
            if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (one + 1.5f * x); }
            else { id = 3; x = -1.0f / x; }
        }
    }
    z = x * x;
    w = z * z;
    s1 = z * (PGM_AT(0) + w * (PGM_AT(2) + w * (PGM_AT(4) + w * (PGM_AT(6) + w * (PGM_AT(8) + w * PGM_AT(10))))));
    s2 = w * (PGM_AT(1) + w * (PGM_AT(3) + w * (PGM_AT(5) + w * (PGM_AT(7) + w * PGM_AT(9)))));
#undef PGM_AT
    if (id < 0) return x - x * (s1 + s2);
    z = pgm_asfloat(atanhi[id]) - ((x * (s1 + s2) - pgm_asfloat(atanlo[id])) - x);
    return hx < 0 ? -z : z;
}
PG_LIBM_FN float pg_atan2f(float y, float x) {
    const float tiny = 1.0e-30f, pi_o_4 = pgm_asfloat(0x3f490fdb), pi_o_2 = pgm_asfloat(0x3fc90fdb), pi = pgm_asfloat(0x40490fdb), pi_lo = pgm_asfloat(0xb3bbbd2e);
    float z;
    const int32_t hx = (int32_t)pgm_asuint(x), ix = hx & 0x7fffffff, hy = (int32_t)pgm_asuint(y), iy = hy & 0x7fffffff;
    if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
    if (hx == 0x3f800000) return pg_atanf(y);
    const int m = ((hy >> 31) & 1) | ((hx >> 30) & 2);  // 2 * sign(x) + sign(y)
    if (iy == 0) {
        switch (m) {
        case 0: case 1: return y;
        case 2: return pi + tiny;
        default: return -pi - tiny;
        }
    }
    if (ix == 0) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    if (ix == 0x7f800000) {
        if (iy == 0x7f800000) {
            switch (m) {
            case 0: return pi_o_4 + tiny;
            case 1: return -pi_o_4 - tiny;
            case 2: return 3.0f * pi_o_4 + tiny;
            default: return -3.0f * pi_o_4 - tiny;
            }
        } else {
            switch (m) {
            case 0: return 0.0f;
            case 1: return -0.0f;
            case 2: return pi + tiny;
            default: return -pi - tiny;
            }
        }
    }
    if (iy == 0x7f800000) return hy < 0 ? -pi_o_2 - tiny : pi_o_2 + tiny;
    const int k = (iy - ix) >> 23;
    if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
    else if (hx < 0 && k < -60) z = 0.0f;
    else z = pg_atanf(__builtin_fabsf(y / x));
    switch (m) {
    case 0: return z;
    case 1: return pgm_asfloat(pgm_asuint(z) ^ 0x80000000);
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}
#endif
