// pg_bssrdf.h -- TabulatedBSSRDF's radial profile (core/bssrdf.cpp:203-236, :352-392) and the spline routines under it
// (core/interpolation.cpp:77-116, :166-226) as device functions over the ABI's PgBSSRDF table.
//
// Used by k_shade<., ., SSS>, k_sss_exit (pg_kernels.hip).  Pinned on the host as well: tests/test_device_headers_on_host.py runs this
// header on the host against the oracle, bit for bit.
#ifndef PG_BSSRDF_H
#define PG_BSSRDF_H
#include "pg_device.h"
#include "../../include/pbrt_gpu.h"

struct DBssrdf {  // one material's TabulatedBSSRDF: the table's five arrays + the constructor's sigma_t, rho (bssrdf.h:141-165)
    const float *rhoSamples, *radiusSamples, *profile, *rhoEff, *profileCDF;
    int nRho, nRadius;
    float sigma_t[3], rho[3], eta;
};
PG_DEV DBssrdf bssrdf_bind(const PgBSSRDF &d, const float *tables) {
    DBssrdf b;
    b.nRho = d.n_rho; b.nRadius = d.n_radius; b.eta = d.eta;
    b.rhoSamples = tables + d.table;
    b.radiusSamples = b.rhoSamples + b.nRho;
    b.profile = b.radiusSamples + b.nRadius;
    b.rhoEff = b.profile + (size_t)b.nRho * b.nRadius;
    b.profileCDF = b.rhoEff + b.nRho;
    for (int c = 0; c < 3; ++c) { b.sigma_t[c] = d.sigma_t[c]; b.rho[c] = d.rho[c]; }
    return b;
}
// FresnelMoment1, bssrdf.cpp:43-52 (the cubic coefficient of the first branch is a double literal in the reference)
PG_DEV float fresnel_moment1(float eta) {
    const float e2 = eta * eta, e3 = e2 * eta, e4 = e3 * eta, e5 = e4 * eta;
    if (eta < 1) return 0.45966f - 1.73965f * eta + 3.37668f * e2 - 3.904945 * e3 + 2.49277f * e4 - 0.68441f * e5;
    return -4.61686f + 11.1136f * eta - 10.4646f * e2 + 5.11455f * e3 - 1.27198f * e4 + 0.12746f * e5;
}
// FindInterval(size, a[i] <= x), pbrt.h:402-415
PG_DEV int find_interval_le(int size, const float *a, float x) {
    int first = 0, len = size;
    while (len > 0) {
        const int half = len >> 1, middle = first + half;
        if (a[middle] <= x) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    const int r = first - 1;
    return r < 0 ? 0 : (r > size - 2 ? size - 2 : r);
}
// CatmullRomWeights, interpolation.cpp:77-116
PG_DEV bool catmull_rom_weights(int size, const float *nodes, float x, int &offset, float w[4]) {
    if (!(x >= nodes[0] && x <= nodes[size - 1])) return false;
    const int idx = find_interval_le(size, nodes, x);
    offset = idx - 1;
    const float x0 = nodes[idx], x1 = nodes[idx + 1];
    const float t = (x - x0) / (x1 - x0), t2 = t * t, t3 = t2 * t;
    w[1] = 2 * t3 - 3 * t2 + 1;
    w[2] = -2 * t3 + 3 * t2;
    if (idx > 0) {
        const float w0 = (t3 - 2 * t2 + t) * (x1 - x0) / (x1 - nodes[idx - 1]);
        w[0] = -w0;
        w[2] += w0;
    } else {
        const float w0 = t3 - 2 * t2 + t;
        w[0] = 0;
        w[1] -= w0;
        w[2] += w0;
    }
    if (idx + 2 < size) {
        const float w3 = (t3 - t2) * (x1 - x0) / (nodes[idx + 2] - x0);
        w[1] -= w3;
        w[3] = w3;
    } else {
        const float w3 = t3 - t2;
        w[1] -= w3;
        w[2] += w3;
        w[3] = 0;
    }
    return true;
}
PG_DEV float cr2d_interp(const float *array, int size2, int offset, const float w[4], int idx) {  // the lambda of SampleCatmullRom2D
    float value = 0;
    for (int i = 0; i < 4; ++i)
        if (w[i] != 0) value += array[(offset + i) * size2 + idx] * w[i];
    return value;
}
// SampleCatmullRom2D, interpolation.cpp:166-226 (fval / pdf are not requested by Sample_Sr)
PG_DEV float sample_catmull_rom_2d(int size1, int size2, const float *nodes1, const float *nodes2, const float *values, const float *cdf, float alpha, float u) {
    int offset;
    float w[4];
    if (!catmull_rom_weights(size1, nodes1, alpha, offset, w)) return 0;
    const float maximum = cr2d_interp(cdf, size2, offset, w, size2 - 1);
    u *= maximum;
    int first = 0, len = size2;
    while (len > 0) {
        const int half = len >> 1, middle = first + half;
        if (cr2d_interp(cdf, size2, offset, w, middle) <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int idx = first - 1;
    idx = idx < 0 ? 0 : (idx > size2 - 2 ? size2 - 2 : idx);
    const float f0 = cr2d_interp(values, size2, offset, w, idx), f1 = cr2d_interp(values, size2, offset, w, idx + 1);
    const float x0 = nodes2[idx], x1 = nodes2[idx + 1];
    const float width = x1 - x0;
    float d0, d1;
    u = (u - cr2d_interp(cdf, size2, offset, w, idx)) / width;
    if (idx > 0) d0 = width * (f1 - cr2d_interp(values, size2, offset, w, idx - 1)) / (x1 - nodes2[idx - 1]);
    else d0 = f1 - f0;
    if (idx + 2 < size2) d1 = width * (cr2d_interp(values, size2, offset, w, idx + 2) - f0) / (nodes2[idx + 2] - x0);
    else d1 = f1 - f0;
    float t;
    if (f0 != f1) t = (f0 - sqrtf(pmax(0.f, f0 * f0 + 2 * u * (f1 - f0)))) / (f0 - f1);
    else t = u / f0;
    float a = 0, b = 1, Fhat, fhat;
    for (;;) {
        if (!(t >= a && t <= b)) t = 0.5f * (a + b);
        Fhat = t * (f0 + t * (.5f * d0 + t * ((1.f / 3.f) * (-2 * d0 - d1) + f1 - f0 + t * (.25f * (d0 + d1) + .5f * (f0 - f1)))));
        fhat = f0 + t * (d0 + t * (-2 * d0 - d1 + 3 * (f1 - f0) + t * (d0 + d1 + 2 * (f0 - f1))));
        if (fabsf(Fhat - u) < 1e-6f || b - a < 1e-6f) break;
        if (Fhat - u < 0) a = t;
        else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + width * t;
}
// TabulatedBSSRDF::Sr, bssrdf.cpp:203-236
PG_DEV Spec bssrdf_sr(const DBssrdf &b, float r) {
    float out[3] = {0, 0, 0};
    for (int ch = 0; ch < 3; ++ch) {
        const float rOptical = r * b.sigma_t[ch];
        int rhoOffset, radiusOffset;
        float rhoW[4], radiusW[4];
        if (!catmull_rom_weights(b.nRho, b.rhoSamples, b.rho[ch], rhoOffset, rhoW) ||
            !catmull_rom_weights(b.nRadius, b.radiusSamples, rOptical, radiusOffset, radiusW))
            continue;
        float sr = 0;
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) {
                const float weight = rhoW[i] * radiusW[j];
                if (weight != 0) sr += weight * b.profile[(rhoOffset + i) * b.nRadius + radiusOffset + j];
            }
        if (rOptical != 0) sr /= 2 * PG_PI * rOptical;
        out[ch] = sr;
    }
    for (int ch = 0; ch < 3; ++ch) {
        out[ch] *= b.sigma_t[ch] * b.sigma_t[ch];
        if (out[ch] < 0) out[ch] = 0;  // Spectrum::Clamp()
    }
    return sp3(out[0], out[1], out[2]);
}
// TabulatedBSSRDF::Sample_Sr, bssrdf.cpp:352-359
PG_DEV float bssrdf_sample_sr(const DBssrdf &b, int ch, float u) {
    if (b.sigma_t[ch] == 0) return -1;
    return sample_catmull_rom_2d(b.nRho, b.nRadius, b.rhoSamples, b.radiusSamples, b.profile, b.profileCDF, b.rho[ch], u) / b.sigma_t[ch];
}
// TabulatedBSSRDF::Pdf_Sr, bssrdf.cpp:361-392
PG_DEV float bssrdf_pdf_sr(const DBssrdf &b, int ch, float r) {
    const float rOptical = r * b.sigma_t[ch];
    int rhoOffset, radiusOffset;
    float rhoW[4], radiusW[4];
    if (!catmull_rom_weights(b.nRho, b.rhoSamples, b.rho[ch], rhoOffset, rhoW) ||
        !catmull_rom_weights(b.nRadius, b.radiusSamples, rOptical, radiusOffset, radiusW))
        return 0.f;
    float sr = 0, rhoEff = 0;
    for (int i = 0; i < 4; ++i) {
        if (rhoW[i] == 0) continue;
        rhoEff += b.rhoEff[rhoOffset + i] * rhoW[i];
        for (int j = 0; j < 4; ++j) {
            if (radiusW[j] == 0) continue;
            sr += b.profile[(rhoOffset + i) * b.nRadius + radiusOffset + j] * rhoW[i] * radiusW[j];
        }
    }
    if (rOptical != 0) sr /= 2 * PG_PI * rOptical;
    return pmax(0.f, sr * b.sigma_t[ch] * b.sigma_t[ch] / rhoEff);
}
// InvertCatmullRom, interpolation.cpp:255-315 (SubsurfaceFromDiffuse of a textured kdsubsurface material, per hit)
PG_DEV float invert_catmull_rom(int n, const float *x, const float *values, float u) {
    if (!(u > values[0])) return x[0];
    if (!(u < values[n - 1])) return x[n - 1];
    const int i = find_interval_le(n, values, u);
    const float x0 = x[i], x1 = x[i + 1], f0 = values[i], f1 = values[i + 1];
    const float width = x1 - x0;
    const float d0 = i > 0 ? width * (f1 - values[i - 1]) / (x1 - x[i - 1]) : f1 - f0;
    const float d1 = i + 2 < n ? width * (values[i + 2] - f0) / (x[i + 2] - x0) : f1 - f0;
    float a = 0, b = 1, t = .5f;
    for (;;) {
        if (!(t > a && t < b)) t = 0.5f * (a + b);
        const float t2 = t * t, t3 = t2 * t;
        const float Fhat = (2 * t3 - 3 * t2 + 1) * f0 + (-2 * t3 + 3 * t2) * f1 + (t3 - 2 * t2 + t) * d0 + (t3 - t2) * d1;
        const float fhat = (6 * t2 - 6 * t) * f0 + (-6 * t2 + 6 * t) * f1 + (3 * t2 - 4 * t + 1) * d0 + (3 * t2 - 2 * t) * d1;
        if (fabsf(Fhat - u) < 1e-6f || b - a < 1e-6f) break;
        if (Fhat - u < 0) a = t;
        else b = t;
        t -= (Fhat - u) / fhat;
    }
    return x0 + t * width;
}
// SeparableBSSRDF::Pdf_Sp, bssrdf.cpp:330-350: the density of an exit point pi (geometric normal n) under the three projection axes
// and three channels Sample_Sp chooses from; (ss, ts, ns) = the entry point's shading frame
PG_DEV float bssrdf_pdf_sp(const DBssrdf &b, V3 ss, V3 ts, V3 ns, V3 poP, V3 piP, V3 piN) {
    const V3 d = poP - piP;
    const float dLocal[3] = {dot(ss, d), dot(ts, d), dot(ns, d)};
    const float nLocal[3] = {dot(ss, piN), dot(ts, piN), dot(ns, piN)};
    const float rProj[3] = {sqrtf(dLocal[1] * dLocal[1] + dLocal[2] * dLocal[2]), sqrtf(dLocal[2] * dLocal[2] + dLocal[0] * dLocal[0]),
                            sqrtf(dLocal[0] * dLocal[0] + dLocal[1] * dLocal[1])};
    const float axisProb[3] = {.25f, .25f, .5f};
    const float chProb = 1 / (float)3;
    float pdf = 0;
    for (int axis = 0; axis < 3; ++axis)
        for (int ch = 0; ch < 3; ++ch) pdf += bssrdf_pdf_sr(b, ch, rProj[axis]) * fabsf(nLocal[axis]) * chProb * axisProb[axis];
    return pdf;
}
// The first half of SeparableBSSRDF::Sample_Sp, bssrdf.cpp:257-292: projection axis, channel, radius, angle -> the probe segment
// [baseP, pTarget] through the sphere of radius rMax around po.  u1 comes back remapped for the choice among the hits
// (selected = Clamp((int)(u1 * nFound), 0, nFound - 1), :321).  cos / sin in double, rounded once (DESIGN.md "libm").
PG_DEV bool bssrdf_probe_segment(const DBssrdf &b, V3 ss, V3 ts, V3 ns, V3 poP, float &u1, float u2x, float u2y, V3 &baseP, V3 &pTarget) {
    V3 vx, vy, vz;
    if (u1 < .5f) { vx = ss; vy = ts; vz = ns; u1 *= 2; }
    else if (u1 < .75f) { vx = ts; vy = ns; vz = ss; u1 = (u1 - .5f) * 4; }
    else { vx = ns; vy = ss; vz = ts; u1 = (u1 - .75f) * 4; }
    int ch = (int)(u1 * 3);
    ch = ch < 0 ? 0 : (ch > 2 ? 2 : ch);
    u1 = u1 * 3 - ch;
    const float r = bssrdf_sample_sr(b, ch, u2x);
    if (r < 0) return false;
    const float phi = 2 * PG_PI * u2y;
    const float rMax = bssrdf_sample_sr(b, ch, 0.999f);
    if (r >= rMax) return false;
    const float l = 2 * sqrtf(rMax * rMax - r * r);
    const float c = pg_cosf(phi), sn = pg_sinf(phi);
    baseP = (poP + (vx * c + vy * sn) * r) - (vz * l) * 0.5f;
    pTarget = baseP + vz * l;
    return true;
}
// SeparableBSSRDFAdapter::f (bssrdf.h:214-219) = Sw(wi) * eta^2 under TransportMode::Radiance; Sw: bssrdf.h:97-100.  cosThetaI = wi.z
// in the exit point's shading frame.  (The adapter samples and weighs like a Lambertian lobe: BxDF::Sample_f / Pdf.)
PG_DEV float bssrdf_adapter_f(float eta, float cosThetaI, float frDielectric) {
    const float c = 1 - 2 * fresnel_moment1(1 / eta);
    return ((1 - frDielectric) / (c * PG_PI)) * (eta * eta);
}
#endif
