// bssrdf.cpp -- the tables behind SubsurfaceMaterial / KdSubsurfaceMaterial: the photon beam diffusion profile
// (core/bssrdf.cpp:43-180 of the reference) over BSSRDFTable(100, 64)'s albedo x radius grid, its per-albedo integral and CDF
// (core/interpolation.cpp:228-253), and the albedo inversion of KdSubsurfaceMaterial (bssrdf.cpp:182-191, interpolation.cpp:255-315).
//
// The reference computes these once per material in the material's constructor (materials/subsurface.h:73-75) on the host; so does
// this front end.  The table is handed over through the C ABI (PgBSSRDF, include/pbrt_gpu.h) and read by TabulatedBSSRDF's
// restatements.  Order-constrained arithmetic: every expression below keeps the reference's operand order and operand TYPES --
// including the two places where the reference computes in double without meaning to (a literal without its `f` in
// FresnelMoment1's cubic term, bssrdf.cpp:48; std::exp of an int in the albedo grid's normalisation, :163) -- since the goldens
// are compared bit for bit.  Everything else (one albedo row per task, the shared spline-segment helper) is this file's own.
#include "scene.h"
#include "api.h"

#include <cmath>
#include <thread>

namespace pbrt {

static const Float kPi = 3.14159265358979323846;
static const Float kInv4Pi = 0.07957747154594766788;

// FrDielectric (reflection.cpp:47-69) and PhaseHG (medium.h:69-72): the single-scattering term needs both on the host
static Float fresnelDielectric(Float cosI, Float etaI, Float etaT) {
    cosI = Clamp(cosI, -1, 1);
    if (!(cosI > 0.f)) { std::swap(etaI, etaT); cosI = std::abs(cosI); }
    const Float sinI = std::sqrt(std::max((Float)0, 1 - cosI * cosI));
    const Float sinT = etaI / etaT * sinI;
    if (sinT >= 1) return 1;
    const Float cosT = std::sqrt(std::max((Float)0, 1 - sinT * sinT));
    const Float parl = ((etaT * cosI) - (etaI * cosT)) / ((etaT * cosI) + (etaI * cosT));
    const Float perp = ((etaI * cosI) - (etaT * cosT)) / ((etaI * cosI) + (etaT * cosT));
    return (parl * parl + perp * perp) / 2;
}
static Float henyeyGreenstein(Float cosTheta, Float g) {
    const Float denom = 1 + g * g + 2 * g * cosTheta;
    return kInv4Pi * (1 - g * g) / (denom * std::sqrt(denom));
}

Float FresnelMoment1(Float eta) {  // bssrdf.cpp:43-52; the eta3 coefficient is a double literal there
    const Float e2 = eta * eta, e3 = e2 * eta, e4 = e3 * eta, e5 = e4 * eta;
    if (eta < 1) return 0.45966f - 1.73965f * eta + 3.37668f * e2 - 3.904945 * e3 + 2.49277f * e4 - 0.68441f * e5;
    return -4.61686f + 11.1136f * eta - 10.4646f * e2 + 5.11455f * e3 - 1.27198f * e4 + 0.12746f * e5;
}
Float FresnelMoment2(Float eta) {  // bssrdf.cpp:54-66
    const Float e2 = eta * eta, e3 = e2 * eta, e4 = e3 * eta, e5 = e4 * eta;
    if (eta < 1) return 0.27614f - 0.87350f * eta + 1.12077f * e2 - 0.65095f * e3 + 0.07883f * e4 + 0.04860f * e5;
    const Float r1 = 1 / eta, r2 = r1 * r1, r3 = r2 * r1;
    return -547.033f + 45.3087f * r3 - 218.725f * r2 + 458.843f * r1 + 404.557f * eta - 189.519f * e2 + 54.9327f * e3 - 9.00603f * e4 + 0.63942f * e5;
}

static const int kBeamSamples = 100;
// depth of the i-th of the 100 exponentially spaced samples along the beam, before the division by the coefficient
static inline Float beamSampleLog(int i) { return std::log(1 - (i + .5f) / kBeamSamples); }

// multiple scattering: the dipole of Equation (15.27) integrated along the beam, bssrdf.cpp:68-124
static Float beamDiffusionMultiple(Float sigma_s, Float sigma_a, Float g, Float eta, Float r) {
    const Float sigmap_s = sigma_s * (1 - g);
    const Float sigmap_t = sigma_a + sigmap_s;
    const Float rhop = sigmap_s / sigmap_t;
    const Float D_g = (2 * sigma_a + sigmap_s) / (3 * sigmap_t * sigmap_t);
    const Float sigma_tr = std::sqrt(sigma_a / D_g);
    const Float fm1 = FresnelMoment1(eta), fm2 = FresnelMoment2(eta);
    const Float ze = -2 * D_g * (1 + 3 * fm2) / (1 - 2 * fm1);
    const Float cPhi = .25f * (1 - 2 * fm1), cE = .5f * (1 - 3 * fm2);
    Float Ed = 0;
    for (int i = 0; i < kBeamSamples; ++i) {
        const Float zr = -beamSampleLog(i) / sigmap_t;  // real source depth
        const Float zv = -zr + 2 * ze;                  // virtual source
        const Float dr = std::sqrt(r * r + zr * zr), dv = std::sqrt(r * r + zv * zv);
        const Float phiD = kInv4Pi / D_g * (std::exp(-sigma_tr * dr) / dr - std::exp(-sigma_tr * dv) / dv);
        const Float EDn = kInv4Pi * (zr * (1 + sigma_tr * dr) * std::exp(-sigma_tr * dr) / (dr * dr * dr) -
                                     zv * (1 + sigma_tr * dv) * std::exp(-sigma_tr * dv) / (dv * dv * dv));
        const Float E = phiD * cPhi + EDn * cE;
        const Float kappa = 1 - std::exp(-2 * sigmap_t * (dr + zr));
        Ed += kappa * rhop * rhop * E;
    }
    return Ed / kBeamSamples;
}
// single scattering below the critical angle, bssrdf.cpp:126-147
static Float beamDiffusionSingle(Float sigma_s, Float sigma_a, Float g, Float eta, Float r) {
    const Float sigma_t = sigma_a + sigma_s, rho = sigma_s / sigma_t;
    const Float tCrit = r * std::sqrt(eta * eta - 1);
    Float Ess = 0;
    for (int i = 0; i < kBeamSamples; ++i) {
        const Float ti = tCrit - beamSampleLog(i) / sigma_t;
        const Float d = std::sqrt(r * r + ti * ti);
        const Float cosThetaO = ti / d;
        Ess += rho * std::exp(-sigma_t * (d + tCrit)) / (d * d) * henyeyGreenstein(cosThetaO, g) *
               (1 - fresnelDielectric(-cosThetaO, 1, eta)) * std::abs(cosThetaO);
    }
    return Ess / kBeamSamples;
}

// One Catmull-Rom segment [x_i, x_i+1] of a non-uniform spline: end values and the derivative estimates the reference's
// three spline routines share (interpolation.cpp:234-247, :266-279)
struct SplineSegment { Float x0, width, f0, f1, d0, d1; };
static SplineSegment splineSegment(int n, const Float *x, const Float *f, int i) {
    SplineSegment s;
    s.x0 = x[i];
    const Float x1 = x[i + 1];
    s.f0 = f[i]; s.f1 = f[i + 1];
    s.width = x1 - s.x0;
    s.d0 = i > 0 ? s.width * (s.f1 - f[i - 1]) / (x1 - x[i - 1]) : s.f1 - s.f0;
    s.d1 = i + 2 < n ? s.width * (f[i + 2] - s.f0) / (x[i + 2] - s.x0) : s.f1 - s.f0;
    return s;
}
// IntegrateCatmullRom, interpolation.cpp:228-253: running integral at every node, total returned
static Float integrateSpline(int n, const Float *x, const Float *f, Float *cdf) {
    Float sum = 0;
    cdf[0] = 0;
    for (int i = 0; i < n - 1; ++i) {
        const SplineSegment s = splineSegment(n, x, f, i);
        sum += ((s.d0 - s.d1) * (1.f / 12.f) + (s.f0 + s.f1) * .5f) * s.width;
        cdf[i + 1] = sum;
    }
    return sum;
}
// InvertCatmullRom, interpolation.cpp:255-315: the x at which the (monotone) spline takes the value u -- Newton steps kept
// inside a bisection bracket
static Float invertSpline(int n, const Float *x, const Float *f, Float u) {
    if (!(u > f[0])) return x[0];
    if (!(u < f[n - 1])) return x[n - 1];
    // FindInterval(n, f[i] <= u), pbrt.h:402-415
    int first = 0, len = n;
    while (len > 0) {
        const int half = len >> 1, middle = first + half;
        if (f[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    const int i = Clamp(first - 1, 0, n - 2);
    const SplineSegment s = splineSegment(n, x, f, i);
    Float a = 0, b = 1, t = .5f;
    for (;;) {
        if (!(t > a && t < b)) t = 0.5f * (a + b);
        const Float t2 = t * t, t3 = t2 * t;
        const Float Fhat = (2 * t3 - 3 * t2 + 1) * s.f0 + (-2 * t3 + 3 * t2) * s.f1 + (t3 - 2 * t2 + t) * s.d0 + (t3 - t2) * s.d1;
        const Float fhat = (6 * t2 - 6 * t) * s.f0 + (-6 * t2 + 6 * t) * s.f1 + (3 * t2 - 4 * t + 1) * s.d0 + (3 * t2 - 2 * t) * s.d1;
        if (std::abs(Fhat - u) < 1e-6f || b - a < 1e-6f) break;
        if (Fhat - u < 0) a = t; else b = t;
        t -= (Fhat - u) / fhat;
    }
    return s.x0 + t * s.width;
}

// ComputeBeamDiffusionBSSRDF(g, eta, &BSSRDFTable(nRho, nRadius)), bssrdf.cpp:149-180.  Layout of `out` (PgBSSRDF.table):
// rhoSamples[nRho], radiusSamples[nRadius], profile[nRho * nRadius], rhoEff[nRho], profileCDF[nRho * nRadius].
void ComputeBeamDiffusionTable(Float g, Float eta, int nRho, int nRadius, std::vector<float> *out) {
    out->assign((size_t)nRho + nRadius + 2 * (size_t)nRho * nRadius + nRho, 0.f);
    float *rhoSamples = out->data(), *radius = rhoSamples + nRho, *profile = radius + nRadius;
    float *rhoEff = profile + (size_t)nRho * nRadius, *cdf = rhoEff + nRho;
    radius[0] = 0;
    radius[1] = 2.5e-3f;
    for (int j = 2; j < nRadius; ++j) radius[j] = radius[j - 1] * 1.2f;
    // (the denominator is a double in the reference: std::exp(int))
    for (int i = 0; i < nRho; ++i) rhoSamples[i] = (1 - std::exp(-8 * i / (Float)(nRho - 1))) / (1 - std::exp((double)-8));
    // one albedo row per task (the reference: ParallelFor over the rows; every row is independent)
    int nt = PbrtOptions.nThreads > 0 ? PbrtOptions.nThreads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, nRho));
    auto rows = [&](int t) {
        for (int i = t; i < nRho; i += nt) {
            const Float rho = rhoSamples[i];
            for (int j = 0; j < nRadius; ++j) {
                const Float r = radius[j];
                profile[(size_t)i * nRadius + j] = 2 * kPi * r * (beamDiffusionSingle(rho, 1 - rho, g, eta, r) + beamDiffusionMultiple(rho, 1 - rho, g, eta, r));
            }
            rhoEff[i] = integrateSpline(nRadius, radius, profile + (size_t)i * nRadius, cdf + (size_t)i * nRadius);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < nt; ++t) pool.emplace_back(rows, t);
    rows(0);
    for (auto &th : pool) th.join();
}

// SubsurfaceFromDiffuse, bssrdf.cpp:182-191: per channel the single-scattering albedo whose effective albedo is rhoEff[c]
void SubsurfaceFromDiffuse(const float *table, int nRho, int nRadius, const Float kd[3], const Float mfp[3], Float sigma_a[3], Float sigma_s[3]) {
    const float *rhoSamples = table, *rhoEff = table + nRho + nRadius + (size_t)nRho * nRadius;
    for (int c = 0; c < 3; ++c) {
        const Float rho = invertSpline(nRho, rhoSamples, rhoEff, kd[c]);
        sigma_s[c] = rho / mfp[c];
        sigma_a[c] = (1 - rho) / mfp[c];
    }
}

}  // namespace pbrt
