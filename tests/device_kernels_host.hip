// TEST INFRASTRUCTURE: pbrt-v3_amd/csrc/pg_kernels.hip -- the shading kernels' translation unit -- compiled for the HOST with the same
// shim as tests/device_headers_host.hip: its __device__ functions (samplers, camera, BSDFs, lights ...) become callable without a GPU;
// the __global__ kernels themselves are parsed but never run here.  tests/test_device_headers_on_host.py drives the BxDF library.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstring>
__host__ inline unsigned int __float_as_uint(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
__host__ inline float __uint_as_float(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
__host__ inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
__host__ inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
__host__ inline bool isinf(float v) { return __builtin_isinf(v); }
__host__ inline bool isnan(float v) { return __builtin_isnan(v); }
__host__ inline unsigned long long __brevll(unsigned long long v) { return __builtin_bitreverse64(v); }
#undef __device__
#define __device__ __attribute__((device)) __attribute__((host))
// the kernels themselves: never launched here.  As unused static host+device functions they are parsed and dropped (no host stubs,
// no fat-binary registration), so the library loads without a HIP runtime device.
#undef __global__
#define __global__ __attribute__((device)) __attribute__((host)) static
#undef __launch_bounds__
#define __launch_bounds__(...)
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(...) ((void)0)  // the launch wrappers of the translation unit compile to nothing
#include "../pbrt-v3_amd/csrc/pg_kernels.hip"
#include "../pbrt-v3_amd/csrc/pg_bssrdf.h"

extern "C" {
// f and pdf of one BxDF in the local frame (f rgb, pdf) as the kernels take them: lobe_f_pdf -- one evaluation for both, and for
// either alone -- and the separate lobe_f / lobe_pdf; a difference between any two of them comes back as NaNs
static bool same_bits(float a, float b) { return std::memcmp(&a, &b, 4) == 0; }
void hostdev_lobe_f_pdf(const PgBxDF *b, const float *wo, const float *wi, float *out) {
    const V3 o = mk(wo[0], wo[1], wo[2]), i = mk(wi[0], wi[1], wi[2]);
    Spec f, fOnly, fNone;
    float pdf, pdfOnly, pdfNone;
    lobe_f_pdf(*b, b->type, o, i, true, true, f, pdf);
    lobe_f_pdf(*b, b->type, o, i, true, false, fOnly, pdfNone);
    lobe_f_pdf(*b, b->type, o, i, false, true, fNone, pdfOnly);
    const Spec fs = lobe_f(*b, b->type, o, i);
    const float ps = lobe_pdf(*b, b->type, o, i);
    out[0] = f.r; out[1] = f.g; out[2] = f.b; out[3] = pdf;
    const bool ok = same_bits(f.r, fOnly.r) && same_bits(f.g, fOnly.g) && same_bits(f.b, fOnly.b) && same_bits(pdf, pdfOnly) &&
                    same_bits(f.r, fs.r) && same_bits(f.g, fs.g) && same_bits(f.b, fs.b) && same_bits(pdf, ps) &&
                    pdfNone == 0 && fNone.r == 0 && fNone.g == 0 && fNone.b == 0;
    if (!ok) out[0] = out[1] = out[2] = out[3] = __builtin_nanf("");
}
// lobe_sample_f: f rgb, pdf, wi; returns the sampled type.  Without the value (wantF = false, what BSDF::Sample_f asks of the
// non-specular BxDFs) direction, pdf and type must be the same
int hostdev_lobe_sample_f(const PgBxDF *b, const float *wo, float u0, float u1, float *out) {
    V3 wi = mk(0, 0, 0), wi2 = mk(0, 0, 0);
    float pdf = 0, pdf2 = 0;
    int sampledType = lobe_type(b->type), sampledType2 = sampledType;
    const Spec f = lobe_sample_f(*b, b->type, mk(wo[0], wo[1], wo[2]), wi, u0, u1, pdf, sampledType);
    lobe_sample_f(*b, b->type, mk(wo[0], wo[1], wo[2]), wi2, u0, u1, pdf2, sampledType2, false);
    out[0] = f.r; out[1] = f.g; out[2] = f.b; out[3] = pdf; out[4] = wi.x; out[5] = wi.y; out[6] = wi.z;
    if (!(same_bits(pdf, pdf2) && same_bits(wi.x, wi2.x) && same_bits(wi.y, wi2.y) && same_bits(wi.z, wi2.z) && sampledType == sampledType2)) out[3] = __builtin_nanf("");
    return sampledType;
}
// HenyeyGreenstein (core/medium.h:69-72, medium.cpp:194-213) and the Halton sample index of a pixel (samplers/halton.cpp:92-116)
float hostdev_phase_hg(float cosTheta, float g) { return phase_hg(cosTheta, g); }
float hostdev_hg_sample_p(float g, const float *wo, float u0, float u1, float *wi) {
    V3 w = mk(0, 0, 0);
    const float p = hg_sample_p(g, mk(wo[0], wo[1], wo[2]), w, u0, u1);
    wi[0] = w.x; wi[1] = w.y; wi[2] = w.z;
    return p;
}
long long hostdev_halton_index(const PgRenderDesc *rd, int px, int py, long long sampleNum) { return (long long)halton_index(*rd, px, py, (uint64_t)sampleNum); }
// Camera::GenerateRay as k_generate computes it: o, d, tMax
void hostdev_camera_ray(const PgRenderDesc *rd, float fx, float fy, float lx, float ly, float *out) {
    V3 o, d;
    float tMax;
    camera_ray(*rd, rd->camera_to_world, fx, fy, lx, ly, o, d, tMax);
    out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = d.x; out[4] = d.y; out[5] = d.z; out[6] = tMax;
}
// SeparableBSSRDFAdapter::f with the shading kernels' own FrDielectric
float hostdev_bssrdf_adapter_f(float eta, float cosThetaI) { return bssrdf_adapter_f(eta, cosThetaI, fr_dielectric(cosThetaI, 1.f, eta)); }
}
