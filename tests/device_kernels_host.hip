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
// lobe_f + lobe_pdf of one BxDF in the local frame: f rgb, pdf
void hostdev_lobe_f_pdf(const PgBxDF *b, const float *wo, const float *wi, float *out) {
    const V3 o = mk(wo[0], wo[1], wo[2]), i = mk(wi[0], wi[1], wi[2]);
    const Spec f = lobe_f(*b, o, i);
    out[0] = f.r; out[1] = f.g; out[2] = f.b; out[3] = lobe_pdf(*b, o, i);
}
// lobe_sample_f: f rgb, pdf, wi; returns the sampled type
int hostdev_lobe_sample_f(const PgBxDF *b, const float *wo, float u0, float u1, float *out) {
    V3 wi = mk(0, 0, 0);
    float pdf = 0;
    int sampledType = lobe_type(*b);
    const Spec f = lobe_sample_f(*b, mk(wo[0], wo[1], wo[2]), wi, u0, u1, pdf, sampledType);
    out[0] = f.r; out[1] = f.g; out[2] = f.b; out[3] = pdf; out[4] = wi.x; out[5] = wi.y; out[6] = wi.z;
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
    camera_ray(*rd, fx, fy, lx, ly, o, d, tMax);
    out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = d.x; out[4] = d.y; out[5] = d.z; out[6] = tMax;
}
// SeparableBSSRDFAdapter::f with the shading kernels' own FrDielectric
float hostdev_bssrdf_adapter_f(float eta, float cosThetaI) { return bssrdf_adapter_f(eta, cosThetaI, fr_dielectric(cosThetaI, 1.f, eta)); }
}
