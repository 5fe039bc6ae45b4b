// TEST INFRASTRUCTURE: instantiates every function of pg_grid.h / pg_bssrdf.h in a kernel so that tests/test_device_headers_on_host.py can
// check that the headers are valid gfx950 device code (hipcc --cuda-device-only -c), ahead of their integration into the volpath kernels.
#include "../pbrt-v3_amd/csrc/pg_grid.h"
#include "../pbrt-v3_amd/csrc/pg_bssrdf.h"
struct Lcg { uint32_t s; PG_DEV float operator()() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) * 0x1p-24f; } };
__global__ void k_instantiate(const PgDensityGrid *g, const float *den, const PgBSSRDF *bs, const float *tables, const float *rays, float *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 o = mk(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2]), d = mk(rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5]);
    Lcg draw{(uint32_t)i};
    float t = 0;
    const float Tr = grid_tr(*g, den, o, d, PG_INF, draw);
    const bool hit = grid_sample(*g, den, o, d, PG_INF, draw, t);
    const DBssrdf b = bssrdf_bind(*bs, tables);
    const Spec sr = bssrdf_sr(b, t);
    V3 base, target;
    float u1 = draw();
    const bool probe = bssrdf_probe_segment(b, mk(1, 0, 0), mk(0, 1, 0), mk(0, 0, 1), o, u1, draw(), draw(), base, target);
    out[i] = Tr + (hit ? t : 0.f) + sr.r + bssrdf_pdf_sr(b, 1, t) + bssrdf_sample_sr(b, 2, draw()) + fresnel_moment1(1 / b.eta) + grid_density(*g, den, o) +
             bssrdf_pdf_sp(b, mk(1, 0, 0), mk(0, 1, 0), mk(0, 0, 1), o, o + d, d) + (probe ? base.x + target.y + u1 : 0.f) +
             invert_catmull_rom(b.nRho, b.rhoSamples, b.rhoEff, draw()) + bssrdf_adapter_f(b.eta, d.z, 0.04f);
}
