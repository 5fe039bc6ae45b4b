// pg_grid.h -- GridDensityMedium (media/grid.{h,cpp} of the reference) as device functions over the ABI's PgDensityGrid.
//
// Used by k_shade<., ., ., GRID> (delta tracking at the start of a vertex) and k_through<., GRID> (ratio tracking of the transmittance
// rays) in pg_kernels.hip.  Pinned on the host as well: tests/test_device_headers_on_host.py compiles this header for the host and
// compares every function with the oracle's correctly-rounded-libm build, bit for bit, on random grids, rays and draw streams.
//
// `Draw` is the path's sampler.Get1D() (a callable returning float); the loops call it in the reference's order.  logf is
// evaluated in double and rounded once, like the other libm calls of the device path (DESIGN.md "libm").
#ifndef PG_GRID_H
#define PG_GRID_H
#include "pg_sphere.h"  // m4_point_err, m4_vec

// GridDensityMedium::D, grid.h:77-81
PG_DEV float grid_d(const PgDensityGrid &g, const float *den, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= g.nx || y >= g.ny || z >= g.nz) return 0.f;
    return den[((size_t)z * g.ny + y) * g.nx + x];
}
// GridDensityMedium::Density, grid.cpp:45-59: trilinear, samples at voxel centres
PG_DEV float grid_density(const PgDensityGrid &g, const float *den, V3 p) {
    const float sx = p.x * g.nx - .5f, sy = p.y * g.ny - .5f, sz = p.z * g.nz - .5f;
    const int ix = (int)floorf(sx), iy = (int)floorf(sy), iz = (int)floorf(sz);
    const float dx = sx - (float)ix, dy = sy - (float)iy, dz = sz - (float)iz;
    const float d00 = plerp(dx, grid_d(g, den, ix, iy, iz), grid_d(g, den, ix + 1, iy, iz));
    const float d10 = plerp(dx, grid_d(g, den, ix, iy + 1, iz), grid_d(g, den, ix + 1, iy + 1, iz));
    const float d01 = plerp(dx, grid_d(g, den, ix, iy, iz + 1), grid_d(g, den, ix + 1, iy, iz + 1));
    const float d11 = plerp(dx, grid_d(g, den, ix, iy + 1, iz + 1), grid_d(g, den, ix + 1, iy + 1, iz + 1));
    return plerp(dz, plerp(dy, d00, d10), plerp(dy, d01, d11));
}
// The head of Sample and Tr (grid.cpp:64-70, :92-98): the world ray with a normalized direction carried into the medium's unit cube
// (Transform::operator()(Ray), transform.h:249-262) and clipped against it (Bounds3::IntersectP, geometry.h:1388-1409).
PG_DEV bool grid_enter(const PgDensityGrid &g, V3 ro, V3 rd, float rtMax, V3 &o, V3 &d, float &tMin, float &tMax) {
    const float l = sqrtf(lensq(rd));
    const V3 dn = normalize(rd);
    float rayTMax = rtMax * l;
    V3 oError;
    o = m4_point_err(g.world_to_medium, ro, oError);
    d = m4_vec(g.world_to_medium, dn);
    const float lengthSquared = lensq(d);
    if (lengthSquared > 0) {
        const float dt = dot(vabs(d), oError) / lengthSquared;
        o = o + d * dt;
        rayTMax -= dt;
    }
    const float oc[3] = {o.x, o.y, o.z}, dc[3] = {d.x, d.y, d.z};
    float t0 = 0, t1 = rayTMax;
    for (int i = 0; i < 3; ++i) {
        const float invRayDir = 1 / dc[i];
        float tNear = (0 - oc[i]) * invRayDir, tFar = (1 - oc[i]) * invRayDir;
        if (tNear > tFar) { const float t = tNear; tNear = tFar; tFar = t; }
        tFar *= 1 + 2 * pgamma(3);
        t0 = tNear > t0 ? tNear : t0;
        t1 = tFar < t1 ? tFar : t1;
        if (t0 > t1) return false;
    }
    tMin = t0; tMax = t1;
    return true;
}
PG_DEV float grid_step(const PgDensityGrid &g, float u) {  // -log(1 - u) * invMaxDensity / sigma_t, log in double
    return pg_logf((1 - u)) * g.inv_max_density / g.sigma_t;
}
// GridDensityMedium::Tr, grid.cpp:88-120: ratio tracking with Russian roulette below 0.1
template <class Draw>
PG_DEV float grid_tr(const PgDensityGrid &g, const float *den, V3 ro, V3 rd, float rtMax, Draw &&draw) {
    V3 o, d;
    float tMin, tMax;
    if (!grid_enter(g, ro, rd, rtMax, o, d, tMin, tMax)) return 1.f;
    float Tr = 1, t = tMin;
    for (;;) {
        t -= grid_step(g, draw());
        if (t >= tMax) break;
        const float density = grid_density(g, den, o + d * t);
        Tr *= 1 - pmax(0.f, density * g.inv_max_density);
        if (Tr < .1f) {
            const float q = pmax(.05f, 1 - Tr);
            if (draw() < q) return 0.f;
            Tr /= 1 - q;
        }
    }
    return Tr;
}
// GridDensityMedium::Sample, grid.cpp:61-86: delta tracking; true = a medium interaction at world-ray parameter t
template <class Draw>
PG_DEV bool grid_sample(const PgDensityGrid &g, const float *den, V3 ro, V3 rd, float rtMax, Draw &&draw, float &t) {
    V3 o, d;
    float tMin, tMax;
    if (!grid_enter(g, ro, rd, rtMax, o, d, tMin, tMax)) return false;
    t = tMin;
    for (;;) {
        t -= grid_step(g, draw());
        if (t >= tMax) return false;
        if (grid_density(g, den, o + d * t) * g.inv_max_density > draw()) return true;
    }
}
#endif
