// pg_texture.h -- texture evaluation on the device, shared by the shading kernels (materials' texture parameters,
// pg_kernels.hip) and the traversal kernel (alpha masks of triangle meshes, pg_traverse.hip).
#ifndef PG_TEXTURE_H
#define PG_TEXTURE_H
#include "pg_device.h"
#include "pg_sphere.h"
#include "pg_kernels.h"

PG_DEV float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }  // pbrt.h:305-311
PG_DEV Spec sp_of(const float *c) { return sp3(c[0], c[1], c[2]); }
#define PG_INV2PI 0.15915494309189533577f
PG_DEV float spherical_theta(V3 v) { return pg_acosf(clampf(v.z, -1, 1)); }  // geometry.h:1468-1470
PG_DEV float spherical_phi(V3 v) { float p = pg_atan2f(v.y, v.x); return (p < 0) ? (p + 2 * PG_PI) : p; }  // :1472-1475
// ===========================================================================
// Textures (core/texture.{h,cpp}, textures/{scale,mix,checkerboard,uv,bilerp}.h) and the per-hit evaluation of a textured
// material's ComputeScatteringFunctions.  Texture nodes reference their operands; the nesting is unrolled over a template
// depth (PG_TEX_DEPTH levels, the host front end refuses deeper graphs) so the call graph has no recursion.
// ===========================================================================
#define PG_TEX_DEPTH 3
#define PG_DEV_CALL __device__ __noinline__
struct TexHit { V3 p, dpdx, dpdy; float u, v, dudx, dvdx, dudy, dvdy; };
PG_DEV void tex_sphere(const PgTexture &t, V3 p, float &s0, float &s1) {  // SphericalMapping2D::sphere, texture.cpp:122-127
    const V3 vec = normalize(m4_point(t.w2t, p) - mk(0, 0, 0));
    const float theta = spherical_theta(vec), phi = spherical_phi(vec);
    s0 = theta * PG_INVPI; s1 = phi * PG_INV2PI;
}
PG_DEV void tex_cylinder(const PgTexture &t, V3 p, float &s0, float &s1) {  // CylindricalMapping2D::cylinder, texture.h:93-96
    const V3 vec = normalize(m4_point(t.w2t, p) - mk(0, 0, 0));
    s0 = (PG_PI + pg_atan2f(vec.y, vec.x)) * PG_INV2PI; s1 = vec.z;
}
PG_DEV void tex_map2d(const PgTexture &t, const TexHit &h, float st[2], float dstdx[2], float dstdy[2]) {
    if (t.mapping == PG_MAP_UV) {  // UVMapping2D::Map, texture.cpp:93-100
        dstdx[0] = t.su * h.dudx; dstdx[1] = t.sv * h.dvdx;
        dstdy[0] = t.su * h.dudy; dstdy[1] = t.sv * h.dvdy;
        st[0] = t.su * h.u + t.du; st[1] = t.sv * h.v + t.dv;
    } else if (t.mapping == PG_MAP_PLANAR) {  // PlanarMapping2D::Map, texture.cpp:150-157
        const V3 vs = mk(t.vs[0], t.vs[1], t.vs[2]), vt = mk(t.vt[0], t.vt[1], t.vt[2]);
        dstdx[0] = dot(h.dpdx, vs); dstdx[1] = dot(h.dpdx, vt);
        dstdy[0] = dot(h.dpdy, vs); dstdy[1] = dot(h.dpdy, vt);
        st[0] = t.du + dot(h.p, vs); st[1] = t.dv + dot(h.p, vt);
    } else {  // SphericalMapping2D::Map (texture.cpp:102-120) / CylindricalMapping2D::Map (:129-148)
        const bool sph = t.mapping == PG_MAP_SPHERICAL;
        const float delta = sph ? .1f : .01f;
        float sx0, sx1, sy0, sy1;
        if (sph) { tex_sphere(t, h.p, st[0], st[1]); tex_sphere(t, h.p + h.dpdx * delta, sx0, sx1); tex_sphere(t, h.p + h.dpdy * delta, sy0, sy1); }
        else { tex_cylinder(t, h.p, st[0], st[1]); tex_cylinder(t, h.p + h.dpdx * delta, sx0, sx1); tex_cylinder(t, h.p + h.dpdy * delta, sy0, sy1); }
        const float inv = 1.f / delta;
        dstdx[0] = (sx0 - st[0]) * inv; dstdx[1] = (sx1 - st[1]) * inv;
        dstdy[0] = (sy0 - st[0]) * inv; dstdy[1] = (sy1 - st[1]) * inv;
        if ((double)dstdx[1] > .5) dstdx[1] = 1 - dstdx[1];
        else if (dstdx[1] < -.5f) dstdx[1] = -(dstdx[1] + 1);
        if ((double)dstdy[1] > .5) dstdy[1] = 1 - dstdy[1];
        else if (dstdy[1] < -.5f) dstdy[1] = -(dstdy[1] + 1);
    }
}
// ---- MIPMap<T>::Lookup (core/mipmap.h:189-331) over the pyramid the host built; Spec carries 1 (r only) or 3 channels
// Mod(a, b), pbrt.h:314-317.  A texel coordinate is inside [0, b) except at the map's border: there the result is a itself and the integer
// division (about forty vector instructions, twice per texel) is skipped -- the same value either way.
PG_DEV int mod_i(int a, int b) { if ((unsigned)a < (unsigned)b) return a; int r = a - (a / b) * b; return (r < 0) ? r + b : r; }
PG_DEV Spec mip_texel(const DScene &sc, const PgImage &im, int level, int s, int t) {  // mipmap.h:189-212
    const int sRes = max(1, im.width >> level), tRes = max(1, im.height >> level);
    if (im.wrap == 0) { s = mod_i(s, sRes); t = mod_i(t, tRes); }
    else if (im.wrap == 2) { s = s < 0 ? 0 : (s > sRes - 1 ? sRes - 1 : s); t = t < 0 ? 0 : (t > tRes - 1 ? tRes - 1 : t); }
    else if (s < 0 || s >= sRes || t < 0 || t >= tRes) return sp(0);
    const float *p = sc.texels + im.level_offset[level] + ((size_t)t * sRes + s) * (im.is_float ? 1 : 3);
    return im.is_float ? sp3(p[0], 0, 0) : sp3(p[0], p[1], p[2]);
}
PG_DEV Spec mip_triangle(const DScene &sc, const PgImage &im, int level, float st0, float st1) {  // mipmap.h:231-243
    level = level < 0 ? 0 : (level > im.n_levels - 1 ? im.n_levels - 1 : level);
    const int sRes = max(1, im.width >> level), tRes = max(1, im.height >> level);
    const float s = st0 * sRes - 0.5f, t = st1 * tRes - 0.5f;
    const int s0 = (int)floorf(s), t0 = (int)floorf(t);
    const float ds = s - s0, dt = t - t0;
    return mip_texel(sc, im, level, s0, t0) * ((1 - ds) * (1 - dt)) + mip_texel(sc, im, level, s0, t0 + 1) * ((1 - ds) * dt) +
           mip_texel(sc, im, level, s0 + 1, t0) * (ds * (1 - dt)) + mip_texel(sc, im, level, s0 + 1, t0 + 1) * (ds * dt);
}
PG_DEV float log2_pbrt(float x) { return pg_logf(x) * 1.442695040888963387004650940071f; }  // pbrt.h:328-331
PG_DEV Spec mip_ewa(const DScene &sc, const PgImage &im, int level, float st0, float st1, float d00, float d01, float d10, float d11) {  // mipmap.h:276-327
    if (level >= im.n_levels) return mip_texel(sc, im, im.n_levels - 1, 0, 0);
    const int sRes = max(1, im.width >> level), tRes = max(1, im.height >> level);
    st0 = st0 * sRes - 0.5f; st1 = st1 * tRes - 0.5f;
    d00 *= sRes; d01 *= tRes; d10 *= sRes; d11 *= tRes;
    float A = d01 * d01 + d11 * d11 + 1;
    float B = -2 * (d00 * d01 + d10 * d11);
    float C = d00 * d00 + d10 * d10 + 1;
    const float invF = 1 / (A * C - B * B * 0.25f);
    A *= invF; B *= invF; C *= invF;
    const float det = -B * B + 4 * A * C;
    const float invDet = 1 / det;
    const float uSqrt = sqrtf(det * C), vSqrt = sqrtf(A * det);
    const int s0 = (int)ceilf(st0 - 2 * invDet * uSqrt), s1 = (int)floorf(st0 + 2 * invDet * uSqrt);
    const int t0 = (int)ceilf(st1 - 2 * invDet * vSqrt), t1 = (int)floorf(st1 + 2 * invDet * vSqrt);
    Spec sum = sp(0);
    float sumWts = 0;
    for (int it = t0; it <= t1; ++it) {
        const float tt = it - st1;
        for (int is = s0; is <= s1; ++is) {
            const float ss = is - st0;
            const float r2 = A * ss * ss + B * ss * tt + C * tt * tt;
            if (r2 < 1) {
                int index = (int)(r2 * 128);
                if (index > 127) index = 127;
                const float weight = sc.ewaLut[index];
                sum = sum + mip_texel(sc, im, level, is, it) * weight;
                sumWts += weight;
            }
        }
    }
    return sum / sumWts;
}
PG_DEV Spec mip_lookup(const DScene &sc, const PgImage &im, const float st[2], const float dstdx[2], const float dstdy[2]) {
    if (im.trilinear) {  // mipmap.h:245-251 -> :214-229
        const float width = pmax(pmax(fabsf(dstdx[0]), fabsf(dstdx[1])), pmax(fabsf(dstdy[0]), fabsf(dstdy[1])));
        const float level = im.n_levels - 1 + log2_pbrt(pmax(width, 1e-8f));
        if (level < 0) return mip_triangle(sc, im, 0, st[0], st[1]);
        if (level >= im.n_levels - 1) return mip_texel(sc, im, im.n_levels - 1, 0, 0);
        const int iLevel = (int)floorf(level);
        const float delta = level - iLevel;
        return mip_triangle(sc, im, iLevel, st[0], st[1]) * (1 - delta) + mip_triangle(sc, im, iLevel + 1, st[0], st[1]) * delta;
    }
    float d00 = dstdx[0], d01 = dstdx[1], d10 = dstdy[0], d11 = dstdy[1];
    if (d00 * d00 + d01 * d01 < d10 * d10 + d11 * d11) { float a = d00, b = d01; d00 = d10; d01 = d11; d10 = a; d11 = b; }
    const float majorLength = sqrtf(d00 * d00 + d01 * d01);
    float minorLength = sqrtf(d10 * d10 + d11 * d11);
    if (minorLength * im.max_anisotropy < majorLength && minorLength > 0) {
        const float scale = majorLength / (minorLength * im.max_anisotropy);
        d10 *= scale; d11 *= scale;
        minorLength *= scale;
    }
    if (minorLength == 0) return mip_triangle(sc, im, 0, st[0], st[1]);
    const float lod = pmax(0.f, im.n_levels - 1.f + log2_pbrt(minorLength));
    const int ilod = (int)floorf(lod);
    const float dl = lod - ilod;
    return mip_ewa(sc, im, ilod, st[0], st[1], d00, d01, d10, d11) * (1 - dl) + mip_ewa(sc, im, ilod + 1, st[0], st[1], d00, d01, d10, d11) * dl;
}
PG_DEV float tex_bump_int(float x) {  // checkerboard.h:92-96
    return (float)(int)floorf(x / 2) + 2 * pmax(x / 2 - (float)(int)floorf(x / 2) - 0.5f, 0.f);
}
// Checkerboard2DTexture::Evaluate up to the choice / blend of its operands: which = 0 (tex1), 1 (tex2), 2 (blend by area2)
PG_DEV int tex_checker2d(const PgTexture &t, const TexHit &h, float &area2) {
    float st[2], dstdx[2], dstdy[2];
    tex_map2d(t, h, st, dstdx, dstdy);
    const int point = (((int)floorf(st[0]) + (int)floorf(st[1])) % 2 == 0) ? 0 : 1;
    if (t.aa_none) return point;
    const float ds = pmax(fabsf(dstdx[0]), fabsf(dstdy[0]));
    const float dt = pmax(fabsf(dstdx[1]), fabsf(dstdy[1]));
    const float s0 = st[0] - ds, s1 = st[0] + ds;
    const float t0 = st[1] - dt, t1 = st[1] + dt;
    if (floorf(s0) == floorf(s1) && floorf(t0) == floorf(t1)) return point;
    const float sint = (tex_bump_int(s1) - tex_bump_int(s0)) / (2 * ds);
    const float tint = (tex_bump_int(t1) - tex_bump_int(t0)) / (2 * dt);
    area2 = sint + tint - 2 * sint * tint;
    if (ds > 1 || dt > 1) area2 = .5f;
    return 2;
}
PG_DEV int tex_checker3d(const PgTexture &t, const TexHit &h) {  // checkerboard.h:120-129, IdentityMapping3D texture.cpp:159-164
    const V3 p = m4_point(t.w2t, h.p);
    return (((int)floorf(p.x) + (int)floorf(p.y) + (int)floorf(p.z)) % 2 == 0) ? 0 : 1;
}
// ---- Perlin noise (core/texture.cpp:164-252): Noise, FBm, Turbulence over the reference's permutation table -----------------
PG_DEV float noise_grad(const int *perm, int x, int y, int z, float dx, float dy, float dz) {  // texture.cpp:194-200
    int h = perm[perm[perm[x] + y] + z];
    h &= 15;
    const float u = h < 8 || h == 12 || h == 13 ? dx : dy;
    const float v = h < 4 || h == 12 || h == 13 ? dy : dz;
    return ((h & 1) ? -u : u) + ((h & 2) ? -v : v);
}
PG_DEV float noise_weight(float t) {  // texture.cpp:202-206
    const float t3 = t * t * t;
    const float t4 = t3 * t;
    return 6 * t4 * t - 15 * t4 + 10 * t3;
}
PG_DEV float noise3(const int *perm, float x, float y, float z) {  // texture.cpp:164-191
    int ix = (int)floorf(x), iy = (int)floorf(y), iz = (int)floorf(z);
    const float dx = x - ix, dy = y - iy, dz = z - iz;
    ix &= 255; iy &= 255; iz &= 255;
    const float w000 = noise_grad(perm, ix, iy, iz, dx, dy, dz);
    const float w100 = noise_grad(perm, ix + 1, iy, iz, dx - 1, dy, dz);
    const float w010 = noise_grad(perm, ix, iy + 1, iz, dx, dy - 1, dz);
    const float w110 = noise_grad(perm, ix + 1, iy + 1, iz, dx - 1, dy - 1, dz);
    const float w001 = noise_grad(perm, ix, iy, iz + 1, dx, dy, dz - 1);
    const float w101 = noise_grad(perm, ix + 1, iy, iz + 1, dx - 1, dy, dz - 1);
    const float w011 = noise_grad(perm, ix, iy + 1, iz + 1, dx, dy - 1, dz - 1);
    const float w111 = noise_grad(perm, ix + 1, iy + 1, iz + 1, dx - 1, dy - 1, dz - 1);
    const float wx = noise_weight(dx), wy = noise_weight(dy), wz = noise_weight(dz);
    const float x00 = plerp(wx, w000, w100), x10 = plerp(wx, w010, w110), x01 = plerp(wx, w001, w101), x11 = plerp(wx, w011, w111);
    const float y0 = plerp(wy, x00, x10), y1 = plerp(wy, x01, x11);
    return plerp(wz, y0, y1);
}
PG_DEV float smooth_step(float lo, float hi, float value) {  // texture.cpp:41-44
    const float v = clampf((value - lo) / (hi - lo), 0, 1);
    return v * v * (-2 * v + 3);
}
PG_DEV float noise_octaves(V3 dpdx, V3 dpdy, int maxOctaves) {  // texture.cpp:210-213
    const float len2 = pmax(lensq(dpdx), lensq(dpdy));
    const float n = -1 - .5f * log2_pbrt(len2);
    return n < 0 ? 0.f : (n > maxOctaves ? (float)maxOctaves : n);
}
// FBm (texture.cpp:208-225) with TURB = false, Turbulence (:227-252) with TURB = true
template <bool TURB>
PG_DEV float noise_sum(const int *perm, V3 p, V3 dpdx, V3 dpdy, float omega, int maxOctaves) {
    const float n = noise_octaves(dpdx, dpdy, maxOctaves);
    const int nInt = (int)floorf(n);
    float sum = 0, lambda = 1, o = 1;
    for (int i = 0; i < nInt; ++i) {
        const float v = noise3(perm, lambda * p.x, lambda * p.y, lambda * p.z);
        sum += o * (TURB ? fabsf(v) : v);
        lambda *= 1.99f;
        o *= omega;
    }
    const float nPartial = n - nInt;
    const float last = noise3(perm, lambda * p.x, lambda * p.y, lambda * p.z);
    if (!TURB) return sum + o * smooth_step(.3f, .7f, nPartial) * last;
    sum += o * plerp(smooth_step(.3f, .7f, nPartial), 0.2f, fabsf(last));
    for (int i = nInt; i < maxOctaves; ++i) {
        sum += o * 0.2f;
        o *= omega;
    }
    return sum;
}
// FBmTexture / WrinkledTexture / WindyTexture::Evaluate (fbm.h:56-60, wrinkled.h:56-60, windy.h:55-61) over IdentityMapping3D
PG_DEV float tex_noise(const DScene &sc, const PgTexture &t, const TexHit &h) {
    const V3 P = m4_point(t.w2t, h.p), dpdx = m4_vec(t.w2t, h.dpdx), dpdy = m4_vec(t.w2t, h.dpdy);
    if (t.type == PG_TEX_FBM) return noise_sum<false>(sc.noisePerm, P, dpdx, dpdy, t.omega, t.octaves);
    if (t.type == PG_TEX_WRINKLED) return noise_sum<true>(sc.noisePerm, P, dpdx, dpdy, t.omega, t.octaves);
    const float windStrength = noise_sum<false>(sc.noisePerm, P * .1f, dpdx * .1f, dpdy * .1f, .5f, 3);
    const float waveHeight = noise_sum<false>(sc.noisePerm, P, dpdx, dpdy, .5f, 6);
    return fabsf(windStrength) * waveHeight;
}
PG_DEV Spec tex_marble(const DScene &sc, const PgTexture &t, const TexHit &h) {  // marble.h:60-91
    V3 p = m4_point(t.w2t, h.p);
    const V3 dpdx = m4_vec(t.w2t, h.dpdx), dpdy = m4_vec(t.w2t, h.dpdy);
    p = p * t.noise_scale;
    const float marble = p.y + t.variation * noise_sum<false>(sc.noisePerm, p, dpdx * t.noise_scale, dpdy * t.noise_scale, t.omega, t.octaves);
    float tt = .5f + .5f * pg_sinf(marble);
    const float c[9][3] = {{.58f, .58f, .6f}, {.58f, .58f, .6f}, {.58f, .58f, .6f}, {.5f, .5f, .5f}, {.6f, .59f, .58f},
                           {.58f, .58f, .6f}, {.58f, .58f, .6f}, {.2f, .2f, .33f}, {.58f, .58f, .6f}};
    int first = (int)floorf(tt * 6);  // NSEG = 6
    if (first > 1) first = 1;
    tt = (tt * 6 - first);
    const Spec c0 = sp_of(c[first]), c1 = sp_of(c[first + 1]), c2 = sp_of(c[first + 2]), c3 = sp_of(c[first + 3]);
    Spec s0 = c0 * (1.f - tt) + c1 * tt;  // de Casteljau
    Spec s1 = c1 * (1.f - tt) + c2 * tt;
    const Spec s2 = c2 * (1.f - tt) + c3 * tt;
    s0 = s0 * (1.f - tt) + s1 * tt;
    s1 = s1 * (1.f - tt) + s2 * tt;
    return (s0 * (1.f - tt) + s1 * tt) * 1.5f;
}
// DotsTexture::Evaluate up to the choice of its operand (dots.h:58-79): true = insideDot
PG_DEV bool tex_dots_inside(const DScene &sc, const PgTexture &t, const TexHit &h) {
    float st[2], dstdx[2], dstdy[2];
    tex_map2d(t, h, st, dstdx, dstdy);
    const int sCell = (int)floorf(st[0] + .5f), tCell = (int)floorf(st[1] + .5f);
    if (noise3(sc.noisePerm, sCell + .5f, tCell + .5f, .5f) > 0) {
        const float radius = .35f;
        const float maxShift = 0.5f - radius;
        const float sCenter = sCell + maxShift * noise3(sc.noisePerm, sCell + 1.5f, tCell + 2.8f, .5f);
        const float tCenter = tCell + maxShift * noise3(sc.noisePerm, sCell + 4.5f, tCell + 9.8f, .5f);
        const float ds = st[0] - sCenter, dt = st[1] - tCenter;
        if (ds * ds + dt * dt < radius * radius) return true;
    }
    return false;
}
// W: a tag that gives a kernel its own copies of the evaluators -- the compiler allots a CALLED function the registers of the most
// permissive kernel that reaches it, so k_material (four waves per SIMD) must not share them with k_shade<2, .> (two)
template <int D, int W = 0> struct TexEval {
    static PG_DEV_CALL float f(const DScene &sc, const PgTexRef &r, const TexHit &h) {
        if (r.tex < 0) return r.v[0];
        const PgTexture &t = sc.textures[r.tex];
        switch (t.type) {
        case PG_TEX_SCALE: return TexEval<D - 1, W>::f(sc, t.tex1, h) * TexEval<D - 1, W>::f(sc, t.tex2, h);  // scale.h:58-60
        case PG_TEX_MIX: {  // mix.h:57-61
            const float t1 = TexEval<D - 1, W>::f(sc, t.tex1, h), t2 = TexEval<D - 1, W>::f(sc, t.tex2, h);
            const float amt = TexEval<D - 1, W>::f(sc, t.amount, h);
            return (1 - amt) * t1 + amt * t2;
        }
        case PG_TEX_IMAGEMAP: {  // ImageTexture::Evaluate, imagemap.h:86-93
            float st[2], dx[2], dy[2];
            tex_map2d(t, h, st, dx, dy);
            return mip_lookup(sc, sc.images[t.image], st, dx, dy).r;
        }
        case PG_TEX_BILERP: {  // bilerp.h:56-61
            float st[2], dx[2], dy[2];
            tex_map2d(t, h, st, dx, dy);
            return (1 - st[0]) * (1 - st[1]) * t.v00[0] + (1 - st[0]) * (st[1]) * t.v01[0] + (st[0]) * (1 - st[1]) * t.v10[0] + (st[0]) * (st[1]) * t.v11[0];
        }
        case PG_TEX_FBM: case PG_TEX_WRINKLED: case PG_TEX_WINDY: return tex_noise(sc, t, h);
        case PG_TEX_DOTS: return tex_dots_inside(sc, t, h) ? TexEval<D - 1, W>::f(sc, t.tex2, h) : TexEval<D - 1, W>::f(sc, t.tex1, h);
        case PG_TEX_CHECKERBOARD_3D: return tex_checker3d(t, h) == 0 ? TexEval<D - 1, W>::f(sc, t.tex1, h) : TexEval<D - 1, W>::f(sc, t.tex2, h);
        case PG_TEX_CHECKERBOARD_2D: {  // checkerboard.h:63-103
            float area2 = 0;
            const int which = tex_checker2d(t, h, area2);
            if (which == 0) return TexEval<D - 1, W>::f(sc, t.tex1, h);
            if (which == 1) return TexEval<D - 1, W>::f(sc, t.tex2, h);
            return (1 - area2) * TexEval<D - 1, W>::f(sc, t.tex1, h) + area2 * TexEval<D - 1, W>::f(sc, t.tex2, h);
        }
        }
        return 0;
    }
    static PG_DEV_CALL Spec s(const DScene &sc, const PgTexRef &r, const TexHit &h) {
        if (r.tex < 0) return sp3(r.v[0], r.v[1], r.v[2]);
        const PgTexture &t = sc.textures[r.tex];
        switch (t.type) {
        case PG_TEX_SCALE: return TexEval<D - 1, W>::s(sc, t.tex1, h) * TexEval<D - 1, W>::s(sc, t.tex2, h);
        case PG_TEX_MIX: {
            const Spec t1 = TexEval<D - 1, W>::s(sc, t.tex1, h), t2 = TexEval<D - 1, W>::s(sc, t.tex2, h);
            const float amt = TexEval<D - 1, W>::f(sc, t.amount, h);
            return t1 * (1 - amt) + t2 * amt;
        }
        case PG_TEX_IMAGEMAP: {
            float st[2], dx[2], dy[2];
            tex_map2d(t, h, st, dx, dy);
            return mip_lookup(sc, sc.images[t.image], st, dx, dy);
        }
        case PG_TEX_UV: {  // uv.h:54-60
            float st[2], dx[2], dy[2];
            tex_map2d(t, h, st, dx, dy);
            return sp3(st[0] - floorf(st[0]), st[1] - floorf(st[1]), 0);
        }
        case PG_TEX_BILERP: {
            float st[2], dx[2], dy[2];
            tex_map2d(t, h, st, dx, dy);
            return sp_of(t.v00) * ((1 - st[0]) * (1 - st[1])) + sp_of(t.v01) * ((1 - st[0]) * (st[1])) + sp_of(t.v10) * ((st[0]) * (1 - st[1])) +
                   sp_of(t.v11) * ((st[0]) * (st[1]));
        }
        case PG_TEX_FBM: case PG_TEX_WRINKLED: case PG_TEX_WINDY: return sp(tex_noise(sc, t, h));
        case PG_TEX_MARBLE: return tex_marble(sc, t, h);
        case PG_TEX_DOTS: return tex_dots_inside(sc, t, h) ? TexEval<D - 1, W>::s(sc, t.tex2, h) : TexEval<D - 1, W>::s(sc, t.tex1, h);
        case PG_TEX_CHECKERBOARD_3D: return tex_checker3d(t, h) == 0 ? TexEval<D - 1, W>::s(sc, t.tex1, h) : TexEval<D - 1, W>::s(sc, t.tex2, h);
        case PG_TEX_CHECKERBOARD_2D: {
            float area2 = 0;
            const int which = tex_checker2d(t, h, area2);
            if (which == 0) return TexEval<D - 1, W>::s(sc, t.tex1, h);
            if (which == 1) return TexEval<D - 1, W>::s(sc, t.tex2, h);
            return TexEval<D - 1, W>::s(sc, t.tex1, h) * (1 - area2) + TexEval<D - 1, W>::s(sc, t.tex2, h) * area2;
        }
        }
        return sp(0);
    }
};
template <int W> struct TexEval<0, W> {  // the innermost level: operands must be constants (the host front end enforces the depth)
    static PG_DEV float f(const DScene &, const PgTexRef &r, const TexHit &) { return r.v[0]; }
    static PG_DEV Spec s(const DScene &, const PgTexRef &r, const TexHit &) { return sp3(r.v[0], r.v[1], r.v[2]); }
};
#endif
