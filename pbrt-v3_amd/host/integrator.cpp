// Camera / sampler / integrator factories and the host half of
// Integrator::Render for the MI355X path.
//   CreatePerspectiveCamera  cameras/perspective.cpp:215-273, camera.h:87-108
//   CreateHaltonSampler      samplers/halton.cpp:65-92,133-139
//   Halton permutations      lowdiscrepancy.cpp:2490-2504, rng.h:61-144, sampling.h:151-157
//   CreatePathIntegrator     integrators/path.cpp:190-213
//   Render                   core/integrator.cpp:228-339 (tile loop -> device)
#include <dlfcn.h>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include "api.h"
#include "error.h"
#include "scene.h"

namespace pbrt {

PerspectiveCamera *CreatePerspectiveCamera(const ParamSet &params, const Transform &cam2world, Film *film, bool orthographic) {
    Float shutteropen = params.FindOneFloat("shutteropen", 0.f);
    Float shutterclose = params.FindOneFloat("shutterclose", 1.f);
    if (shutterclose < shutteropen) {
        Warning("Shutter close time [%f] < shutter open [%f].  Swapping them.", shutterclose, shutteropen);
        std::swap(shutterclose, shutteropen);
    }
    Float lensradius = params.FindOneFloat("lensradius", 0.f);
    Float focaldistance = params.FindOneFloat("focaldistance", 1e6);
    Float frame = params.FindOneFloat("frameaspectratio", Float(film->fullResolution[0]) / Float(film->fullResolution[1]));
    Float sxmin, sxmax, symin, symax;
    if (frame > 1.f) { sxmin = -frame; sxmax = frame; symin = -1.f; symax = 1.f; }
    else { sxmin = -1.f; sxmax = 1.f; symin = -1.f / frame; symax = 1.f / frame; }
    const std::vector<Float> *sw = params.FindFloat("screenwindow");
    if (sw) {
        if (sw->size() == 4) { sxmin = (*sw)[0]; sxmax = (*sw)[1]; symin = (*sw)[2]; symax = (*sw)[3]; }
        else Error("\"screenwindow\" should have four values");
    }
    Float fov = 90.f;
    if (!orthographic) {  // perspective.cpp:253-257
        fov = params.FindOneFloat("fov", 90.);
        Float halffov = params.FindOneFloat("halffov", -1.f);
        if (halffov > 0.f) fov = 2.f * halffov;
    }
    PerspectiveCamera *cam = new PerspectiveCamera;
    cam->orthographic = orthographic;
    cam->film.reset(film);
    cam->CameraToWorld = cam2world;
    cam->lensRadius = lensradius; cam->focalDistance = focaldistance;
    cam->shutterOpen = shutteropen; cam->shutterClose = shutterclose;
    // ProjectiveCamera ctor, camera.h:98-107
    // perspective.cpp:52 Perspective(fov, 1e-2f, 1000.f); orthographic.cpp:53 Orthographic(0, 1) = Scale(1, 1, 1/(1-0)) * Translate(0, 0, -0)
    Transform CameraToScreen = orthographic ? Scale(1, 1, 1 / (1.f - 0.f)) * Translate(Vector3f(0, 0, -0.f)) : Perspective(fov, 1e-2f, 1000.f);
    Transform ScreenToRaster = Scale(film->fullResolution[0], film->fullResolution[1], 1) *
                               Scale(1 / (sxmax - sxmin), 1 / (symin - symax), 1) *
                               Translate(Vector3f(-sxmin, -symax, 0));
    Transform RasterToScreen = Inverse(ScreenToRaster);
    cam->RasterToCamera = Inverse(CameraToScreen) * RasterToScreen;
    return cam;
}

// ---- Halton ---------------------------------------------------------------
static void extendedGCD(uint64_t a, uint64_t b, int64_t *x, int64_t *y) {  // halton.cpp:52-62
    if (b == 0) { *x = 1; *y = 0; return; }
    int64_t d = a / b, xp, yp;
    extendedGCD(b, a % b, &xp, &yp);
    *x = yp;
    *y = xp - (d * yp);
}
static uint64_t multiplicativeInverse(int64_t a, int64_t n) {  // halton.cpp:46-50; Mod() from pbrt.h
    int64_t x, y;
    extendedGCD(a, n, &x, &y);
    int64_t r = x - (x / n) * n;
    return (uint64_t)((r < 0) ? r + n : r);
}
HaltonSampler *CreateHaltonSampler(const ParamSet &params, const int sb[4]) {
    int nsamp = params.FindOneInt("pixelsamples", 16);
    if (PbrtOptions.quickRender) nsamp = 1;
    bool sampleAtCenter = params.FindOneBool("samplepixelcenter", false);
    HaltonSampler *s = new HaltonSampler;
    s->samplesPerPixel = nsamp;
    s->sampleAtPixelCenter = sampleAtCenter;
    const int kMaxResolution = 128;
    int res[2] = {sb[2] - sb[0], sb[3] - sb[1]};
    for (int i = 0; i < 2; ++i) {
        int base = (i == 0) ? 2 : 3;
        int scale = 1, exp = 0;
        while (scale < std::min(res[i], kMaxResolution)) { scale *= base; ++exp; }
        s->baseScales[i] = scale;
        s->baseExponents[i] = exp;
    }
    s->sampleStride = s->baseScales[0] * s->baseScales[1];
    s->multInverse[0] = (int)multiplicativeInverse(s->baseScales[1], s->baseScales[0]);
    s->multInverse[1] = (int)multiplicativeInverse(s->baseScales[0], s->baseScales[1]);
    return s;
}

HaltonSampler *CreateSobolSampler(const ParamSet &params, const int sb[4]) {  // sobol.cpp:64-69, SobolSampler ctor sobol.h:52-63
    int nsamp = params.FindOneInt("pixelsamples", 16);
    if (PbrtOptions.quickRender) nsamp = 1;
    auto roundUpPow2 = [](int64_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v |= v >> 32; return v + 1; };  // pbrt.h:369-388
    HaltonSampler *s = new HaltonSampler;
    memset(s->baseScales, 0, sizeof(s->baseScales)); memset(s->baseExponents, 0, sizeof(s->baseExponents)); memset(s->multInverse, 0, sizeof(s->multInverse));
    s->sampleStride = 0; s->sampleAtPixelCenter = false;
    s->sobol = true;
    s->samplesPerPixel = (int)roundUpPow2(nsamp);
    if (s->samplesPerPixel != nsamp) Warning("Non power-of-two sample count rounded up to %d for SobolSampler.", s->samplesPerPixel);
    s->resolution = (int)roundUpPow2((int32_t)std::max(sb[2] - sb[0], sb[3] - sb[1]));  // RoundUpPow2(int32_t)
    s->log2Resolution = 0;
    while ((1 << (s->log2Resolution + 1)) <= s->resolution) ++s->log2Resolution;  // Log2Int, pbrt.h:333-343
    return s;
}

// The samplers of the reference that consume one RNG stream per tile.  Only their parameters are read here; the streams are
// generated where the paths are traced (oracle / device), tile by tile.
HaltonSampler *CreateTileSerialSampler(const std::string &name, const ParamSet &params) {
    auto roundUpPow2 = [](int64_t v) { v--; v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16; v |= v >> 32; return v + 1; };  // pbrt.h:369-388
    auto isPow2 = [](int64_t v) { return v && !(v & (v - 1)); };
    HaltonSampler *s = new HaltonSampler;
    memset(s->baseScales, 0, sizeof(s->baseScales)); memset(s->baseExponents, 0, sizeof(s->baseExponents)); memset(s->multInverse, 0, sizeof(s->multInverse));
    s->sampleStride = 0; s->sampleAtPixelCenter = false;
    if (name == "random") {  // random.cpp:72-75: no "dimensions", no quick-render clamp
        s->kind = PG_SAMPLER_RANDOM;
        s->samplesPerPixel = params.FindOneInt("pixelsamples", 4);
    } else if (name == "stratified") {  // stratified.cpp:79-87
        s->kind = PG_SAMPLER_STRATIFIED;
        s->jitterSamples = params.FindOneBool("jitter", true);
        s->xPixelSamples = params.FindOneInt("xsamples", 4);
        s->yPixelSamples = params.FindOneInt("ysamples", 4);
        s->nSampledDimensions = params.FindOneInt("dimensions", 4);
        if (PbrtOptions.quickRender) s->xPixelSamples = s->yPixelSamples = 1;
        s->samplesPerPixel = s->xPixelSamples * s->yPixelSamples;
    } else {  // zerotwosequence.cpp:78-83, maxmin.cpp:82-87
        int nsamp = params.FindOneInt("pixelsamples", 16);
        s->nSampledDimensions = params.FindOneInt("dimensions", 4);
        if (PbrtOptions.quickRender) nsamp = 1;
        int64_t spp = nsamp;
        if (name == "maxmindist") {  // the constructor's lambda, maxmin.h:54-71: at most 2^17 - 1, then a power of two
            s->kind = PG_SAMPLER_MAXMINDIST;
            int cIndex = 0;
            while (((int64_t)1 << (cIndex + 1)) <= spp) ++cIndex;  // Log2Int
            if (cIndex >= 17) { Warning("No more than %d samples per pixel are supported with MaxMinDistSampler. Rounding down.", (1 << 17) - 1); spp = (1 << 17) - 1; }
            if (!isPow2(spp)) { spp = roundUpPow2(spp); Warning("Non power-of-two sample count rounded up to %lld for MaxMinDistSampler.", (long long)spp); }
        } else {
            s->kind = PG_SAMPLER_ZEROTWO;
            if (!isPow2(spp)) Warning("Pixel samples being rounded up to power of 2 (from %lld to %lld).", (long long)spp, (long long)roundUpPow2(spp));
            spp = roundUpPow2(spp);
        }
        s->samplesPerPixel = (int)spp;
    }
    if (s->samplesPerPixel < 1 || s->nSampledDimensions < 0) { Error("Sampler \"%s\": needs at least one sample per pixel and a non-negative \"dimensions\".", name.c_str()); Fatal(); }
    return s;
}

namespace {
struct RNG {  // PCG32, rng.h:61-144
    uint64_t state = 0x853c49e6748fea9bULL, inc = 0xda3e39cb94b95bdbULL;
    uint32_t UniformUInt32() {
        uint64_t oldstate = state;
        state = oldstate * 0x5851f42d4c957f2dULL + inc;
        uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
        uint32_t rot = (uint32_t)(oldstate >> 59u);
        return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
    }
    uint32_t UniformUInt32(uint32_t b) {
        uint32_t threshold = (~b + 1u) % b;
        while (true) { uint32_t r = UniformUInt32(); if (r >= threshold) return r % b; }
    }
};
}  // namespace
void ComputeRadicalInversePermutations(int nDims, std::vector<uint16_t> *perms, std::vector<int32_t> *sums) {
    const int PrimeTableSize = 1000;  // lowdiscrepancy.h:52
    if (nDims > PrimeTableSize) nDims = PrimeTableSize;
    // first 1000 primes (lowdiscrepancy.cpp:40-122 tabulates them)
    std::vector<int> primes;
    for (int c = 2; (int)primes.size() < PrimeTableSize; ++c) {
        bool isPrime = true;
        for (int p : primes) { if (p * p > c) break; if (c % p == 0) { isPrime = false; break; } }
        if (isPrime) primes.push_back(c);
    }
    // The RNG stream runs through all bases in order, so bases < nDims get the
    // same permutations as in the full 1000-base table.
    RNG rng;
    perms->clear(); sums->clear();
    int sum = 0;
    for (int i = 0; i < nDims; ++i) {
        sums->push_back(sum);
        size_t base = perms->size();
        for (int j = 0; j < primes[i]; ++j) perms->push_back((uint16_t)j);
        uint16_t *p = &(*perms)[base];
        int count = primes[i];
        for (int k = 0; k < count; ++k) {  // Shuffle(p, count, 1, rng), sampling.h:151-157
            int other = k + rng.UniformUInt32(count - k);
            std::swap(p[k], p[other]);
        }
        sum += primes[i];
    }
    sums->push_back(sum);
}

// ---- integrator -------------------------------------------------------------
GpuPathIntegrator::GpuPathIntegrator(int maxDepth, std::shared_ptr<PerspectiveCamera> camera,
                                     std::shared_ptr<HaltonSampler> sampler, const int pb[4], Float rrThreshold,
                                     const std::string &lightSampleStrategy)
    : camera(camera), sampler(sampler), maxDepth(maxDepth), rrThreshold(rrThreshold), lightSampleStrategy(lightSampleStrategy) {
    for (int i = 0; i < 4; ++i) pixelBounds[i] = pb[i];
}
GpuPathIntegrator *CreatePathIntegrator(const ParamSet &params, std::shared_ptr<HaltonSampler> sampler,
                                        std::shared_ptr<PerspectiveCamera> camera) {
    int maxDepth = params.FindOneInt("maxdepth", 5);
    int pixelBounds[4];
    camera->film->GetSampleBounds(pixelBounds);
    const std::vector<int> *pb = params.FindInt("pixelbounds");
    if (pb) {
        if (pb->size() != 4) Error("Expected four values for \"pixelbounds\" parameter. Got %d.", (int)pb->size());
        else {
            int b[4] = {(*pb)[0], (*pb)[2], (*pb)[1], (*pb)[3]};
            pixelBounds[0] = std::max(pixelBounds[0], std::min(b[0], b[2]));
            pixelBounds[1] = std::max(pixelBounds[1], std::min(b[1], b[3]));
            pixelBounds[2] = std::min(pixelBounds[2], std::max(b[0], b[2]));
            pixelBounds[3] = std::min(pixelBounds[3], std::max(b[1], b[3]));
            if ((pixelBounds[2] - pixelBounds[0]) * (pixelBounds[3] - pixelBounds[1]) == 0) Error("Degenerate \"pixelbounds\" specified.");
        }
    }
    Float rrThreshold = params.FindOneFloat("rrthreshold", 1.);
    std::string lightStrategy = params.FindOneString("lightsamplestrategy", "spatial");
    return new GpuPathIntegrator(maxDepth, camera, sampler, pixelBounds, rrThreshold, lightStrategy);
}

void GpuPathIntegrator::Flatten(const Scene &scene, FlatScene *flat) const {
    const BVHAccel &bvh = *scene.aggregate;
    // primitives: the top-level BVH's (in its order), then every object definition's in first-use order
    std::vector<const GeometricPrimitive *> all;
    for (const auto &prim : bvh.primitives) all.push_back(&prim);
    flat->nodes = bvh.nodes;
    std::map<const ObjectDefinition *, int> objectIndex;
    for (size_t pi = 0; pi < all.size(); ++pi) {  // (`all` grows: a moving shape inside an object definition is an object of its own, found here)
        const GeometricPrimitive &prim = *all[pi];
        if (!prim.object) continue;
        const ObjectDefinition *od = prim.object.get();
        if (!objectIndex.count(od)) {
            objectIndex[od] = (int)flat->objects.size();
            PgObject o;
            o.first_prim = (int)all.size();
            o.first_node = (int)flat->nodes.size();
            if (od->accel) {
                o.n_nodes = (int)od->accel->nodes.size();
                o.n_prims = (int)od->accel->primitives.size();
                flat->nodes.insert(flat->nodes.end(), od->accel->nodes.begin(), od->accel->nodes.end());
                for (const auto &p2 : od->accel->primitives) all.push_back(&p2);
            } else {
                o.n_nodes = 0; o.n_prims = (int)od->prims.size();
                for (const auto &p2 : od->prims) all.push_back(&p2);
            }
            flat->objects.push_back(o);
        }
    }
    const size_t nTop = bvh.primitives.size(), nTris = all.size();
    flat->indices.resize(3 * nTris);
    flat->triFlags.resize(nTris);
    flat->triMaterial.resize(nTris);
    flat->triLight.resize(nTris);
    // concatenate the meshes' vertex arrays in first-use order
    std::map<const TriangleMesh *, int> meshBase;
    bool anyN = false, anyUV = false, anyS = false;
    std::map<const Sphere *, int> sphereIndex;
    for (const GeometricPrimitive *pp : all) {
        const GeometricPrimitive &prim = *pp;
        if (prim.object) continue;
        if (prim.sphere) {
            const Sphere *sp = prim.sphere.get();
            if (sphereIndex.count(sp)) continue;
            sphereIndex[sp] = (int)flat->spheres.size();
            PgSphere g;
            memset(&g, 0, sizeof(g));
            const Matrix4x4 &m = sp->ObjectToWorld.GetMatrix(), &mi = sp->WorldToObject.GetMatrix();
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { g.o2w[4 * r + c] = m.m[r][c]; g.w2o[4 * r + c] = mi.m[r][c]; }
            g.radius = sp->radius; g.z_min = sp->zMin; g.z_max = sp->zMax;
            g.theta_min = sp->thetaMin; g.theta_max = sp->thetaMax; g.phi_max = sp->phiMax;
            g.reverse_orientation = sp->reverseOrientation; g.swaps_handedness = sp->transformSwapsHandedness;
            g.shape = sp->shape; g.height = sp->height; g.inner_radius = sp->innerRadius; g.area = sp->Area();
            for (int k = 0; k < 3; ++k) { g.p1[k] = sp->p1[k]; g.p2[k] = sp->p2[k]; }
            g.ah = sp->ah; g.ch = sp->ch;
            flat->spheres.push_back(g);
            continue;
        }
        const TriangleMesh *m = prim.shape.mesh.get();
        if (meshBase.count(m)) continue;
        meshBase[m] = (int)(flat->P.size() / 3);
        for (const Point3f &p : m->p) { flat->P.push_back(p.x); flat->P.push_back(p.y); flat->P.push_back(p.z); }
        anyN |= !m->n.empty(); anyUV |= !m->uv.empty(); anyS |= !m->s.empty();
    }
    size_t nVerts = flat->P.size() / 3;
    if (anyN) flat->N.assign(3 * nVerts, 0.f);
    if (anyUV) flat->UV.assign(2 * nVerts, 0.f);
    if (anyS) flat->S.assign(3 * nVerts, 0.f);
    for (auto &kv : meshBase) {
        const TriangleMesh *m = kv.first;
        size_t b = kv.second;
        for (size_t i = 0; i < m->n.size(); ++i) { flat->N[3 * (b + i)] = m->n[i].x; flat->N[3 * (b + i) + 1] = m->n[i].y; flat->N[3 * (b + i) + 2] = m->n[i].z; }
        for (size_t i = 0; i < m->s.size(); ++i) { flat->S[3 * (b + i)] = m->s[i].x; flat->S[3 * (b + i) + 1] = m->s[i].y; flat->S[3 * (b + i) + 2] = m->s[i].z; }
        for (size_t i = 0; i < m->uv.size(); ++i) flat->UV[2 * b + i] = m->uv[i];
    }
    for (size_t k = 0; k < nTris; ++k) {
        const GeometricPrimitive &prim = *all[k];
        flat->triMaterial[k] = prim.material < 0 ? 0 : prim.material;
        if (prim.mediumInside >= 0 || prim.mediumOutside >= 0) {
            if (flat->triMediumInside.empty()) { flat->triMediumInside.assign(nTris, -1); flat->triMediumOutside.assign(nTris, -1); }
            flat->triMediumInside[k] = prim.mediumInside; flat->triMediumOutside[k] = prim.mediumOutside;
        }
        flat->triLight[k] = prim.areaLight;
        if (prim.object) {  // TransformedPrimitive: one PgInstance per use
            PgInstance inst;
            memset(&inst, 0, sizeof(inst));
            const Matrix4x4 &m = prim.xf->InstanceToWorld.GetMatrix(), &mi = prim.xf->WorldToInstance.GetMatrix();
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { inst.i2w[4 * r + c] = m.m[r][c]; inst.w2i[4 * r + c] = mi.m[r][c]; }
            inst.object = objectIndex[prim.object.get()];
            inst.identity = prim.xf->InstanceToWorld.IsIdentity() ? 1 : 0;
            if (prim.xf->animated) {  // AnimatedTransform's constructor, transform.cpp:396-411
                inst.animated = 1;
                inst.time[0] = prim.xf->time[0]; inst.time[1] = prim.xf->time[1];
                const Matrix4x4 &me = prim.xf->InstanceToWorldEnd.GetMatrix(), &mie = prim.xf->WorldToInstanceEnd.GetMatrix();
                for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { inst.i2w_end[4 * r + c] = me.m[r][c]; inst.w2i_end[4 * r + c] = mie.m[r][c]; }
                DecomposeTransform(m, inst.T[0], inst.R[0], inst.S[0]);
                DecomposeTransform(me, inst.T[1], inst.R[1], inst.S[1]);
                const float *q0 = inst.R[0];
                float *q1 = inst.R[1];
                if ((q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2]) + q0[3] * q1[3] < 0) for (int i = 0; i < 4; ++i) q1[i] = -q1[i];  // the shortest path
            }
            flat->indices[3 * k] = (int)flat->instances.size();
            flat->indices[3 * k + 1] = flat->indices[3 * k + 2] = 0;
            flat->triFlags[k] = PG_PRIM_INSTANCE;
            flat->instances.push_back(inst);
            continue;
        }
        if (prim.sphere) {
            flat->indices[3 * k] = sphereIndex[prim.sphere.get()];
            flat->indices[3 * k + 1] = flat->indices[3 * k + 2] = 0;
            flat->triFlags[k] = PG_PRIM_SPHERE;
            continue;
        }
        const TriangleMesh *m = prim.shape.mesh.get();
        int base = meshBase[m];
        const int *v = prim.shape.v();
        for (int j = 0; j < 3; ++j) flat->indices[3 * k + j] = base + v[j];
        uint32_t f = 0;
        if (m->reverseOrientation ^ m->transformSwapsHandedness) f |= PG_TRI_FLIP_NORMAL;
        if (m->reverseOrientation) f |= PG_TRI_REVERSE_ORIENTATION;
        if (!m->n.empty()) f |= PG_TRI_HAS_N;
        if (!m->uv.empty()) f |= PG_TRI_HAS_UV;
        if (!m->s.empty()) f |= PG_TRI_HAS_S;
        if (m->alphaMask >= 0) { f |= PG_TRI_ALPHA; if (flat->triAlpha.empty()) flat->triAlpha.assign(nTris, -1); flat->triAlpha[k] = m->alphaMask; }
        flat->triFlags[k] = f;
    }
    flat->materials = scene.materials;
    flat->bxdfs = scene.bxdfs;
    flat->textures = scene.textures;
    flat->textured = scene.textured;
    flat->images = scene.images;
    flat->texels = scene.texels;
    flat->envTables = scene.envTables;
    flat->alphas = scene.alphas;
    flat->media = scene.media;
    flat->mediaGrid = scene.mediaGrid; flat->grids = scene.grids; flat->gridDensity = scene.gridDensity;
    flat->bssrdfs = scene.bssrdfs; flat->materialBssrdf = scene.materialBssrdf; flat->bssrdfTables = scene.bssrdfTables;
    EWAWeightLut(flat->ewaLut);
    flat->lights = scene.lights;
    // Light::Preprocess (scene.h:57-60): DistantLight keeps the world's bounding sphere (distant.h:55-57, geometry.h:803-806)
    if (!flat->nodes.empty()) {
        const PgBVHNode &root = flat->nodes[0];
        Point3f pMin(root.bmin[0], root.bmin[1], root.bmin[2]), pMax(root.bmax[0], root.bmax[1], root.bmax[2]);
        Point3f center = (pMin + pMax) / 2;
        bool inside = center.x >= pMin.x && center.x <= pMax.x && center.y >= pMin.y && center.y <= pMax.y && center.z >= pMin.z &&
                      center.z <= pMax.z;
        Float radius = inside ? (center - pMax).Length() : 0;
        for (PgLight &l : flat->lights) if (l.type == PG_LIGHT_DISTANT || l.type == PG_LIGHT_INFINITE) l.world_radius = radius;
    }
    // dimensions a path can consume: 5 camera + per bounce (1+2+2 direct, 2 bsdf, 1 rr); 1000 max (halton.h:71-76)
    int nDims = (int)std::min(1000LL, std::max(5LL, 5 + 8 * ((long long)maxDepth + 2)));  // (64-bit: "maxdepth" is the file's to choose)
    if (volumetric) nDims = 1000;  // volpath.cpp:77-78,119-123: medium sampling consumes dimensions on uncounted bounces too
    if (!scene.bssrdfs.empty()) nDims = 1000;  // path.cpp:152-174: a subsurface vertex draws 10 more values (Sample_S, lights at pi, the exit direction)
    ComputeRadicalInversePermutations(nDims, &flat->perms, &flat->permSums);
    PgSceneDesc &d = flat->desc;
    memset(&d, 0, sizeof(d));
    d.abi_version = PG_ABI_VERSION;
    d.n_nodes = (int)bvh.nodes.size(); d.nodes = flat->nodes.data();
    d.n_nodes_all = (int)flat->nodes.size(); d.n_prims_all = (int)nTris;
    d.n_objects = (int)flat->objects.size(); d.objects = flat->objects.data();
    d.n_instances = (int)flat->instances.size(); d.instances = flat->instances.data();
    d.n_tris = (int)nTop; d.indices = flat->indices.data(); d.tri_flags = flat->triFlags.data();
    d.tri_material = flat->triMaterial.data(); d.tri_light = flat->triLight.data();
    d.n_verts = (int)nVerts; d.P = flat->P.data();
    d.N = anyN ? flat->N.data() : nullptr; d.UV = anyUV ? flat->UV.data() : nullptr; d.S = anyS ? flat->S.data() : nullptr;
    d.n_materials = (int)flat->materials.size(); d.materials = flat->materials.data();
    d.n_lights = (int)flat->lights.size(); d.lights = flat->lights.data();
    // CreateLightSampleDistribution, lightdistrib.cpp:48-66
    if (lightSampleStrategy == "uniform" || flat->lights.size() == 1) d.light_strategy = PG_LIGHTS_UNIFORM;
    else if (lightSampleStrategy == "power") d.light_strategy = PG_LIGHTS_POWER;
    else if (lightSampleStrategy == "spatial") d.light_strategy = PG_LIGHTS_SPATIAL;
    else {
        Error("Light sample distribution type \"%s\" unknown. Using \"spatial\".", lightSampleStrategy.c_str());
        d.light_strategy = PG_LIGHTS_SPATIAL;
    }
    d.n_perm_dims = (int)flat->permSums.size() - 1; d.perms = flat->perms.data(); d.perm_sums = flat->permSums.data();
    d.n_spheres = (int)flat->spheres.size(); d.spheres = flat->spheres.data();
    d.n_bxdfs = (int)flat->bxdfs.size(); d.bxdfs = flat->bxdfs.data();
    d.n_textures = (int)flat->textures.size(); d.textures = flat->textures.data();
    d.n_textured = (int)flat->textured.size(); d.textured = flat->textured.data();
    d.n_images = (int)flat->images.size(); d.images = flat->images.data();
    d.n_texel_floats = (int64_t)flat->texels.size(); d.texels = flat->texels.data();
    d.ewa_lut = flat->ewaLut;
    d.n_env_floats = (int64_t)flat->envTables.size(); d.env_tables = flat->envTables.data();
    d.n_media = (int)flat->media.size(); d.media = flat->media.data();
    if (!flat->bssrdfs.empty()) {  // subsurface scattering tables (ABI 24)
        d.n_bssrdfs = (int)flat->bssrdfs.size(); d.bssrdfs = flat->bssrdfs.data(); d.material_bssrdf = flat->materialBssrdf.data();
        d.n_bssrdf_floats = (int64_t)flat->bssrdfTables.size(); d.bssrdf_tables = flat->bssrdfTables.data();
    }
    if (!flat->grids.empty()) {  // GridDensityMedium tables (ABI 23)
        d.n_grids = (int)flat->grids.size(); d.grids = flat->grids.data(); d.media_grid = flat->mediaGrid.data();
        d.n_density_floats = (int64_t)flat->gridDensity.size(); d.grid_density = flat->gridDensity.data();
    }
    if (scene.usesNoise) d.noise_perm = GetNoisePermutation();
    if (sampler->kind == PG_SAMPLER_MAXMINDIST) d.cmaxmin = GetMaxMinDistTable();
    if (sampler->sobol) { const SobolTables &t = GetSobolTables(); d.sobol_matrices = t.matrices32; d.vdc_sobol = t.vdc; d.vdc_sobol_inv = t.vdcInv; }
    d.tri_medium_inside = flat->triMediumInside.empty() ? nullptr : flat->triMediumInside.data();
    d.tri_medium_outside = flat->triMediumOutside.empty() ? nullptr : flat->triMediumOutside.data();
    d.n_alphas = (int)flat->alphas.size(); d.alphas = flat->alphas.data();
    d.tri_alpha = flat->triAlpha.empty() ? nullptr : flat->triAlpha.data();
}

void GpuPathIntegrator::FillRenderDesc(PgRenderDesc *rd) const {
    memset(rd, 0, sizeof(*rd));
    rd->abi_version = PG_ABI_VERSION;
    const Film &film = *camera->film;
    rd->camera_type = camera->environment ? 2 : (camera->orthographic ? 1 : 0);
    rd->integrator = volumetric ? 1 : 0;
    rd->camera_medium = cameraMedium;
    memcpy(rd->raster_to_camera, camera->RasterToCamera.GetMatrix().m, 16 * sizeof(float));
    memcpy(rd->camera_to_world, camera->CameraToWorld.GetMatrix().m, 16 * sizeof(float));
    rd->camera_animated = camera->animated ? 1 : 0;
    if (camera->animated) {  // AnimatedTransform's constructor, transform.cpp:396-411
        rd->camera_time[0] = camera->transformStartTime; rd->camera_time[1] = camera->transformEndTime;
        memcpy(rd->camera_to_world_end, camera->CameraToWorldEnd.GetMatrix().m, 16 * sizeof(float));
        DecomposeTransform(camera->CameraToWorld.GetMatrix(), rd->camera_T[0], rd->camera_R[0], rd->camera_S[0]);
        DecomposeTransform(camera->CameraToWorldEnd.GetMatrix(), rd->camera_T[1], rd->camera_R[1], rd->camera_S[1]);
        const float *q0 = rd->camera_R[0];
        float *q1 = rd->camera_R[1];
        if ((q0[0] * q1[0] + q0[1] * q1[1] + q0[2] * q1[2]) + q0[3] * q1[3] < 0) for (int i = 0; i < 4; ++i) q1[i] = -q1[i];  // the shortest path
    }
    {  // dxCamera / dyCamera: perspective.cpp:60-63 (difference of two points), orthographic.cpp:57-58 (a transformed vector)
        Vector3f dx, dy;
        if (camera->orthographic) { dx = camera->RasterToCamera.Vec(Vector3f(1, 0, 0)); dy = camera->RasterToCamera.Vec(Vector3f(0, 1, 0)); }
        else {
            dx = camera->RasterToCamera.Pt(Point3f(1, 0, 0)) - camera->RasterToCamera.Pt(Point3f(0, 0, 0));
            dy = camera->RasterToCamera.Pt(Point3f(0, 1, 0)) - camera->RasterToCamera.Pt(Point3f(0, 0, 0));
        }
        rd->dx_camera[0] = dx.x; rd->dx_camera[1] = dx.y; rd->dx_camera[2] = dx.z;
        rd->dy_camera[0] = dy.x; rd->dy_camera[1] = dy.y; rd->dy_camera[2] = dy.z;
    }
    rd->lens_radius = camera->lensRadius; rd->focal_distance = camera->focalDistance;
    rd->shutter_open = camera->shutterOpen; rd->shutter_close = camera->shutterClose;
    rd->full_res[0] = film.fullResolution[0]; rd->full_res[1] = film.fullResolution[1];
    for (int i = 0; i < 4; ++i) rd->cropped_pixel_bounds[i] = film.croppedPixelBounds[i];
    film.GetSampleBounds(rd->sample_bounds);
    rd->filter_radius[0] = film.filterRadius[0]; rd->filter_radius[1] = film.filterRadius[1];
    rd->filter_general = film.filterGeneral ? 1 : 0;
    memcpy(rd->filter_table, film.filterTable, sizeof(rd->filter_table));
    rd->film_scale = film.scale; rd->max_sample_luminance = film.maxSampleLuminance;
    rd->spp = sampler->samplesPerPixel;
    for (int i = 0; i < 2; ++i) {
        rd->base_scales[i] = sampler->baseScales[i]; rd->base_exponents[i] = sampler->baseExponents[i];
        rd->mult_inverse[i] = sampler->multInverse[i];
    }
    rd->sample_stride = sampler->sampleStride;
    rd->sample_at_pixel_center = sampler->sampleAtPixelCenter ? 1 : 0;
    rd->sampler = sampler->kind ? sampler->kind : (sampler->sobol ? 1 : 0);
    rd->sampler_dims = sampler->nSampledDimensions;
    rd->strat_samples[0] = sampler->xPixelSamples; rd->strat_samples[1] = sampler->yPixelSamples;
    rd->strat_jitter = sampler->jitterSamples ? 1 : 0;
    rd->sobol_resolution = sampler->resolution; rd->sobol_log2_resolution = sampler->log2Resolution;
    rd->max_depth = maxDepth; rd->rr_threshold = rrThreshold;
    for (int i = 0; i < 4; ++i) rd->pixel_bounds[i] = pixelBounds[i];
    rd->tile_first = 0; rd->tile_step = 1;
    // The box filter's fast film path (filter_general = 0: every sample into its own pixel's sum, the rare sample that also lands
    // in a second pixel appended to that pixel's sum afterwards) has the reference's summation order only while no sample lands in
    // a pixel the reference visits LATER than the sample's own: FilmTile::AddSample adds in the order the samples are taken
    // (film.h:121-161), so a sample of pixel x whose film position x + u rounds UP to x + 1 is added to pixel x + 1 BEFORE that
    // pixel's own samples.  Whether that can happen is a property of sampler, resolution and sample count
    // (pgh_box_filter_needs_gather, include/pbrt_gpu.h); if it can, the frame takes the gathering film path.
    if (!rd->filter_general && pgh_box_filter_needs_gather(rd)) rd->filter_general = 1;
    if (rd->filter_general) film.TileHalo(rd->tile_halo);
    rd->tile_pixels = rd->filter_general ? (16 + rd->tile_halo[0] + rd->tile_halo[2]) * (16 + rd->tile_halo[1] + rd->tile_halo[3]) : 256;
}

// The C ABI is bound at run time, the way a pbrt maintainer's plugin loader
// would bind it; a missing library is a hard error, never a CPU fallback.
namespace {
struct GpuApi {
    void *lib = nullptr;
    decltype(&pg_set_device) set_device = nullptr;
    decltype(&pg_last_error) last_error = nullptr;
    decltype(&pg_scene_create) scene_create = nullptr;
    decltype(&pg_scene_destroy) scene_destroy = nullptr;
    decltype(&pg_render_tile_count) render_tile_count = nullptr;
    decltype(&pg_render) render = nullptr;
    decltype(&pg_render_sharded) render_sharded = nullptr;
    decltype(&pg_counters) counters = nullptr;
    decltype(&pg_hlbvh_build) hlbvh_build = nullptr;
    bool Load() {
        if (lib) return true;
        std::string path;
        if (const char *e = getenv("PBRT_GPU_LIB")) path = e;
        else {
            Dl_info info;
            if (dladdr((void *)&CreatePathIntegrator, &info) && info.dli_fname) path = DirectoryContaining(info.dli_fname) + "/libpbrt_gpu.so";
            else path = "libpbrt_gpu.so";
        }
        lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!lib) { Error("Unable to load the HIP back end \"%s\": %s", path.c_str(), dlerror()); return false; }
#define BIND(n) n = (decltype(n))dlsym(lib, "pg_" #n); if (!n) { Error("libpbrt_gpu.so lacks symbol pg_" #n); return false; }
        BIND(set_device) BIND(last_error) BIND(scene_create) BIND(scene_destroy) BIND(render_tile_count) BIND(render) BIND(render_sharded) BIND(counters) BIND(hlbvh_build)
#undef BIND
        return true;
    }
};
GpuApi gpuApi;
}  // namespace

bool DeviceHLBVHBuild(int n, const float *bounds, int maxPrimsInNode, std::vector<PgBVHNode> *nodes, std::vector<int> *order) {
    if (!gpuApi.Load()) return false;
    if (gpuApi.set_device(PbrtOptions.device) != PG_OK) { Error("pg_set_device: %s", gpuApi.last_error()); return false; }
    nodes->resize(2 * (size_t)n);
    order->resize((size_t)n);
    int nNodes = 0;
    if (gpuApi.hlbvh_build(n, bounds, maxPrimsInNode, nodes->data(), &nNodes, order->data()) != PG_OK) { Error("pg_hlbvh_build: %s", gpuApi.last_error()); return false; }
    nodes->resize((size_t)nNodes);
    return true;
}

// Prints the reference's own statistics (scene.cpp:40-42, integrator.cpp:48, triangle.cpp:45) summed over the devices used.
static void ReportStatistics(const std::vector<PgCounters> &all, double sec) {
    if (PbrtOptions.quiet) return;
    PgCounters c;
    memset(&c, 0, sizeof(c));
    for (const PgCounters &k : all) {
        c.camera_rays += k.camera_rays; c.closest_rays += k.closest_rays; c.shadow_rays += k.shadow_rays;
        c.tri_tests += k.tri_tests; c.node_visits += k.node_visits;
    }
    const double rays = double(c.closest_rays + c.shadow_rays);
    fprintf(stderr, "Statistics:\n  Integrator/Camera rays traced %llu\n  Intersections/Regular ray intersection tests %llu\n"
                    "  Intersections/Shadow ray intersection tests %llu\n  Intersections/Ray-triangle intersection tests %llu\n"
                    "  BVH/Node fetches %llu\n  Integrator::Render() %.3f s on %d GPU(s)  (%.2f Mrays/s, %.2f Msamples/s)\n",
            (unsigned long long)c.camera_rays, (unsigned long long)c.closest_rays, (unsigned long long)c.shadow_rays,
            (unsigned long long)c.tri_tests, (unsigned long long)c.node_visits, sec, (int)all.size(), rays / sec / 1e6, double(c.camera_rays) / sec / 1e6);
}

void GpuPathIntegrator::Render(const Scene &scene) {
    if (!gpuApi.Load()) { Error("Rendering aborted: no HIP back end (there is no CPU fallback)."); Fatal(); }
    FlatScene flat;
    Flatten(scene, &flat);
    PgRenderDesc rd;
    FillRenderDesc(&rd);
    // The devices of this node the frame is sharded over: tile t of the full-frame tiling goes to device t mod N, as the
    // reference shards a frame over machines with crop windows (main/pbrt.cpp:94-100); one host thread per device inside
    // pg_render_sharded, the shards gathered peer-to-peer on the first device.
    std::vector<int> devices = PbrtOptions.devices;
    if (devices.empty()) devices.push_back(PbrtOptions.device);
    const int n = (int)devices.size();
    std::vector<PgScene *> dev((size_t)n, nullptr);
    {   // the scene replicated on every device, uploads in parallel (pg_set_device selects the device per thread)
        std::vector<std::string> err((size_t)n);
        auto create = [&](int r) {
            if (gpuApi.set_device(devices[r]) != PG_OK || gpuApi.scene_create(&flat.desc, &dev[r]) != PG_OK) { err[r] = gpuApi.last_error(); dev[r] = nullptr; }
        };
        std::vector<std::thread> threads;
        for (int r = 1; r < n; ++r) threads.emplace_back(create, r);
        create(0);
        for (auto &t : threads) t.join();
        for (int r = 0; r < n; ++r)
            if (!dev[r]) {  // the scenes that were created on the other devices go before the error unwinds (Fatal is an exception at the C ABI)
                Error("pg_scene_create on device %d: %s", devices[r], err[r].c_str());
                for (PgScene *d : dev) if (d) gpuApi.scene_destroy(d);
                Fatal();
            }
    }
    std::vector<std::vector<PgFilmPixel>> film((size_t)n);
    std::vector<std::vector<PgStraySample>> strays((size_t)n);
    std::vector<PgFilmPixel *> filmPtr((size_t)n);
    std::vector<PgStraySample *> strayPtr((size_t)n);
    std::vector<int> nStrays((size_t)n, 0);
    std::vector<PgRenderDesc> shard((size_t)n, rd);
    int maxStrays = 0;
    for (int r = 0; r < n; ++r) {
        shard[r].tile_first = r; shard[r].tile_step = n;
        const int nTiles = gpuApi.render_tile_count(&shard[r]);
        film[r].resize((size_t)nTiles * (size_t)rd.tile_pixels);
        // stray samples are rare, except under the MaxMinDistSampler: every pixel's first sample is (0, 0) and lands in up to three neighbours
        maxStrays = std::max(maxStrays, nTiles * (rd.sampler == PG_SAMPLER_MAXMINDIST ? 256 * 4 : 256 / 8) + 1024);
    }
    for (int r = 0; r < n; ++r) { strays[r].resize((size_t)maxStrays); filmPtr[r] = film[r].data(); strayPtr[r] = strays[r].data(); }
    auto t0 = std::chrono::steady_clock::now();
    int st;
    if (n == 1) {
        if (gpuApi.set_device(devices[0]) != PG_OK) { Error("pg_set_device: %s", gpuApi.last_error()); Fatal(); }
        st = gpuApi.render(dev[0], &rd, filmPtr[0], strayPtr[0], maxStrays, &nStrays[0], PG_MEM_HOST, nullptr);
    } else st = gpuApi.render_sharded(dev.data(), n, &rd, filmPtr.data(), strayPtr.data(), maxStrays, nStrays.data());
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (st != PG_OK) { Error("pg_render: %s", gpuApi.last_error()); for (PgScene *d : dev) gpuApi.scene_destroy(d); Fatal(); }
    std::vector<PgCounters> counters((size_t)n);
    for (int r = 0; r < n; ++r) if (gpuApi.counters(dev[r], &counters[r]) != PG_OK) memset(&counters[r], 0, sizeof(PgCounters));
    ReportStatistics(counters, sec);
    for (PgScene *d : dev) gpuApi.scene_destroy(d);
    camera->film->MergeShards(rd, n, filmPtr.data(), strayPtr.data(), nStrays.data());  // Film::MergeFilmTile per tile, in the frame's tile order whatever n is
    camera->film->WriteImage();  // integrator.cpp:338
}
}  // namespace pbrt
