// gpupath_binding.cpp -- the REFERENCE-SIDE binding of the MI355X path, compiled.
//
// TEST INFRASTRUCTURE.  This file is what a pbrt-v3 maintainer would add to the reference tree: an Integrator subclass
// (src/core/integrator.h:53-58) whose Render() flattens the reference's own, already built Scene -- its parser's output,
// its BVHAccel's node array and primitive order, its TriangleMesh vertex arrays, its materials' BxDF lists, its lights,
// camera, film and sampler objects -- into PgSceneDesc / PgRenderDesc, renders through the C ABI of include/pbrt_gpu.h
// (libpbrt_gpu.so, bound at run time) and hands the result to the reference's Film::MergeFilmTile / WriteImage
// (src/core/film.cpp:117-130, :169-211).  Nothing of this repository's own front end (parser, BVH builder, Film) is
// involved, so a render through this binary proves the drop-in end to end.
//
// Build (oracle/Makefile.ref, target _ref/pbrt_gpubind): this file with -fno-access-control (the members it reads are
// private in the reference: accelerators/bvh.h:93-94, core/primitive.h:85-88, shapes/triangle.h:111-113 ...; a
// maintainer would add a friend declaration instead), the reference's own src/core/api.cpp compiled once more from where
// it lies with -DCreatePathIntegrator=GpuBind_CreatePathIntegrator -DCreateVolPathIntegrator=GpuBind_CreateVolPathIntegrator
// (the "two lines in api.cpp" of INTEGRATION.md, expressed without touching the file), src/main/pbrt.cpp and libpbrt_ref.a.
// `Integrator "path"` / `"volpath"` of any .pbrt file then selects the device path.
//
// Closed set of this binding: triangle meshes (with N / S / uv), spheres, cylinders and disks; materials whose parameters are
// constant textures (their BxDF lists and, for the subsurface materials, their TabulatedBSSRDF are read off Material::ComputeScatteringFunctions); diffuse area lights, point, spot
// and distant lights; perspective and orthographic cameras; all six samplers; every pixel filter; homogeneous
// and grid-density media.  Anything else is reported with Error() and the process exits -- there is no CPU fallback here either.
#include <dlfcn.h>
#include <unistd.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "pbrt.h"
#include "accelerators/bvh.h"
#include "api.h"
#include "bssrdf.h"
#include "camera.h"
#include "cameras/orthographic.h"
#include "cameras/perspective.h"
#include "film.h"
#include "filters/box.h"
#include "integrator.h"
#include "integrators/path.h"
#include "integrators/volpath.h"
#include "light.h"
#include "lights/diffuse.h"
#include "lights/distant.h"
#include "lights/point.h"
#include "lights/spot.h"
#include "lowdiscrepancy.h"
#include "material.h"
#include "media/grid.h"
#include "media/homogeneous.h"
#include "memory.h"
#include "microfacet.h"
#include "paramset.h"
#include "primitive.h"
#include "reflection.h"
#include "sampler.h"
#include "stats.h"
#include "samplers/halton.h"
#include "samplers/maxmin.h"
#include "samplers/random.h"
#include "samplers/sobol.h"
#include "samplers/stratified.h"
#include "samplers/zerotwosequence.h"
#include "scene.h"
#include "shapes/cylinder.h"
#include "shapes/disk.h"
#include "shapes/sphere.h"
#include "shapes/triangle.h"
#include "sobolmatrices.h"

#include "../include/pbrt_gpu.h"

namespace pbrt {
namespace {

// exit() would wait for ever on the reference's parked worker threads (core/parallel.cpp:305-322): leave at once
[[noreturn]] void Die() { fflush(stdout); fflush(stderr); _exit(1); }
[[noreturn]] void Unsupported(const char *what) {
    Error("gpupath binding: %s is outside the device path's closed set; there is no CPU fallback.", what);
    Die();
}

// ---- the C ABI, bound at run time ---------------------------------------------------------------------------------------------
struct Abi {
    decltype(&pg_set_device) set_device;
    decltype(&pg_last_error) last_error;
    decltype(&pg_scene_create) scene_create;
    decltype(&pg_scene_destroy) scene_destroy;
    decltype(&pg_render_tile_count) render_tile_count;
    decltype(&pg_render) render;
    decltype(&pg_counters) counters;
    Abi() {
        const char *path = getenv("PBRT_GPU_LIB");
        void *lib = dlopen(path ? path : "libpbrt_gpu.so", RTLD_NOW | RTLD_LOCAL);
        if (!lib) { Error("gpupath binding: cannot load the HIP back end (%s)", dlerror()); Die(); }
#define BIND(n) n = (decltype(n))dlsym(lib, "pg_" #n); if (!n) { Error("libpbrt_gpu.so lacks pg_" #n); Die(); }
        BIND(set_device) BIND(last_error) BIND(scene_create) BIND(scene_destroy) BIND(render_tile_count) BIND(render) BIND(counters)
#undef BIND
    }
};

void CopyMatrix(const Matrix4x4 &m, float *dst) { for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) dst[4 * r + c] = m.m[r][c]; }
void Copy3x3(const Matrix4x4 &m, float *dst) { for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) dst[3 * r + c] = m.m[r][c]; }
void CopyRGB(const Spectrum &s, float *dst) { Float rgb[3]; s.ToRGB(rgb); for (int c = 0; c < 3; ++c) dst[c] = rgb[c]; }

// LinearBVHNode is defined inside accelerators/bvh.cpp (:95-104); PgBVHNode is byte-identical to it.  The node count is
// not kept by BVHAccel: the end of the depth-first layout is found by walking it.
int CountNodes(const PgBVHNode *nodes) {
    int end = 1, i = 0;
    for (;;) {
        if (nodes[i].nprims > 0) { end = std::max(end, i + 1); break; }  // the last leaf of the right spine ends the array
        i = nodes[i].offset;  // second child: its subtree is laid out after the first child's
    }
    return end;
}

// ---- materials: a constant-texture material IS the BxDF list its ComputeScatteringFunctions adds (reflection.h:164-213) ----------
PgBxDF Lobe(int type) {
    PgBxDF b;
    memset(&b, 0, sizeof(b));
    b.type = type; b.eta_a = b.eta_b = 1; b.alpha_x = b.alpha_y = 1; b.on_a = 1;
    return b;
}
void SetDistribution(PgBxDF *b, const MicrofacetDistribution *d) {
    const TrowbridgeReitzDistribution *tr = dynamic_cast<const TrowbridgeReitzDistribution *>(d);
    if (!tr) Unsupported("a microfacet distribution other than TrowbridgeReitz");
    b->alpha_x = tr->alphax; b->alpha_y = tr->alphay;
}
void SetFresnel(PgBxDF *b, const Fresnel *f) {
    if (const FresnelDielectric *fd = dynamic_cast<const FresnelDielectric *>(f)) { b->fresnel = PG_FRESNEL_DIELECTRIC; b->eta_a = fd->etaI; b->eta_b = fd->etaT; }
    else if (const FresnelConductor *fc = dynamic_cast<const FresnelConductor *>(f)) {
        b->fresnel = PG_FRESNEL_CONDUCTOR;
        CopyRGB(fc->etaT, b->cond_eta); CopyRGB(fc->k, b->cond_k);
    } else if (dynamic_cast<const FresnelNoOp *>(f)) b->fresnel = PG_FRESNEL_NOOP;
    else Unsupported("a Fresnel term other than dielectric / conductor / no-op");
}
PgBxDF ConvertBxDF(const BxDF *bx) {
    if (const ScaledBxDF *s = dynamic_cast<const ScaledBxDF *>(bx)) {  // MixMaterial's wrappers, innermost first
        PgBxDF b = ConvertBxDF(s->bxdf);
        if (b.n_scales == PG_MAX_BXDF_SCALES) Unsupported("mix materials nested more than three deep");
        CopyRGB(s->scale, b.scale[b.n_scales++]);
        return b;
    }
    if (const LambertianReflection *l = dynamic_cast<const LambertianReflection *>(bx)) { PgBxDF b = Lobe(PG_BXDF_LAMBERT_R); CopyRGB(l->R, b.R); return b; }
    if (const LambertianTransmission *l = dynamic_cast<const LambertianTransmission *>(bx)) { PgBxDF b = Lobe(PG_BXDF_LAMBERT_T); CopyRGB(l->T, b.T); return b; }
    if (const OrenNayar *o = dynamic_cast<const OrenNayar *>(bx)) { PgBxDF b = Lobe(PG_BXDF_OREN_NAYAR); CopyRGB(o->R, b.R); b.on_a = o->A; b.on_b = o->B; return b; }
    if (const SpecularReflection *s = dynamic_cast<const SpecularReflection *>(bx)) { PgBxDF b = Lobe(PG_BXDF_SPECULAR_R); CopyRGB(s->R, b.R); SetFresnel(&b, s->fresnel); return b; }
    if (const SpecularTransmission *s = dynamic_cast<const SpecularTransmission *>(bx)) {
        PgBxDF b = Lobe(PG_BXDF_SPECULAR_T); CopyRGB(s->T, b.T); b.eta_a = s->etaA; b.eta_b = s->etaB; return b;
    }
    if (const FresnelSpecular *s = dynamic_cast<const FresnelSpecular *>(bx)) {
        PgBxDF b = Lobe(PG_BXDF_FRESNEL_SPECULAR); CopyRGB(s->R, b.R); CopyRGB(s->T, b.T); b.eta_a = s->etaA; b.eta_b = s->etaB; return b;
    }
    if (const MicrofacetReflection *m = dynamic_cast<const MicrofacetReflection *>(bx)) {
        PgBxDF b = Lobe(PG_BXDF_MICROFACET_R); CopyRGB(m->R, b.R); SetFresnel(&b, m->fresnel); SetDistribution(&b, m->distribution); return b;
    }
    if (const MicrofacetTransmission *m = dynamic_cast<const MicrofacetTransmission *>(bx)) {
        PgBxDF b = Lobe(PG_BXDF_MICROFACET_T); CopyRGB(m->T, b.T); b.eta_a = m->etaA; b.eta_b = m->etaB; SetDistribution(&b, m->distribution); return b;
    }
    if (const FresnelBlend *f = dynamic_cast<const FresnelBlend *>(bx)) {
        PgBxDF b = Lobe(PG_BXDF_FRESNEL_BLEND); CopyRGB(f->Rd, b.R); CopyRGB(f->Rs, b.T); SetDistribution(&b, f->distribution); return b;
    }
    Unsupported("a BxDF outside reflection.h's Lambertian / Oren-Nayar / specular / microfacet / Fresnel-blend set");
}
// What a subsurface material leaves in si->bssrdf (subsurface.cpp:87-90): TabulatedBSSRDF's members (bssrdf.h:141-165)
struct BssrdfProbe {
    bool present = false;
    float eta = 0, sigma_t[3] = {0, 0, 0}, rho[3] = {0, 0, 0};
    const BSSRDFTable *table = nullptr;
    const Material *material = nullptr;  // SeparableBSSRDF::material: what the probe rays' hits are compared with (bssrdf.cpp:301)
    bool operator==(const BssrdfProbe &o) const {
        return present == o.present && eta == o.eta && !memcmp(sigma_t, o.sigma_t, sizeof(sigma_t)) && !memcmp(rho, o.rho, sizeof(rho)) && table == o.table && material == o.material;
    }
};
// the material's BxDF list at one (arbitrary) surface point
void ListBxDFs(const Material *mat, Float u, Float v, std::vector<PgBxDF> *out, float *eta, BssrdfProbe *bss = nullptr) {
    MemoryArena arena;
    SurfaceInteraction si(Point3f(u, v, 0.25f), Vector3f(0, 0, 0), Point2f(u, v), Vector3f(0, 0, 1), Vector3f(1, 0, 0), Vector3f(0, 1, 0),
                          Normal3f(0, 0, 0), Normal3f(0, 0, 0), 0, nullptr);
    mat->ComputeScatteringFunctions(&si, arena, TransportMode::Radiance, true);
    out->clear();
    *eta = 1;
    if (bss && si.bssrdf) {
        const TabulatedBSSRDF *t = dynamic_cast<const TabulatedBSSRDF *>(si.bssrdf);
        if (!t) Unsupported("a BSSRDF other than TabulatedBSSRDF");
        bss->present = true; bss->eta = t->eta; bss->table = &t->table; bss->material = t->material;
        CopyRGB(t->sigma_t, bss->sigma_t); CopyRGB(t->rho, bss->rho);
    }
    if (!si.bsdf) return;
    for (int i = 0; i < si.bsdf->nBxDFs; ++i) out->push_back(ConvertBxDF(si.bsdf->bxdfs[i]));
    *eta = si.bsdf->eta;
}

struct Flat {
    std::vector<PgBVHNode> nodes;
    std::vector<int32_t> indices, triMaterial, triLight, triMedIn, triMedOut;
    std::vector<uint32_t> triFlags;
    std::vector<float> P, N, UV, S;
    std::vector<PgMaterial> materials;
    std::vector<PgBxDF> bxdfs;
    std::vector<PgLight> lights;
    std::vector<PgSphere> spheres;
    std::vector<PgObject> objects;      // the primitives TransformedPrimitives wrap: a BVHAccel's nodes / primitives appended, or one primitive
    std::vector<PgInstance> instances;  // one per TransformedPrimitive of the scene's BVH
    std::vector<PgMedium> media;
    std::vector<int32_t> mediaGrid;   // per medium: index into grids, -1 = HomogeneousMedium
    std::vector<PgDensityGrid> grids;
    std::vector<float> gridDensity;
    std::map<const Medium *, int> mediumIndex;
    std::vector<PgBSSRDF> bssrdfs;          // one per material with a TabulatedBSSRDF
    std::vector<int32_t> materialBssrdf;    // per material, -1 = none
    std::vector<const Material *> bssrdfMaterial;  // per PgBSSRDF: the Material its probe rays look for
    std::vector<float> bssrdfTables;
    std::map<const BSSRDFTable *, int64_t> bssrdfTableOf;
    std::vector<int32_t> permSums;
    PgSceneDesc desc;
};

int InternMedium(const Medium *m, Flat *flat) {
    if (!m) return -1;
    auto it = flat->mediumIndex.find(m);
    if (it != flat->mediumIndex.end()) return it->second;
    PgMedium pm;
    int grid = -1;
    if (const HomogeneousMedium *h = dynamic_cast<const HomogeneousMedium *>(m)) {
        CopyRGB(h->sigma_a, pm.sigma_a); CopyRGB(h->sigma_s, pm.sigma_s); CopyRGB(h->sigma_t, pm.sigma_t);
        pm.g = h->g;
    } else if (const GridDensityMedium *gm = dynamic_cast<const GridDensityMedium *>(m)) {  // media/grid.h:49-96
        CopyRGB(gm->sigma_a, pm.sigma_a); CopyRGB(gm->sigma_s, pm.sigma_s); CopyRGB(gm->sigma_a + gm->sigma_s, pm.sigma_t);
        pm.g = gm->g;
        PgDensityGrid gd;
        gd.nx = gm->nx; gd.ny = gm->ny; gd.nz = gm->nz; gd.reserved = 0;
        gd.density_offset = (int64_t)flat->gridDensity.size();
        gd.sigma_t = gm->sigma_t; gd.inv_max_density = gm->invMaxDensity;
        const Matrix4x4 &w2m = gm->WorldToMedium.GetMatrix();
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) gd.world_to_medium[4 * r + c] = w2m.m[r][c];
        const Float *den = gm->density.get();
        flat->gridDensity.insert(flat->gridDensity.end(), den, den + (size_t)gm->nx * gm->ny * gm->nz);
        grid = (int)flat->grids.size();
        flat->grids.push_back(gd);
    } else Unsupported("a medium other than HomogeneousMedium / GridDensityMedium");
    flat->media.push_back(pm);
    flat->mediaGrid.push_back(grid);
    return flat->mediumIndex[m] = (int)flat->media.size() - 1;
}
// the media tables of the description (again after the camera's medium was added)
void SetMediaTables(Flat *flat) {
    PgSceneDesc &d = flat->desc;
    d.n_media = (int)flat->media.size(); d.media = flat->media.data();
    if (!flat->grids.empty()) {
        d.n_grids = (int)flat->grids.size(); d.grids = flat->grids.data(); d.media_grid = flat->mediaGrid.data();
        d.n_density_floats = (int64_t)flat->gridDensity.size(); d.grid_density = flat->gridDensity.data();
    }
}

void FlattenScene(const Scene &scene, int maxDepth, bool volumetric, const std::string &strategy, const Sampler &sampler, Flat *flat) {
    const BVHAccel *bvh = dynamic_cast<const BVHAccel *>(scene.aggregate.get());
    if (!bvh) Unsupported("an aggregate other than BVHAccel (Accelerator \"kdtree\")");
    if (bvh->primitives.empty() || !bvh->nodes) Unsupported("an empty scene");
    const PgBVHNode *nodes = reinterpret_cast<const PgBVHNode *>(bvh->nodes);
    flat->nodes.assign(nodes, nodes + CountNodes(nodes));
    const size_t nTop = bvh->primitives.size();
    const int nTopNodes = (int)flat->nodes.size();
    // every primitive the device sees: the scene BVH's own (BVHAccel::primitives order), then -- once per distinct wrapped primitive -- what the
    // TransformedPrimitives among them wrap (api.cpp:1399-1419 a moving shape's primitives, :1567-1586 an object instance): a BVHAccel's nodes
    // and primitives appended verbatim, or a lone GeometricPrimitive (include/pbrt_gpu.h, PgObject / PgInstance)
    std::vector<const Primitive *> all;
    for (const auto &pr : bvh->primitives) all.push_back(pr.get());
    std::map<const Primitive *, int> objectIndex;
    // (`all` grows inside the loop: a TransformedPrimitive among an object's primitives -- a moving shape inside an object definition,
    // api.cpp:1386-1419 -- wraps an object of its own, found when the loop reaches it: PG_PRIM_INSTANCE inside an object's run, ABI 29)
    std::vector<int> instanceOf;
    for (size_t k = 0; k < all.size(); ++k) {
        instanceOf.resize(all.size(), -1);
        const TransformedPrimitive *tp = dynamic_cast<const TransformedPrimitive *>(all[k]);
        if (!tp) continue;
        const Primitive *inner = tp->primitive.get();
        if (!objectIndex.count(inner)) {
            PgObject o;
            o.first_prim = (int)all.size(); o.first_node = (int)flat->nodes.size();
            if (const BVHAccel *ob = dynamic_cast<const BVHAccel *>(inner)) {
                const PgBVHNode *on = reinterpret_cast<const PgBVHNode *>(ob->nodes);
                o.n_nodes = CountNodes(on); o.n_prims = (int)ob->primitives.size();
                flat->nodes.insert(flat->nodes.end(), on, on + o.n_nodes);
                for (const auto &pr : ob->primitives) all.push_back(pr.get());
            } else if (dynamic_cast<const GeometricPrimitive *>(inner) || (k < nTop && dynamic_cast<const TransformedPrimitive *>(inner))) {
                // one primitive, no accelerator (api.cpp:1567) -- a shape, or the definition's only primitive is a moving shape's TransformedPrimitive
                o.n_nodes = 0; o.n_prims = 1; all.push_back(inner);
            }
            else Unsupported("a TransformedPrimitive over something other than a BVHAccel or one GeometricPrimitive");
            objectIndex[inner] = (int)flat->objects.size();
            flat->objects.push_back(o);
        }
        PgInstance in;
        memset(&in, 0, sizeof(in));
        const AnimatedTransform &a = tp->PrimitiveToWorld;
        CopyMatrix(a.startTransform->m, in.i2w); CopyMatrix(a.startTransform->mInv, in.w2i);
        in.object = objectIndex[inner];
        in.identity = a.startTransform->IsIdentity() ? 1 : 0;
        if (a.actuallyAnimated) {  // the reference's own decomposition (AnimatedTransform's constructor, transform.cpp:396-411)
            in.animated = 1;
            in.time[0] = a.startTime; in.time[1] = a.endTime;
            CopyMatrix(a.endTransform->m, in.i2w_end); CopyMatrix(a.endTransform->mInv, in.w2i_end);
            for (int e = 0; e < 2; ++e) {
                in.T[e][0] = a.T[e].x; in.T[e][1] = a.T[e].y; in.T[e][2] = a.T[e].z;
                in.R[e][0] = a.R[e].v.x; in.R[e][1] = a.R[e].v.y; in.R[e][2] = a.R[e].v.z; in.R[e][3] = a.R[e].w;
                for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) in.S[e][3 * i + j] = a.S[e].m[i][j];
            }
        }
        instanceOf.resize(all.size(), -1);
        instanceOf[k] = (int)flat->instances.size();
        flat->instances.push_back(in);
    }
    const size_t n = all.size();
    instanceOf.resize(n, -1);
    flat->indices.assign(3 * n, 0); flat->triFlags.assign(n, 0); flat->triMaterial.assign(n, 0); flat->triLight.assign(n, -1);
    std::map<const Light *, int> lightIndex;
    for (size_t i = 0; i < scene.lights.size(); ++i) lightIndex[scene.lights[i].get()] = (int)i;
    flat->lights.resize(scene.lights.size());
    for (PgLight &l : flat->lights) { memset(&l, 0, sizeof(l)); l.prim = -1; l.env_image = -1; }
    std::map<const TriangleMesh *, int> meshBase;
    std::map<const Material *, int> materialIndex;
    std::vector<const TriangleMesh *> meshes;
    bool anyN = false, anyUV = false, anyS = false, anyMedium = false;
    // pass 1: vertex arrays of the meshes in first-use order
    for (size_t k = 0; k < n; ++k) {
        if (instanceOf[k] >= 0) continue;
        const GeometricPrimitive *gp = dynamic_cast<const GeometricPrimitive *>(all[k]);
        if (!gp) Unsupported("a primitive that is neither a GeometricPrimitive nor a TransformedPrimitive");
        if (const Triangle *tri = dynamic_cast<const Triangle *>(gp->shape.get())) {
            const TriangleMesh *m = tri->mesh.get();
            if (!meshBase.count(m)) {
                meshBase[m] = (int)(flat->P.size() / 3);
                meshes.push_back(m);
                for (int i = 0; i < m->nVertices; ++i) { flat->P.push_back(m->p[i].x); flat->P.push_back(m->p[i].y); flat->P.push_back(m->p[i].z); }
                anyN |= (bool)m->n; anyUV |= (bool)m->uv; anyS |= (bool)m->s;
                if (m->alphaMask || m->shadowAlphaMask) Unsupported("alpha-mask textures in this binding");
            }
        }
    }
    const size_t nVerts = flat->P.size() / 3;
    if (anyN) flat->N.assign(3 * nVerts, 0.f);
    if (anyUV) flat->UV.assign(2 * nVerts, 0.f);
    if (anyS) flat->S.assign(3 * nVerts, 0.f);
    for (const TriangleMesh *m : meshes) {
        const size_t b = meshBase[m];
        for (int i = 0; i < m->nVertices; ++i) {
            if (m->n) { flat->N[3 * (b + i)] = m->n[i].x; flat->N[3 * (b + i) + 1] = m->n[i].y; flat->N[3 * (b + i) + 2] = m->n[i].z; }
            if (m->s) { flat->S[3 * (b + i)] = m->s[i].x; flat->S[3 * (b + i) + 1] = m->s[i].y; flat->S[3 * (b + i) + 2] = m->s[i].z; }
            if (m->uv) { flat->UV[2 * (b + i)] = m->uv[i].x; flat->UV[2 * (b + i) + 1] = m->uv[i].y; }
        }
    }
    // pass 2: primitives in BVHAccel::primitives (= orderedPrims) order
    for (size_t k = 0; k < n; ++k) {
        if (instanceOf[k] >= 0) {  // TransformedPrimitive: primitive.h:92-117
            flat->indices[3 * k] = instanceOf[k];
            flat->triFlags[k] = PG_PRIM_INSTANCE;
            continue;
        }
        const GeometricPrimitive *gp = static_cast<const GeometricPrimitive *>(all[k]);
        const Shape *shape = gp->shape.get();
        if (const Triangle *tri = dynamic_cast<const Triangle *>(shape)) {
            const TriangleMesh *m = tri->mesh.get();
            const int base = meshBase[m];
            for (int j = 0; j < 3; ++j) flat->indices[3 * k + j] = base + tri->v[j];
            uint32_t f = 0;
            if (tri->reverseOrientation ^ tri->transformSwapsHandedness) f |= PG_TRI_FLIP_NORMAL;
            if (tri->reverseOrientation) f |= PG_TRI_REVERSE_ORIENTATION;
            if (m->n) f |= PG_TRI_HAS_N;
            if (m->uv) f |= PG_TRI_HAS_UV;
            if (m->s) f |= PG_TRI_HAS_S;
            flat->triFlags[k] = f;
        } else {
            PgSphere g;
            memset(&g, 0, sizeof(g));
            CopyMatrix(shape->ObjectToWorld->m, g.o2w); CopyMatrix(shape->WorldToObject->m, g.w2o);
            g.reverse_orientation = shape->reverseOrientation; g.swaps_handedness = shape->transformSwapsHandedness;
            g.area = shape->Area();
            if (const Sphere *s = dynamic_cast<const Sphere *>(shape)) {
                g.shape = PG_SHAPE_SPHERE; g.radius = s->radius; g.z_min = s->zMin; g.z_max = s->zMax;
                g.theta_min = s->thetaMin; g.theta_max = s->thetaMax; g.phi_max = s->phiMax;
            } else if (const Cylinder *c = dynamic_cast<const Cylinder *>(shape)) {
                g.shape = PG_SHAPE_CYLINDER; g.radius = c->radius; g.z_min = c->zMin; g.z_max = c->zMax; g.phi_max = c->phiMax;
            } else if (const Disk *d = dynamic_cast<const Disk *>(shape)) {
                g.shape = PG_SHAPE_DISK; g.radius = d->radius; g.height = d->height; g.inner_radius = d->innerRadius; g.phi_max = d->phiMax;
                g.z_min = g.z_max = d->height;
            } else Unsupported("a shape other than triangle / sphere / cylinder / disk in this binding");
            flat->indices[3 * k] = (int)flat->spheres.size();
            flat->triFlags[k] = PG_PRIM_SPHERE;
            flat->spheres.push_back(g);
        }
        // GeometricPrimitive::material -> an interned PgMaterial carrying its BxDF list
        const Material *mat = gp->material.get();
        auto mi = materialIndex.find(mat);
        if (mi == materialIndex.end()) {
            PgMaterial pm;
            memset(&pm, 0, sizeof(pm));
            pm.bsdf_eta = 1; pm.textured_index = -1;
            if (!mat) pm.type = PG_MAT_NONE;  // a medium boundary (path.cpp:107-113)
            else {
                std::vector<PgBxDF> a, b;
                float etaA, etaB;
                BssrdfProbe sa, sb;
                ListBxDFs(mat, 0.125f, 0.25f, &a, &etaA, &sa);
                ListBxDFs(mat, 0.625f, 0.75f, &b, &etaB, &sb);  // a second point: a non-constant texture shows up as a different list
                if (a.size() != b.size() || etaA != etaB || (a.size() && memcmp(a.data(), b.data(), a.size() * sizeof(PgBxDF))) || !(sa == sb))
                    Unsupported("a material with non-constant textures in this binding");
                if (sa.present) {  // SubsurfaceMaterial / KdSubsurfaceMaterial: the table is the material's own BSSRDFTable (subsurface.h:73-75)
                    PgBSSRDF pb;
                    memset(&pb, 0, sizeof(pb));
                    pb.eta = sa.eta;
                    pb.a.tex = pb.b.tex = -1;  // constants (textured == 0): sigma_t / rho below are final
                    for (int c = 0; c < 3; ++c) { pb.sigma_t[c] = sa.sigma_t[c]; pb.rho[c] = sa.rho[c]; }
                    const BSSRDFTable &t = *sa.table;
                    pb.n_rho = t.nRhoSamples; pb.n_radius = t.nRadiusSamples;
                    auto ti = flat->bssrdfTableOf.find(&t);
                    if (ti == flat->bssrdfTableOf.end()) {
                        ti = flat->bssrdfTableOf.insert({&t, (int64_t)flat->bssrdfTables.size()}).first;
                        auto put = [&](const Float *v, size_t n) { flat->bssrdfTables.insert(flat->bssrdfTables.end(), v, v + n); };
                        const size_t nr = t.nRhoSamples, nd = t.nRadiusSamples;
                        put(t.rhoSamples.get(), nr); put(t.radiusSamples.get(), nd); put(t.profile.get(), nr * nd); put(t.rhoEff.get(), nr); put(t.profileCDF.get(), nr * nd);
                    }
                    pb.table = ti->second;
                    pb.match_material = -1;  // resolved below: TabulatedBSSRDF::material among the interned materials
                    flat->materialBssrdf.resize(flat->materials.size() + 1, -1);
                    flat->materialBssrdf[flat->materials.size()] = (int)flat->bssrdfs.size();
                    flat->bssrdfs.push_back(pb);
                    flat->bssrdfMaterial.push_back(sa.material);
                }
                pm.type = PG_MAT_LOBES; pm.first_bxdf = (int)flat->bxdfs.size(); pm.n_bxdfs = (int)a.size(); pm.bsdf_eta = etaA;
                flat->bxdfs.insert(flat->bxdfs.end(), a.begin(), a.end());
            }
            flat->materials.push_back(pm);
            mi = materialIndex.insert({mat, (int)flat->materials.size() - 1}).first;
        }
        flat->triMaterial[k] = mi->second;
        if (gp->areaLight) {
            const DiffuseAreaLight *al = dynamic_cast<const DiffuseAreaLight *>(gp->areaLight.get());
            if (!al) Unsupported("an area light other than DiffuseAreaLight");
            const int li = lightIndex.at(al);
            PgLight &l = flat->lights[li];
            l.type = PG_LIGHT_AREA; l.prim = (int)k; CopyRGB(al->Lemit, l.L); l.two_sided = al->twoSided; l.area = al->area;
            flat->triLight[k] = li;
        }
        const int mIn = InternMedium(gp->mediumInterface.inside, flat), mOut = InternMedium(gp->mediumInterface.outside, flat);
        if (mIn >= 0 || mOut >= 0) {
            if (!anyMedium) { flat->triMedIn.assign(n, -1); flat->triMedOut.assign(n, -1); anyMedium = true; }
            flat->triMedIn[k] = mIn; flat->triMedOut[k] = mOut;
        }
    }
    // the other lights of scene.lights, in their order
    for (size_t i = 0; i < scene.lights.size(); ++i) {
        const Light *lt = scene.lights[i].get();
        PgLight &l = flat->lights[i];
        if (dynamic_cast<const DiffuseAreaLight *>(lt)) { if (l.prim < 0) Unsupported("an area light whose shape is not in the scene"); continue; }
        if (const PointLight *p = dynamic_cast<const PointLight *>(lt)) { l.type = PG_LIGHT_POINT; CopyRGB(p->I, l.L); l.pos[0] = p->pLight.x; l.pos[1] = p->pLight.y; l.pos[2] = p->pLight.z; }
        else if (const SpotLight *s = dynamic_cast<const SpotLight *>(lt)) {
            l.type = PG_LIGHT_SPOT; CopyRGB(s->I, l.L); l.pos[0] = s->pLight.x; l.pos[1] = s->pLight.y; l.pos[2] = s->pLight.z;
            Copy3x3(s->WorldToLight.m, l.w2l); l.cos_total_width = s->cosTotalWidth; l.cos_falloff_start = s->cosFalloffStart;
        } else if (const DistantLight *d = dynamic_cast<const DistantLight *>(lt)) {
            l.type = PG_LIGHT_DISTANT; CopyRGB(d->L, l.L); l.pos[0] = d->wLight.x; l.pos[1] = d->wLight.y; l.pos[2] = d->wLight.z;
            l.world_radius = d->worldRadius;  // as Light::Preprocess left it (distant.h:55-57)
        } else Unsupported("a light other than diffuse area / point / spot / distant in this binding");
    }
    PgSceneDesc &d = flat->desc;
    memset(&d, 0, sizeof(d));
    d.abi_version = PG_ABI_VERSION;
    d.n_nodes = nTopNodes; d.n_nodes_all = (int)flat->nodes.size(); d.nodes = flat->nodes.data();
    d.n_tris = (int)nTop; d.n_prims_all = (int)n; d.indices = flat->indices.data(); d.tri_flags = flat->triFlags.data();
    d.n_objects = (int)flat->objects.size(); d.objects = flat->objects.data();
    d.n_instances = (int)flat->instances.size(); d.instances = flat->instances.data();
    d.tri_material = flat->triMaterial.data(); d.tri_light = flat->triLight.data();
    d.n_verts = (int)nVerts; d.P = flat->P.data();
    d.N = anyN ? flat->N.data() : nullptr; d.UV = anyUV ? flat->UV.data() : nullptr; d.S = anyS ? flat->S.data() : nullptr;
    d.n_materials = (int)flat->materials.size(); d.materials = flat->materials.data();
    d.n_lights = (int)flat->lights.size(); d.lights = flat->lights.data();
    // CreateLightSampleDistribution (lightdistrib.cpp:48-66)
    if (strategy == "uniform" || scene.lights.size() == 1) d.light_strategy = PG_LIGHTS_UNIFORM;
    else if (strategy == "power") d.light_strategy = PG_LIGHTS_POWER;
    else d.light_strategy = PG_LIGHTS_SPATIAL;
    // the reference's own digit permutations (HaltonSampler::radicalInversePermutations, halton.cpp:69-72) and PrimeSums
    if (dynamic_cast<const HaltonSampler *>(&sampler)) {
        const int nDims = volumetric || !flat->bssrdfs.empty() ? PrimeTableSize : std::min(PrimeTableSize, 5 + 8 * (maxDepth + 2));  // (a subsurface vertex draws 10 more values)
        for (int i = 0; i <= nDims; ++i) flat->permSums.push_back(i < PrimeTableSize ? PrimeSums[i] : PrimeSums[PrimeTableSize - 1] + Primes[PrimeTableSize - 1]);
        d.n_perm_dims = nDims; d.perms = HaltonSampler::radicalInversePermutations.data(); d.perm_sums = flat->permSums.data();
    } else if (dynamic_cast<const SobolSampler *>(&sampler)) {  // the reference's generator matrices
        d.sobol_matrices = SobolMatrices32; d.vdc_sobol = &VdCSobolMatrices[0][0]; d.vdc_sobol_inv = &VdCSobolMatricesInv[0][0];
    } else {  // a sampler over one RNG stream per tile: nothing but (for maxmindist) its generator matrices
        d.sobol_matrices = SobolMatrices32; d.vdc_sobol = &VdCSobolMatrices[0][0]; d.vdc_sobol_inv = &VdCSobolMatricesInv[0][0];  // (lets the Halton table be absent)
        d.cmaxmin = &CMaxMinDist[0][0];
    }
    d.n_spheres = (int)flat->spheres.size(); d.spheres = flat->spheres.data();
    d.n_bxdfs = (int)flat->bxdfs.size(); d.bxdfs = flat->bxdfs.data();
    SetMediaTables(flat);
    if (!flat->bssrdfs.empty()) {  // subsurface scattering tables (ABI 24)
        flat->materialBssrdf.resize(flat->materials.size(), -1);
        for (size_t i = 0; i < flat->bssrdfs.size(); ++i) {  // a component of a mix that no primitive carries itself matches nothing
            auto mi = materialIndex.find(flat->bssrdfMaterial[i]);
            flat->bssrdfs[i].match_material = mi != materialIndex.end() ? mi->second : -2;
        }
        d.n_bssrdfs = (int)flat->bssrdfs.size(); d.bssrdfs = flat->bssrdfs.data(); d.material_bssrdf = flat->materialBssrdf.data();
        d.n_bssrdf_floats = (int64_t)flat->bssrdfTables.size(); d.bssrdf_tables = flat->bssrdfTables.data();
    }
    d.tri_medium_inside = anyMedium ? flat->triMedIn.data() : nullptr; d.tri_medium_outside = anyMedium ? flat->triMedOut.data() : nullptr;
}

void FillRenderDesc(const Camera &camera, const Sampler &sampler, const Bounds2i &pixelBounds, int maxDepth, Float rrThreshold, bool volumetric,
                    int cameraMedium, PgRenderDesc *rd) {
    memset(rd, 0, sizeof(*rd));
    rd->abi_version = PG_ABI_VERSION;
    rd->integrator = volumetric ? 1 : 0;
    rd->camera_medium = cameraMedium;
    const ProjectiveCamera *pc = dynamic_cast<const ProjectiveCamera *>(&camera);
    if (!pc) Unsupported("a camera other than perspective / orthographic in this binding");
    CopyMatrix(pc->RasterToCamera.m, rd->raster_to_camera);
    CopyMatrix(camera.CameraToWorld.startTransform->m, rd->camera_to_world);
    if (camera.CameraToWorld.actuallyAnimated) {  // the reference's own decomposition (AnimatedTransform's constructor, transform.cpp:396-411)
        const AnimatedTransform &a = camera.CameraToWorld;
        rd->camera_animated = 1;
        rd->camera_time[0] = a.startTime; rd->camera_time[1] = a.endTime;
        CopyMatrix(a.endTransform->m, rd->camera_to_world_end);
        for (int k = 0; k < 2; ++k) {
            rd->camera_T[k][0] = a.T[k].x; rd->camera_T[k][1] = a.T[k].y; rd->camera_T[k][2] = a.T[k].z;
            rd->camera_R[k][0] = a.R[k].v.x; rd->camera_R[k][1] = a.R[k].v.y; rd->camera_R[k][2] = a.R[k].v.z; rd->camera_R[k][3] = a.R[k].w;
            for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) rd->camera_S[k][3 * i + j] = a.S[k].m[i][j];
        }
    }
    Vector3f dx, dy;
    if (const PerspectiveCamera *p = dynamic_cast<const PerspectiveCamera *>(pc)) { rd->camera_type = 0; dx = p->dxCamera; dy = p->dyCamera; }
    else if (const OrthographicCamera *o = dynamic_cast<const OrthographicCamera *>(pc)) { rd->camera_type = 1; dx = o->dxCamera; dy = o->dyCamera; }
    else Unsupported("a projective camera other than perspective / orthographic");
    rd->dx_camera[0] = dx.x; rd->dx_camera[1] = dx.y; rd->dx_camera[2] = dx.z;
    rd->dy_camera[0] = dy.x; rd->dy_camera[1] = dy.y; rd->dy_camera[2] = dy.z;
    rd->lens_radius = pc->lensRadius; rd->focal_distance = pc->focalDistance;
    rd->shutter_open = camera.shutterOpen; rd->shutter_close = camera.shutterClose;
    const Film &film = *camera.film;
    rd->full_res[0] = film.fullResolution.x; rd->full_res[1] = film.fullResolution.y;
    const Bounds2i &cb = film.croppedPixelBounds;
    rd->cropped_pixel_bounds[0] = cb.pMin.x; rd->cropped_pixel_bounds[1] = cb.pMin.y; rd->cropped_pixel_bounds[2] = cb.pMax.x; rd->cropped_pixel_bounds[3] = cb.pMax.y;
    const Bounds2i sb = film.GetSampleBounds();
    rd->sample_bounds[0] = sb.pMin.x; rd->sample_bounds[1] = sb.pMin.y; rd->sample_bounds[2] = sb.pMax.x; rd->sample_bounds[3] = sb.pMax.y;
    const Vector2f radius = film.filter->radius;
    rd->filter_radius[0] = radius.x; rd->filter_radius[1] = radius.y;
    const bool box = dynamic_cast<const BoxFilter *>(film.filter.get()) != nullptr;
    rd->filter_general = !(box && radius.x <= 0.5f && radius.y <= 0.5f && radius.x > 0 && radius.y > 0);
    if (rd->filter_general) {  // the extent of a 16x16 tile's FilmTile beyond the tile (Film::GetFilmTile, film.cpp:95-106)
        rd->tile_halo[0] = -(int)std::ceil(-0.5f - radius.x); rd->tile_halo[1] = -(int)std::ceil(-0.5f - radius.y);
        rd->tile_halo[2] = (int)std::floor(-0.5f + radius.x) + 1; rd->tile_halo[3] = (int)std::floor(-0.5f + radius.y) + 1;
        rd->tile_pixels = (16 + rd->tile_halo[0] + rd->tile_halo[2]) * (16 + rd->tile_halo[1] + rd->tile_halo[3]);
    } else rd->tile_pixels = 256;
    static_assert(Film::filterTableWidth == 16, "Film::filterTable is 16x16");
    memcpy(rd->filter_table, film.filterTable, sizeof(rd->filter_table));
    rd->film_scale = film.scale; rd->max_sample_luminance = film.maxSampleLuminance;
    rd->spp = (int)sampler.samplesPerPixel;
    if (const HaltonSampler *h = dynamic_cast<const HaltonSampler *>(&sampler)) {
        rd->sampler = 0;
        for (int i = 0; i < 2; ++i) { rd->base_scales[i] = h->baseScales[i]; rd->base_exponents[i] = h->baseExponents[i]; rd->mult_inverse[i] = h->multInverse[i]; }
        rd->sample_stride = h->sampleStride; rd->sample_at_pixel_center = h->sampleAtPixelCenter;
    } else if (const SobolSampler *s = dynamic_cast<const SobolSampler *>(&sampler)) {
        rd->sampler = 1; rd->sobol_resolution = s->resolution; rd->sobol_log2_resolution = s->log2Resolution;
    } else if (dynamic_cast<const RandomSampler *>(&sampler)) rd->sampler = PG_SAMPLER_RANDOM;
    else if (const PixelSampler *ps = dynamic_cast<const PixelSampler *>(&sampler)) {  // its per-tile clones are seeded by the device, tile by tile
        rd->sampler_dims = (int)ps->samples1D.size();
        if (const StratifiedSampler *st = dynamic_cast<const StratifiedSampler *>(ps)) {
            rd->sampler = PG_SAMPLER_STRATIFIED; rd->strat_samples[0] = st->xPixelSamples; rd->strat_samples[1] = st->yPixelSamples; rd->strat_jitter = st->jitterSamples;
        } else if (dynamic_cast<const ZeroTwoSequenceSampler *>(ps)) rd->sampler = PG_SAMPLER_ZEROTWO;
        else if (dynamic_cast<const MaxMinDistSampler *>(ps)) rd->sampler = PG_SAMPLER_MAXMINDIST;
        else Unsupported("an unknown PixelSampler");
    } else Unsupported("a sampler outside halton / sobol / random / stratified / 02sequence / maxmindist");
    rd->max_depth = maxDepth; rd->rr_threshold = rrThreshold;
    rd->pixel_bounds[0] = pixelBounds.pMin.x; rd->pixel_bounds[1] = pixelBounds.pMin.y; rd->pixel_bounds[2] = pixelBounds.pMax.x; rd->pixel_bounds[3] = pixelBounds.pMax.y;
    // a box-filter frame in which a film position can round up onto the next pixel takes the gathering film path (include/pbrt_gpu.h)
    if (!rd->filter_general && pgh_box_filter_needs_gather(rd)) {
        rd->filter_general = 1;
        rd->tile_halo[0] = -(int)std::ceil(-0.5f - radius.x); rd->tile_halo[1] = -(int)std::ceil(-0.5f - radius.y);
        rd->tile_halo[2] = (int)std::floor(-0.5f + radius.x) + 1; rd->tile_halo[3] = (int)std::floor(-0.5f + radius.y) + 1;
        rd->tile_pixels = (16 + rd->tile_halo[0] + rd->tile_halo[2]) * (16 + rd->tile_halo[1] + rd->tile_halo[3]);
    }
    rd->tile_first = 0; rd->tile_step = 1;
}

// The device's per-tile film blocks back into the reference's Film: one FilmTile per 16x16 tile, filled with the sums
// FilmTile::AddSample would have left in it (film.h:121-161), merged by Film::MergeFilmTile (film.cpp:117-130).
void MergeIntoFilm(Film *film, const PgRenderDesc &rd, const std::vector<PgFilmPixel> &px, const PgStraySample *strays, int nStrays) {
    const int sx0 = rd.sample_bounds[0], sy0 = rd.sample_bounds[1];
    const int nTilesX = (rd.sample_bounds[2] - sx0 + 15) / 16, nTilesY = (rd.sample_bounds[3] - sy0 + 15) / 16;
    const int hx = rd.filter_general ? rd.tile_halo[0] : 0, hy = rd.filter_general ? rd.tile_halo[1] : 0;
    const int tw = rd.filter_general ? 16 + rd.tile_halo[0] + rd.tile_halo[2] : 16;
    for (int t = 0; t < nTilesX * nTilesY; ++t) {
        const int x0 = sx0 + (t % nTilesX) * 16, y0 = sy0 + (t / nTilesX) * 16;
        const int x1 = std::min(x0 + 16, rd.sample_bounds[2]), y1 = std::min(y0 + 16, rd.sample_bounds[3]);
        std::unique_ptr<FilmTile> tile = film->GetFilmTile(Bounds2i(Point2i(x0, y0), Point2i(x1, y1)));
        const Bounds2i tb = tile->GetPixelBounds();
        for (Point2i p : tb) {
            const int bx = p.x - (x0 - hx), by = p.y - (y0 - hy);
            if (bx < 0 || by < 0 || bx >= tw || by >= rd.tile_pixels / tw) continue;  // box filter: the FilmTile's one-pixel rim only ever receives strays
            if (!rd.filter_general && (p.x < x0 || p.x >= x1 || p.y < y0 || p.y >= y1)) continue;
            const PgFilmPixel &fp = px[(size_t)t * rd.tile_pixels + (size_t)by * tw + bx];
            FilmTilePixel &tp = tile->GetPixel(p);
            tp.contribSum = Spectrum::FromRGB(fp.rgb);
            tp.filterWeightSum = fp.weight;
        }
        for (int i = 0; i < nStrays; ++i) {  // samples of this tile that also cover a neighbouring pixel (film.h:127-132)
            const PgStraySample &s = strays[i];
            if (s.src_px < x0 || s.src_px >= x1 || s.src_py < y0 || s.src_py >= y1 || !InsideExclusive(Point2i(s.px, s.py), tb)) continue;
            FilmTilePixel &tp = tile->GetPixel(Point2i(s.px, s.py));
            tp.contribSum += Spectrum::FromRGB(s.rgb);
            tp.filterWeightSum += s.weight;
        }
        film->MergeFilmTile(std::move(tile));
    }
}

// The device's counters under the reference's own statistics: the counters Scene::Intersect[P], SamplerIntegrator::Render and the integrators' Li
// keep (scene.cpp:40-42, integrator.cpp:48, path.cpp:45-46, volpath.cpp:45-47) are registered here once more under the SAME titles --
// StatsAccumulator adds what carries one title (stats.h:62-118) -- and set from PgCounters, so that the statistics this binary prints after a
// render are the ones the CPU integrator would have printed.  (Set on the thread that renders: pbrtWorldEnd reports its thread's statistics.)
STAT_COUNTER("Intersections/Regular ray intersection tests", devIntersectionTests);
STAT_COUNTER("Intersections/Shadow ray intersection tests", devShadowTests);
STAT_COUNTER("Integrator/Camera rays traced", devCameraRays);
STAT_PERCENT("Integrator/Zero-radiance paths", devZeroRadiancePaths, devTotalPaths);
STAT_INT_DISTRIBUTION("Integrator/Path length", devPathLength);
STAT_COUNTER("Integrator/Volume interactions", devVolumeInteractions);
STAT_COUNTER("Integrator/Surface interactions", devSurfaceInteractions);
static void ReportDeviceStats(const PgCounters &c) {
    devIntersectionTests += (int64_t)c.closest_rays; devShadowTests += (int64_t)c.shadow_rays; devCameraRays += (int64_t)c.camera_rays;
    devZeroRadiancePaths += (int64_t)c.paths_zero_radiance; devTotalPaths += (int64_t)c.paths_total;
    devVolumeInteractions += (int64_t)c.volume_interactions; devSurfaceInteractions += (int64_t)c.surface_interactions;
    if (c.path_length_count > 0) {  // ReportValue's four accumulators (stats.h:340-346), all of the frame's paths at once
        devPathLengthsum += (int64_t)c.path_length_sum; devPathLengthcount += (int64_t)c.path_length_count;
        devPathLengthmin = std::min(devPathLengthmin, (int64_t)c.path_length_min); devPathLengthmax = std::max(devPathLengthmax, (int64_t)c.path_length_max);
    }
}

void RenderOnDevice(const Scene &scene, const Camera &camera, const Sampler &sampler, const Bounds2i &pixelBounds, int maxDepth, Float rrThreshold,
                    const std::string &strategy, bool volumetric) {
    static Abi abi;
    Flat flat;
    FlattenScene(scene, maxDepth, volumetric, strategy, sampler, &flat);
    // Camera::medium: one of the media already interned through the primitives (the same Medium object), or a new one
    const int cameraMedium = InternMedium(camera.medium, &flat);
    SetMediaTables(&flat);
    PgRenderDesc rd;
    FillRenderDesc(camera, sampler, pixelBounds, maxDepth, rrThreshold, volumetric, cameraMedium, &rd);
    const int device = getenv("PBRT_GPU_DEVICE") ? atoi(getenv("PBRT_GPU_DEVICE")) : 0;
    if (abi.set_device(device) != PG_OK) { Error("pg_set_device: %s", abi.last_error()); Die(); }
    PgScene *dev = nullptr;
    if (abi.scene_create(&flat.desc, &dev) != PG_OK) { Error("pg_scene_create: %s", abi.last_error()); Die(); }
    const int nTiles = abi.render_tile_count(&rd);
    std::vector<PgFilmPixel> film((size_t)nTiles * (size_t)rd.tile_pixels);
    std::vector<PgStraySample> strays((size_t)nTiles * (rd.sampler == PG_SAMPLER_MAXMINDIST ? 1024 : 32) + 1024);
    int32_t nStrays = 0;
    if (abi.render(dev, &rd, film.data(), strays.data(), (int32_t)strays.size(), &nStrays, PG_MEM_HOST, nullptr) != PG_OK) {
        Error("pg_render: %s", abi.last_error());
        Die();
    }
    PgCounters c;
    if (abi.counters(dev, &c) == PG_OK) ReportDeviceStats(c);
    if (abi.counters(dev, &c) == PG_OK)
        fprintf(stderr, "gpupath binding: %llu camera rays, %llu regular + %llu shadow ray intersection tests, %.1f ms on the device\n",
                (unsigned long long)c.camera_rays, (unsigned long long)c.closest_rays, (unsigned long long)c.shadow_rays, c.render_ms);
    abi.scene_destroy(dev);
    MergeIntoFilm(camera.film, rd, film, strays.data(), nStrays);
    camera.film->WriteImage();  // integrator.cpp:338, unchanged
}

// Integrator "path": a PathIntegrator whose Render() goes to the device; Li() is never called.
class GpuPathIntegrator : public PathIntegrator {
  public:
    GpuPathIntegrator(const PathIntegrator &host, std::shared_ptr<const Camera> camera, std::shared_ptr<Sampler> sampler)
        : PathIntegrator(host.maxDepth, camera, sampler, host.pixelBounds, host.rrThreshold, host.lightSampleStrategy), cam(camera), smp(sampler) {}
    void Render(const Scene &scene) override { RenderOnDevice(scene, *cam, *smp, pixelBounds, maxDepth, rrThreshold, lightSampleStrategy, false); }
  private:
    std::shared_ptr<const Camera> cam;
    std::shared_ptr<Sampler> smp;
};
class GpuVolPathIntegrator : public VolPathIntegrator {
  public:
    GpuVolPathIntegrator(const VolPathIntegrator &host, std::shared_ptr<const Camera> camera, std::shared_ptr<Sampler> sampler)
        : VolPathIntegrator(host.maxDepth, camera, sampler, host.pixelBounds, host.rrThreshold, host.lightSampleStrategy), cam(camera), smp(sampler) {}
    void Render(const Scene &scene) override { RenderOnDevice(scene, *cam, *smp, pixelBounds, maxDepth, rrThreshold, lightSampleStrategy, true); }
  private:
    std::shared_ptr<const Camera> cam;
    std::shared_ptr<Sampler> smp;
};
}  // namespace

// What the re-compiled api.cpp calls in place of CreatePathIntegrator / CreateVolPathIntegrator (api.cpp:1682, :1693): the
// reference's own factories read the parameters, the device-backed subclass takes them over.
PathIntegrator *GpuBind_CreatePathIntegrator(const ParamSet &params, std::shared_ptr<Sampler> sampler, std::shared_ptr<const Camera> camera) {
    std::unique_ptr<PathIntegrator> host(CreatePathIntegrator(params, sampler, camera));
    return new GpuPathIntegrator(*host, camera, sampler);
}
VolPathIntegrator *GpuBind_CreateVolPathIntegrator(const ParamSet &params, std::shared_ptr<Sampler> sampler, std::shared_ptr<const Camera> camera) {
    std::unique_ptr<VolPathIntegrator> host(CreateVolPathIntegrator(params, sampler, camera));
    return new GpuVolPathIntegrator(*host, camera, sampler);
}
}  // namespace pbrt
