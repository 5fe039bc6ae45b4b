// pg_kernels.h -- shared declarations between the HIP kernels (pg_kernels.hip)
// and the C-ABI host code (pg_abi.hip).
#ifndef PG_KERNELS_H
#define PG_KERNELS_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include "../../include/pbrt_gpu.h"

// Per-triangle flag bits stored in tris[3*i].w, on top of PG_TRI_*.
#ifndef PG_TRI_STRIDE
#define PG_TRI_STRIDE 4
#endif
#define PG_TRI_BOGUS 0x100u  // Triangle::Intersect rejects every hit (triangle.cpp:309-317)

// An alpha / shadow-alpha mask that is a constant or a plain float image map under a (u, v) mapping -- what cut-out foliage uses --
// flattened for k_trace: without ray differentials both of MIPMap::Lookup's filters reduce to the bilinear `triangle` lookup on
// level 0 (mipmap.h:245-251 with width 0; :263 minorLength == 0), so the traversal kernel evaluates four texels inline instead
// of calling the general texture evaluator (whose registers and scratch every ray of the kernel would pay for).
struct DAlphaTex {
    float su, sv, du, dv;  // UVMapping2D
    int width, height;     // level 0 of the MIPMap<Float>
    int wrap;              // ImageWrap: 0 repeat, 1 black, 2 clamp
    int image;             // -1: the constant `constant`; -2: no such mask; >= 0: image map (texels at `offset`, in floats)
    long long offset;
    float constant;
    int pad;
};
// One object definition (instancing) on the device: its BVHAccel's root, or its lone primitive when nNodes == 0.
struct DObject { float box[6]; int rootRef; int firstPrim; int nNodes; int pad; };
// What k_trace needs to enter a (still) instance, in one 96-byte record: rows 0 - 2 of WorldToInstance, its object's root box and references.
// affine: row 3 of the matrix is (0, 0, 0, 1) -- the kernel then forms w from those constants (the same operations, no loads); a moving
// instance or a projective matrix goes through PgInstance as before.
struct DInstEntry { float w2i[12]; float box[6]; int rootRef; int firstPrim; int nNodes; int affineStill; int pad[2]; };
#define TR_NO_ROOT 0x7fffffff

// One tile's sampler when the sampler draws from one RNG stream per tile (PgSamplerKind 2 .. 5): the tile's PCG32 state
// (core/rng.h:61-144), PixelSampler's current1DDimension / current2DDimension / currentPixelSampleIndex (sampler.cpp:100-134),
// the pixel the tile is at, and the camera sample's lens point (kept for the ray differentials at the first hit).
struct TileSamplerState {
    unsigned long long state, inc;
    int cur1D, cur2D, sampleIndex, active;  // active: the current pixel exists and lies inside the integrator's pixel bounds
    float lens0, lens1;
    int px, py;
    unsigned int draws;  // RNG::UniformUInt32 calls so far (k_ts_start_tile measures what a StartPixel consumed)
    float time;          // the camera sample's time number (kept like the lens point: a moving camera's differentials need its transform again)
};

// Device-resident scene.  All pointers are device memory.
struct DScene {
    // Linearised BVH, 32 B/node exactly as PgBVHNode: two float4 per node
    //   n[0] = (bmin.x, bmin.y, bmin.z, bmax.x)   n[1] = (bmax.y, bmax.z, offset, nprims|axis<<16)
    const float4 *nodes;
    // Child-pair records (pg_traverse.hip), 64 B per INTERIOR node, depth-first order:
    //   w[0] = (c0.lo.x, c0.hi.x, c1.lo.x, c1.hi.x)  w[1] = same for y  w[2] = same for z
    //   w[3] = (ref0, ref1, split axis, 0)   c0 = first child (index+1), c1 = second child (offset)
    // ref >= 0: record index of an interior child; ref < 0: leaf child, ~ref = firstPrim << leafBits | (nPrims-1)
    const float4 *wnodes;
    float rootBox[6];  // nodes[0].bounds: (min.xyz, max.xyz)
    int rootRef;       // ref of nodes[0]
    int leafBits;
    // Triangles in BVH order, 48 B each: three float4
    //   t[0] = (p0, flags)  t[1] = (p1, material)  t[2] = (p2, light)
    // stored PG_TRI_STRIDE float4 apart: 4 puts every record into one 64-B line (of 48-B records packed back to back half
    // straddle two lines, and the traversal is bound by lines fetched)
    const float4 *tris;
    // per-vertex shading normals N and (u, v) de-indexed per triangle, BVH order, in ONE 64-byte record (one line where two arrays
    // -- 48 B and 24 B per triangle, records straddling lines -- were 2.6 lines per hit: profiles/r06l_*):
    //   sixteen floats: n0.xyz n1.xyz n2.xyz 0 | u0 v0 u1 v1 u2 v2   (the (u, v) alone -- k_trace's alpha masks -- are 24 contiguous bytes)
    // nullptr when no mesh has either; attrN / attrUV: some mesh has normals / uv (a triangle's own PG_TRI_HAS_N / _UV bit says whether
    // it has; default uv, triangle.h:104-106).  triS: per-vertex tangents S (3 float4 per triangle), or nullptr when no mesh has them
    const float4 *triAttr, *triS;
    int attrN, attrUV;
    const float *alphaUV;  // the (u, v) once more, 6 floats per triangle, for k_trace's alpha masks (scenes with alpha-masked meshes), else nullptr
    const PgMaterial *materials;
    const PgLight *lights;
    // what the shading kernel needs of a light the moment it is chosen, 80 B per light, read in ONE round trip: h[0] = (type,
    // prim, two_sided, area), h[1] = (L.rgb, 0), h[2..4] = an area light's primitive record (tris[3 prim ..]) -- instead of
    // PgLight fields, then the emitter's index, then its vertices, each a dependent access
    const float4 *lightHot;
    int nNodes, nTris, nLights, nMaterials;
    const PgSphere *spheres;  // Shape "sphere" primitives: tris[3*k] = (sphere index, 0, 0, flags | PG_PRIM_SPHERE)
    int nSpheres;
    const PgInstance *instances;  // TransformedPrimitives: tris[3*k] = (instance index, 0, 0, PG_PRIM_INSTANCE); inside an object definition only with hasNest
    const DObject *objects;
    const DInstEntry *instEntry;  // one per instance (see DInstEntry)
    int nInstances;
    int *hitInst;  // per closest-hit result: the instance the hit primitive was reached through, or -1 (written by k_trace<.., true>)
    // Moving shapes / instances (PgInstance::animated; TransformedPrimitive over an AnimatedTransform): hasMotion says the scene has one -- every
    // ray queue then carries its rays' times (RayQueue::time) and k_trace runs its XP_ANIM instantiation, which interpolates the instance's
    // transform at the ray's time when it enters one and leaves, per closest-hit result on a moving instance, the interpolated matrices for the
    // shading kernels: animXf[PG_XF_STRIDE * i] = InterpolatedPrimToWorld, + 16 its inverse, + 32 IsIdentity (pg_motion.h)
    int hasMotion;
    int rayTimes;  // the render's queues carry their rays' times (queue_times); 0 in the unit entry points' copy: their rays have time 0
    float *animXf;
    // ABI 29 -- a moving shape inside an object definition (api.cpp:1405-1418): a TransformedPrimitive among an instance's primitives, i.e. a hit can
    // lie under TWO transforms.  hasNest says the scene has one: k_trace runs its XP_NEST instantiation (a second saved traversal context), hitInst then
    // holds outer + nInstances * (inner + 1) (PG_NEST_OUTER / PG_NEST_INNER below; inner = -1: a hit one level deep) and the inner instance's interpolated
    // matrices wait at animXf[PG_XF_STRIDE * (nestXfOff + i)] -- the buffer has two halves, every pointer offset applied to it moves both alike.
    // Shading runs in MODE 2 (pg_shade_mode), the one kernel family that carries the second transform.
    int hasNest;
    int nestXfOff;
    const PgAlphaMask *alphas;           // alpha / shadow-alpha textures of meshes; triAlpha[k] indexes it for PG_TRI_ALPHA triangles
    const int *triAlpha;
    int hasAlpha;
    // alphaTex[2 k] / [2 k + 1] = the alpha / shadow-alpha mask of alphas[k] in DAlphaTex form, or nullptr when some mask of the
    // scene is a texture DAlphaTex cannot express (k_trace then evaluates masks through TexEval)
    const DAlphaTex *alphaTex;
    const PgImage *images;               // MIPMaps of the image textures (pyramid levels in texels[])
    const float *texels;
    const float *ewaLut;                 // MIPMap::weightLut (128)
    const float *envTables;              // the infinite lights' Distribution2D tables (PgLight.env_table)
    const PgTexture *textures;           // texture nodes
    const int *noisePerm;                // NoisePerm (512 entries) of the Perlin-noise textures, or nullptr
    const PgTexturedMaterial *textured;  // materials evaluated per hit (PG_MAT_TEXTURED)
    int hasTextured;
    // This structure's scene tables once more, in device memory: what the kernels hand to the functions that are CALLED (the texture
    // and material evaluators, pg_texture.h) -- a reference to a kernel's own by-value DScene would make every lane write the whole
    // structure to its scratch memory first (688 B per lane per launch: it was most of k_shade<2>'s write traffic).  Only the
    // tables set by pg_scene_create are valid in it (not hitInst, ts*, which later calls set in the host's copy).
    const DScene *self;
    const unsigned char *primClass;      // per primitive record: the shading class of its material (k_shade_order), or nullptr
    const PgDensityGrid *grids;          // GridDensityMedium (ABI 23): mediaGrid[m] = index of medium m's grid, -1 = homogeneous
    const int *mediaGrid;
    const float *gridDensity;
    int nGrids;
    const PgMedium *media;               // HomogeneousMedium table; triMediumIn/Out[k] = the primitive's MediumInterface (-1 = none), or nullptr
    const int *triMediumIn, *triMediumOut;
    const PgBxDF *bxdfs;      // the materials' BxDF lists (PgMaterial.first_bxdf / n_bxdfs)
    // the same lists as PkLobe records for k_shade<3> (scenes whose textured materials are evaluated ahead, MatPre): material m's list starts
    // at record matPk[m].x and takes matPk[m].y (1 or 2: its lobes carry ScaledBxDF factors) records per lobe; nullptr otherwise
    const float4 *bxdfsPk;
    const int2 *matPk;
    // subsurface scattering (ABI 24): materialBssrdf[m] >= 0 = ComputeScatteringFunctions of material m also sets si->bssrdf
    const PgBSSRDF *bssrdfs;
    const int *materialBssrdf;
    const float *bssrdfTables;
    int nBssrdfs;
    int ext;          // the EXT shading kernels are needed: spheres, infinite lights or PG_MAT_LOBES materials (or PG_FORCE_EXT=1)
    int hasInfinite;  // some light is an InfiniteAreaLight (Scene::infiniteLights non-empty)
    // light sampling distributions (lightdistrib.cpp): strategy + tables
    int lightStrategy;
    int nVoxels[3];
    float bmin[3], bmax[3];
    const float *distTable;  // per distribution: func[nl], cdf[nl+1], funcInt  (stride 2*nl+2)
    // Sparse "spatial" tables (SpatialLightDistribution fills its voxels on first touch, lightdistrib.cpp:135-230): voxelSlot[v]
    // >= 0: the voxel's distribution is distTable + slot * stride; -1: nobody asked yet; -2: requested.  A shading lane that
    // meets a missing voxel appends it to voxelRequests (counter voxelCounters[0]) and its own queue index to the retry list
    // (RenderParams.retryList, counter voxelCounters[1]); the host computes the requested voxels and re-runs those lanes.
    int sparseLights;
    int *voxelSlot, *voxelRequests, *voxelCounters;
    // Halton tables
    const uint16_t *perms;
    const int32_t *permSums;
    const int32_t *primes;
    int nPermDims;
    // per Halton dimension two int4: (base, offset of its permutation in perms, m, L) -- floor(a / base) = (t + ((a - t) >> 1))
    // >> (L - 1) with t = mulhi(m, a), exact for every 32-bit a: the digit loops divide by multiplying -- and, as float bits,
    // (1 / base, (1 / base) * perm[0] / (1 - 1 / base), 0, 0): ScrambledRadicalInverse's two constants (pg_abi.hip builds the table)
    const int4 *haltonDims;
    // SobolSampler tables (core/sobolmatrices.h:49-52), nullptr unless the scene was created with them
    const uint32_t *sobolMatrices;
    const uint64_t *vdcSobol, *vdcSobolInv;
    // Tile-serial samplers: set by pg_render for the duration of a frame.  One path per tile is in flight and slot = the
    // tile's local index; ts1 / ts2 hold the current pixel's sample arrays, per tile [tsDims][tsSpp] floats / float pairs.
    TileSamplerState *ts;
    float *ts1, *ts2;
    int tsDims, tsSpp;
    // tsBatched: the PixelSamplers when "dimensions" covers every draw a path can make -- no path touches its tile's stream after
    // StartPixel, so the sample arrays of ALL pixels are generated ahead (ts1 / ts2 indexed by pixel = local tile * 256 + pixel in tile
    // instead of by tile) and the paths run as an ordinary wavefront; tsOverflow: set by a draw beyond the arrays (must never happen)
    int tsBatched;
    int *tsOverflow;
    const uint32_t *cmaxmin;  // CMaxMinDist [17][32]
};

// Triangle `prim`'s (u, v) / per-vertex normals from its attribute record (DScene::triAttr)
__device__ __forceinline__ void tri_attr_uv(const DScene &sc, int prim, float uv[6]) {
    const float4 *a = sc.triAttr + 4 * (size_t)prim;
    const float2 q = *reinterpret_cast<const float2 *>(reinterpret_cast<const float *>(a) + 10);
    const float4 r = a[3];
    uv[0] = q.x; uv[1] = q.y; uv[2] = r.x; uv[3] = r.y; uv[4] = r.z; uv[5] = r.w;
}
__device__ __forceinline__ void tri_attr_normals(const DScene &sc, int prim, float n[9]) {
    const float4 *a = sc.triAttr + 4 * (size_t)prim;
    const float4 q0 = a[0], q1 = a[1];
    n[0] = q0.x; n[1] = q0.y; n[2] = q0.z; n[3] = q0.w; n[4] = q1.x; n[5] = q1.y; n[6] = q1.z; n[7] = q1.w; n[8] = reinterpret_cast<const float *>(a)[8];
}

// A queue of rays in SoA float4 pairs: 32 B per ray
//   o[i] = (o.x, o.y, o.z, tMax)   d[i] = (d.x, d.y, d.z, slot id bits)
// The queue is PG_REGIONS sub-queues ("regions"), one per XCD: region r owns entries [r*regionCap, r*regionCap + count(r))
// and its own append counter on its own 128-B line.  A producer block b appends to region b % 8 (= the XCD it runs on)
// with ONE atomic per block, so no counter sees more than 1/8 of the blocks; consumers walk the regions the same way.
#define PG_REGIONS 8
#define PG_COUNT_STRIDE 32  // ints between two region counters (128 B)
#define PG_XF_STRIDE 36  // floats per entry of DScene::animXf: matrix, inverse, IsIdentity (+ 3 of padding: 16-byte aligned entries)
struct RayQueue {
    float4 *o;
    float4 *d;
    int *counts;    // PG_REGIONS counters, PG_COUNT_STRIDE ints apart
    int regionCap;  // entries per region (multiple of 256)
};
__host__ __device__ inline int queue_region_count(const RayQueue &q, int r) { return q.counts[r * PG_COUNT_STRIDE]; }
// Ray::time per entry, in scenes with moving shapes / instances only: one float per entry BEHIND the queue's `d` array (the buffers are allocated
// with that room, pg_abi.hip) -- not a member of RayQueue: four more pointers among the kernel arguments cost the still scenes' shading kernels
// 30 - 60 spilled scalar registers each
#define PG_QUEUE_TIMES(sc, q) ((sc).rayTimes ? reinterpret_cast<float *>((q).d + (size_t)(q).regionCap * PG_REGIONS) : nullptr)

// Path state in QUEUE order (PathIntegrator only): entry i belongs to ray i of the main queue it accompanies, so the shading
// kernel reads it with the ray (one coalesced, independent load) instead of gathering it by slot behind the ray's own load.
struct QueueState {
    float4 *L;     // (L.rgb, pFilm.x)
    float4 *beta;  // (beta.rgb, pFilm.y)
    int4 *meta;    // as PathState::meta
    int *medium;   // volpath: the ray's medium (index + 1, 0 = none), as VolState::medium; nullptr under PathIntegrator
};

// Per-path state, indexed by slot.
struct PathState {
    float4 *L;      // (L.rgb, pFilm.x)
    float4 *beta;   // (beta.rgb, pFilm.y)
    int4 *meta;     // (haltonIndexLo, haltonIndexHi, etaScale bits, dimension<<20 | flags | bounces)
    // pending direct-lighting estimate of the current bounce (EstimateDirect, integrator.cpp:108-215)
    float4 *pdLight;  // (f*Li*weight/lightPdf rgb, light-selection pdf)
    float4 *pdMis;    // (f*|wi.n| rgb, scatteringPdf)
    float4 *pdBeta;   // (beta before the bounce rgb, MIS weight)
    int4 *pdInfo;     // (shadow queue pos or -1, mis queue pos or -1, lightNum, unused)
    // PathIntegrator (not volpath): L / beta / meta travel with the ray in queue order (qs[0] / qs[1] accompany the two main
    // queues); L[slot] then only receives a path's FINAL radiance, the pd* terms are indexed by the ray's queue position, and
    // pdInfo.w says where the path's L lives when k_resolve adds the direct lighting: >= 0 entry of the next queue's state,
    // < 0: ~slot (the path ended).  VolPathIntegrator without BSSRDF materials or grid media: the same, with the ray's medium as a
    // fourth array, the pending terms of VolState by the ray's queue position too, and the transmittance rays carrying that position
    // instead of the slot.  qs[0].L == nullptr: everything by slot (volpath scenes with BSSRDF materials or a GridDensityMedium).
    QueueState qs[2];
};

// Extra per-path state of the VolPathIntegrator (integrators/volpath.cpp), indexed by slot.  A "through" ray is a ray of
// VisibilityTester::Tr (light.cpp:63-81, kind 0) or Scene::IntersectTr (scene.cpp:57-70, kind 1): it is re-traced through
// surfaces without a material until it is blocked, arrives or escapes, accumulating the media's transmittance.
struct VolState {
    int *medium;        // medium (index + 1, 0 = none) of the path's current ray
    float4 *trAcc[2];   // per kind: (Tr so far rgb, the through ray's medium as int bits)
    float4 *p1[3];      // kind 0: the light sample the ray is heading to (p, pError, n); queue-order state: p1[0].w = the light sample's MIS weight (-1: delta light)
    float4 *misLi;      // kind 1, once finished: radiance of the sampled light along the ray (rgb)
    float4 *pdLi;       // (Li rgb of the light sample, lightPdf)
};

// The BSSRDF branch of PathIntegrator::Li / VolPathIntegrator::Li (path.cpp:152-174, volpath.cpp:151-176) between the vertex po where
// the path enters the medium and the exit vertex pi that SeparableBSSRDF::Sample_Sp finds (bssrdf.cpp:253-328): indexed by slot.
// The probe segment of Sample_Sp is a chain of closest-hit queries that keeps every hit on po's material and then picks one
// uniformly: the chain is walked twice through k_trace -- once to count, once up to the chosen hit.
struct SssState {
    float4 *po;        // (po.p, u1 as Sample_Sp leaves it for the choice among the hits)
    float4 *frame[3];  // (ns, eta) (ss, BSSRDF index as int bits) (ts, SeparableBSSRDF::material as int bits): po's shading frame
    float4 *coef[2];   // TabulatedBSSRDF::sigma_t, ::rho at po (per hit for textured materials)
    float4 *target;    // (pTarget, 0): the far end of the probe segment
    int2 *count;       // (hits on the material found by the first walk, hits on it seen so far by the second)
    float4 *hit;       // the chosen hit's record as k_trace wrote it, the probe ray that found it (o, d) and the instance it was in
    float4 *hitO, *hitD;
    int *hitInst;
    float *hitXf;      // scenes with moving instances: the chosen hit's interpolated instance matrices (PG_XF_STRIDE floats per slot, as DScene::animXf), else nullptr
    int hitXfNest;     // DScene::hasNest: the inner TransformedPrimitive's matrices of a chosen hit two levels deep wait at hitXf[PG_XF_STRIDE * (hitXfNest + slot)]
    int2 *medium;      // volpath: (medium of the probe ray under way, medium of the probe ray that found the chosen hit); index + 1, 0 = none
    RayQueue qjob;     // the first probe ray of every path that entered this branch at the current bounce (appended by k_shade)
};

// One BxDF of a list in 48 bytes: what k_material writes for a hit on a material with textured parameters and what k_shade<3> reads --
// and, packed once at pg_scene_create, the constant lists of the other materials for the same kernel (DScene::bxdfsPk).  The members carry
// PgBxDF's names over a layout that depends on the type, so the BxDF functions (lobe_f, lobe_pdf, lobe_sample_f, lobe_fresnel: templates on
// the record type) read each field where it is used:
//   hdr = type | fresnel << 4 | n_scales << 6        R
//   T | the conductor's eta | OrenNayar's A, B       alpha_x
//   eta_a, eta_b | the conductor's k                 alpha_y
// Every field a BxDF of that type reads, nothing else.  A MixMaterial's ScaledBxDF factors (scale[0 .. 2], nine floats) take a record of their
// own BEHIND the lobe's; a hit's lobes all have one or none has.
struct PkLobe {
    uint32_t hdr;
    float R[3];
    union { float T[3]; float cond_eta[3]; struct { float on_a, on_b, on_unused; }; };
    float alpha_x;
    union { struct { float eta_a, eta_b, eta_unused; }; float cond_k[3]; };
    float alpha_y;
};
static_assert(sizeof(PkLobe) == 48, "PkLobe is three float4");
__host__ __device__ inline void pg_pack_lobe(const PgBxDF &b, float *q) {  // q[12]
    const uint32_t hdr = (uint32_t)b.type | ((uint32_t)b.fresnel << 4) | ((uint32_t)b.n_scales << 6);
    memcpy(&q[0], &hdr, 4);
    q[1] = b.R[0]; q[2] = b.R[1]; q[3] = b.R[2];
    q[7] = b.alpha_x; q[11] = b.alpha_y;
    if (b.type == PG_BXDF_OREN_NAYAR) { q[4] = b.on_a; q[5] = b.on_b; q[6] = 0.f; q[8] = b.eta_a; q[9] = b.eta_b; q[10] = 0.f; }
    else if (b.fresnel == PG_FRESNEL_CONDUCTOR) { for (int c = 0; c < 3; ++c) { q[4 + c] = b.cond_eta[c]; q[8 + c] = b.cond_k[c]; } }
    else { for (int c = 0; c < 3; ++c) q[4 + c] = b.T[c]; q[8] = b.eta_a; q[9] = b.eta_b; q[10] = 0.f; }
}
__host__ __device__ inline void pg_pack_lobe_scales(const PgBxDF &b, float *q) {  // q[12]: the record behind the lobe's
    for (int i = 0; i < PG_MAX_BXDF_SCALES; ++i) for (int c = 0; c < 3; ++c) q[3 * i + c] = b.scale[i][c];
    q[9] = q[10] = q[11] = 0.f;
}

// Materials evaluated ahead of the shading launch (k_material, pg_kernels.hip): for every main-queue entry whose hit has a
// material with textured parameters (PG_MAT_TEXTURED) and whose path is still alive, Material::ComputeScatteringFunctions' outputs
// -- the BxDF list, BSDF::eta and the shading frame Material::Bump leaves -- at the POSITION p the entry has in the launch's shading
// order (RenderParams::order: k_material and the shading kernel walk the queue in the same order, thread for thread), one PLANE of
// N = regionCap * PG_REGIONS records per record index:
//   head[p] = (shading.n, BSDF::eta)   head[N + p] = (shading.dpdu, number of BxDFs | 0x100 for a mix, as int bits)   record r of the list = lobes[(r * N + p) * 3 ..]
// Neighbouring lanes write and read neighbouring 48-B records, and a hit with one BxDF touches one plane: every line that moves is used
// whole.  (Until round 6 a hit's records lay side by side at the stride of the scene's longest list, by entry: a one-lobe hit used 48 B
// of the 240 its slot took, profiles/r06k_*.)  stride = the planes the scene's longest list needs (PgScene: counted per material kind,
// <= PG_MAX_BXDFS lobes, two records per lobe of a mix).  lobes == nullptr: the scene has no such material, or the buffers did not fit
// -- the shading kernel then evaluates materials itself (k_shade<2, .>).
struct MatPre { float4 *lobes; float4 *head; int stride; };  // lobes: packed 48-B records (LobeBsdfT, pg_kernels.hip), `stride` planes

#define PG_META_SPECULAR 0x10000
#define PG_META_DONE 0x20000
#define PG_META_HASDIFF 0x80000  // the ray still is the camera's RayDifferential (cleared by the first SpawnRay)

struct RenderParams {
    PgRenderDesc rd;
    int nTilesX, nTilesY;
    // batch description: tiles [tileLocal0, tileLocal0+nTilesBatch) of this shard, samples [s0, s0+sCount)
    int tileLocal0, nTilesBatch, s0, sCount;
    int capacity;  // slots in this batch = nTilesBatch * sCount * 256
    // sparse light tables: lanes that met a missing voxel leave their queue index in retryList; a launch with retryCount > 0
    // shades exactly those entries
    int *retryList;
    int retryCount;
    // shading order (k_shade_order): order[position] = the main-queue entry the thread at that position shades; nullptr: its own
    const int *order;
    // volpath, GlobalSamplers: k_shade_order has drawn HomogeneousMedium::Sample's two numbers for every entry whose ray is in a
    // medium -- volPre[entry] = (channel as int bits, sampled distance); the shading kernel takes them from here.  nullptr: it draws.
    const float2 *volPre;
    MatPre matPre;    // k_material's outputs for this launch's queue (lobes == nullptr: none)
    int tsGuessSkew;  // tests (PG_TS_GUESS_SKEW): added to k_ts_start_tile's first guess of a StartPixel's consumption, so that its correction passes run
};

// The shading kernels count Triangle::Intersect calls of light.Pdf_Li (the reference's nTests statistic) into PG_LIGHT_TEST_SHARDS
// counters, one per 128-B line, picked by the block index: every wave of a shading launch adds to the count, and two million
// atomics per launch on ONE word (about 88 per microsecond, MI355X_MICROARCH.md) had made k_shade wait for them for half its time.
#define PG_LIGHT_TEST_SHARDS 256
#define PG_LIGHT_TEST_STRIDE 16  // unsigned long longs between two shards (128 B)
// The integrators' own statistics (PgCounters, ABI 28) in the same shards, words 1 .. 8 of a shard's line -- one set of atomics per WAVE
// that has something to report, from wave-wide ballots / reductions: sums are added; the shortest path is kept as 65535 - length and the
// longest as length + 1 under atomicMax, so that a zeroed shard means "no path yet"
#define PG_STAT_LEN_SUM 1
#define PG_STAT_LEN_COUNT 2
#define PG_STAT_LEN_MINC 3
#define PG_STAT_LEN_MAXP 4
#define PG_STAT_PATHS 5
#define PG_STAT_PATHS_ZERO 6
#define PG_STAT_VOLUME 7
#define PG_STAT_SURFACE 8
struct TraceCounters {
    unsigned long long node_visits, tri_tests;
};

// Tunables of k_trace: LDS stack entries per lane, rays per wave segment, idle-lane count that triggers a refill.
// cullK: closest-hit far-child early-cull margin (pg_traverse.hip); exact while a ray's tMax never grows by more than
// this factor through rounding (each accepted hit can raise it by <= 3 roundings, i.e. ~5000 successive raises).
// maxAccepted: accepted hits per ray beyond which the margin's proof no longer holds (<= 4096 for cullK = 1 + 2^-10)
struct TraceConfig { int depth, segRays /* rays per chunk */, refillAt, triW; float cullK; int gridBlocks; int maxAccepted;
                     int refillAtAny, triWAny;  /* the same two for any-hit launches: their rays end at the first hit, a wave waits for more idle lanes and more leaf lanes (profiles/r03z_*) */
                     int anyhitFree;  /* any-hit rays visit the nearer child first instead of the reference's order (same answers) */
                     int refillAtInst, triWInst, refillAtAnyInst;  /* scenes with object instances: closest-hit refill threshold and triangle-step weight, any-hit refill threshold */ };
// The defaults, with the PG_TRACE_* environment overrides of experiments and tests applied.  Every scene carries its own copy
// (PgScene::trace): the exact-fallback retry of one scene must not change what another host thread's launches use.
TraceConfig default_trace_config();
// cursorInit (device, optional): q's region counters as they were before entries were appended -- only the appended entries are traced
void launch_closest(const DScene &sc, const TraceConfig &c, RayQueue q, float4 *hits, float *tOut, TraceCounters *cn, int *cursors, int *cullGuard, hipStream_t s,
                    const int *cursorInit = nullptr);
// two queues in one launch; q1's results land at hits[hitOffset1 + i]
void launch_closest2(const DScene &sc, const TraceConfig &c, RayQueue q0, RayQueue q1, float4 *hits, int hitOffset1, TraceCounters *cn, int *cursors, int *cullGuard,
                     hipStream_t s, float *tOut = nullptr, const int *cursorInit = nullptr);
void launch_anyhit(const DScene &sc, const TraceConfig &c, RayQueue q, int *occluded, TraceCounters *cn, int *cursors, hipStream_t s);
void launch_generate(const DScene &sc, const RenderParams &rp, PathState st, RayQueue q, hipStream_t s);
// tile-serial samplers: seed the tiles' streams; StartPixel for pixel (lx, ly) of every tile; the camera sample + ray of sample
// `sampleIndex` of that pixel; FilmTile::AddSample of the finished paths
void launch_ts_init(const DScene &sc, const RenderParams &rp, hipStream_t s);
void launch_ts_start_tile(const DScene &sc, const RenderParams &rp, hipStream_t s);  // tsBatched: StartPixel for every pixel of every tile, in the tile's order
void launch_ts_start_pixel(const DScene &sc, const RenderParams &rp, int lx, int ly, hipStream_t s);
void launch_ts_generate(const DScene &sc, const RenderParams &rp, PathState st, RayQueue q, int sampleIndex, hipStream_t s);
void launch_ts_film(const DScene &sc, const RenderParams &rp, PathState st, PgFilmPixel *film, PgStraySample *strays, int maxStrays, int *nStrays,
                    hipStream_t s);
// cur: which of st.qs[] accompanies qin (the other one accompanies qnext)
// Shading order for scenes with textured / BxDF-list materials: inside windows of PG_ORDER_WINDOW consecutive entries (measured: 1024 / 4096 / 16384 -> 42.8 / 37.8 / 36.6 ms per launch, profiles/r03l) of a region the
// shading threads take the entries grouped by the hit's material class (a stable counting sort of entry indices; the entries stay
// where they are), so that a shading wave evaluates one material's textures and BxDF list instead of a handful.
#ifndef PG_ORDER_WINDOW
#define PG_ORDER_WINDOW 16384
#endif
#define PG_ORDER_CLASSES 16  // 0 .. 12 material classes, 13 = scattered in a medium (volpath), 14 = the ray escaped, 15 = no entry
void launch_shade_order(const DScene &sc, RayQueue qin, const float4 *hits, int *order, hipStream_t s);
// volpath: additionally draws the medium sample of every entry (RenderParams::volPre) and puts the entries that scatter in the medium
// into a class of their own (class 13), so that a wave shades medium vertices or surface vertices, not both one after the other
void launch_shade_order_vol(const DScene &sc, const RenderParams &rp, PathState st, VolState vs, RayQueue qin, const float4 *hits, const float *hitT,
                            int *order, float2 *volPre, hipStream_t s, int cur = 0);
// Which k_shade<MODE, ...> launch_shade / launch_shade_vol run for a frame (PgCounters::shading_modes records it): 0 the baked-in BxDF shapes,
// 1 a material's BxDF list, 3 the packed lists k_material wrote ahead of the launch, 2 the evaluators inside the shading kernel -- what scenes
// with BSSRDF materials or grid media run, and what a textured scene FALLS BACK to when k_material's lists do not fit in memory.
inline int pg_shade_mode(const DScene &sc, const RenderParams &rp, bool vol, bool sss, bool gridPhase) {
    if (sc.hasNest) return 2;  // hits under two transforms (a moving shape inside an object definition): the general kernels carry the second one
    if (gridPhase || (sss && sc.nBssrdfs > 0)) return sc.hasTextured ? 2 : 1;
    if (sc.hasTextured) return rp.matPre.lobes ? 3 : 2;
    return (vol || sc.ext) ? 1 : 0;
}
void launch_shade(const DScene &sc, const RenderParams &rp, PathState st, RayQueue qin, const float4 *hits, RayQueue qnext,
                  RayQueue qshadow, RayQueue qmis, unsigned long long *lightTriTests, hipStream_t s, int cur = 0, const SssState *sss = nullptr);
// Subsurface scattering: one step of the probe chains (pass 1: count the hits on the material; pass 2: stop at the chosen one) over
// the rays of qin (results at hits[i], instances at sc.hitInst[i]), continued rays to qout; then the exit vertices of the jobs of
// sss.qjob: Pdf_Sp / Sr, direct lighting (shadow / MIS rays, pending terms at the job's queue index) and the next ray (appended to qnext)
void launch_sss_probe(const DScene &sc, SssState sss, int pass, RayQueue qin, const float4 *hits, RayQueue qout, hipStream_t s, bool vol = false, bool first = false);
void launch_sss_exit(const DScene &sc, const RenderParams &rp, PathState st, SssState sss, RayQueue qnext, RayQueue qshadow, RayQueue qmis,
                     unsigned long long *lightTriTests, hipStream_t s, int nxt, bool vol, VolState vs, int phase = 0, float4 *exitVertex = nullptr);
void launch_resolve(const DScene &sc, PathState st, RayQueue qin, RayQueue qmis, const int *occluded, const float4 *misHits, hipStream_t s,
                    int cur, unsigned long long *stats = nullptr);
// VolPathIntegrator: the shading step with medium sampling (hitT = the hits' ray parameters), one step of the through rays of
// `kind` (results of qin at hits[hitBase + i]; continued rays go to qout), and EstimateDirect's sums with transmittance
// gridVertex / phase: scenes with a grid medium shade a vertex in two launches (phase 1, transmittance rays, resolve, phase 2); 0 = one pass
void launch_shade_vol(const DScene &sc, const RenderParams &rp, PathState st, VolState vs, RayQueue qin, const float4 *hits, const float *hitT,
                      RayQueue qnext, RayQueue qshadow, RayQueue qmis, unsigned long long *lightTriTests, hipStream_t s, const SssState *sss = nullptr,
                      float4 *gridVertex = nullptr, int phase = 0, int cur = 0);
void launch_through(const DScene &sc, PathState st, VolState vs, int kind, RayQueue qin, const float4 *hits, const float *hitT, int hitBase,
                    RayQueue qout, hipStream_t s, const RenderParams *rpGrid = nullptr);
void launch_resolve_vol(const DScene &sc, PathState st, VolState vs, RayQueue qin, hipStream_t s, int cur = 0);
void launch_fill_int(int *p, int value, int n, hipStream_t s);
void launch_film(const RenderParams &rp, PathState st, PgFilmPixel *film, PgStraySample *strays, int maxStrays, int *nStrays,
                 hipStream_t s);
void launch_film_general(const RenderParams &rp, PathState st, PgFilmPixel *film, hipStream_t s);
void launch_light_tables(const DScene &sc, float *table, int nDistributions, hipStream_t s);
// the distributions of voxels requests[0 .. n) into pool slots firstSlot ..; also publishes voxelSlot[]
void launch_light_tables_sparse(const DScene &sc, float *pool, const int *requests, int n, int firstSlot, hipStream_t s);
int pgSetError(int code, const char *msg);  // pg_abi.hip: sets pg_last_error()
#endif
