// pg_abi.hip -- the C ABI of include/pbrt_gpu.h: scene upload, the wavefront
// render loop (SamplerIntegrator::Render, integrator.cpp:228-339, re-ordered
// into batches of camera samples that advance one bounce per launch), and the
// batched Scene::Intersect/IntersectP entry points.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <dlfcn.h>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>
#include "pg_device.h"
#include "pg_kernels.h"

static thread_local std::string g_lastError;
static int setError(int code, const char *fmt, ...) {
    char buf[1024];
    va_list a;
    va_start(a, fmt);
    vsnprintf(buf, sizeof(buf), fmt, a);
    va_end(a);
    g_lastError = buf;
    return code;
}
// for the other translation units of this library (pg_hlbvh.hip)
int pgSetError(int code, const char *msg) { g_lastError = msg; return code; }
#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) return setError(PG_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

struct DeviceBuffer {
    void *p = nullptr;
    size_t bytes = 0;
    hipError_t alloc(size_t n) {
        release();
        bytes = n;
        if (n == 0) return hipSuccess;
        return hipMalloc(&p, n);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
    ~DeviceBuffer() { release(); }
};

struct PgScene {
    int device = 0;
    DScene d;
    TraceConfig trace;  // k_trace's tunables for THIS scene's launches (the exact-fallback retry changes them for one call)
    DeviceBuffer nodes, wnodes, tris, spheres, bxdfs, objects, instances, instEntry, textures, textured, images, texels, ewaLut, envTables, alphas, alphaTex, triAlpha, triAttr, triS, uv, materials, lights, distTable, perms, permSums, primes, media, triMediumIn, triMediumOut, sobolMatrices, vdcSobol, vdcSobolInv, noisePerm;
    // work buffers (sized on first render, reused)
    int capacity = 0;
    DeviceBuffer sceneCopy;  // DScene::self
    DeviceBuffer tsOverflow;  // tsBatched: the flag a draw beyond the sample arrays raises
    DeviceBuffer bxdfsPk, matPk;     // the constant BxDF lists as PkLobe records + where each material's starts (k_shade<3>)
    DeviceBuffer matLobes, matHead;  // k_material: the BxDF lists and shading frames of the hits on materials with textured parameters (MatPre)
    int matStride = 0;               // the largest such list of this scene, 0: materials are evaluated inside the shading kernel
    DeviceBuffer shadeOrder, primClass, volPre;  // k_shade_order: the order buffer of the main queue, the primitives' material classes, volpath's pre-drawn medium samples
    bool volOrder = false;  // volpath launches shade medium vertices and surface vertices in separate waves (scenes with homogeneous media only)
    DeviceBuffer animXf;  // scenes with moving instances: the interpolated matrices per closest-hit result
    DeviceBuffer qo[4], qd[4], counts, hitsMain, hitInst, occluded, stL, stBeta, stMeta, pdLight, pdMis, pdBeta, pdInfo, traceCn,
        lightTests, filmDev, straysDev, nStraysDev, cullGuard, cursors, cursors2;
    hipStream_t shadowStream = nullptr;  // any-hit launches run here, concurrently with the next closest-hit launch
    hipEvent_t evShaded = nullptr, evShadowed = nullptr;
    bool cullTripped = false;  // the last call raised k_trace's cull guard (checkCullGuard)
    bool overlapShadow = false;  // PG_OVERLAP_SHADOW=1: any-hit launch on a second stream beside the closest-hit launch (per-kernel times then overlap)
    // test-path buffers
    DeviceBuffer tO, tD, tT, tPrim, tHit, tOcc, tCount;
    PgCounters counters;
    std::vector<hipEvent_t> events;
    bool hasNullMaterial = false;
    // VolPathIntegrator work buffers (sized on the first volpath render)
    int volCapacity = 0;
    int nMedia = 0;
    DeviceBuffer vqo[2], vqd[2], vCounts, volMedium, trAcc[2], volP1[3], misLi, pdLi, hitT;
    DeviceBuffer qsL[2], qsBeta[2], qsMeta[2], qsMedium[2];  // path state in queue order (qsMedium: volpath)
    DeviceBuffer bssrdfs, materialBssrdf, bssrdfTables;  // subsurface scattering (ABI 24)
    DeviceBuffer grids, mediaGrid, gridDensity, gridVertex;  // GridDensityMedium (ABI 23); the two-phase shading's per-slot vertex record
    // the BSSRDF branch of Li: per-slot state between entry and exit vertex (SssState), the job queue, two probe queues
    DeviceBuffer sssPo, sssFrame[3], sssCoef[2], sssTarget, sssCount, sssHit, sssHitO, sssHitD, sssHitInst, sssHitXf, sssMedium, sssQo[3], sssQd[3], sssCounts, sssTail;
    int sssCapacity = 0;
    DeviceBuffer lightHot;  // DScene::lightHot
    DeviceBuffer haltonDims;  // DScene::haltonDims
    DeviceBuffer cmaxmin, tsState, ts1, ts2;  // tile-serial samplers: CMaxMinDist, the tiles' sampler states and sample arrays
    DeviceBuffer voxelSlot, voxelRequests, voxelCounters, retryList;  // sparse "spatial" light tables (DScene::sparseLights)
    int poolSlots = 0, poolUsed = 0, nVoxelsTotal = 0;
    DeviceBuffer shardFilm, gatherDev;  // pg_render_sharded: this device's packed shard [film | strays | count]; on rank 0's device the gathered frame
    int qsCapacity = 0;
};

extern "C" {

const char *pg_last_error(void) { return g_lastError.c_str(); }

int pg_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return setError(PG_ERR_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    return n;
}
int pg_set_device(int device) {
    HIP_TRY(hipSetDevice(device));
    return PG_OK;
}

void pg_scene_destroy(PgScene *s) {
    if (!s) return;
    (void)hipSetDevice(s->device);
    for (hipEvent_t e : s->events) (void)hipEventDestroy(e);
    if (s->evShaded) (void)hipEventDestroy(s->evShaded);
    if (s->evShadowed) (void)hipEventDestroy(s->evShadowed);
    if (s->shadowStream) (void)hipStreamDestroy(s->shadowStream);
    delete s;
}

int pg_scene_create(const PgSceneDesc *desc, PgScene **out) {
    if (!desc || !out) return setError(PG_ERR_INVALID, "pg_scene_create: null argument");
    *out = nullptr;
    if (desc->abi_version != PG_ABI_VERSION) return setError(PG_ERR_INVALID, "ABI version %d, expected %d", desc->abi_version, PG_ABI_VERSION);
    // (P may be absent when no primitive is a triangle -- a scene of quadrics only: every triangle's indices are checked against n_verts below)
    if (desc->n_tris < 0 || desc->n_nodes < 0 || desc->n_verts < 0 || (desc->n_tris > 0 && (!desc->nodes || !desc->indices || (desc->n_verts > 0 && !desc->P))))
        return setError(PG_ERR_INVALID, "pg_scene_create: malformed geometry arrays");
    if (desc->n_grids < 0 || (desc->n_grids > 0 && (!desc->grids || !desc->media_grid || !desc->grid_density)))
        return setError(PG_ERR_INVALID, "pg_scene_create: malformed grid-medium tables");
    if (desc->n_bssrdfs < 0 || (desc->n_bssrdfs > 0 && (!desc->bssrdfs || !desc->material_bssrdf || !desc->bssrdf_tables)))
        return setError(PG_ERR_INVALID, "pg_scene_create: malformed BSSRDF tables");
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (ndev <= 0) return setError(PG_ERR_DEVICE, "no HIP device visible (there is no CPU fallback)");
    PgScene *s = new PgScene;
    HIP_TRY(hipGetDevice(&s->device));
    memset(&s->counters, 0, sizeof(s->counters));
    DScene &d = s->d;
    memset(&d, 0, sizeof(d));
    // primitives / nodes of object definitions follow the top-level ones (hosts that know no instancing leave the _all counts 0)
    const int nt = desc->n_prims_all > desc->n_tris ? desc->n_prims_all : desc->n_tris;
    const int nnAll = desc->n_nodes_all > desc->n_nodes ? desc->n_nodes_all : desc->n_nodes;
#define FAIL(code, ...) do { int c_ = setError(code, __VA_ARGS__); pg_scene_destroy(s); return c_; } while (0)
#define HIP_TRY_S(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) FAIL(PG_ERR_DEVICE, "%s: %s", #expr, hipGetErrorString(e_)); } while (0)
    // --- nodes: uploaded verbatim (32 B/node, same bytes as pbrt's LinearBVHNode)
    HIP_TRY_S(s->nodes.alloc(sizeof(PgBVHNode) * (size_t)nnAll));
    if (nnAll) HIP_TRY_S(hipMemcpy(s->nodes.p, desc->nodes, s->nodes.bytes, hipMemcpyHostToDevice));
    // --- child-pair records for k_trace (pg_traverse.hip): one 64-B record per interior node
    {
        int maxLeaf = 1;
        for (int i = 0; i < nnAll; ++i) if (desc->nodes[i].nprims > maxLeaf) maxLeaf = desc->nodes[i].nprims;
        int leafBits = 0;
        while ((1 << leafBits) < maxLeaf) ++leafBits;
        if ((uint64_t)nt >= ((uint64_t)1 << (31 - leafBits)) - 1)
            FAIL(PG_ERR_UNSUPPORTED, "%d triangles with up to %d per leaf exceed the 31-bit leaf reference", nt, maxLeaf);
        std::vector<float4> w;
        // one BVHAccel's nodes [firstNode, +nn) -> records appended to w; leaf references carry GLOBAL primitive indices
        // (firstPrim + the node's own offset).  Returns the reference of the BVH's root.
        bool badChildren = false;
        int worldPending = 0, objectPending = 0;  // stack entries the world BVH / the deepest object BVH can have pending
        // which records share a 128-B line (measured, profiles/r03n_record_layout_ab.txt: 1 is 1 - 3 % faster than 0, 2 is no gain)
        const int recordLayout = getenv("PG_RECORD_LAYOUT") ? atoi(getenv("PG_RECORD_LAYOUT")) : 1;
        auto buildRecords = [&](int firstNode, int nn, int firstPrim) -> int {
            const PgBVHNode *nodes = desc->nodes + firstNode;
            // The node array comes from the caller: before anything is indexed by it, require the reference's layout
            // (bvh.cpp:640-658: first child at i + 1, second child later in the array) and that it is a tree -- every node but
            // the root the child of exactly one interior node -- so that every interior node is reached from the root once.
            // The same pass takes the greatest number of entries a traversal can have pending, which the stack must hold.
            {
                std::vector<int> level((size_t)nn, 0);
                std::vector<unsigned char> parents((size_t)nn, 0);
                int deepest = 0;
                for (int i = 0; i < nn && !badChildren; ++i) {
                    if (i > 0 && parents[i] != 1) { badChildren = true; break; }
                    if (nodes[i].nprims != 0) continue;
                    const int c0 = i + 1;
                    const long long c1 = nodes[i].offset;
                    if (c0 >= nn || c1 <= c0 || c1 >= nn || parents[c0] || parents[c1] || nodes[i].axis > 2) { badChildren = true; break; }
                    parents[c0] = parents[c1] = 1;
                    level[c0] = level[c1] = level[i] + 1;  // entries pending while a child of node i is visited: <= level
                    deepest = std::max(deepest, level[i] + 1);
                }
                if (badChildren) return TR_NO_ROOT;
                (firstNode == 0 ? worldPending : objectPending) = std::max(firstNode == 0 ? worldPending : objectPending, deepest);
            }
            std::vector<int> recIndex((size_t)nn, -1);
            int nInterior = 0;
            if (recordLayout != 0 && ((w.size() / 4) & 1)) w.resize(w.size() + 4, make_float4(0, 0, 0, 0));
            const int base = (int)(w.size() / 4);
            for (int i = 0; i < nn; ++i) if (nodes[i].nprims == 0) ++nInterior;
            if (recordLayout == 0) {  // depth-first: a record's line mate is the next interior node of the reference's array
                int k = 0;
                for (int i = 0; i < nn; ++i) if (nodes[i].nprims == 0) recIndex[i] = base + k++;
            } else {
                // Two records share a 128-B line, and an L2 miss fills the whole line: choose the line mates.  1: a node with the
                // child a ray through it is likelier to visit (the one with the larger surface area); 2: the two children of a
                // node.  Nodes left alone (no interior child / sibling) pair up among themselves in the order they are met.
                auto area = [&](int i) { const PgBVHNode &b = nodes[i]; const float x = b.bmax[0] - b.bmin[0], y = b.bmax[1] - b.bmin[1], z = b.bmax[2] - b.bmin[2]; return x * y + y * z + z * x; };
                int nPlaced = 0;  // (base is even: lines are pairs of absolute record indices, and nodes are placed two at a time)
                auto place = [&](int n) { recIndex[n] = base + nPlaced++; };
                std::vector<int> singles, stack;
                if (nn > 0 && nodes[0].nprims == 0) stack.push_back(0);
                auto placeSingle = [&](int n) { singles.push_back(n); if (singles.size() == 2) { place(singles[0]); place(singles[1]); singles.clear(); } };
                while (!stack.empty()) {
                    const int n = stack.back(); stack.pop_back();
                    const int c0 = n + 1, c1 = nodes[n].offset;
                    if (c0 >= nn || c1 <= n || c1 >= nn) continue;  // (reported below)
                    const bool i0 = nodes[c0].nprims == 0, i1 = nodes[c1].nprims == 0;
                    if (recordLayout == 1) {  // `n` is the head of a line unless it was placed as its parent's mate
                        int h = -1, o = -1;
                        if (i0 && i1) { h = area(c0) >= area(c1) ? c0 : c1; o = h == c0 ? c1 : c0; } else if (i0) h = c0; else if (i1) h = c1;
                        if (recIndex[n] < 0) {
                            if (h < 0) { placeSingle(n); continue; }
                            place(n); place(h);
                            if (o >= 0) stack.push_back(o);
                            // h's own children head new lines
                            const int h0 = h + 1, h1 = nodes[h].offset;
                            if (h1 > h && h1 < nn && h0 < nn) { if (nodes[h1].nprims == 0) stack.push_back(h1); if (nodes[h0].nprims == 0) stack.push_back(h0); }
                        }
                    } else {  // siblings share a line
                        if (n == 0) placeSingle(n);  // (the root; every other node is placed, or waits among the singles, when it is pushed)
                        if (i0 && i1) { place(c0); place(c1); stack.push_back(c1); stack.push_back(c0); }
                        else if (i0) { placeSingle(c0); stack.push_back(c0); }
                        else if (i1) { placeSingle(c1); stack.push_back(c1); }
                    }
                }
                if (singles.size() == 1) place(singles[0]);
                if (nPlaced != nInterior) badChildren = true;  // every interior node has exactly one slot
            }
            auto refOf = [&](int i) -> int {
                const PgBVHNode &nd = nodes[i];
                return nd.nprims == 0 ? recIndex[i] : ~(((firstPrim + nd.offset) << leafBits) | (nd.nprims - 1));
            };
            w.resize((size_t)(base + nInterior) * 4);
            for (int i = 0; i < nn; ++i) {
                const PgBVHNode &nd = nodes[i];
                if (nd.nprims != 0) continue;
                const int c0 = i + 1, c1 = nd.offset;
                if (c0 >= nn || c1 <= i || c1 >= nn) { badChildren = true; continue; }
                const PgBVHNode &a = nodes[c0], &b = nodes[c1];
                float4 *r = &w[(size_t)recIndex[i] * 4];
                r[0] = make_float4(a.bmin[0], a.bmax[0], b.bmin[0], b.bmax[0]);
                r[1] = make_float4(a.bmin[1], a.bmax[1], b.bmin[1], b.bmax[1]);
                r[2] = make_float4(a.bmin[2], a.bmax[2], b.bmin[2], b.bmax[2]);
                int r0 = refOf(c0), r1 = refOf(c1), ax = nd.axis;
                float f0, f1, f2;
                memcpy(&f0, &r0, 4); memcpy(&f1, &r1, 4); memcpy(&f2, &ax, 4);
                r[3] = make_float4(f0, f1, f2, 0.f);
            }
            return nn > 0 ? refOf(0) : TR_NO_ROOT;
        };
        const int nn = desc->n_nodes;
        s->trace = default_trace_config();
        const int topRef = buildRecords(0, nn, 0);
        // object definitions (instancing): each with its own records, root box and root reference
        if (desc->n_objects > 0 && !desc->objects) FAIL(PG_ERR_INVALID, "pg_scene_create: n_objects = %d without an objects array", desc->n_objects);
        std::vector<DObject> objs((size_t)(desc->n_objects > 0 ? desc->n_objects : 0));
        for (size_t k = 0; k < objs.size(); ++k) {
            const PgObject &o = desc->objects[k];
            if (o.first_prim < desc->n_tris || o.n_prims < 1 || o.first_prim + o.n_prims > nt || o.n_nodes < 0 ||
                (o.n_nodes > 0 && (o.first_node < desc->n_nodes || o.first_node + o.n_nodes > nnAll)) || (o.n_nodes == 0 && o.n_prims != 1))
                FAIL(PG_ERR_INVALID, "object %d: nodes [%d, +%d) / primitives [%d, +%d) out of range", (int)k, o.first_node, o.n_nodes, o.first_prim, o.n_prims);
            DObject &dobj = objs[k];
            memset(&dobj, 0, sizeof(dobj));
            dobj.firstPrim = o.first_prim;
            dobj.nNodes = o.n_nodes;
            if (o.n_nodes > 0) {
                dobj.rootRef = buildRecords(o.first_node, o.n_nodes, o.first_prim);
                for (int c = 0; c < 3; ++c) { dobj.box[c] = desc->nodes[o.first_node].bmin[c]; dobj.box[3 + c] = desc->nodes[o.first_node].bmax[c]; }
            }
        }
        if (badChildren) FAIL(PG_ERR_INVALID, "the BVH node array is not a tree in the reference's layout (a child out of range, shared or unreachable)");
        // k_trace's stack (csrc/pg_traverse.hip): TR_STACK_TOTAL entries behind the LDS part, shared by the world traversal and the
        // instance traversal above it; the reference-order any-hit kernel keeps one bit per pending entry of ONE tree in a 64-bit
        // mask.  The reference itself has int nodesToVisit[64] per BVH (bvh.cpp:670, :708) and no check at all.
        if (worldPending > 64 || objectPending > 64)
            FAIL(PG_ERR_UNSUPPORTED, "a BVH is %d levels deep: more than the 64 pending nodes of the reference's traversal stack (bvh.cpp:670)",
                 std::max(worldPending, objectPending));
        if (worldPending + objectPending > 64 + s->trace.depth)
            FAIL(PG_ERR_UNSUPPORTED, "world BVH (%d levels) + object BVH (%d levels) exceed the traversal stack of %d entries", worldPending, objectPending,
                 64 + s->trace.depth);
        for (int k = 0; k < nt; ++k) {
            const uint32_t f = desc->tri_flags ? desc->tri_flags[k] : 0;
            if (!(f & PG_PRIM_INSTANCE)) continue;
            const int ii = desc->indices[3 * k];
            if (ii < 0 || ii >= desc->n_instances || !desc->instances || desc->instances[ii].object < 0 || desc->instances[ii].object >= desc->n_objects)
                FAIL(PG_ERR_INVALID, "primitive %d: instance %d / its object out of range", k, ii);
            if (k >= desc->n_tris) {
                // ABI 29: a TransformedPrimitive among an object definition's primitives (a moving shape inside ObjectBegin / ObjectEnd, api.cpp:1386-1419):
                // ONE level -- what it wraps holds shapes only, as in the reference, whose ObjectInstance cannot appear inside a definition (api.cpp:1549-1552)
                const PgObject &inner = desc->objects[desc->instances[ii].object];
                for (int q = inner.first_prim; q < inner.first_prim + inner.n_prims; ++q)
                    if (q < 0 || q >= nt || (desc->tri_flags[q] & PG_PRIM_INSTANCE))
                        FAIL(PG_ERR_UNSUPPORTED, "primitive %d: a TransformedPrimitive inside an object definition wraps another one (more than two levels)", k);
                if (k >= inner.first_prim && k < inner.first_prim + inner.n_prims) FAIL(PG_ERR_INVALID, "primitive %d: an object definition contains itself", k);
                d.hasNest = 1;
            }
        }
        if (d.hasNest) {
            // hitInst = outer + n_instances * (inner + 1) in an int; three BVHs share k_trace's stack
            if ((int64_t)desc->n_instances * ((int64_t)desc->n_instances + 1) >= ((int64_t)1 << 31))
                FAIL(PG_ERR_UNSUPPORTED, "%d instances in a scene with TransformedPrimitives inside object definitions: the pair (outer, inner) does not fit a hit's instance word", desc->n_instances);
            if (worldPending + 2 * objectPending > 64 + s->trace.depth)
                FAIL(PG_ERR_UNSUPPORTED, "world BVH (%d levels) + two object BVHs (%d levels) exceed the traversal stack of %d entries", worldPending, objectPending, 64 + s->trace.depth);
        }
        if (!objs.empty()) {
            HIP_TRY_S(s->objects.alloc(sizeof(DObject) * objs.size()));
            HIP_TRY_S(hipMemcpy(s->objects.p, objs.data(), s->objects.bytes, hipMemcpyHostToDevice));
        }
        if (desc->n_instances > 0 && desc->instances) {
            HIP_TRY_S(s->instances.alloc(sizeof(PgInstance) * (size_t)desc->n_instances));
            HIP_TRY_S(hipMemcpy(s->instances.p, desc->instances, s->instances.bytes, hipMemcpyHostToDevice));
        }
        d.instEntry = nullptr;
        if (desc->n_instances > 0 && desc->instances && !objs.empty()) {
            std::vector<DInstEntry> ent((size_t)desc->n_instances);
            for (int i = 0; i < desc->n_instances; ++i) {
                const PgInstance &in = desc->instances[i];
                DInstEntry &e = ent[i];
                memset(&e, 0, sizeof(e));
                if (in.object < 0 || in.object >= desc->n_objects) continue;  // (never referenced: the primitives' instances were checked above)
                const DObject &ob = objs[in.object];
                memcpy(e.w2i, in.w2i, sizeof(e.w2i));
                memcpy(e.box, ob.box, sizeof(e.box));
                e.rootRef = ob.rootRef; e.firstPrim = ob.firstPrim; e.nNodes = ob.nNodes;
                const float last[4] = {0.f, 0.f, 0.f, 1.f};
                e.affineStill = (!in.animated && memcmp(in.w2i + 12, last, sizeof(last)) == 0) ? 1 : 0;  // (bitwise: -0 is not 0 here)
            }
            HIP_TRY_S(s->instEntry.alloc(sizeof(DInstEntry) * ent.size()));
            HIP_TRY_S(hipMemcpy(s->instEntry.p, ent.data(), s->instEntry.bytes, hipMemcpyHostToDevice));
            d.instEntry = (const DInstEntry *)s->instEntry.p;
        }
        d.objects = (const DObject *)s->objects.p;
        d.instances = (const PgInstance *)s->instances.p;
        d.nInstances = s->instances.p ? desc->n_instances : 0;
        d.hasMotion = 0; d.rayTimes = 0; d.animXf = nullptr;
        for (int i = 0; i < d.nInstances; ++i) if (desc->instances[i].animated) d.hasMotion = d.rayTimes = 1;
        HIP_TRY_S(s->wnodes.alloc(sizeof(float4) * w.size()));
        if (!w.empty()) HIP_TRY_S(hipMemcpy(s->wnodes.p, w.data(), s->wnodes.bytes, hipMemcpyHostToDevice));
        d.wnodes = (const float4 *)s->wnodes.p;
        d.leafBits = leafBits;
        if (nn > 0) {
            for (int k = 0; k < 3; ++k) { d.rootBox[k] = desc->nodes[0].bmin[k]; d.rootBox[3 + k] = desc->nodes[0].bmax[k]; }
            d.rootRef = topRef;
        }
    }
    // --- triangles: gather vertices into BVH order, 48 B per triangle
    std::vector<float4> tris((size_t)nt * PG_TRI_STRIDE, make_float4(0, 0, 0, 0));
    std::vector<float> uv;
    bool anyUV = false;
    for (int k = 0; k < nt; ++k) anyUV |= desc->UV && desc->tri_flags && (desc->tri_flags[k] & PG_TRI_HAS_UV);
    if (anyUV) uv.resize((size_t)nt * 6);
    for (int k = 0; k < nt; ++k) {
        const int32_t *v = &desc->indices[3 * k];
        uint32_t flags = desc->tri_flags ? desc->tri_flags[k] : 0;
        int mat = desc->tri_material ? desc->tri_material[k] : 0;
        int light = desc->tri_light ? desc->tri_light[k] : -1;
        if (mat < 0 || mat >= desc->n_materials) FAIL(PG_ERR_INVALID, "triangle %d has out-of-range material %d", k, mat);
        if (light >= desc->n_lights) FAIL(PG_ERR_INVALID, "triangle %d has out-of-range light %d", k, light);
        if (flags & PG_PRIM_INSTANCE) {  // TransformedPrimitive: the record carries the instance's index
            float iw, fw, mw, lw;
            const int none = -1;
            flags = PG_PRIM_INSTANCE;
            memcpy(&iw, &v[0], 4); memcpy(&fw, &flags, 4); memcpy(&mw, &mat, 4); memcpy(&lw, &none, 4);
            tris[PG_TRI_STRIDE * (size_t)k] = make_float4(iw, 0, 0, fw);
            tris[PG_TRI_STRIDE * (size_t)k + 1] = make_float4(0, 0, 0, mw);
            tris[PG_TRI_STRIDE * (size_t)k + 2] = make_float4(0, 0, 0, lw);
            if (anyUV) { const float duv[6] = {0, 0, 1, 0, 1, 1}; memcpy(&uv[(size_t)k * 6], duv, sizeof(duv)); }
            continue;
        }
        if (flags & PG_PRIM_SPHERE) {  // Shape "sphere": the record carries the sphere's index where a triangle has p0.x
            if (v[0] < 0 || v[0] >= desc->n_spheres || !desc->spheres) FAIL(PG_ERR_INVALID, "primitive %d has out-of-range sphere index %d", k, v[0]);
            float iw, fw, mw, lw;
            flags = PG_PRIM_SPHERE;
            memcpy(&iw, &v[0], 4); memcpy(&fw, &flags, 4); memcpy(&mw, &mat, 4); memcpy(&lw, &light, 4);
            tris[PG_TRI_STRIDE * (size_t)k] = make_float4(iw, 0, 0, fw);
            tris[PG_TRI_STRIDE * (size_t)k + 1] = make_float4(0, 0, 0, mw);
            tris[PG_TRI_STRIDE * (size_t)k + 2] = make_float4(0, 0, 0, lw);
            if (anyUV) { const float duv[6] = {0, 0, 1, 0, 1, 1}; memcpy(&uv[(size_t)k * 6], duv, sizeof(duv)); }
            continue;
        }
        for (int j = 0; j < 3; ++j)
            if (v[j] < 0 || v[j] >= desc->n_verts) FAIL(PG_ERR_INVALID, "triangle %d has out-of-range vertex index %d", k, v[j]);
        V3 p[3];
        for (int j = 0; j < 3; ++j) p[j] = mk(desc->P[3 * v[j]], desc->P[3 * v[j] + 1], desc->P[3 * v[j] + 2]);
        float tuv[6] = {0, 0, 1, 0, 1, 1};
        if (anyUV && (flags & PG_TRI_HAS_UV))
            for (int j = 0; j < 3; ++j) { tuv[2 * j] = desc->UV[2 * v[j]]; tuv[2 * j + 1] = desc->UV[2 * v[j] + 1]; }
        if (anyUV) memcpy(&uv[(size_t)k * 6], tuv, sizeof(tuv));
        V3 dpdu;
        if (!tri_dpdu(p[0], p[1], p[2], tuv, dpdu)) flags |= PG_TRI_BOGUS;  // triangle.cpp:309-317
        float fw, mw, lw;
        memcpy(&fw, &flags, 4); memcpy(&mw, &mat, 4); memcpy(&lw, &light, 4);
        tris[PG_TRI_STRIDE * (size_t)k] = make_float4(p[0].x, p[0].y, p[0].z, fw);
        tris[PG_TRI_STRIDE * (size_t)k + 1] = make_float4(p[1].x, p[1].y, p[1].z, mw);
        tris[PG_TRI_STRIDE * (size_t)k + 2] = make_float4(p[2].x, p[2].y, p[2].z, lw);
    }
    // per-vertex normals and (u, v), de-indexed like the positions, one 64-byte record per triangle (DScene::triAttr; only when some mesh has either);
    // tangents in an array of their own (only when some mesh has them)
    bool anyN = false, anyS = false;
    for (int k = 0; k < nt; ++k) {
        anyN |= desc->N && desc->tri_flags && (desc->tri_flags[k] & PG_TRI_HAS_N);
        anyS |= desc->S && desc->tri_flags && (desc->tri_flags[k] & PG_TRI_HAS_S);
    }
    if (anyN || anyUV) {
        std::vector<float> a((size_t)nt * 16, 0.f);
        for (int k = 0; k < nt; ++k) {
            float *r = &a[16 * (size_t)k];
            if (anyUV) memcpy(r + 10, &uv[(size_t)k * 6], 6 * sizeof(float));
            if (!anyN || !(desc->tri_flags[k] & PG_TRI_HAS_N) || (desc->tri_flags[k] & (PG_PRIM_INSTANCE | PG_PRIM_SPHERE))) continue;
            const int32_t *v = &desc->indices[3 * k];
            for (int j = 0; j < 3; ++j) for (int c = 0; c < 3; ++c) r[3 * j + c] = desc->N[3 * v[j] + c];
        }
        bool anyAlpha = false;  // k_trace's alpha masks read the (u, v) alone: from the compact array (24 B per triangle) they had before the attribute records
        for (int k = 0; k < nt && anyUV; ++k) anyAlpha |= (desc->tri_flags[k] & PG_TRI_ALPHA) != 0;
        if (anyAlpha) {
            HIP_TRY_S(s->uv.alloc(sizeof(float) * uv.size()));
            HIP_TRY_S(hipMemcpy(s->uv.p, uv.data(), s->uv.bytes, hipMemcpyHostToDevice));
        }
        HIP_TRY_S(s->triAttr.alloc(sizeof(float) * a.size()));
        HIP_TRY_S(hipMemcpy(s->triAttr.p, a.data(), s->triAttr.bytes, hipMemcpyHostToDevice));
    }
    if (anyS) {
        std::vector<float4> a((size_t)nt * 3, make_float4(0, 0, 0, 0));
        for (int k = 0; k < nt; ++k) {
            if (!(desc->tri_flags[k] & PG_TRI_HAS_S) || (desc->tri_flags[k] & (PG_PRIM_INSTANCE | PG_PRIM_SPHERE))) continue;
            const int32_t *v = &desc->indices[3 * k];
            for (int j = 0; j < 3; ++j) a[3 * (size_t)k + j] = make_float4(desc->S[3 * v[j]], desc->S[3 * v[j] + 1], desc->S[3 * v[j] + 2], 0.f);
        }
        HIP_TRY_S(s->triS.alloc(sizeof(float4) * a.size()));
        HIP_TRY_S(hipMemcpy(s->triS.p, a.data(), s->triS.bytes, hipMemcpyHostToDevice));
    }
    {  // texture graph: operands are earlier nodes (no cycles), nesting <= 3 levels (the unrolled evaluator of pg_kernels.hip)
        std::vector<int> depth((size_t)(desc->n_textures > 0 ? desc->n_textures : 0), 1);
        auto refDepth = [&](const PgTexRef &r, int self) -> int { return r.tex < 0 ? 0 : ((r.tex >= self) ? 1000 : depth[r.tex]); };
        for (int i = 0; i < (int)depth.size(); ++i) {
            const PgTexture &t = desc->textures[i];
            if (t.type < PG_TEX_SCALE || t.type > PG_TEX_DOTS) FAIL(PG_ERR_UNSUPPORTED, "texture %d: unknown type %d", i, t.type);
            if (t.type >= PG_TEX_FBM && !desc->noise_perm) FAIL(PG_ERR_INVALID, "texture %d is a Perlin-noise texture, but the scene has no noise_perm table", i);
            if (t.type == PG_TEX_MARBLE && t.is_float) FAIL(PG_ERR_UNSUPPORTED, "texture %d: marble is a spectrum texture only (marble.cpp:39-42)", i);
            if (t.type == PG_TEX_IMAGEMAP && (t.image < 0 || t.image >= desc->n_images || desc->images[t.image].is_float != (t.is_float ? 1 : 0)))
                FAIL(PG_ERR_INVALID, "texture %d: image %d out of range or of the wrong texel type", i, t.image);
            int dmax = std::max(refDepth(t.tex1, i), std::max(refDepth(t.tex2, i), refDepth(t.amount, i)));
            if (dmax >= 1000) FAIL(PG_ERR_INVALID, "texture %d refers to a texture that is not defined before it", i);
            depth[i] = 1 + dmax;
            if (depth[i] > 3) FAIL(PG_ERR_UNSUPPORTED, "texture %d: operands nested %d deep (this build evaluates 3 levels)", i, depth[i]);
        }
        for (int i = 0; i < desc->n_textured; ++i) {
            const PgTexturedMaterial &tm = desc->textured[i];
            if (tm.kind < PG_KIND_MATTE || tm.kind > PG_KIND_MIX) FAIL(PG_ERR_UNSUPPORTED, "textured material %d: unknown kind %d", i, tm.kind);
            for (int k = 0; k < 5; ++k) if (tm.s[k].tex >= desc->n_textures) FAIL(PG_ERR_INVALID, "textured material %d: texture %d out of range", i, tm.s[k].tex);
            for (int k = 0; k < 4; ++k) if (tm.f[k].tex >= desc->n_textures) FAIL(PG_ERR_INVALID, "textured material %d: texture %d out of range", i, tm.f[k].tex);
            if (tm.kind == PG_KIND_MIX) {
                for (int j = 0; j < 2; ++j) {
                    if (tm.sub[j] < 0 || tm.sub[j] >= desc->n_materials) FAIL(PG_ERR_INVALID, "textured material %d: mixed material %d out of range", i, tm.sub[j]);
                    const PgMaterial &sm = desc->materials[tm.sub[j]];
                    if (sm.type == PG_MAT_TEXTURED && sm.textured_index >= 0 && sm.textured_index < desc->n_textured && desc->textured[sm.textured_index].kind == PG_KIND_MIX) {
                        const PgTexturedMaterial &t2 = desc->textured[sm.textured_index];
                        for (int q = 0; q < 2; ++q)
                            if (t2.sub[q] >= 0 && t2.sub[q] < desc->n_materials && desc->materials[t2.sub[q]].type == PG_MAT_TEXTURED)
                                FAIL(PG_ERR_UNSUPPORTED, "textured material %d: textured mix materials nested more than two deep", i);
                    }
                }
            }
        }
    }
    if (desc->n_images > 0) {
        if (!desc->images || !desc->texels || !desc->ewa_lut) FAIL(PG_ERR_INVALID, "images without texels / EWA weight table");
        for (int i = 0; i < desc->n_images; ++i) {
            const PgImage &im = desc->images[i];
            if (im.n_levels < 1 || im.n_levels > PG_MAX_MIP_LEVELS || im.width < 1 || im.height < 1 || (im.width & (im.width - 1)) || (im.height & (im.height - 1)))
                FAIL(PG_ERR_INVALID, "image %d: %d levels of %d x %d (MIPMap levels are powers of two)", i, im.n_levels, im.width, im.height);
            for (int l = 0; l < im.n_levels; ++l) {
                const int64_t sRes = std::max(1, im.width >> l), tRes = std::max(1, im.height >> l);
                if (im.level_offset[l] < 0 || im.level_offset[l] + sRes * tRes * (im.is_float ? 1 : 3) > desc->n_texel_floats)
                    FAIL(PG_ERR_INVALID, "image %d: level %d lies outside the texel array", i, l);
            }
        }
        HIP_TRY_S(s->images.alloc(sizeof(PgImage) * (size_t)desc->n_images));
        HIP_TRY_S(hipMemcpy(s->images.p, desc->images, s->images.bytes, hipMemcpyHostToDevice));
        HIP_TRY_S(s->texels.alloc(sizeof(float) * (size_t)desc->n_texel_floats));
        HIP_TRY_S(hipMemcpy(s->texels.p, desc->texels, s->texels.bytes, hipMemcpyHostToDevice));
        HIP_TRY_S(s->ewaLut.alloc(sizeof(float) * 128));
        HIP_TRY_S(hipMemcpy(s->ewaLut.p, desc->ewa_lut, s->ewaLut.bytes, hipMemcpyHostToDevice));
    }
    if (desc->n_textures > 0 && desc->textures) {
        HIP_TRY_S(s->textures.alloc(sizeof(PgTexture) * (size_t)desc->n_textures));
        HIP_TRY_S(hipMemcpy(s->textures.p, desc->textures, s->textures.bytes, hipMemcpyHostToDevice));
    }
    if (desc->n_textured > 0 && desc->textured) {
        HIP_TRY_S(s->textured.alloc(sizeof(PgTexturedMaterial) * (size_t)desc->n_textured));
        HIP_TRY_S(hipMemcpy(s->textured.p, desc->textured, s->textured.bytes, hipMemcpyHostToDevice));
    }
    if (desc->n_bxdfs > 0 && desc->bxdfs) {
        HIP_TRY_S(s->bxdfs.alloc(sizeof(PgBxDF) * (size_t)desc->n_bxdfs));
        HIP_TRY_S(hipMemcpy(s->bxdfs.p, desc->bxdfs, s->bxdfs.bytes, hipMemcpyHostToDevice));
    }
    if (desc->n_spheres > 0) {
        HIP_TRY_S(s->spheres.alloc(sizeof(PgSphere) * (size_t)desc->n_spheres));
        HIP_TRY_S(hipMemcpy(s->spheres.p, desc->spheres, s->spheres.bytes, hipMemcpyHostToDevice));
    }
    HIP_TRY_S(s->tris.alloc(sizeof(float4) * tris.size()));
    if (nt) HIP_TRY_S(hipMemcpy(s->tris.p, tris.data(), s->tris.bytes, hipMemcpyHostToDevice));
    // --- materials / lights
    bool anyLobeMaterial = false, anyTextured = false;
    for (int i = 0; i < desc->n_materials; ++i) {
        const PgMaterial &m = desc->materials[i];
        if (m.type == PG_MAT_NONE) s->hasNullMaterial = true;
        else if (m.type < PG_MAT_MATTE || m.type > PG_MAT_TEXTURED)
            FAIL(PG_ERR_UNSUPPORTED, "material %d: unknown type %d", i, m.type);
        if (m.type == PG_MAT_LOBES) anyLobeMaterial = true;
        if (m.type == PG_MAT_TEXTURED) {
            anyTextured = true;
            if (m.textured_index < 0 || m.textured_index >= desc->n_textured || !desc->textured)
                FAIL(PG_ERR_INVALID, "material %d: textured_index %d out of range", i, m.textured_index);
        }
        if (m.n_bxdfs < 0 || m.n_bxdfs > PG_MAX_BXDFS || (m.n_bxdfs > 0 && (m.first_bxdf < 0 || m.first_bxdf + m.n_bxdfs > desc->n_bxdfs || !desc->bxdfs)))
            FAIL(PG_ERR_INVALID, "material %d: BxDF list [%d, +%d) is outside the scene's %d BxDFs", i, m.first_bxdf, m.n_bxdfs, desc->n_bxdfs);
        for (int j = 0; j < m.n_bxdfs; ++j) {
            const PgBxDF &bx = desc->bxdfs[m.first_bxdf + j];
            if (bx.type < PG_BXDF_LAMBERT_R || bx.type > PG_BXDF_FRESNEL_BLEND || bx.fresnel < PG_FRESNEL_NOOP || bx.fresnel > PG_FRESNEL_CONDUCTOR ||
                bx.n_scales < 0 || bx.n_scales > PG_MAX_BXDF_SCALES)
                FAIL(PG_ERR_UNSUPPORTED, "material %d: BxDF %d has unknown type %d / fresnel %d / %d scales", i, j, bx.type, bx.fresnel, bx.n_scales);
        }
    }
    // device copy: a plastic's `roughness` becomes the TrowbridgeReitz alpha.  RoughnessToAlpha (microfacet.h:127-132) calls
    // logf; evaluating it here on the host uses the same libm as the reference build.
    std::vector<PgMaterial> devMaterials(desc->materials, desc->materials + desc->n_materials);
    for (PgMaterial &m : devMaterials) {
        if (m.type != PG_MAT_PLASTIC) continue;
        float rough = m.roughness;
        if (m.remap_roughness) {
            rough = (rough < 1e-3f) ? 1e-3f : rough;  // std::max(roughness, (Float)1e-3)
            float x = logf(rough);
            rough = 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
        }
        m.roughness = (0.001f < rough) ? rough : 0.001f;  // TrowbridgeReitzDistribution ctor: std::max(Float(0.001), alpha)
    }
    d.hasInfinite = 0;
    for (int i = 0; i < desc->n_lights; ++i)
        if (desc->lights[i].type == PG_LIGHT_INFINITE) {
            d.hasInfinite = 1;
            const PgLight &l = desc->lights[i];
            const int64_t need = (int64_t)(2 * (int64_t)l.env_nu + 2) * l.env_nv + 2 * (int64_t)l.env_nv + 2;
            if (l.env_image < 0 || l.env_image >= desc->n_images || !desc->images || desc->images[l.env_image].is_float || l.env_nu < 1 || l.env_nv < 1 ||
                l.env_table < 0 || l.env_table + need > desc->n_env_floats || !desc->env_tables)
                FAIL(PG_ERR_INVALID, "light %d: infinite light without a valid radiance map / sampling distribution", i);
        }
    bool anyImageLight = false;  // projection / goniometric lights: sampled by the general shading kernels only
    for (int i = 0; i < desc->n_lights; ++i)
        if (desc->lights[i].type == PG_LIGHT_PROJECTION || desc->lights[i].type == PG_LIGHT_GONIO) {
            anyImageLight = true;
            const PgLight &l = desc->lights[i];
            if (l.env_image >= desc->n_images || (l.env_image >= 0 && (!desc->images || desc->images[l.env_image].is_float)))
                FAIL(PG_ERR_INVALID, "light %d: map %d out of range or not an RGB image", i, l.env_image);
            if (l.type == PG_LIGHT_PROJECTION && !(l.screen[2] > l.screen[0] && l.screen[3] > l.screen[1])) FAIL(PG_ERR_INVALID, "light %d: empty projection screen bounds", i);
        }
    for (int i = 0; i < desc->n_lights; ++i)
        if (desc->lights[i].type < PG_LIGHT_AREA || desc->lights[i].type > PG_LIGHT_GONIO) FAIL(PG_ERR_UNSUPPORTED, "light %d: unknown type %d", i, desc->lights[i].type);
        else if (desc->lights[i].type == PG_LIGHT_AREA && (desc->lights[i].prim < 0 || desc->lights[i].prim >= nt))
            FAIL(PG_ERR_INVALID, "light %d has no emitting triangle", i);
    HIP_TRY_S(s->materials.alloc(sizeof(PgMaterial) * (size_t)desc->n_materials));
    if (desc->n_materials) HIP_TRY_S(hipMemcpy(s->materials.p, devMaterials.data(), s->materials.bytes, hipMemcpyHostToDevice));
    HIP_TRY_S(s->lights.alloc(sizeof(PgLight) * (size_t)desc->n_lights));
    if (desc->n_lights) HIP_TRY_S(hipMemcpy(s->lights.p, desc->lights, s->lights.bytes, hipMemcpyHostToDevice));
    {   // DScene::lightHot
        std::vector<float4> hot((size_t)desc->n_lights * 5, make_float4(0, 0, 0, 0));
        for (int l = 0; l < desc->n_lights; ++l) {
            const PgLight &L = desc->lights[l];
            float ti, pi_, si;
            memcpy(&ti, &L.type, 4); memcpy(&pi_, &L.prim, 4); memcpy(&si, &L.two_sided, 4);
            hot[5 * (size_t)l] = make_float4(ti, pi_, si, L.area);
            hot[5 * (size_t)l + 1] = make_float4(L.L[0], L.L[1], L.L[2], 0.f);
            if (L.type == PG_LIGHT_AREA && L.prim >= 0 && L.prim < nt) for (int k = 0; k < 3; ++k) hot[5 * (size_t)l + 2 + k] = tris[PG_TRI_STRIDE * (size_t)L.prim + k];
        }
        HIP_TRY_S(s->lightHot.alloc(sizeof(float4) * hot.size()));
        if (!hot.empty()) HIP_TRY_S(hipMemcpy(s->lightHot.p, hot.data(), s->lightHot.bytes, hipMemcpyHostToDevice));
    }
    // --- Halton tables
    // (a scene rendered with the Sobol' sampler only may come without one: pg_render then refuses sampler = halton)
    const bool haltonTable = !(desc->n_perm_dims == 0 && desc->sobol_matrices);
    if (haltonTable && (desc->n_perm_dims < 5 || !desc->perms || !desc->perm_sums)) FAIL(PG_ERR_INVALID, "Halton permutation table missing (need >= 5 dims)");
    std::vector<int32_t> primes;
    for (int c = 2; (int)primes.size() < desc->n_perm_dims; ++c) {
        bool is = true;
        for (int p : primes) { if (p * p > c) break; if (c % p == 0) { is = false; break; } }
        if (is) primes.push_back(c);
    }
    for (int i = 0; i < desc->n_perm_dims; ++i)
        if (desc->perm_sums[i + 1] - desc->perm_sums[i] != primes[i]) FAIL(PG_ERR_INVALID, "perm_sums[%d] does not match prime base %d", i, primes[i]);
    if (haltonTable) {
        HIP_TRY_S(s->perms.alloc(sizeof(uint16_t) * (size_t)desc->perm_sums[desc->n_perm_dims]));
        HIP_TRY_S(hipMemcpy(s->perms.p, desc->perms, s->perms.bytes, hipMemcpyHostToDevice));
        HIP_TRY_S(s->permSums.alloc(sizeof(int32_t) * (size_t)(desc->n_perm_dims + 1)));
        HIP_TRY_S(hipMemcpy(s->permSums.p, desc->perm_sums, s->permSums.bytes, hipMemcpyHostToDevice));
        HIP_TRY_S(s->primes.alloc(sizeof(int32_t) * primes.size()));
        HIP_TRY_S(hipMemcpy(s->primes.p, primes.data(), s->primes.bytes, hipMemcpyHostToDevice));
        // per dimension, two int4: (base, offset of its digit permutation, m, L) with floor(a / base) = (t + ((a - t) >> 1)) >> (L - 1),
        // t = mulhi(m, a), for every 32-bit a (division by an invariant integer with a 33-bit multiplier: L = ceil(log2 base),
        // m = floor(2^32 (2^L - base) / base) + 1), and (invBase, invBase * perm[0] / (1 - invBase), 0, 0) as float bits: the two
        // per-dimension constants of ScrambledRadicalInverse (lowdiscrepancy.cpp:411, :422), evaluated here with the same float
        // operations in the same order as the kernels would (contraction off, IEEE division)
        std::vector<int32_t> hd(8 * primes.size(), 0);
        for (size_t i = 0; i < primes.size(); ++i) {
            const uint64_t b = (uint64_t)primes[i];
            int L = 0;
            while (((uint64_t)1 << L) < b) ++L;
            const uint64_t m = (((uint64_t)1 << 32) * (((uint64_t)1 << L) - b)) / b + 1;
            hd[8 * i] = primes[i]; hd[8 * i + 1] = desc->perm_sums[i]; hd[8 * i + 2] = (int32_t)(uint32_t)m; hd[8 * i + 3] = L;
            volatile float invBase = 1.f / (float)(uint32_t)primes[i];
            volatile float t1 = invBase * (float)desc->perms[desc->perm_sums[i]];
            volatile float t2 = 1 - invBase;
            volatile float tail = t1 / t2;
            const float ib = invBase, tl = tail;
            memcpy(&hd[8 * i + 4], &ib, 4); memcpy(&hd[8 * i + 5], &tl, 4);
        }
        HIP_TRY_S(s->haltonDims.alloc(sizeof(int32_t) * hd.size()));
        HIP_TRY_S(hipMemcpy(s->haltonDims.p, hd.data(), s->haltonDims.bytes, hipMemcpyHostToDevice));
    }

    if (desc->cmaxmin) {  // the MaxMinDistSampler's generator matrices
        HIP_TRY_S(s->cmaxmin.alloc(17 * 32 * sizeof(uint32_t)));
        HIP_TRY_S(hipMemcpy(s->cmaxmin.p, desc->cmaxmin, s->cmaxmin.bytes, hipMemcpyHostToDevice));
    }

    d.nodes = (const float4 *)s->nodes.p; d.tris = (const float4 *)s->tris.p; d.spheres = (const PgSphere *)s->spheres.p; d.nSpheres = desc->n_spheres > 0 ? desc->n_spheres : 0;
    d.bxdfs = (const PgBxDF *)s->bxdfs.p;
    if (desc->n_env_floats > 0 && desc->env_tables) {
        HIP_TRY_S(s->envTables.alloc(sizeof(float) * (size_t)desc->n_env_floats));
        HIP_TRY_S(hipMemcpy(s->envTables.p, desc->env_tables, s->envTables.bytes, hipMemcpyHostToDevice));
    }
    d.envTables = (const float *)s->envTables.p;
    {  // alpha masks of triangle meshes
        bool anyAlpha = false;
        for (int k = 0; k < nt && desc->tri_flags; ++k)
            if ((desc->tri_flags[k] & PG_TRI_ALPHA) && !(desc->tri_flags[k] & (PG_PRIM_SPHERE | PG_PRIM_INSTANCE))) {
                anyAlpha = true;
                if (!desc->tri_alpha || !desc->alphas || desc->tri_alpha[k] < 0 || desc->tri_alpha[k] >= desc->n_alphas)
                    FAIL(PG_ERR_INVALID, "triangle %d: PG_TRI_ALPHA without a valid alpha mask", k);
                const PgAlphaMask &am = desc->alphas[desc->tri_alpha[k]];
                if (am.alpha.tex >= desc->n_textures || am.shadow_alpha.tex >= desc->n_textures) FAIL(PG_ERR_INVALID, "triangle %d: alpha texture out of range", k);
            }
        if (anyAlpha) {
            HIP_TRY_S(s->alphas.alloc(sizeof(PgAlphaMask) * (size_t)desc->n_alphas));
            HIP_TRY_S(hipMemcpy(s->alphas.p, desc->alphas, s->alphas.bytes, hipMemcpyHostToDevice));
            HIP_TRY_S(s->triAlpha.alloc(sizeof(int) * (size_t)nt));
            HIP_TRY_S(hipMemcpy(s->triAlpha.p, desc->tri_alpha, s->triAlpha.bytes, hipMemcpyHostToDevice));
        }
        d.alphas = (const PgAlphaMask *)s->alphas.p; d.triAlpha = (const int *)s->triAlpha.p; d.hasAlpha = anyAlpha ? 1 : 0;
        if (anyAlpha) {  // the masks in DAlphaTex form, if every one of them is a constant or a float image map under a (u, v) mapping
            std::vector<DAlphaTex> at((size_t)desc->n_alphas * 2);
            bool simple = getenv("PG_ALPHA_GENERAL") == nullptr;  // (tests: force the general evaluator)
            for (int k = 0; k < desc->n_alphas && simple; ++k)
                for (int which = 0; which < 2; ++which) {
                    const PgAlphaMask &am = desc->alphas[k];
                    const PgTexRef &r = which ? am.shadow_alpha : am.alpha;
                    DAlphaTex &a = at[2 * (size_t)k + which];
                    memset(&a, 0, sizeof(a));
                    if (!(which ? am.has_shadow_alpha : am.has_alpha)) { a.image = -2; continue; }
                    if (r.tex < 0) { a.image = -1; a.constant = r.v[0]; continue; }
                    const PgTexture &tx = desc->textures[r.tex];
                    if (tx.type != PG_TEX_IMAGEMAP || tx.mapping != PG_MAP_UV || tx.image < 0 || tx.image >= desc->n_images) { simple = false; break; }
                    const PgImage &im = desc->images[tx.image];
                    // (27 levels: beyond that MIPMap::Lookup's trilinear branch no longer lands on level 0 for a zero filter width)
                    if (!im.is_float || im.n_levels < 1 || im.n_levels > 26 || im.width < 1 || im.height < 1) { simple = false; break; }
                    a.su = tx.su; a.sv = tx.sv; a.du = tx.du; a.dv = tx.dv;
                    a.width = im.width; a.height = im.height; a.wrap = im.wrap; a.image = tx.image; a.offset = im.level_offset[0];
                }
            if (simple) {
                HIP_TRY_S(s->alphaTex.alloc(sizeof(DAlphaTex) * at.size()));
                HIP_TRY_S(hipMemcpy(s->alphaTex.p, at.data(), s->alphaTex.bytes, hipMemcpyHostToDevice));
                d.alphaTex = (const DAlphaTex *)s->alphaTex.p;
            }
        }
    }
    if (desc->n_bssrdfs > 0) {  // subsurface scattering: the BSSRDFs, which material has one, the beam-diffusion tables
        for (int k = 0; k < desc->n_bssrdfs; ++k) {
            const PgBSSRDF &b = desc->bssrdfs[k];
            const int64_t need = (int64_t)b.n_rho + b.n_radius + 2 * (int64_t)b.n_rho * b.n_radius + b.n_rho;
            if (b.n_rho < 2 || b.n_radius < 2 || b.table < 0 || b.table + need > desc->n_bssrdf_floats || b.match_material < 0 || b.match_material >= desc->n_materials ||
                b.a.tex >= desc->n_textures || b.b.tex >= desc->n_textures)
                FAIL(PG_ERR_INVALID, "BSSRDF %d: table / material / texture out of range", k);
        }
        for (int m = 0; m < desc->n_materials; ++m)
            if (desc->material_bssrdf[m] >= desc->n_bssrdfs) FAIL(PG_ERR_INVALID, "material %d: BSSRDF index out of range", m);
        HIP_TRY_S(s->bssrdfs.alloc(sizeof(PgBSSRDF) * (size_t)desc->n_bssrdfs));
        HIP_TRY_S(hipMemcpy(s->bssrdfs.p, desc->bssrdfs, s->bssrdfs.bytes, hipMemcpyHostToDevice));
        HIP_TRY_S(s->materialBssrdf.alloc(sizeof(int32_t) * (size_t)desc->n_materials));
        HIP_TRY_S(hipMemcpy(s->materialBssrdf.p, desc->material_bssrdf, s->materialBssrdf.bytes, hipMemcpyHostToDevice));
        HIP_TRY_S(s->bssrdfTables.alloc(sizeof(float) * (size_t)desc->n_bssrdf_floats));
        HIP_TRY_S(hipMemcpy(s->bssrdfTables.p, desc->bssrdf_tables, s->bssrdfTables.bytes, hipMemcpyHostToDevice));
        d.bssrdfs = (const PgBSSRDF *)s->bssrdfs.p; d.materialBssrdf = (const int *)s->materialBssrdf.p; d.bssrdfTables = (const float *)s->bssrdfTables.p;
        d.nBssrdfs = desc->n_bssrdfs;
    }
    {  // participating media (HomogeneousMedium) and the primitives' MediumInterfaces
        s->nMedia = desc->n_media > 0 ? desc->n_media : 0;
        if (s->nMedia > 0 && !desc->media) FAIL(PG_ERR_INVALID, "n_media = %d without a media table", desc->n_media);
        if ((desc->tri_medium_inside != nullptr) != (desc->tri_medium_outside != nullptr)) FAIL(PG_ERR_INVALID, "tri_medium_inside and tri_medium_outside go together");
        if (desc->tri_medium_inside)
            for (int k = 0; k < nt; ++k)
                if (desc->tri_medium_inside[k] < -1 || desc->tri_medium_inside[k] >= s->nMedia || desc->tri_medium_outside[k] < -1 || desc->tri_medium_outside[k] >= s->nMedia)
                    FAIL(PG_ERR_INVALID, "primitive %d: medium index out of range", k);
        if (s->nMedia > 0) {
            HIP_TRY_S(s->media.alloc(sizeof(PgMedium) * (size_t)s->nMedia));
            HIP_TRY_S(hipMemcpy(s->media.p, desc->media, s->media.bytes, hipMemcpyHostToDevice));
        }
        if (desc->tri_medium_inside) {
            HIP_TRY_S(s->triMediumIn.alloc(sizeof(int) * (size_t)nt));
            HIP_TRY_S(s->triMediumOut.alloc(sizeof(int) * (size_t)nt));
            HIP_TRY_S(hipMemcpy(s->triMediumIn.p, desc->tri_medium_inside, s->triMediumIn.bytes, hipMemcpyHostToDevice));
            HIP_TRY_S(hipMemcpy(s->triMediumOut.p, desc->tri_medium_outside, s->triMediumOut.bytes, hipMemcpyHostToDevice));
        }
        d.media = (const PgMedium *)s->media.p; d.triMediumIn = (const int *)s->triMediumIn.p; d.triMediumOut = (const int *)s->triMediumOut.p;
        if (desc->n_grids > 0) {  // GridDensityMedium: the grids, which medium has one, the density values
            int64_t nDensity = 0;
            for (int k = 0; k < desc->n_grids; ++k) {
                const PgDensityGrid &g = desc->grids[k];
                if (g.nx < 1 || g.ny < 1 || g.nz < 1 || g.density_offset < 0 || !(g.sigma_t > 0) || !(g.inv_max_density > 0))
                    FAIL(PG_ERR_INVALID, "grid medium %d: malformed (%d x %d x %d, sigma_t %g)", k, g.nx, g.ny, g.nz, (double)g.sigma_t);
                nDensity = std::max<int64_t>(nDensity, g.density_offset + (int64_t)g.nx * g.ny * g.nz);
            }
            if (nDensity > desc->n_density_floats) FAIL(PG_ERR_INVALID, "grid media: %lld density values, %lld given", (long long)nDensity, (long long)desc->n_density_floats);
            for (int m = 0; m < s->nMedia; ++m) if (desc->media_grid[m] >= desc->n_grids) FAIL(PG_ERR_INVALID, "medium %d: grid index out of range", m);
            HIP_TRY_S(s->grids.alloc(sizeof(PgDensityGrid) * (size_t)desc->n_grids));
            HIP_TRY_S(hipMemcpy(s->grids.p, desc->grids, s->grids.bytes, hipMemcpyHostToDevice));
            HIP_TRY_S(s->mediaGrid.alloc(sizeof(int32_t) * (size_t)std::max(1, s->nMedia)));
            if (s->nMedia > 0) HIP_TRY_S(hipMemcpy(s->mediaGrid.p, desc->media_grid, sizeof(int32_t) * (size_t)s->nMedia, hipMemcpyHostToDevice));
            HIP_TRY_S(s->gridDensity.alloc(sizeof(float) * (size_t)nDensity));
            HIP_TRY_S(hipMemcpy(s->gridDensity.p, desc->grid_density, s->gridDensity.bytes, hipMemcpyHostToDevice));
            d.grids = (const PgDensityGrid *)s->grids.p; d.mediaGrid = (const int *)s->mediaGrid.p; d.gridDensity = (const float *)s->gridDensity.p;
            d.nGrids = desc->n_grids;
        }
    }
    if (desc->noise_perm) {  // NoisePerm of the Perlin-noise textures
        for (int i = 0; i < 512; ++i) if (desc->noise_perm[i] < 0 || desc->noise_perm[i] > 255) FAIL(PG_ERR_INVALID, "noise_perm[%d] = %d is not a byte", i, desc->noise_perm[i]);
        HIP_TRY_S(s->noisePerm.alloc(sizeof(int) * 512));
        HIP_TRY_S(hipMemcpy(s->noisePerm.p, desc->noise_perm, s->noisePerm.bytes, hipMemcpyHostToDevice));
    }
    d.noisePerm = (const int *)s->noisePerm.p;
    if (desc->sobol_matrices) {  // SobolSampler tables
        if (!desc->vdc_sobol || !desc->vdc_sobol_inv) FAIL(PG_ERR_INVALID, "sobol_matrices without vdc_sobol / vdc_sobol_inv");
        HIP_TRY_S(s->sobolMatrices.alloc(sizeof(uint32_t) * 1024 * 52));
        HIP_TRY_S(s->vdcSobol.alloc(sizeof(uint64_t) * 25 * 52));
        HIP_TRY_S(s->vdcSobolInv.alloc(sizeof(uint64_t) * 26 * 52));
        HIP_TRY_S(hipMemcpy(s->sobolMatrices.p, desc->sobol_matrices, s->sobolMatrices.bytes, hipMemcpyHostToDevice));
        HIP_TRY_S(hipMemcpy(s->vdcSobol.p, desc->vdc_sobol, s->vdcSobol.bytes, hipMemcpyHostToDevice));
        HIP_TRY_S(hipMemcpy(s->vdcSobolInv.p, desc->vdc_sobol_inv, s->vdcSobolInv.bytes, hipMemcpyHostToDevice));
    }
    d.sobolMatrices = (const uint32_t *)s->sobolMatrices.p; d.vdcSobol = (const uint64_t *)s->vdcSobol.p; d.vdcSobolInv = (const uint64_t *)s->vdcSobolInv.p;
    d.images = (const PgImage *)s->images.p; d.texels = (const float *)s->texels.p; d.ewaLut = (const float *)s->ewaLut.p;
    d.textures = (const PgTexture *)s->textures.p; d.textured = (const PgTexturedMaterial *)s->textured.p;
    d.hasTextured = anyTextured ? 1 : 0;
    s->matStride = 0;
    {   // k_material (materials evaluated ahead of the shading launch): room per hit = the longest BxDF list a material of this scene can
        // build, counted as the materials' ComputeScatteringFunctions add them (materials/*.cpp; MatEval, pg_kernels.hip).  Not for scenes
        // with BSSRDF materials or grid media (their kernels evaluate inside), PG_MAT_PRE=0: nowhere.
        const char *mp = getenv("PG_MAT_PRE");
        if (anyTextured && desc->n_bssrdfs == 0 && d.nGrids == 0 && !d.hasNest && !(mp && atoi(mp) == 0)) {  // (hasNest: MODE 2 carries the second transform)
            std::function<int(int, int)> lobes = [&](int mi, int depth) -> int {
                if (mi < 0 || mi >= desc->n_materials) return 0;
                const PgMaterial &m = desc->materials[mi];
                if (m.type != PG_MAT_TEXTURED) return std::min(std::max(m.n_bxdfs, 0), PG_MAX_BXDFS);
                const PgTexturedMaterial &tm = desc->textured[m.textured_index];
                switch (tm.kind) {
                case PG_KIND_MATTE: case PG_KIND_MIRROR: case PG_KIND_METAL: case PG_KIND_SUBSTRATE: return 1;
                case PG_KIND_PLASTIC: case PG_KIND_GLASS: return 2;
                case PG_KIND_UBER: return 5;
                case PG_KIND_TRANSLUCENT: return 4;
                case PG_KIND_MIX: return depth >= 3 ? PG_MAX_BXDFS : std::min(PG_MAX_BXDFS, lobes(tm.sub[0], depth + 1) + lobes(tm.sub[1], depth + 1));
                }
                return PG_MAX_BXDFS;
            };
            // in RECORDS of 48 B (LobeBsdfT, pg_kernels.hip): one per BxDF, two where the hit's material is a mix (the ScaledBxDF factors)
            int stride = 1;
            for (int i = 0; i < desc->n_materials; ++i)
                if (desc->materials[i].type == PG_MAT_TEXTURED)
                    stride = std::max(stride, lobes(i, 0) * (desc->textured[desc->materials[i].textured_index].kind == PG_KIND_MIX ? 2 : 1));
            s->matStride = stride;
            // the constant lists once more as PkLobe records, for the kernel that reads k_material's (k_shade<3>)
            std::vector<float> pk;
            std::vector<int2> matPk((size_t)desc->n_materials, make_int2(0, 1));
            for (int i = 0; i < desc->n_materials; ++i) {
                const PgMaterial &m = desc->materials[i];
                if (m.type == PG_MAT_TEXTURED || m.n_bxdfs <= 0 || !desc->bxdfs) continue;
                bool scaled = false;
                for (int k = 0; k < m.n_bxdfs; ++k) scaled |= desc->bxdfs[m.first_bxdf + k].n_scales > 0;
                matPk[i] = make_int2((int)(pk.size() / 12), scaled ? 2 : 1);
                for (int k = 0; k < m.n_bxdfs; ++k) {
                    const PgBxDF &b = desc->bxdfs[m.first_bxdf + k];
                    pk.resize(pk.size() + (scaled ? 24 : 12));
                    float *q = pk.data() + pk.size() - (scaled ? 24 : 12);
                    pg_pack_lobe(b, q);
                    if (scaled) pg_pack_lobe_scales(b, q + 12);
                }
            }
            if (pk.empty()) pk.resize(12, 0.f);
            HIP_TRY_S(s->bxdfsPk.alloc(pk.size() * sizeof(float)));
            HIP_TRY_S(hipMemcpy(s->bxdfsPk.p, pk.data(), s->bxdfsPk.bytes, hipMemcpyHostToDevice));
            HIP_TRY_S(s->matPk.alloc(matPk.size() * sizeof(int2)));
            HIP_TRY_S(hipMemcpy(s->matPk.p, matPk.data(), s->matPk.bytes, hipMemcpyHostToDevice));
            d.bxdfsPk = (const float4 *)s->bxdfsPk.p; d.matPk = (const int2 *)s->matPk.p;
        }
    }
    {   // shading classes (k_shade_order): scenes whose materials evaluate textures / BxDF lists shade grouped by material.  Few
        // materials: each is a class (its textures stay with its waves as well); many: materials that run the same code
        // (type, kind, bump) share one.  PG_SHADE_ORDER=0: queue order, as scenes without such materials are shaded.
        const char *so = getenv("PG_SHADE_ORDER");
        s->volOrder = s->nMedia > 0 && d.nGrids == 0 && !(so && atoi(so) == 0);
        if (d.hasTextured && desc->n_materials > 1 && !(so && atoi(so) == 0)) {
            const int nClasses = PG_ORDER_CLASSES - 3;  // 0 .. 12; 13 = scattered in a medium (volpath), 14 = the ray escaped, 15 = no entry
            std::vector<unsigned char> matClass((size_t)desc->n_materials, 0);
            if (desc->n_materials <= nClasses) for (int i = 0; i < desc->n_materials; ++i) matClass[i] = (unsigned char)i;
            else {
                std::vector<int> sigs;
                for (int i = 0; i < desc->n_materials; ++i) {
                    const PgMaterial &m = desc->materials[i];
                    int sig = m.type;
                    if (m.type == PG_MAT_TEXTURED) { const PgTexturedMaterial &tm = desc->textured[m.textured_index]; sig |= (tm.kind << 8) | (tm.has_bump ? 1 << 16 : 0); }
                    size_t k = 0;
                    while (k < sigs.size() && sigs[k] != sig) ++k;
                    if (k == sigs.size()) sigs.push_back(sig);
                    matClass[i] = (unsigned char)(k % (size_t)nClasses);
                }
            }
            std::vector<unsigned char> pc((size_t)nt);
            for (int k = 0; k < nt; ++k) pc[k] = matClass[desc->tri_material ? desc->tri_material[k] : 0];
            HIP_TRY_S(s->primClass.alloc(pc.size()));
            HIP_TRY_S(hipMemcpy(s->primClass.p, pc.data(), pc.size(), hipMemcpyHostToDevice));
            d.primClass = (const unsigned char *)s->primClass.p;
        }
    }
    {  // PG_FORCE_EXT=1 runs the general kernels on scenes that do not need them (tests: both paths agree bit for bit)
        const char *fe = getenv("PG_FORCE_EXT");
        d.ext = (d.hasTextured || d.nSpheres > 0 || d.nInstances > 0 || d.hasInfinite || anyImageLight || anyLobeMaterial || (fe && atoi(fe) != 0)) ? 1 : 0;
    }
    d.triAttr = (const float4 *)s->triAttr.p; d.triS = (const float4 *)s->triS.p; d.attrN = anyN ? 1 : 0; d.attrUV = anyUV ? 1 : 0; d.alphaUV = (const float *)s->uv.p;
    d.materials = (const PgMaterial *)s->materials.p; d.lights = (const PgLight *)s->lights.p;
    d.nNodes = desc->n_nodes; d.nTris = nt; d.nLights = desc->n_lights; d.nMaterials = desc->n_materials;
    d.perms = (const uint16_t *)s->perms.p; d.permSums = (const int32_t *)s->permSums.p; d.primes = (const int32_t *)s->primes.p; d.haltonDims = (const int4 *)s->haltonDims.p;
    d.nPermDims = desc->n_perm_dims;
    d.lightHot = (const float4 *)s->lightHot.p;
    d.cmaxmin = (const uint32_t *)s->cmaxmin.p;
    d.lightStrategy = desc->light_strategy;
    // --- light sampling distributions (lightdistrib.cpp)
    const int nl = desc->n_lights;
    if (nl > 0) {
        const size_t stride = 2 * (size_t)nl + 2;
        if (desc->light_strategy == PG_LIGHTS_SPATIAL) {
            // SpatialLightDistribution ctor, lightdistrib.cpp:96-125 (maxVoxels = 64)
            // a scene without geometry has no bound to divide into voxels and no surface to look a voxel up from: one voxel
            PgBVHNode root;
            memset(&root, 0, sizeof(root));
            if (desc->n_nodes > 0 && desc->nodes) root = desc->nodes[0];
            float diag[3];
            for (int i = 0; i < 3; ++i) { d.bmin[i] = root.bmin[i]; d.bmax[i] = root.bmax[i]; diag[i] = root.bmax[i] - root.bmin[i]; }
            int me = (diag[0] > diag[1] && diag[0] > diag[2]) ? 0 : ((diag[1] > diag[2]) ? 1 : 2);
            float bmax = diag[me];
            size_t total = 1;
            for (int i = 0; i < 3; ++i) {
                int nv = bmax > 0 ? (int)roundf(diag[i] / bmax * 64) : 1;
                d.nVoxels[i] = nv > 1 ? nv : 1;
                total *= (size_t)d.nVoxels[i];
            }
            // Dense table (every voxel up front: they are pure functions of the voxel) while it is small; beyond that the
            // voxels are computed on first touch like the reference's hash table (lightdistrib.cpp:135-230), into a pool.
            size_t denseLimit = (size_t)1 << 30;
            if (const char *e = getenv("PG_SPARSE_LIGHTS")) { if (atoi(e) != 0) denseLimit = 0; }
            if (total * stride * sizeof(float) > denseLimit) {
                const size_t budget = (size_t)16 << 30;
                s->nVoxelsTotal = (int)total;
                s->poolSlots = (int)std::min<size_t>(total, std::max<size_t>(1, budget / (stride * sizeof(float))));
                HIP_TRY_S(s->distTable.alloc((size_t)s->poolSlots * stride * sizeof(float)));
                HIP_TRY_S(s->voxelSlot.alloc(total * sizeof(int)));
                HIP_TRY_S(hipMemset(s->voxelSlot.p, 0xff, total * sizeof(int)));  // -1: not requested
                HIP_TRY_S(s->voxelRequests.alloc(total * sizeof(int)));
                HIP_TRY_S(s->voxelCounters.alloc(2 * sizeof(int)));
                HIP_TRY_S(hipMemset(s->voxelCounters.p, 0, 2 * sizeof(int)));
                d.distTable = (const float *)s->distTable.p;
                d.sparseLights = 1;
                // (the exit vertices of subsurface paths look their light distribution up without the deferral the shading kernel has)
                if (d.nBssrdfs > 0) FAIL(PG_ERR_UNSUPPORTED, "subsurface scattering with a spatial light distribution beyond the dense table's size: use \"lightsamplestrategy\" \"power\" or \"uniform\"");
                if (d.nGrids > 0) FAIL(PG_ERR_UNSUPPORTED, "a grid medium with a spatial light distribution beyond the dense table's size: use \"lightsamplestrategy\" \"power\" or \"uniform\"");
                d.voxelSlot = (int *)s->voxelSlot.p; d.voxelRequests = (int *)s->voxelRequests.p; d.voxelCounters = (int *)s->voxelCounters.p;
            } else {
                HIP_TRY_S(s->distTable.alloc(total * stride * sizeof(float)));
                d.distTable = (const float *)s->distTable.p;
                launch_light_tables(d, (float *)s->distTable.p, (int)total, 0);
                HIP_TRY_S(hipGetLastError());
                HIP_TRY_S(hipDeviceSynchronize());
            }
        } else {
            // UniformLightDistribution (lightdistrib.cpp:68-71) / PowerLightDistribution (integrator.cpp:217-225, diffuse.cpp:64-66)
            std::vector<float> tab(stride);
            float *func = tab.data(), *cdf = func + nl;
            for (int i = 0; i < nl; ++i) {
                if (desc->light_strategy == PG_LIGHTS_POWER) {
                    const PgLight &l = desc->lights[i];
                    float P[3];  // Light::Power(): diffuse.cpp:64-66, point.cpp:54, spot.cpp:74-76, distant.cpp:62-64
                    for (int c = 0; c < 3; ++c) {
                        float v = l.L[c];
                        if (l.type == PG_LIGHT_POINT) v *= 4 * PG_PI;
                        else if (l.type == PG_LIGHT_SPOT) { v *= 2; v *= PG_PI; v *= (1 - .5f * (l.cos_falloff_start + l.cos_total_width)); }
                        else if (l.type == PG_LIGHT_DISTANT) { v *= PG_PI; v *= l.world_radius; v *= l.world_radius; }
                        else if (l.type == PG_LIGHT_PROJECTION) { v = l.env_power[c] * v; v *= 2; v *= PG_PI; v *= (1.f - l.cos_total_width); }  // projection.cpp:93-99
                        else if (l.type == PG_LIGHT_GONIO) v = (v * (4 * PG_PI)) * l.env_power[c];  // goniometric.cpp:54-58
                        else if (l.type == PG_LIGHT_INFINITE) v = l.env_power[c] * (PG_PI * l.world_radius * l.world_radius);  // infinite.cpp:87-91
                        else { v *= (float)(l.two_sided ? 2 : 1); v *= l.area; v *= PG_PI; }
                        P[c] = v;
                    }
                    func[i] = 0.212671f * P[0] + 0.715160f * P[1] + 0.072169f * P[2];
                } else func[i] = 1;
            }
            cdf[0] = 0;
            for (int i = 1; i < nl + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / nl;
            float funcInt = cdf[nl];
            if (funcInt == 0) { for (int i = 1; i < nl + 1; ++i) cdf[i] = (float)i / (float)nl; }
            else { for (int i = 1; i < nl + 1; ++i) cdf[i] /= funcInt; }
            tab[2 * nl + 1] = funcInt;
            HIP_TRY_S(s->distTable.alloc(stride * sizeof(float)));
            HIP_TRY_S(hipMemcpy(s->distTable.p, tab.data(), stride * sizeof(float), hipMemcpyHostToDevice));
            d.distTable = (const float *)s->distTable.p;
        }
    }
    HIP_TRY_S(s->traceCn.alloc(sizeof(TraceCounters) * 3));  // closest hit, any hit, and launches that are not the reference's (second walk of a BSSRDF probe chain)
    HIP_TRY_S(hipMemset(s->traceCn.p, 0, s->traceCn.bytes));
    HIP_TRY_S(s->cursors.alloc(2 * PG_REGIONS * PG_COUNT_STRIDE * sizeof(int)));
    HIP_TRY_S(s->cursors2.alloc(2 * PG_REGIONS * PG_COUNT_STRIDE * sizeof(int)));
    HIP_TRY_S(hipStreamCreateWithFlags(&s->shadowStream, hipStreamNonBlocking));
    HIP_TRY_S(hipEventCreateWithFlags(&s->evShaded, hipEventDisableTiming));
    HIP_TRY_S(hipEventCreateWithFlags(&s->evShadowed, hipEventDisableTiming));
    if (const char *e = getenv("PG_OVERLAP_SHADOW")) s->overlapShadow = atoi(e) != 0;
    HIP_TRY_S(s->cullGuard.alloc(sizeof(int) * 2 + 8 * sizeof(unsigned long long)));  // the guard word (+ the counters of the PG_TRACE_STATS experiment build)
    HIP_TRY_S(hipMemset(s->cullGuard.p, 0, s->cullGuard.bytes));
    HIP_TRY_S(s->lightTests.alloc(sizeof(unsigned long long) * PG_LIGHT_TEST_SHARDS * PG_LIGHT_TEST_STRIDE));
    HIP_TRY_S(hipMemset(s->lightTests.p, 0, s->lightTests.bytes));
    HIP_TRY_S(s->sceneCopy.alloc(sizeof(DScene)));
    d.self = (const DScene *)s->sceneCopy.p;
    HIP_TRY_S(hipMemcpy(s->sceneCopy.p, &d, sizeof(DScene), hipMemcpyHostToDevice));
    *out = s;
    return PG_OK;
#undef FAIL
#undef HIP_TRY_S
}

static void traceClosest(PgScene *s, RayQueue q, float4 *hits, float *tOut, TraceCounters *cn, hipStream_t st) {
    DScene d = s->d;
    d.hitInst = nullptr; d.animXf = nullptr; d.rayTimes = 0;  // the unit entry points report primitive, t and barycentrics only; their rays have time 0
    launch_closest(d, s->trace, q, hits, tOut, cn, (int *)s->cursors.p, (int *)s->cullGuard.p, st);
}
static void traceAnyhit(PgScene *s, RayQueue q, int *occluded, TraceCounters *cn, hipStream_t st) {
    // the unit entry point answers in the reference's visiting order: its counters are the reference's (tests compare them)
    TraceConfig c = s->trace;
    c.anyhitFree = 0;
    DScene d = s->d;
    d.rayTimes = 0;  // the unit entry points' rays have time 0 (their queues carry no times)
    launch_anyhit(d, c, q, occluded, cn, (int *)s->cursors.p, st);
}

// k_trace's early-cull margin is exact while no ray accepts more than TR_MAX_ACCEPTED hits (pg_traverse.hip); otherwise
// the kernel raises this flag and the call fails instead of returning a possibly different image.
static int checkCullGuard(PgScene *s) {
    int g = 0;
    HIP_TRY(hipMemcpy(&g, s->cullGuard.p, sizeof(int), hipMemcpyDeviceToHost));
    if (g) {
        HIP_TRY(hipMemset(s->cullGuard.p, 0, sizeof(int)));
        s->cullTripped = true;  // the entry points then repeat the call without the margin (exact by construction, slower)
        return setError(PG_ERR_OVERFLOW, "a ray accepted more than 4096 successive hits: the far-child cull margin is no longer provably exact");
    }
    return PG_OK;
}
// Runs `call`; if a ray outran the early-cull margin's proof (sorted stacks of alpha cards can do that), runs it again with the
// margin disabled -- every far child then waits on the stack for the reference's own test at pop time -- with the counters put
// back to where they were, so the result and its statistics are those of one exact pass.
static int withExactFallback(PgScene *s, const std::function<int()> &call) {
    const PgCounters saved = s->counters;
    TraceCounters savedDev[2];
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemcpy(savedDev, s->traceCn.p, sizeof(savedDev), hipMemcpyDeviceToHost));
    std::vector<unsigned long long> savedLt(PG_LIGHT_TEST_SHARDS * PG_LIGHT_TEST_STRIDE);
    HIP_TRY(hipMemcpy(savedLt.data(), s->lightTests.p, s->lightTests.bytes, hipMemcpyDeviceToHost));
    s->cullTripped = false;
    int st = call();
    if (st != PG_ERR_OVERFLOW || !s->cullTripped) return st;
    s->counters = saved;
    HIP_TRY(hipMemcpy(s->traceCn.p, savedDev, sizeof(savedDev), hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(s->lightTests.p, savedLt.data(), s->lightTests.bytes, hipMemcpyHostToDevice));
    const TraceConfig cfg = s->trace;  // per scene: another host thread's scene (pg_render_sharded) keeps its own margin
    s->trace.cullK = 3e38f;
    s->trace.maxAccepted = 0x7fffffff;  // nothing to guard: no far child is culled early any more
    s->cullTripped = false;
    st = call();
    s->trace = cfg;
    return st;
}

static int tileCount(const PgRenderDesc *rd) {
    int nx = (rd->sample_bounds[2] - rd->sample_bounds[0] + 15) / 16, ny = (rd->sample_bounds[3] - rd->sample_bounds[1] + 15) / 16;
    if (nx <= 0 || ny <= 0 || rd->tile_step <= 0 || rd->tile_first < 0) return 0;
    int total = nx * ny;
    return rd->tile_first >= total ? 0 : (total - rd->tile_first + rd->tile_step - 1) / rd->tile_step;
}
int pg_render_tile_count(const PgRenderDesc *desc) {
    if (!desc) return setError(PG_ERR_INVALID, "pg_render_tile_count: null argument");
    return tileCount(desc);
}

// Queue geometry for a batch of `capacity` path slots: PG_REGIONS regions of regionCap entries (multiple of 256) such
// that the blocks of one XCD (b % 8) can never overflow their region.
// `slack`: twice the room (sparse light tables: entries re-shaded in a second pass append from other blocks than their own)
static int regionCapFor(int capacity, bool slack = false) {
    int nblk = (capacity + 255) / 256;
    return ((nblk + PG_REGIONS - 1) / PG_REGIONS) * 256 * (slack ? 2 : 1);
}
static int ensureWorkBuffers(PgScene *s, int capacity) {
    if (s->capacity >= capacity) return PG_OK;
    const size_t n = (size_t)regionCapFor(capacity, s->d.sparseLights != 0) * PG_REGIONS;  // >= capacity
    if (s->d.sparseLights) HIP_TRY(s->retryList.alloc(n * sizeof(int)));
    // (scenes with moving instances: one float per entry behind `d`, the rays' times: PG_QUEUE_TIMES)
    for (int i = 0; i < 4; ++i) { HIP_TRY(s->qo[i].alloc(n * sizeof(float4))); HIP_TRY(s->qd[i].alloc(n * (sizeof(float4) + (s->d.hasMotion ? sizeof(float) : 0)))); }
    HIP_TRY(s->counts.alloc(4 * PG_REGIONS * PG_COUNT_STRIDE * sizeof(int)));
    // (grid media: a third part -- the transmittance rays must leave the main rays' hits alone for the second shading phase)
    const size_t hitParts = s->d.nGrids > 0 ? 3 : 2;
    HIP_TRY(s->hitsMain.alloc(hitParts * n * sizeof(float4)));
    if (s->d.primClass || s->volOrder) HIP_TRY(s->shadeOrder.alloc(n * sizeof(int)));
    if (s->volOrder) HIP_TRY(s->volPre.alloc(n * sizeof(float2)));
    if (s->matStride > 0) {  // k_material's lists and frames, one set per main-queue entry; no room: the shading kernel evaluates materials itself
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { freeB = 0; (void)hipGetLastError(); }
        const size_t recBytes = 3 * sizeof(float4);  // one packed BxDF record
        const size_t want = n * ((size_t)s->matStride * recBytes + 2 * sizeof(float4)), have = s->matLobes.bytes + s->matHead.bytes;
        s->matLobes.release(); s->matHead.release();
        // (everything else of this function and the integrator's own state -- about 600 B per slot -- is still to be allocated)
        // (a list's record index r * plane + p is a 32-bit product in the kernels: plane * stride must stay below 2^31)
        const bool fits = want + n * 700 + ((size_t)1 << 30) <= freeB + have && n * (size_t)s->matStride < ((size_t)1 << 31);
        if (!fits || s->matLobes.alloc(n * (size_t)s->matStride * recBytes) != hipSuccess || s->matHead.alloc(n * 2 * sizeof(float4)) != hipSuccess) {
            s->matLobes.release(); s->matHead.release(); (void)hipGetLastError();
        }
    }
    if (s->d.nInstances > 0) { HIP_TRY(s->hitInst.alloc(hitParts * n * sizeof(int))); s->d.hitInst = (int *)s->hitInst.p; }
    if (s->d.hasMotion) {  // InterpolatedPrimToWorld per closest-hit result on a moving instance (pg_motion.h)
        HIP_TRY(s->animXf.alloc((s->d.hasNest ? 2 : 1) * hitParts * n * PG_XF_STRIDE * sizeof(float)));  // (hasNest: a second half for the inner transform)
        s->d.animXf = (float *)s->animXf.p;
        s->d.nestXfOff = (int)(hitParts * n);
    }  // main-queue hits, then MIS-queue hits at offset n (one launch fills both)
    HIP_TRY(s->occluded.alloc(n * sizeof(int)));
    HIP_TRY(s->stL.alloc(n * sizeof(float4)));
    HIP_TRY(s->stBeta.alloc(n * sizeof(float4)));
    HIP_TRY(s->stMeta.alloc(n * sizeof(int4)));
    HIP_TRY(s->pdLight.alloc(n * sizeof(float4)));
    HIP_TRY(s->pdMis.alloc(n * sizeof(float4)));
    HIP_TRY(s->pdBeta.alloc(n * sizeof(float4)));
    HIP_TRY(s->pdInfo.alloc(n * sizeof(int4)));
    s->capacity = capacity;
    return PG_OK;
}

static hipEvent_t getEvent(PgScene *s, size_t idx) {
    while (s->events.size() <= idx) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) return nullptr;
        s->events.push_back(e);
    }
    return s->events[idx];
}

static int renderFrame(PgScene *s, const PgRenderDesc *rd, PgFilmPixel *film, PgStraySample *strays, int32_t maxStrays, int32_t *nStrays, int mem,
                       void *streamPtr);
int pg_render(PgScene *s, const PgRenderDesc *rd, PgFilmPixel *film, PgStraySample *strays, int32_t maxStrays, int32_t *nStrays,
              int mem, void *streamPtr) {
    if (!s || !rd || !film || !nStrays || (maxStrays > 0 && !strays)) return setError(PG_ERR_INVALID, "pg_render: null argument");
    return withExactFallback(s, [&]() { return renderFrame(s, rd, film, strays, maxStrays, nStrays, mem, streamPtr); });
}
static int renderFrame(PgScene *s, const PgRenderDesc *rd, PgFilmPixel *film, PgStraySample *strays, int32_t maxStrays, int32_t *nStrays, int mem,
                       void *streamPtr) {
    if (rd->abi_version != PG_ABI_VERSION) return setError(PG_ERR_INVALID, "ABI version %d, expected %d", rd->abi_version, PG_ABI_VERSION);
    if (rd->filter_radius[0] <= 0 || rd->filter_radius[1] <= 0) return setError(PG_ERR_INVALID, "pg_render: filter radius must be positive");
    if (!rd->filter_general && (rd->filter_radius[0] > 0.5f || rd->filter_radius[1] > 0.5f || rd->tile_pixels != 256))
        return setError(PG_ERR_INVALID, "pg_render: filter_general = 0 is the box filter of radius <= 0.5 with 256-entry tile blocks");
    if (rd->filter_general && rd->tile_pixels != (16 + rd->tile_halo[0] + rd->tile_halo[2]) * (16 + rd->tile_halo[1] + rd->tile_halo[3]))
        return setError(PG_ERR_INVALID, "pg_render: tile_pixels does not match tile_halo");
    if (rd->spp <= 0 || rd->max_depth < 0 || rd->tile_step <= 0) return setError(PG_ERR_INVALID, "pg_render: bad spp/maxdepth/tile_step");
    if (rd->integrator != 0 && rd->integrator != 1) return setError(PG_ERR_INVALID, "pg_render: integrator %d (0 = path, 1 = volpath)", rd->integrator);
    const bool vol = rd->integrator == 1;
    // (overlapShadow: the environment's PG_OVERLAP_SHADOW at pg_scene_create, then pg_scene_set_option -- a caller such as bench.py times
    // frames with the overlap and takes per-kernel times from a serialised frame of the same scene)
    if (rd->camera_medium < -1 || rd->camera_medium >= s->nMedia) return setError(PG_ERR_INVALID, "pg_render: camera_medium %d out of range", rd->camera_medium);
    if (rd->sampler < PG_SAMPLER_HALTON || rd->sampler > PG_SAMPLER_MAXMINDIST) return setError(PG_ERR_INVALID, "pg_render: sampler %d (PgSamplerKind 0 .. 5)", rd->sampler);
    // The PixelSamplers (stratified, 02sequence, maxmindist) fall back to their tile's RNG stream only for draws beyond their
    // "dimensions" (sampler.cpp:108-134).  PathIntegrator::Li draws at most 1 + 2 maxdepth one-dimensional numbers (time; light choice and
    // roulette per vertex) and 2 + 3 maxdepth two-dimensional ones (film, lens; uLight, uScattering, the next direction per vertex):
    // with that many sampled dimensions StartPixel alone consumes the stream, every pixel's arrays can be generated ahead, and the
    // paths run as one wavefront like the GlobalSamplers' (tsBatched).  Not for volpath (a ray through material-less surfaces samples
    // its medium an unbounded number of times), materials with a BSSRDF, or sparse light tables (their deferred vertices re-draw).
    bool tsBatched = false;
    if (rd->sampler > PG_SAMPLER_RANDOM && rd->integrator == 0 && s->d.nBssrdfs == 0 && !s->d.sparseLights && rd->sampler_dims <= 63 &&
        rd->sampler_dims >= 2 + 3 * (long long)rd->max_depth && !(getenv("PG_TS_BATCHED") && atoi(getenv("PG_TS_BATCHED")) == 0)) {
        // The arrays of all local tiles are one allocation: taken here, before any state is set, so that a device without the room
        // (less memory free, a large scene beside them) renders tile by tile as before instead of failing with PG_ERR_DEVICE.
        const size_t nArr = (size_t)tileCount(rd) * 256 * (size_t)rd->sampler_dims * (size_t)rd->spp;
        size_t freeB = 0, totalB = 0;
        if (hipMemGetInfo(&freeB, &totalB) != hipSuccess) { freeB = 0; (void)hipGetLastError(); }
        const size_t have = s->ts1.bytes + s->ts2.bytes;  // (buffers of an earlier frame are given back first)
        if (nArr * 12 <= ((size_t)48 << 30) && nArr * 12 + ((size_t)2 << 30) <= freeB + have) {
            s->ts1.release(); s->ts2.release();
            tsBatched = s->ts1.alloc(sizeof(float) * (nArr + 1)) == hipSuccess && s->ts2.alloc(sizeof(float) * 2 * (nArr + 1)) == hipSuccess;
            if (!tsBatched) { s->ts1.release(); s->ts2.release(); (void)hipGetLastError(); }
        }
    }
    const bool tileSerial = rd->sampler >= PG_SAMPLER_RANDOM && !tsBatched;
    if (rd->sampler >= PG_SAMPLER_RANDOM) {
        if (rd->sampler != PG_SAMPLER_RANDOM && (rd->sampler_dims < 0 || rd->sampler_dims > 4096)) return setError(PG_ERR_INVALID, "pg_render: sampler_dims %d", rd->sampler_dims);
        if (rd->sampler == PG_SAMPLER_STRATIFIED && (rd->strat_samples[0] < 1 || rd->strat_samples[1] < 1 || rd->strat_samples[0] * rd->strat_samples[1] != rd->spp))
            return setError(PG_ERR_INVALID, "pg_render: stratified sampler %d x %d samples, spp %d", rd->strat_samples[0], rd->strat_samples[1], rd->spp);
        if ((rd->sampler == PG_SAMPLER_ZEROTWO || rd->sampler == PG_SAMPLER_MAXMINDIST) && (rd->spp & (rd->spp - 1)))
            return setError(PG_ERR_INVALID, "pg_render: sampler %d needs a power-of-two spp (the reference rounds up), got %d", rd->sampler, rd->spp);
        if (rd->sampler == PG_SAMPLER_MAXMINDIST && (!s->cmaxmin.p || rd->sampler_dims < 1 || rd->spp >= (1 << 17)))
            return setError(PG_ERR_INVALID, "pg_render: maxmindist needs PgSceneDesc.cmaxmin, sampler_dims >= 1 and spp < 2^17");
    }
    if (rd->sampler == 1) {
        if (!s->d.sobolMatrices) return setError(PG_ERR_INVALID, "pg_render: sampler = sobol, but the scene was created without the Sobol' tables");
        if (rd->sobol_log2_resolution < 0 || rd->sobol_log2_resolution > 26 || rd->sobol_resolution != (1 << rd->sobol_log2_resolution))
            return setError(PG_ERR_INVALID, "pg_render: sobol_resolution %d / sobol_log2_resolution %d", rd->sobol_resolution, rd->sobol_log2_resolution);
    }
    if (rd->sampler == 0 && (!s->d.perms || (5 + 8 * ((long long)rd->max_depth + 1) > s->d.nPermDims && s->d.nPermDims < 1000)))
        return setError(PG_ERR_INVALID, "Halton table has %d dimensions; maxdepth %d needs %lld", s->d.nPermDims, rd->max_depth, 5 + 8 * ((long long)rd->max_depth + 1));
    if (!rd->filter_general && pgh_box_filter_needs_gather(rd))
        return setError(PG_ERR_INVALID, "pg_render: filter_general = 0, but in this frame a film position can round up onto the next pixel "
                                        "(pg_box_filter_needs_gather, include/pbrt_gpu.h): render it with filter_general = 1");
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = (hipStream_t)streamPtr;
    const int nLocalTiles = tileCount(rd);
    RenderParams rp;
    memset(&rp, 0, sizeof(rp));
    rp.rd = *rd;
    s->d.tsBatched = 0; s->d.tsOverflow = nullptr;  // (a call that failed half-way may have left them set)
    rp.nTilesX = (rd->sample_bounds[2] - rd->sample_bounds[0] + 15) / 16;
    rp.nTilesY = (rd->sample_bounds[3] - rd->sample_bounds[1] + 15) / 16;

    // device film / stray buffers (caller's when mem == DEVICE)
    PgFilmPixel *dFilm = film;
    PgStraySample *dStrays = strays;
    int *dNStrays = nStrays;
    const size_t filmBytes = sizeof(PgFilmPixel) * (size_t)rd->tile_pixels * (size_t)nLocalTiles;
    if (mem == PG_MEM_HOST) {
        HIP_TRY(s->filmDev.alloc(filmBytes));
        HIP_TRY(s->straysDev.alloc(sizeof(PgStraySample) * (size_t)(maxStrays > 0 ? maxStrays : 1)));
        HIP_TRY(s->nStraysDev.alloc(sizeof(int)));
        dFilm = (PgFilmPixel *)s->filmDev.p; dStrays = (PgStraySample *)s->straysDev.p; dNStrays = (int *)s->nStraysDev.p;
    }
    if (filmBytes) HIP_TRY(hipMemsetAsync(dFilm, 0, filmBytes, stream));
    HIP_TRY(hipMemsetAsync(dNStrays, 0, sizeof(int), stream));
    if (nLocalTiles == 0) {
        if (mem == PG_MEM_HOST) *nStrays = 0;
        return PG_OK;
    }
    // batch shape: as many whole tiles x samples as fit the path budget
    size_t budget = (size_t)1 << 27;
    if (const char *e = getenv("PG_BATCH_PATHS")) { long v = atol(e); if (v >= 256) budget = (size_t)v; }
    int sPerBatch = rd->spp, tilesPerBatch = nLocalTiles;
    if ((size_t)tilesPerBatch * 256 * sPerBatch > budget) {
        if (rd->filter_general) {
            // the gathering film kernel needs all samples of a tile in one batch (reference summation order): split by tiles
            tilesPerBatch = (int)(budget / ((size_t)256 * sPerBatch));
            if (tilesPerBatch < 1) tilesPerBatch = 1;
        } else {
            // prefer all tiles with fewer samples (keeps primary rays coherent and every pixel busy)
            sPerBatch = (int)(budget / ((size_t)tilesPerBatch * 256));
            if (sPerBatch < 1) { sPerBatch = 1; tilesPerBatch = (int)(budget / 256); if (tilesPerBatch < 1) tilesPerBatch = 1; }
        }
    }
    // tile-serial samplers: one path per tile in flight, slot = the tile's local index
    const int capacity = tileSerial ? std::max(nLocalTiles, 256) : tilesPerBatch * 256 * sPerBatch;
    int st = ensureWorkBuffers(s, capacity);
    if (st != PG_OK) return st;
    const int QSTRIDE = PG_REGIONS * PG_COUNT_STRIDE;  // ints of counter storage per queue
    VolState vs;
    memset(&vs, 0, sizeof(vs));
    RayQueue vq[2];  // second halves of the through-ray ping-pong (the first halves are q[2] and q[3])
    if (vol) {
        const size_t n = (size_t)regionCapFor(capacity, s->d.sparseLights != 0) * PG_REGIONS;
        if (s->volCapacity < capacity) {
            for (int i = 0; i < 2; ++i) { HIP_TRY(s->vqo[i].alloc(n * sizeof(float4))); HIP_TRY(s->vqd[i].alloc(n * (sizeof(float4) + (s->d.hasMotion ? sizeof(float) : 0)))); HIP_TRY(s->trAcc[i].alloc(n * sizeof(float4))); }
            for (int i = 0; i < 3; ++i) HIP_TRY(s->volP1[i].alloc(n * sizeof(float4)));
            HIP_TRY(s->vCounts.alloc(2 * QSTRIDE * sizeof(int)));
            HIP_TRY(s->volMedium.alloc(n * sizeof(int)));
            HIP_TRY(s->misLi.alloc(n * sizeof(float4)));
            HIP_TRY(s->pdLi.alloc(n * sizeof(float4)));
            HIP_TRY(s->hitT.alloc((s->d.nGrids > 0 ? 3 : 2) * n * sizeof(float)));
            s->volCapacity = capacity;
        }
        if (s->d.nGrids > 0 && s->gridVertex.bytes < n * sizeof(float4)) HIP_TRY(s->gridVertex.alloc(n * sizeof(float4)));
        vs.medium = (int *)s->volMedium.p;
        for (int i = 0; i < 2; ++i) vs.trAcc[i] = (float4 *)s->trAcc[i].p;
        for (int i = 0; i < 3; ++i) vs.p1[i] = (float4 *)s->volP1[i].p;
        vs.misLi = (float4 *)s->misLi.p; vs.pdLi = (float4 *)s->pdLi.p;
        for (int i = 0; i < 2; ++i) { vq[i].o = (float4 *)s->vqo[i].p; vq[i].d = (float4 *)s->vqd[i].p; vq[i].counts = (int *)s->vCounts.p + i * QSTRIDE; }
    }

    float4 *const hitsMis = (float4 *)s->hitsMain.p + (size_t)regionCapFor(capacity, s->d.sparseLights != 0) * PG_REGIONS;
    PathState ps;
    memset(&ps, 0, sizeof(ps));
    // PathIntegrator, and VolPathIntegrator on scenes without BSSRDF materials or grid media (whose probe-chain / two-phase kernels find a
    // path's state by its slot): L / beta / meta (/ the ray's medium) in queue order beside each main queue.  (The kernels are compiled for one
    // or the other: k_shade's QSTATE.)
    const bool volQ = vol && s->d.nBssrdfs == 0 && s->d.nGrids == 0;
    if (!vol || volQ) {
        const size_t n = (size_t)regionCapFor(capacity, s->d.sparseLights != 0) * PG_REGIONS;
        if (s->qsCapacity < capacity) {
            for (int i = 0; i < 2; ++i) { HIP_TRY(s->qsL[i].alloc(n * sizeof(float4))); HIP_TRY(s->qsBeta[i].alloc(n * sizeof(float4))); HIP_TRY(s->qsMeta[i].alloc(n * sizeof(int4))); }
            s->qsCapacity = capacity;
        }
        if (volQ && s->qsMedium[0].bytes < n * sizeof(int)) for (int i = 0; i < 2; ++i) HIP_TRY(s->qsMedium[i].alloc(n * sizeof(int)));
        for (int i = 0; i < 2; ++i) { ps.qs[i].L = (float4 *)s->qsL[i].p; ps.qs[i].beta = (float4 *)s->qsBeta[i].p; ps.qs[i].meta = (int4 *)s->qsMeta[i].p;
                                      ps.qs[i].medium = volQ ? (int *)s->qsMedium[i].p : nullptr; }
    }
    // Subsurface scattering (PathIntegrator): per-slot state of the BSSRDF branch, the job queue and the two probe queues
    const bool sssOn = s->d.nBssrdfs > 0;
    SssState sq;
    memset(&sq, 0, sizeof(sq));
    RayQueue sssP[2];
    if (sssOn) {
        const size_t n = (size_t)regionCapFor(capacity, s->d.sparseLights != 0) * PG_REGIONS;
        if (s->sssCapacity < capacity) {
            HIP_TRY(s->sssPo.alloc(n * sizeof(float4))); HIP_TRY(s->sssTarget.alloc(n * sizeof(float4))); HIP_TRY(s->sssCount.alloc(n * sizeof(int2)));
            for (int i = 0; i < 3; ++i) HIP_TRY(s->sssFrame[i].alloc(n * sizeof(float4)));
            for (int i = 0; i < 2; ++i) HIP_TRY(s->sssCoef[i].alloc(n * sizeof(float4)));
            HIP_TRY(s->sssHit.alloc(n * sizeof(float4))); HIP_TRY(s->sssHitO.alloc(n * sizeof(float4))); HIP_TRY(s->sssHitD.alloc(n * sizeof(float4)));
            HIP_TRY(s->sssHitInst.alloc(n * sizeof(int)));
            if (s->d.hasMotion) HIP_TRY(s->sssHitXf.alloc((s->d.hasNest ? 2 : 1) * n * PG_XF_STRIDE * sizeof(float)));  // the chosen hit's interpolated instance matrices
            HIP_TRY(s->sssMedium.alloc(n * sizeof(int2)));
            for (int i = 0; i < 3; ++i) { HIP_TRY(s->sssQo[i].alloc(n * sizeof(float4))); HIP_TRY(s->sssQd[i].alloc(n * (sizeof(float4) + (s->d.hasMotion ? sizeof(float) : 0)))); }  // (+ the probe rays' times: PG_QUEUE_TIMES)
            HIP_TRY(s->sssCounts.alloc(3 * QSTRIDE * sizeof(int)));
            HIP_TRY(s->sssTail.alloc(QSTRIDE * sizeof(int)));
            s->sssCapacity = capacity;
        }
        sq.po = (float4 *)s->sssPo.p; sq.target = (float4 *)s->sssTarget.p; sq.count = (int2 *)s->sssCount.p;
        for (int i = 0; i < 3; ++i) sq.frame[i] = (float4 *)s->sssFrame[i].p;
        for (int i = 0; i < 2; ++i) sq.coef[i] = (float4 *)s->sssCoef[i].p;
        sq.hit = (float4 *)s->sssHit.p; sq.hitO = (float4 *)s->sssHitO.p; sq.hitD = (float4 *)s->sssHitD.p; sq.hitInst = (int *)s->sssHitInst.p; sq.hitXf = (float *)s->sssHitXf.p; sq.hitXfNest = (int)n; sq.medium = (int2 *)s->sssMedium.p;
        sq.qjob.o = (float4 *)s->sssQo[0].p; sq.qjob.d = (float4 *)s->sssQd[0].p; sq.qjob.counts = (int *)s->sssCounts.p;
        for (int i = 0; i < 2; ++i) { sssP[i].o = (float4 *)s->sssQo[1 + i].p; sssP[i].d = (float4 *)s->sssQd[1 + i].p; sssP[i].counts = (int *)s->sssCounts.p + (1 + i) * QSTRIDE; }
    }
    ps.L = (float4 *)s->stL.p; ps.beta = (float4 *)s->stBeta.p; ps.meta = (int4 *)s->stMeta.p;
    ps.pdLight = (float4 *)s->pdLight.p; ps.pdMis = (float4 *)s->pdMis.p; ps.pdBeta = (float4 *)s->pdBeta.p; ps.pdInfo = (int4 *)s->pdInfo.p;
    int *counts = (int *)s->counts.p;
    RayQueue q[4];
    for (int i = 0; i < 4; ++i) { q[i].o = (float4 *)s->qo[i].p; q[i].d = (float4 *)s->qd[i].p; q[i].counts = counts + i * QSTRIDE; }
    // sum of a queue's region counters in a host copy of the counter block
    auto queueTotal = [&](const int *blk, int qi) { uint64_t t = 0; for (int r = 0; r < PG_REGIONS; ++r) t += (uint64_t)blk[qi * QSTRIDE + r * PG_COUNT_STRIDE]; return t; };
    TraceCounters *cnClosest = (TraceCounters *)s->traceCn.p, *cnShadow = cnClosest + 1;
    unsigned long long *lightTests = (unsigned long long *)s->lightTests.p;

    size_t ev = 0;
    std::vector<std::pair<size_t, int>> timed;  // (event index, kernel: 0 closest-hit, 1 any-hit, 2 shade, 3 resolve, 4 generate, 5 film)
    // HIP events around one launch on `st_` (the stream the kernel runs on); per-kernel times are only meaningful while the
    // any-hit launch does not share the chip with the closest-hit launch (PG_OVERLAP_SHADOW=0, the default)
    // (not in the tile-serial mode: its hundreds of thousands of small launches would each need a pair of events)
    const bool timing = !tileSerial;
#define PG_TIMED(kind_, st_, launch_) do { if (!timing) { launch_; break; } hipEvent_t a_ = getEvent(s, ev), b_ = getEvent(s, ev + 1); \
        if (!a_ || !b_) return setError(PG_ERR_DEVICE, "hipEventCreate failed"); \
        timed.push_back({ev, kind_}); ev += 2; HIP_TRY(hipEventRecord(a_, st_)); launch_; HIP_TRY(hipEventRecord(b_, st_)); } while (0)
    hipEvent_t evStart = getEvent(s, ev++), evStop = getEvent(s, ev++);
    if (!evStart || !evStop) return setError(PG_ERR_DEVICE, "hipEventCreate failed");
    HIP_TRY(hipEventRecord(evStart, stream));
    // Sparse "spatial" light tables: after a shading launch, compute the distributions of the voxels its lanes asked for and
    // shade the entries that waited for them (one host round trip per launch while the table warms up; none once every
    // voxel the image touches exists -- the tables stay with the scene).
    rp.retryList = (int *)s->retryList.p;
    if (s->matLobes.p && s->matHead.p) { rp.matPre.lobes = (float4 *)s->matLobes.p; rp.matPre.head = (float4 *)s->matHead.p; rp.matPre.stride = s->matStride; }
    auto settleLightTables = [&](const std::function<void()> &reshade) -> int {
        if (!s->d.sparseLights) return PG_OK;
        int cnt[2] = {0, 0};
        HIP_TRY(hipMemcpyAsync(cnt, s->voxelCounters.p, sizeof(cnt), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        if (cnt[0] == 0 && cnt[1] == 0) return PG_OK;
        if (s->poolUsed + cnt[0] > s->poolSlots)
            return setError(PG_ERR_UNSUPPORTED, "spatial light distribution: %d voxels x %d lights exceed the table pool (%d voxels); use "
                                                "\"lightsamplestrategy\" \"power\" or \"uniform\"", s->poolUsed + cnt[0], s->d.nLights, s->poolSlots);
        launch_light_tables_sparse(s->d, (float *)s->distTable.p, (const int *)s->voxelRequests.p, cnt[0], s->poolUsed, stream);
        s->poolUsed += cnt[0];
        HIP_TRY(hipMemsetAsync(s->voxelCounters.p, 0, sizeof(cnt), stream));
        if (cnt[1] > 0) { rp.retryCount = cnt[1]; reshade(); rp.retryCount = 0; }
        return PG_OK;
    };
    uint64_t closestRays = 0, shadowRays = 0, cameraRays = 0, closestLaunches = 0, shadowLaunches = 0;
    uint64_t shadeLaunches = 0, resolveLaunches = 0, shadeItems = 0, misRays = 0, shadingModes = 0;
    // PgCounters::shading_modes: which k_shade<MODE> this frame's shading launches are (pg_shade_mode, the launch functions' own choice)
    auto noteShading = [&](const DScene &dsc, bool vol, bool sss, bool gridPhase) {
        const int mode = pg_shade_mode(dsc, rp, vol, sss, gridPhase);
        shadingModes |= 1ull << mode;
        if (mode == 3) shadingModes |= PG_SHADING_MATERIAL_PREPASS;
        if (mode == 2 && !gridPhase && !(sss && dsc.nBssrdfs > 0) && s->matStride > 0) shadingModes |= PG_SHADING_LISTS_DID_NOT_FIT;
    };
    std::vector<int> hostCounts;  // read back once per batch at the end (pinned copy not needed: tiny)
    DeviceBuffer countLog;        // per-bounce queue sizes, copied back after the batch for the ray statistics
    // (64-bit: maxdepth comes from the caller / the scene file.)  Bounce launches are enqueued without looking at the queues, so
    // a huge maxdepth is bounded here: beyond PG_MAX_BLIND_BOUNCES the host looks at the main queue every 32 bounces and stops
    // when it is empty (Russian roulette ends every path), and a frame whose paths outlive PG_MAX_BOUNCES fails loudly.
    const long long PG_MAX_BLIND_BOUNCES = 64, PG_MAX_BOUNCES = 4096;
    const long long wantIters = (long long)rd->max_depth + 1 + (s->hasNullMaterial ? 64 : 0);
    const int maxIters = (int)std::min<long long>(wantIters, PG_MAX_BOUNCES);
    HIP_TRY(countLog.alloc(sizeof(int) * 4 * QSTRIDE * (size_t)(maxIters + 1)));
    std::vector<int> curQueueOfBounce;

    // One batch of paths from their camera rays (`generate` fills the first main queue) to their film samples (`film`): all the
    // bounces of the rp.capacity path slots described by rp.
    auto tracePaths = [&](const std::function<void()> &generate, const std::function<void()> &filmSamples) -> int {
        {
            for (int i = 0; i < 4; ++i) q[i].regionCap = regionCapFor(rp.capacity, s->d.sparseLights != 0);
            if (sssOn) { sq.qjob.regionCap = q[0].regionCap; sssP[0].regionCap = sssP[1].regionCap = q[0].regionCap; }
            curQueueOfBounce.clear();
            HIP_TRY(hipMemsetAsync(counts, 0, 4 * QSTRIDE * sizeof(int), stream));
            int cur = 0;  // main queue index (0/1 ping-pong); 2 = shadow, 3 = MIS
            PG_TIMED(4, stream, generate());
            if (vol) {
                // VolPathIntegrator::Li (volpath.cpp:72-186).  Per loop iteration: closest-hit(main rays, with the hits' ray
                // parameters) -> shade with medium sampling -> the transmittance rays of the light samples and of the
                // BSDF/phase samples, re-traced until none is left under way (light.cpp:63-81, scene.cpp:57-70) -> resolve.
                // Crossing a surface without a material does not count as a bounce, so the loop runs until the queue is empty.
                DScene dv = s->d;
                dv.ext = 1;  // the general shading kernels
                const int n1 = regionCapFor(capacity, s->d.sparseLights != 0) * PG_REGIONS;
                float *hitT = (float *)s->hitT.p;
                for (int i = 0; i < 2; ++i) vq[i].regionCap = q[0].regionCap;
                launch_fill_int(vs.medium, rd->camera_medium + 1, rp.capacity, stream);  // camera rays start in the camera's medium (camera.h:78)
                std::vector<int> blk(4 * QSTRIDE), vblk(2 * QSTRIDE);
                auto readCounts = [&]() -> int {
                    HIP_TRY(hipMemcpyAsync(blk.data(), counts, 4 * QSTRIDE * sizeof(int), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipMemcpyAsync(vblk.data(), s->vCounts.p, 2 * QSTRIDE * sizeof(int), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipStreamSynchronize(stream));
                    return PG_OK;
                };
                if (int e = readCounts()) return e;
                uint64_t nMain = queueTotal(blk.data(), cur);
                cameraRays += nMain;
                for (int iter = 0; nMain > 0; ++iter) {
                    if (iter > 100000) return setError(PG_ERR_DEVICE, "pg_render: volpath loop did not terminate");
                    const int nxt = cur ^ 1;
                    PG_TIMED(0, stream, launch_closest(dv, s->trace, q[cur], (float4 *)s->hitsMain.p, hitT, cnClosest, (int *)s->cursors.p, (int *)s->cullGuard.p, stream));
                    ++closestLaunches; closestRays += nMain; shadeItems += nMain;
                    HIP_TRY(hipMemsetAsync(counts + nxt * QSTRIDE, 0, QSTRIDE * sizeof(int), stream));
                    HIP_TRY(hipMemsetAsync(counts + 2 * QSTRIDE, 0, 2 * QSTRIDE * sizeof(int), stream));
                    if (sssOn) HIP_TRY(hipMemsetAsync(sq.qjob.counts, 0, QSTRIDE * sizeof(int), stream));
                    const SssState *sssArg = sssOn ? &sq : nullptr;
                    // a scene with a grid medium shades in two phases around the transmittance rays (k_shade<., ., ., GRID>)
                    const bool gridOn = s->d.nGrids > 0;
                    float4 *gridVertex = (float4 *)s->gridVertex.p;
                    // (GlobalSamplers, dense light tables: the tile-serial streams and the deferred vertices of sparse tables draw in the shading kernel)
                    float2 *volPre = (s->volOrder && !tileSerial && !s->d.sparseLights) ? (float2 *)s->volPre.p : nullptr;
                    rp.volPre = volPre;
                    rp.order = (s->d.primClass || volPre) ? (const int *)s->shadeOrder.p : nullptr;  // (the second phase of a grid scene takes the same order)
                    PG_TIMED(2, stream, (launch_shade_order_vol(dv, rp, ps, vs, q[cur], (const float4 *)s->hitsMain.p, hitT, (int *)s->shadeOrder.p, volPre, stream, cur), launch_shade_vol(dv, rp, ps, vs, q[cur], (const float4 *)s->hitsMain.p, hitT, q[nxt], q[2], q[3], lightTests, stream, sssArg, gridVertex, gridOn ? 1 : 0, cur)));
                    ++shadeLaunches; noteShading(dv, true, sssArg != nullptr, gridOn);
                    if (int e = settleLightTables([&]() { launch_shade_vol(dv, rp, ps, vs, q[cur], (const float4 *)s->hitsMain.p, hitT, q[nxt], q[2], q[3], lightTests, stream, sssArg, gridVertex, gridOn ? 1 : 0, cur); })) return e;
                    // through rays: kind 0 = light samples (q[2] <-> vq[0]), kind 1 = BSDF / phase samples (q[3] <-> vq[1]), re-traced
                    // until none is left under way
                    auto throughRays = [&]() -> int {
                        if (gridOn) {
                            // ratio tracking draws from the path's sampler: a path's kind-0 ray (visibility.Tr) runs to its end before
                            // its kind-1 ray (IntersectTr after the BSDF sample) starts, as in EstimateDirect; one queue pair at a time
                            for (int kind = 0; kind < 2; ++kind) {
                                RayQueue tk[2] = {kind == 0 ? q[2] : q[3], vq[kind]};
                                int tc = 0;
                                for (int pass = 0;; ++pass) {
                                    if (pass > 100000) return setError(PG_ERR_DEVICE, "pg_render: transmittance loop did not terminate");
                                    if (int e = readCounts()) return e;
                                    const uint64_t nk = tc == 0 ? queueTotal(blk.data(), 2 + kind) : queueTotal(vblk.data(), kind);
                                    if (nk == 0) break;
                                    // (results at the offset the kind's through kernel reads them from)
                                    const size_t off = (size_t)(1 + kind) * n1;  // part 0 keeps the main rays' hits for phase 2
                                    float4 *hk = (float4 *)s->hitsMain.p + off;
                                    DScene dk = dv;
                                    if (dk.hitInst) dk.hitInst += off;
                                    if (dk.animXf) dk.animXf += off * PG_XF_STRIDE;
                                    PG_TIMED(0, stream, launch_closest(dk, s->trace, tk[tc], hk, hitT + off, cnClosest, (int *)s->cursors.p, (int *)s->cullGuard.p, stream));
                                    ++closestLaunches; closestRays += nk;
                                    HIP_TRY(hipMemsetAsync(tk[tc ^ 1].counts, 0, QSTRIDE * sizeof(int), stream));
                                    launch_through(dv, ps, vs, kind, tk[tc], (const float4 *)s->hitsMain.p, hitT, (int)off, tk[tc ^ 1], stream, &rp);
                                    tc ^= 1;
                                }
                            }
                            return PG_OK;
                        }
                        RayQueue tq[2][2] = {{q[2], vq[0]}, {q[3], vq[1]}};
                        int tcur = 0;
                        for (int pass = 0;; ++pass) {
                            if (pass > 100000) return setError(PG_ERR_DEVICE, "pg_render: transmittance loop did not terminate");
                            if (int e = readCounts()) return e;
                            const uint64_t n0 = tcur == 0 ? queueTotal(blk.data(), 2) : queueTotal(vblk.data(), 0);
                            const uint64_t n1q = tcur == 0 ? queueTotal(blk.data(), 3) : queueTotal(vblk.data(), 1);
                            if (n0 + n1q == 0) break;
                            PG_TIMED(0, stream, launch_closest2(dv, s->trace, tq[0][tcur], tq[1][tcur], (float4 *)s->hitsMain.p, n1, cnClosest, (int *)s->cursors.p, (int *)s->cullGuard.p, stream, hitT));
                            ++closestLaunches; closestRays += n0 + n1q;
                            HIP_TRY(hipMemsetAsync(tq[0][tcur ^ 1].counts, 0, QSTRIDE * sizeof(int), stream));
                            HIP_TRY(hipMemsetAsync(tq[1][tcur ^ 1].counts, 0, QSTRIDE * sizeof(int), stream));
                            launch_through(dv, ps, vs, 0, tq[0][tcur], (const float4 *)s->hitsMain.p, hitT, 0, tq[0][tcur ^ 1], stream);
                            launch_through(dv, ps, vs, 1, tq[1][tcur], (const float4 *)s->hitsMain.p, hitT, n1, tq[1][tcur ^ 1], stream);
                            tcur ^= 1;
                        }
                        return PG_OK;
                    };
                    if (int e = throughRays()) return e;
                    PG_TIMED(3, stream, launch_resolve_vol(dv, ps, vs, q[cur], stream, cur));
                    ++resolveLaunches;
                    if (gridOn) {  // phase 2: the vertices' next directions, drawn behind the transmittance rays' numbers
                        PG_TIMED(2, stream, launch_shade_vol(dv, rp, ps, vs, q[cur], (const float4 *)s->hitsMain.p, hitT, q[nxt], q[2], q[3], lightTests, stream, sssArg, gridVertex, 2));
                        ++shadeLaunches;
                        if (int e = readCounts()) return e;
                    }
                    if (sssOn) {
                        // ---- the BSSRDF branch (volpath.cpp:151-176) of the paths k_shade handed over: probe chains (two walks), exit
                        // vertices, their transmittance rays and resolve; the exit vertices' next rays join q[nxt], which is
                        // traced as a whole at the start of the next iteration
                        std::vector<int> jb(QSTRIDE);
                        auto regionSum = [&](const int *dev, uint64_t &total) -> int {
                            HIP_TRY(hipMemcpyAsync(jb.data(), dev, QSTRIDE * sizeof(int), hipMemcpyDeviceToHost, stream));
                            HIP_TRY(hipStreamSynchronize(stream));
                            total = 0;
                            for (int r = 0; r < PG_REGIONS; ++r) total += (uint64_t)jb[r * PG_COUNT_STRIDE];
                            return PG_OK;
                        };
                        uint64_t nJobs = 0;
                        if (int e = regionSum(sq.qjob.counts, nJobs)) return e;
                        if (nJobs > 0) {
                            for (int pass = 1; pass <= 2; ++pass) {
                                RayQueue curQ = sq.qjob;
                                uint64_t nRays = nJobs;
                                for (int step = 0; nRays > 0; ++step) {
                                    if (step > 1000000) return setError(PG_ERR_DEVICE, "pg_render: a BSSRDF probe chain did not terminate");
                                    RayQueue outQ = sssP[step & 1];
                                    HIP_TRY(hipMemsetAsync(outQ.counts, 0, QSTRIDE * sizeof(int), stream));
                                    launch_closest(dv, s->trace, curQ, (float4 *)s->hitsMain.p, nullptr, pass == 1 ? cnClosest : cnClosest + 2, (int *)s->cursors.p, (int *)s->cullGuard.p, stream);
                                    if (pass == 1) { closestRays += nRays; ++closestLaunches; }
                                    launch_sss_probe(dv, sq, pass, curQ, (const float4 *)s->hitsMain.p, outQ, stream, true, step == 0);
                                    if (int e = regionSum(outQ.counts, nRays)) return e;
                                    curQ = outQ;
                                }
                            }
                            HIP_TRY(hipMemsetAsync(counts + 2 * QSTRIDE, 0, 2 * QSTRIDE * sizeof(int), stream));
                            // (a grid medium's ratio tracking draws from the paths' samplers: the exit vertices' transmittance rays run between their direct
                            // lighting and their next directions, as at k_shade's vertices -- k_sss_exit in two phases; gridVertex is free again by now)
                            launch_sss_exit(dv, rp, ps, sq, q[nxt], q[2], q[3], lightTests, stream, nxt, true, vs, gridOn ? 1 : 0, gridVertex);
                            ++shadeLaunches; shadeItems += nJobs;
                            if (int e = throughRays()) return e;
                            launch_resolve_vol(dv, ps, vs, sq.qjob, stream);
                            ++resolveLaunches;
                            if (gridOn) { launch_sss_exit(dv, rp, ps, sq, q[nxt], q[2], q[3], lightTests, stream, nxt, true, vs, 2, gridVertex); ++shadeLaunches; }
                        }
                        if (int e = readCounts()) return e;
                    }
                    // the last pass of the through loop read the counters: the main queue's size comes from the same block
                    nMain = queueTotal(blk.data(), nxt);
                    cur = nxt;
                }
                filmSamples();
                HIP_TRY(hipStreamSynchronize(stream));
                return PG_OK;
            }
            // Launch order per bounce b (one stream): shade(b) -> any-hit(shadow rays of b) -> closest-hit(main rays of b+1
            // and MIS rays of b in ONE launch) -> resolve(b).  The first closest-hit launch traces the camera rays alone.
            auto timedClosest = [&](RayQueue qa, float4 *ha, const RayQueue *qb, float4 *hb) -> int {
                hipEvent_t a = nullptr, b = nullptr;
                if (timing) { a = getEvent(s, ev); b = getEvent(s, ev + 1); timed.push_back({ev, 0}); ev += 2; HIP_TRY(hipEventRecord(a, stream)); }
                if (qb) launch_closest2(s->d, s->trace, qa, *qb, ha, (int)(hb - ha), cnClosest, (int *)s->cursors.p, (int *)s->cullGuard.p, stream);
                else launch_closest(s->d, s->trace, qa, ha, nullptr, cnClosest, (int *)s->cursors.p, (int *)s->cullGuard.p, stream);
                if (timing) HIP_TRY(hipEventRecord(b, stream));
                ++closestLaunches;
                return PG_OK;
            };
            if (int e = timedClosest(q[cur], (float4 *)s->hitsMain.p, nullptr, nullptr)) return e;
            int iters = 0;
            for (int bounce = 0; bounce < maxIters; ++bounce, ++iters) {
                const int nxt = cur ^ 1;
                HIP_TRY(hipMemsetAsync(counts + nxt * QSTRIDE, 0, QSTRIDE * sizeof(int), stream));
                HIP_TRY(hipMemsetAsync(counts + 2 * QSTRIDE, 0, 2 * QSTRIDE * sizeof(int), stream));
                if (sssOn) HIP_TRY(hipMemsetAsync(sq.qjob.counts, 0, QSTRIDE * sizeof(int), stream));
                const SssState *sssArg = sssOn ? &sq : nullptr;
                rp.order = s->d.primClass ? (const int *)s->shadeOrder.p : nullptr;
                PG_TIMED(2, stream, (launch_shade_order(s->d, q[cur], (const float4 *)s->hitsMain.p, (int *)s->shadeOrder.p, stream), launch_shade(s->d, rp, ps, q[cur], (const float4 *)s->hitsMain.p, q[nxt], q[2], q[3], lightTests, stream, cur, sssArg)));
                ++shadeLaunches; noteShading(s->d, false, sssArg != nullptr, false);
                if (int e = settleLightTables([&]() { launch_shade(s->d, rp, ps, q[cur], (const float4 *)s->hitsMain.p, q[nxt], q[2], q[3], lightTests, stream, cur, sssArg); })) return e;
                // paths that reach maxdepth neither continue nor sample lights (path.cpp:104): nothing left to trace
                const bool lastDepth = !s->hasNullMaterial && bounce >= rd->max_depth;
                if (!lastDepth) {
                    // The shadow rays of this bounce and the closest-hit rays of the next depend only on shade(b): the any-hit
                    // launch goes to a second stream so that its blocks fill the chip while the closest-hit launch's
                    // persistent waves drain (and vice versa); resolve(b) joins both.
                    // (tile-serial samplers: a handful of rays per launch, each a chain of dependent fetches -- both launches are
                    // latency-bound and run side by side)
                    const bool overlap = s->overlapShadow || tileSerial;
                    hipStream_t sst = overlap ? s->shadowStream : stream;
                    if (overlap) {
                        HIP_TRY(hipEventRecord(s->evShaded, stream));
                        HIP_TRY(hipStreamWaitEvent(sst, s->evShaded, 0));
                    }
#define PG_Q_SHADOW q[2]
                    hipEvent_t a = nullptr, b = nullptr;
                    if (timing) { a = getEvent(s, ev); b = getEvent(s, ev + 1); timed.push_back({ev, 1}); ev += 2; HIP_TRY(hipEventRecord(a, sst)); }
                    launch_anyhit(s->d, s->trace, PG_Q_SHADOW, (int *)s->occluded.p, cnShadow, (int *)s->cursors2.p, sst);
                    if (timing) HIP_TRY(hipEventRecord(b, sst));
                    ++shadowLaunches;
                    if (overlap) HIP_TRY(hipEventRecord(s->evShadowed, sst));
                    if (int e = timedClosest(q[nxt], (float4 *)s->hitsMain.p, &q[3], hitsMis)) return e;
                    if (overlap) HIP_TRY(hipStreamWaitEvent(stream, s->evShadowed, 0));
                    PG_TIMED(3, stream, launch_resolve(s->d, ps, q[cur], q[3], (const int *)s->occluded.p, (const float4 *)hitsMis, stream, cur, lightTests));
                    ++resolveLaunches;
                }
                // log this bounce's queue sizes
                HIP_TRY(hipMemcpyAsync((int *)countLog.p + 4 * QSTRIDE * (size_t)bounce, counts, 4 * QSTRIDE * sizeof(int), hipMemcpyDeviceToDevice, stream));
                curQueueOfBounce.push_back(cur);
                if (sssOn && !lastDepth) {
                    // ---- the BSSRDF branch of Li (path.cpp:152-174) for the paths k_shade handed over: the probe chains of
                    // SeparableBSSRDF::Sample_Sp walked twice through the traversal kernel (count the hits on the material; stop at
                    // the chosen one), then the exit vertices: their shadow / MIS rays, the tail of the next bounce's queue that
                    // their next rays form, and the resolve of their direct lighting.  One counter read-back per step.
                    std::vector<int> jb(QSTRIDE);
                    auto regionSum = [&](const int *dev, uint64_t &total) -> int {
                        HIP_TRY(hipMemcpyAsync(jb.data(), dev, QSTRIDE * sizeof(int), hipMemcpyDeviceToHost, stream));
                        HIP_TRY(hipStreamSynchronize(stream));
                        total = 0;
                        for (int r = 0; r < PG_REGIONS; ++r) total += (uint64_t)jb[r * PG_COUNT_STRIDE];
                        return PG_OK;
                    };
                    uint64_t nJobs = 0;
                    if (int e = regionSum(sq.qjob.counts, nJobs)) return e;
                    if (nJobs > 0) {
                        const size_t n1 = (size_t)q[0].regionCap * PG_REGIONS;
                        DScene dprobe = s->d;  // the probe rays' hits go where the MIS rays' went (k_resolve is done with those)
                        if (dprobe.hitInst) dprobe.hitInst += n1;
                        if (dprobe.animXf) dprobe.animXf += n1 * PG_XF_STRIDE;
                        for (int pass = 1; pass <= 2; ++pass) {
                            RayQueue curQ = sq.qjob;
                            uint64_t nRays = nJobs;
                            for (int step = 0; nRays > 0; ++step) {
                                if (step > 1000000) return setError(PG_ERR_DEVICE, "pg_render: a BSSRDF probe chain did not terminate");
                                RayQueue outQ = sssP[step & 1];
                                HIP_TRY(hipMemsetAsync(outQ.counts, 0, QSTRIDE * sizeof(int), stream));
                                // the second walk repeats queries the reference makes once: its rays and traversal work are not counted
                                launch_closest(dprobe, s->trace, curQ, hitsMis, nullptr, pass == 1 ? cnClosest : cnClosest + 2, (int *)s->cursors.p, (int *)s->cullGuard.p, stream);
                                if (pass == 1) { closestRays += nRays; ++closestLaunches; }
                                launch_sss_probe(dprobe, sq, pass, curQ, hitsMis, outQ, stream);
                                if (int e = regionSum(outQ.counts, nRays)) return e;
                                curQ = outQ;
                            }
                        }
                        HIP_TRY(hipMemcpyAsync(s->sssTail.p, counts + nxt * QSTRIDE, QSTRIDE * sizeof(int), hipMemcpyDeviceToDevice, stream));
                        HIP_TRY(hipMemsetAsync(counts + 2 * QSTRIDE, 0, 2 * QSTRIDE * sizeof(int), stream));
                        launch_sss_exit(s->d, rp, ps, sq, q[nxt], q[2], q[3], lightTests, stream, nxt, false, vs);
                        ++shadeLaunches; shadeItems += nJobs;
                        launch_anyhit(s->d, s->trace, q[2], (int *)s->occluded.p, cnShadow, (int *)s->cursors2.p, stream);
                        ++shadowLaunches;
                        launch_closest2(s->d, s->trace, q[nxt], q[3], (float4 *)s->hitsMain.p, (int)n1, cnClosest, (int *)s->cursors.p, (int *)s->cullGuard.p, stream, nullptr,
                                        (const int *)s->sssTail.p);
                        ++closestLaunches;
                        launch_resolve(s->d, ps, sq.qjob, q[3], (const int *)s->occluded.p, (const float4 *)hitsMis, stream, cur);
                        ++resolveLaunches;
                        uint64_t nSh = 0, nMis = 0;
                        if (int e = regionSum(counts + 2 * QSTRIDE, nSh)) return e;
                        if (int e = regionSum(counts + 3 * QSTRIDE, nMis)) return e;
                        shadowRays += nSh; closestRays += nMis; misRays += nMis;  // (the next rays are counted with the next bounce's queue)
                    }
                }
                cur = nxt;
                if ((s->hasNullMaterial && bounce >= rd->max_depth) || (bounce >= PG_MAX_BLIND_BOUNCES && bounce % 32 == 0)) {
                    std::vector<int> blk(4 * QSTRIDE);
                    HIP_TRY(hipMemcpyAsync(blk.data(), counts, 4 * QSTRIDE * sizeof(int), hipMemcpyDeviceToHost, stream));
                    HIP_TRY(hipStreamSynchronize(stream));
                    if (queueTotal(blk.data(), cur) == 0) { ++iters; break; }
                }
            }
            if (iters == maxIters && wantIters > maxIters) {  // the last allowed bounce: is anything still alive?
                std::vector<int> blk(4 * QSTRIDE);
                HIP_TRY(hipMemcpyAsync(blk.data(), counts, 4 * QSTRIDE * sizeof(int), hipMemcpyDeviceToHost, stream));
                HIP_TRY(hipStreamSynchronize(stream));
                if (queueTotal(blk.data(), cur) != 0) return setError(PG_ERR_UNSUPPORTED, "paths longer than %lld vertices (maxdepth %d)", PG_MAX_BOUNCES, rd->max_depth);
            }
            PG_TIMED(5, stream, filmSamples());
            hostCounts.resize(4 * QSTRIDE * (size_t)iters);
            HIP_TRY(hipMemcpyAsync(hostCounts.data(), countLog.p, sizeof(int) * 4 * QSTRIDE * (size_t)iters, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            for (int b = 0; b < iters; ++b) {
                const int *blk = hostCounts.data() + 4 * QSTRIDE * (size_t)b;
                closestRays += queueTotal(blk, curQueueOfBounce[b]) + queueTotal(blk, 3);
                shadowRays += queueTotal(blk, 2);
                shadeItems += queueTotal(blk, curQueueOfBounce[b]);
                misRays += queueTotal(blk, 3);
                if (getenv("PG_PRINT_COUNTS"))
                    fprintf(stderr, "pg_render: bounce %d main %llu shadow %llu mis %llu\n", b, (unsigned long long)queueTotal(blk, curQueueOfBounce[b]),
                            (unsigned long long)queueTotal(blk, 2), (unsigned long long)queueTotal(blk, 3));
            }
            if (iters > 0) cameraRays += queueTotal(hostCounts.data(), curQueueOfBounce[0]);
        }
        return PG_OK;
    };
    const std::function<void()> filmBatch = [&]() {
        if (rd->filter_general) launch_film_general(rp, ps, dFilm, stream);
        else launch_film(rp, ps, dFilm, dStrays, maxStrays, dNStrays, stream);
    };
    if (tsBatched) {  // every pixel's sample arrays, one lane per tile (the tile's stream in the reference's pixel order)
        const size_t nArr = (size_t)nLocalTiles * 256 * (size_t)rd->sampler_dims * (size_t)rd->spp;
        HIP_TRY(s->tsState.alloc(sizeof(TileSamplerState) * (size_t)nLocalTiles));
        if (s->ts1.bytes < sizeof(float) * (nArr + 1) || s->ts2.bytes < sizeof(float) * 2 * (nArr + 1)) return setError(PG_ERR_DEVICE, "pg_render: sample arrays not allocated");  // (taken where tsBatched was decided)
        HIP_TRY(s->tsOverflow.alloc(sizeof(int)));
        HIP_TRY(hipMemsetAsync(s->tsOverflow.p, 0, sizeof(int), stream));
        s->d.ts = (TileSamplerState *)s->tsState.p; s->d.ts1 = (float *)s->ts1.p; s->d.ts2 = (float *)s->ts2.p;
        s->d.tsDims = rd->sampler_dims; s->d.tsSpp = rd->spp; s->d.tsBatched = 1; s->d.tsOverflow = (int *)s->tsOverflow.p;
        rp.tileLocal0 = 0; rp.nTilesBatch = nLocalTiles; rp.s0 = 0; rp.sCount = 1; rp.capacity = nLocalTiles;
        launch_ts_init(s->d, rp, stream);
        rp.tsGuessSkew = getenv("PG_TS_GUESS_SKEW") ? atoi(getenv("PG_TS_GUESS_SKEW")) : 0;
        launch_ts_start_tile(s->d, rp, stream);
    }
    if (!tileSerial) {
        for (int tile0 = 0; tile0 < nLocalTiles; tile0 += tilesPerBatch)
            for (int s0 = 0; s0 < rd->spp; s0 += sPerBatch) {
                rp.tileLocal0 = tile0;
                rp.nTilesBatch = std::min(tilesPerBatch, nLocalTiles - tile0);
                rp.s0 = s0;
                rp.sCount = std::min(sPerBatch, rd->spp - s0);
                rp.capacity = rp.nTilesBatch * 256 * rp.sCount;
                if (int e = tracePaths([&]() { launch_generate(s->d, rp, ps, q[0], stream); }, filmBatch)) return e;
            }
    } else {
        // The samplers that draw from one RNG stream per tile (random, stratified, 02sequence, maxmindist): a tile's pixels, a
        // pixel's samples and a sample's draws consume the stream in order, and how many numbers a path takes depends on the
        // path -- so a tile has ONE path in flight, and the wavefront is one path of every tile: pixel (lx, ly) of all tiles,
        // sample by sample (integrator.cpp:247-332).  Tiles clipped by the image skip the pixels they do not have.
        const int nd = rd->sampler == PG_SAMPLER_RANDOM ? 0 : rd->sampler_dims;
        HIP_TRY(s->tsState.alloc(sizeof(TileSamplerState) * (size_t)nLocalTiles));
        HIP_TRY(s->ts1.alloc(sizeof(float) * ((size_t)nLocalTiles * nd * rd->spp + 1)));
        HIP_TRY(s->ts2.alloc(sizeof(float) * 2 * ((size_t)nLocalTiles * nd * rd->spp + 1)));
        s->d.ts = (TileSamplerState *)s->tsState.p; s->d.ts1 = (float *)s->ts1.p; s->d.ts2 = (float *)s->ts2.p;
        s->d.tsDims = nd; s->d.tsSpp = rd->spp;
        rp.tileLocal0 = 0; rp.nTilesBatch = nLocalTiles; rp.s0 = 0; rp.sCount = 1; rp.capacity = nLocalTiles;
        launch_ts_init(s->d, rp, stream);
        int err = PG_OK;
        for (int ly = 0; ly < 16 && !err; ++ly)
            for (int lx = 0; lx < 16 && !err; ++lx) {
                if (rd->sample_bounds[0] + lx >= rd->sample_bounds[2] || rd->sample_bounds[1] + ly >= rd->sample_bounds[3]) continue;  // no tile has this pixel
                launch_ts_start_pixel(s->d, rp, lx, ly, stream);
                for (int sn = 0; sn < rd->spp && !err; ++sn)
                    err = tracePaths([&]() { launch_ts_generate(s->d, rp, ps, q[0], sn, stream); },
                                     [&]() { launch_ts_film(s->d, rp, ps, dFilm, dStrays, maxStrays, dNStrays, stream); });
            }
        s->d.ts = nullptr; s->d.ts1 = s->d.ts2 = nullptr;
        if (err) return err;
    }
    if (tsBatched) {
        int over = 0;
        HIP_TRY(hipMemcpyAsync(&over, s->tsOverflow.p, sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        s->d.ts = nullptr; s->d.ts1 = s->d.ts2 = nullptr; s->d.tsBatched = 0; s->d.tsOverflow = nullptr;
        if (over & 2) return setError(PG_ERR_DEVICE, "pg_render: the start offsets of a tile's pixel sample arrays did not converge (k_ts_start_tile, internal error)");
        if (over) return setError(PG_ERR_DEVICE, "pg_render: a path drew beyond the %d sampled dimensions of the batched pixel sampler (internal error)", rd->sampler_dims);
    }
    HIP_TRY(hipEventRecord(evStop, stream));
    HIP_TRY(hipGetLastError());
    int hostNStrays = 0;
    HIP_TRY(hipMemcpyAsync(&hostNStrays, dNStrays, sizeof(int), hipMemcpyDeviceToHost, stream));
    if (mem == PG_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(film, dFilm, filmBytes, hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        int nCopy = std::min(hostNStrays, (int)maxStrays);
        if (nCopy > 0) HIP_TRY(hipMemcpy(strays, dStrays, sizeof(PgStraySample) * (size_t)nCopy, hipMemcpyDeviceToHost));
        *nStrays = nCopy;
    } else {
        HIP_TRY(hipStreamSynchronize(stream));
        if (hostNStrays > maxStrays) { int v = maxStrays; HIP_TRY(hipMemcpy(dNStrays, &v, sizeof(int), hipMemcpyHostToDevice)); }
    }
    // statistics
    TraceCounters tc[2];
    unsigned long long lt = 0;
    HIP_TRY(hipMemcpy(tc, s->traceCn.p, sizeof(tc), hipMemcpyDeviceToHost));
    {
        std::vector<unsigned long long> shards(PG_LIGHT_TEST_SHARDS * PG_LIGHT_TEST_STRIDE);
        HIP_TRY(hipMemcpy(shards.data(), lightTests, s->lightTests.bytes, hipMemcpyDeviceToHost));
        for (int i = 0; i < PG_LIGHT_TEST_SHARDS; ++i) lt += shards[(size_t)i * PG_LIGHT_TEST_STRIDE];
        // the integrators' own statistics, words 1 .. 8 of the same shards (pg_kernels.h): like the light tests they run on from pg_counters_reset
        PgCounters &c = s->counters;
        c.paths_total = c.paths_zero_radiance = c.path_length_sum = c.path_length_count = c.path_length_min = c.path_length_max = c.volume_interactions = c.surface_interactions = 0;
        unsigned long long minc = 0, maxp = 0;
        for (int i = 0; i < PG_LIGHT_TEST_SHARDS; ++i) {
            const unsigned long long *sh = &shards[(size_t)i * PG_LIGHT_TEST_STRIDE];
            c.path_length_sum += sh[PG_STAT_LEN_SUM]; c.path_length_count += sh[PG_STAT_LEN_COUNT];
            minc = std::max(minc, sh[PG_STAT_LEN_MINC]); maxp = std::max(maxp, sh[PG_STAT_LEN_MAXP]);
            c.paths_total += sh[PG_STAT_PATHS]; c.paths_zero_radiance += sh[PG_STAT_PATHS_ZERO];
            c.volume_interactions += sh[PG_STAT_VOLUME]; c.surface_interactions += sh[PG_STAT_SURFACE];
        }
        if (c.path_length_count > 0) { c.path_length_min = 0xffff - minc; c.path_length_max = maxp - 1; }
    }
    PgCounters &c = s->counters;
    c.camera_rays += cameraRays; c.closest_rays += closestRays; c.shadow_rays += shadowRays;
    c.node_visits = tc[0].node_visits + tc[1].node_visits;
    c.tri_tests = tc[0].tri_tests + tc[1].tri_tests + lt;
    c.light_tri_tests = lt;
    c.closest_node_visits = tc[0].node_visits; c.closest_tri_tests = tc[0].tri_tests;
    c.shadow_node_visits = tc[1].node_visits; c.shadow_tri_tests = tc[1].tri_tests;
    c.closest_launches += closestLaunches; c.shadow_launches += shadowLaunches;
    c.shade_launches += shadeLaunches; c.resolve_launches += resolveLaunches; c.shade_items += shadeItems; c.mis_rays += misRays;
    c.shading_modes |= shadingModes;
    for (auto &te : timed) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, s->events[te.first], s->events[te.first + 1]) != hipSuccess) continue;
        double *acc[6] = {&c.closest_ms, &c.shadow_ms, &c.shade_ms, &c.resolve_ms, &c.generate_ms, &c.film_ms};
        *acc[te.second] += ms;
    }
#undef PG_TIMED
    float ms = 0;
    if (hipEventElapsedTime(&ms, evStart, evStop) == hipSuccess) c.render_ms += ms;
    if (int st2 = checkCullGuard(s)) return st2;
    if (hostNStrays > maxStrays) return setError(PG_ERR_OVERFLOW, "%d stray samples, buffer holds %d", hostNStrays, maxStrays);
    return PG_OK;
}

// ---- the film gather of pg_render_sharded over RCCL (ncclGather, rccl.h:745) ---------------------------------------
// librccl is opened on the first multi-device render (a single-device process never pays for loading it); one communicator per
// device list, kept for the life of the process.  Every rank sends ONE packed shard [film | strays | count] of the same size (the
// largest shard's: tile counts differ by at most one), the first device receives n of them in rank order.
#ifndef HIP_EMU_H
namespace {
typedef struct ncclComm *PgNcclComm;
struct Rccl {
    int (*CommInitAll)(PgNcclComm *, int, const int *) = nullptr;
    int (*Gather)(const void *, void *, size_t, int, int, PgNcclComm, hipStream_t) = nullptr;  // ncclGather, rccl.h:745
    const char *(*GetErrorString)(int) = nullptr;
    bool ok = false;
    std::string why;
};
Rccl *rcclApi() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, [] {
        void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!h) { r.why = std::string("librccl not found: ") + dlerror(); return; }
        r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
        r.Gather = (decltype(r.Gather))dlsym(h, "ncclGather");
        r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
        r.ok = r.CommInitAll && r.Gather && r.GetErrorString;
        if (!r.ok) r.why = "librccl lacks ncclCommInitAll / ncclGather";
    });
    return &r;
}
std::mutex g_commMutex;
std::map<std::vector<int>, std::vector<PgNcclComm>> g_comms;
// communicators of this device list, or null with the reason in `why` (then the caller gathers with peer copies)
const std::vector<PgNcclComm> *shardComms(const std::vector<int> &devices, std::string &why) {
    std::vector<int> sorted = devices;
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) { why = "a device appears twice in the list (RCCL wants one rank per GPU)"; return nullptr; }
    Rccl &r = *rcclApi();
    if (!r.ok) { why = r.why; return nullptr; }
    std::lock_guard<std::mutex> lock(g_commMutex);
    auto it = g_comms.find(devices);
    if (it != g_comms.end()) return &it->second;
    std::vector<PgNcclComm> comms(devices.size(), nullptr);
    const int st = r.CommInitAll(comms.data(), (int)devices.size(), devices.data());
    if (st != 0) { why = std::string("ncclCommInitAll: ") + r.GetErrorString(st); return nullptr; }
    return &(g_comms[devices] = comms);
}
}  // namespace
#endif
// "rccl" / "peer" (+ why RCCL was not used): how the last pg_render_sharded of this process gathered its shards.  The text is
// replaced under a lock and handed out as a pointer that stays valid for the calling thread until its next call.
static std::mutex g_shardTransportMutex;
static std::string g_shardTransport = "none";
const char *pg_shard_transport(void) {
    static thread_local std::string mine;
    std::lock_guard<std::mutex> lock(g_shardTransportMutex);
    mine = g_shardTransport;
    return mine.c_str();
}
int pg_box_filter_needs_gather(const PgRenderDesc *rd) { return rd ? pgh_box_filter_needs_gather(rd) : 0; }

// ---- one frame over several devices of the node, from one host process ---------------------------------------------------------
// One host thread per device (pg_set_device is per thread); every thread renders its tiles into a packed shard on its own device,
// then ONE ncclGather (RCCL over xGMI) brings the n shards to the first device -- or, where RCCL cannot run (a device that repeats
// in the list, no librccl, PG_SHARD_GATHER=peer), one peer-to-peer copy per rank into the same layout.  No per-bounce
// communication, no reduction: tiles are disjoint.
int pg_render_sharded(PgScene *const *scenes, int32_t n, const PgRenderDesc *desc, PgFilmPixel *const *film, PgStraySample *const *strays,
                      int32_t maxStrays, int32_t *nStrays) {
    if (!scenes || n < 1 || !desc || !film || !nStrays || (maxStrays > 0 && !strays)) return setError(PG_ERR_INVALID, "pg_render_sharded: null argument");
    if (desc->tile_first != 0 || desc->tile_step != 1) return setError(PG_ERR_INVALID, "pg_render_sharded: desc must describe the whole frame (tile_first 0, tile_step 1)");
    for (int r = 0; r < n; ++r) if (!scenes[r] || (maxStrays > 0 && !strays[r])) return setError(PG_ERR_INVALID, "pg_render_sharded: null entry for rank %d", r);
    PgScene *root = scenes[0];
    const size_t strayBytes = sizeof(PgStraySample) * (size_t)(maxStrays > 0 ? maxStrays : 1);
    std::vector<PgRenderDesc> rd((size_t)n, *desc);
    std::vector<size_t> filmBytes((size_t)n);
    size_t filmMax = sizeof(PgFilmPixel);  // never an empty buffer, also when no rank owns a tile
    for (int r = 0; r < n; ++r) {
        rd[r].tile_first = r; rd[r].tile_step = n;
        filmBytes[r] = sizeof(PgFilmPixel) * (size_t)desc->tile_pixels * (size_t)tileCount(&rd[r]);
        filmMax = std::max(filmMax, filmBytes[r]);
        // a rank that owns no tile (more devices than tiles: a small image or crop window) has nothing to receive: its film
        // pointer may be null; it still runs pg_render (which handles an empty shard) so that its counters and stray count are set
        if (filmBytes[r] && !film[r]) return setError(PG_ERR_INVALID, "pg_render_sharded: null film buffer for rank %d", r);
    }
    // one packed shard per rank, the same size for all: [film (filmMax) | strays | count]; the gathered frame is n of them in rank order
    const size_t strayOff = filmMax, countOff = strayOff + strayBytes, per = (countOff + sizeof(int) + 255) / 256 * 256;
    const size_t total = per * (size_t)n;
    // one sharded render at a time per DEVICE: the communicators of a device list carry one collective at a time and the scenes' shard / gather
    // buffers are per scene, so two calls that share a device wait for each other; calls over disjoint device sets run side by side.  The
    // devices' locks are taken in ascending order of the device number (no two calls can hold them crosswise).
    static std::mutex deviceMutex[64];
    std::vector<int> lockOrder;
    for (int r = 0; r < n; ++r) lockOrder.push_back(scenes[r]->device & 63);
    std::sort(lockOrder.begin(), lockOrder.end());
    lockOrder.erase(std::unique(lockOrder.begin(), lockOrder.end()), lockOrder.end());
    std::vector<std::unique_lock<std::mutex>> deviceLocks;
    for (int dev : lockOrder) deviceLocks.emplace_back(deviceMutex[dev]);
    HIP_TRY(hipSetDevice(root->device));
    if (root->gatherDev.bytes < total) HIP_TRY(root->gatherDev.alloc(total));  // kept between frames (hipFree + hipMalloc synchronise the device)
    char *gather = (char *)root->gatherDev.p;
    // transport: RCCL unless PG_SHARD_GATHER=peer, a device repeats (tests on a one-GPU box) or librccl cannot be used -- then peer copies
    std::string why;
#ifndef HIP_EMU_H
    const std::vector<PgNcclComm> *comms = nullptr;
    {
        const char *e = getenv("PG_SHARD_GATHER");
        std::vector<int> devices((size_t)n);
        for (int r = 0; r < n; ++r) devices[r] = scenes[r]->device;
        if (e && !strcmp(e, "peer")) why = "PG_SHARD_GATHER=peer";
        else comms = shardComms(devices, why);
        if (!comms && e && !strcmp(e, "rccl")) return setError(PG_ERR_DEVICE, "pg_render_sharded: PG_SHARD_GATHER=rccl, but %s", why.c_str());
    }
#else
    why = "emulated devices";
#endif
    std::vector<int> status((size_t)n, PG_OK);
    std::vector<std::string> message((size_t)n);
    // Every rank reaches the collective or none does: a rank whose render failed (out of memory on one device, say) must not leave
    // the other n - 1 threads waiting in ncclGather for a peer that already returned.  The threads meet at a host barrier between
    // the render and the gather; if any of them failed by then, all skip the gather and the caller gets that rank's error.
    struct { std::mutex m; std::condition_variable cv; int arrived = 0; bool anyFailed = false; } meet;
    auto meetAll = [&](bool failed) -> bool {  // returns whether any rank failed
        std::unique_lock<std::mutex> lock(meet.m);
        meet.anyFailed = meet.anyFailed || failed;
        if (++meet.arrived == n) meet.cv.notify_all();
        else meet.cv.wait(lock, [&] { return meet.arrived == n; });
        return meet.anyFailed;
    };
    auto render = [&](int r) -> bool {  // this rank's tiles into its packed shard; false = failed (status / message set)
        PgScene *s = scenes[r];
        if (hipSetDevice(s->device) != hipSuccess) { status[r] = PG_ERR_DEVICE; message[r] = "hipSetDevice failed"; return false; }
        bool viaRccl = false;
#ifndef HIP_EMU_H
        viaRccl = comms != nullptr;
#endif
        if (!viaRccl && s->device != root->device) {  // direct xGMI writes into the gathering device's buffer
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, s->device, root->device) == hipSuccess && can) {
                hipError_t e = hipDeviceEnablePeerAccess(root->device, 0);
                if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) { status[r] = PG_ERR_DEVICE; message[r] = hipGetErrorString(e); return false; }
                (void)hipGetLastError();
            }
        }
#ifdef PG_TEST_HOOKS  // fault injection for tests/test_emulated_device.py: compiled into the emulated test build only, never into libpbrt_gpu.so
        if (const char *e = getenv("PG_TEST_FAIL_RANK")) if (atoi(e) == r) { status[r] = PG_ERR_DEVICE; message[r] = "PG_TEST_FAIL_RANK (a test's injected failure)"; return false; }
#endif
        if (s->shardFilm.bytes < per && s->shardFilm.alloc(per) != hipSuccess) { status[r] = PG_ERR_DEVICE; message[r] = "out of device memory (packed shard)"; return false; }
        char *packed = (char *)s->shardFilm.p;
        int st = pg_render(s, &rd[r], (PgFilmPixel *)packed, (PgStraySample *)(packed + strayOff), maxStrays, (int32_t *)(packed + countOff), PG_MEM_DEVICE, nullptr);
        if (st != PG_OK) { status[r] = st; message[r] = pg_last_error(); return false; }
        return true;
    };
    auto work = [&](int r) {
        const bool ok = render(r);
        if (meetAll(!ok)) return;  // some rank failed: nobody enters the collective
        PgScene *s = scenes[r];
        char *packed = (char *)s->shardFilm.p;
#ifndef HIP_EMU_H
        if (comms) {
            // one collective per frame: every rank's thread calls it on its own communicator (the threads are the "different
            // threads" of rccl.h:213); root 0 receives rank r's shard at gather + r * per
            const int rs = rcclApi()->Gather(packed, r == 0 ? gather : nullptr, per, /*ncclChar*/ 0, 0, (*comms)[r], nullptr);
            if (rs != 0) { status[r] = PG_ERR_DEVICE; message[r] = std::string("ncclGather: ") + rcclApi()->GetErrorString(rs); return; }
            const hipError_t e = hipStreamSynchronize(nullptr);
            if (e != hipSuccess) { status[r] = PG_ERR_DEVICE; message[r] = std::string("after ncclGather: ") + hipGetErrorString(e); }
            return;
        }
#endif
        const hipError_t e = hipMemcpyPeer(gather + per * (size_t)r, root->device, packed, s->device, per);
        if (e != hipSuccess) { status[r] = PG_ERR_DEVICE; message[r] = std::string("peer copy of the film shard: ") + hipGetErrorString(e); }
    };
    {
        std::vector<std::thread> threads;
        for (int r = 1; r < n; ++r) threads.emplace_back(work, r);
        work(0);
        for (auto &t : threads) t.join();
    }
    for (int r = 0; r < n; ++r) if (status[r] != PG_OK) return setError(status[r], "rank %d: %s", r, message[r].c_str());
    { std::lock_guard<std::mutex> lock(g_shardTransportMutex); g_shardTransport = why.empty() ? "rccl" : "peer (" + why + ")"; }
    // the gathered frame back to the host in one piece
    HIP_TRY(hipSetDevice(root->device));
    std::vector<char> host(total);
    HIP_TRY(hipMemcpy(host.data(), gather, total, hipMemcpyDeviceToHost));
    for (int r = 0; r < n; ++r) {
        const char *shard = host.data() + per * (size_t)r;
        if (filmBytes[r]) memcpy(film[r], shard, filmBytes[r]);
        int cnt = 0;
        memcpy(&cnt, shard + countOff, sizeof(int));
        cnt = std::max(0, std::min(cnt, (int)maxStrays));
        if (cnt > 0) memcpy(strays[r], shard + strayOff, sizeof(PgStraySample) * (size_t)cnt);
        nStrays[r] = cnt;
    }
    return PG_OK;
}

// ---- batched Scene::Intersect / IntersectP -------------------------------------------
static int uploadRays(PgScene *s, int n, const float *o, const float *d, const float *tmax, int mem, hipStream_t stream) {
    // AoS (3 floats) host/device input -> the kernels' float4 SoA queue layout
    std::vector<float4> qo((size_t)n), qd((size_t)n);
    std::vector<float> ho, hd, ht;
    const float *po = o, *pd = d, *pt = tmax;
    if (mem == PG_MEM_DEVICE) {
        ho.resize(3 * (size_t)n); hd.resize(3 * (size_t)n); ht.resize((size_t)n);
        HIP_TRY(hipMemcpy(ho.data(), o, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(hd.data(), d, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(ht.data(), tmax, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost));
        po = ho.data(); pd = hd.data(); pt = ht.data();
    }
    for (int i = 0; i < n; ++i) {
        float idf;
        memcpy(&idf, &i, 4);
        qo[i] = make_float4(po[3 * i], po[3 * i + 1], po[3 * i + 2], pt[i]);
        qd[i] = make_float4(pd[3 * i], pd[3 * i + 1], pd[3 * i + 2], idf);
    }
    HIP_TRY(s->tO.alloc(sizeof(float4) * (size_t)n));
    HIP_TRY(s->tD.alloc(sizeof(float4) * (size_t)n));
    // the rays fill the regions front to back: region r holds rays [r*cap, r*cap + count(r))
    const int cap = regionCapFor(n);
    int hc[PG_REGIONS * PG_COUNT_STRIDE] = {0};
    for (int r = 0; r < PG_REGIONS; ++r) hc[r * PG_COUNT_STRIDE] = std::max(0, std::min(cap, n - r * cap));
    HIP_TRY(s->tCount.alloc(sizeof(hc)));
    HIP_TRY(hipMemcpyAsync(s->tO.p, qo.data(), s->tO.bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(s->tD.p, qd.data(), s->tD.bytes, hipMemcpyHostToDevice, stream));
    HIP_TRY(hipMemcpyAsync(s->tCount.p, hc, sizeof(hc), hipMemcpyHostToDevice, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    return PG_OK;
}
static RayQueue testQueue(PgScene *s, int n) {
    RayQueue q;
    q.o = (float4 *)s->tO.p; q.d = (float4 *)s->tD.p; q.counts = (int *)s->tCount.p; q.regionCap = regionCapFor(n);
    return q;
}

static int intersectBatch(PgScene *s, int32_t n, const float *o, const float *d, const float *tmax, int32_t *prim, float *t, float *bary, int mem,
                          void *streamPtr);
int pg_intersect(PgScene *s, int32_t n, const float *o, const float *d, const float *tmax, int32_t *prim, float *t, float *bary, int mem,
                 void *streamPtr) {
    if (!s || n < 0 || (n > 0 && (!o || !d || !tmax || !prim || !t || !bary))) return setError(PG_ERR_INVALID, "pg_intersect: null argument");
    if (n == 0) return PG_OK;
    return withExactFallback(s, [&]() { return intersectBatch(s, n, o, d, tmax, prim, t, bary, mem, streamPtr); });
}
static int intersectBatch(PgScene *s, int32_t n, const float *o, const float *d, const float *tmax, int32_t *prim, float *t, float *bary, int mem,
                          void *streamPtr) {
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = (hipStream_t)streamPtr;
    int st = uploadRays(s, n, o, d, tmax, mem, stream);
    if (st != PG_OK) return st;
    HIP_TRY(s->tHit.alloc(sizeof(float4) * (size_t)n));
    HIP_TRY(s->tT.alloc(sizeof(float) * (size_t)n));
    RayQueue q = testQueue(s, n);
    hipEvent_t a = getEvent(s, 0), b = getEvent(s, 1);
    HIP_TRY(hipEventRecord(a, stream));
    traceClosest(s, q, (float4 *)s->tHit.p, (float *)s->tT.p, (TraceCounters *)s->traceCn.p, stream);
    HIP_TRY(hipEventRecord(b, stream));
    HIP_TRY(hipGetLastError());
    std::vector<float4> hits((size_t)n);
    std::vector<float> ts((size_t)n);
    HIP_TRY(hipMemcpyAsync(hits.data(), s->tHit.p, sizeof(float4) * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipMemcpyAsync(ts.data(), s->tT.p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::vector<int32_t> hp((size_t)n);
    std::vector<float> hb(3 * (size_t)n);
    for (int i = 0; i < n; ++i) {
        memcpy(&hp[i], &hits[i].x, 4);
        bool hit = hp[i] >= 0;
        hb[3 * i] = hit ? hits[i].y : 0; hb[3 * i + 1] = hit ? hits[i].z : 0; hb[3 * i + 2] = hit ? hits[i].w : 0;
    }
    hipMemcpyKind kind = mem == PG_MEM_DEVICE ? hipMemcpyHostToDevice : hipMemcpyHostToHost;
    HIP_TRY(hipMemcpy(prim, hp.data(), sizeof(int32_t) * (size_t)n, kind));
    HIP_TRY(hipMemcpy(t, ts.data(), sizeof(float) * (size_t)n, kind));
    HIP_TRY(hipMemcpy(bary, hb.data(), sizeof(float) * 3 * (size_t)n, kind));
    TraceCounters tc[2];
    HIP_TRY(hipMemcpy(tc, s->traceCn.p, sizeof(tc), hipMemcpyDeviceToHost));
    PgCounters &c = s->counters;
    c.closest_rays += (uint64_t)n;
    c.closest_node_visits = tc[0].node_visits; c.closest_tri_tests = tc[0].tri_tests;
    c.node_visits = tc[0].node_visits + tc[1].node_visits;
    c.tri_tests = tc[0].tri_tests + tc[1].tri_tests + c.light_tri_tests;
    c.closest_launches += 1;
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess) c.closest_ms += ms;
    return checkCullGuard(s);
}

int pg_intersect_p(PgScene *s, int32_t n, const float *o, const float *d, const float *tmax, uint8_t *occluded, int mem, void *streamPtr) {
    if (!s || n < 0 || (n > 0 && (!o || !d || !tmax || !occluded))) return setError(PG_ERR_INVALID, "pg_intersect_p: null argument");
    if (n == 0) return PG_OK;
    HIP_TRY(hipSetDevice(s->device));
    hipStream_t stream = (hipStream_t)streamPtr;
    int st = uploadRays(s, n, o, d, tmax, mem, stream);
    if (st != PG_OK) return st;
    HIP_TRY(s->tOcc.alloc(sizeof(int) * (size_t)n));
    RayQueue q = testQueue(s, n);
    hipEvent_t a = getEvent(s, 0), b = getEvent(s, 1);
    HIP_TRY(hipEventRecord(a, stream));
    traceAnyhit(s, q, (int *)s->tOcc.p, (TraceCounters *)s->traceCn.p + 1, stream);
    HIP_TRY(hipEventRecord(b, stream));
    HIP_TRY(hipGetLastError());
    std::vector<int> occ((size_t)n);
    HIP_TRY(hipMemcpyAsync(occ.data(), s->tOcc.p, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, stream));
    HIP_TRY(hipStreamSynchronize(stream));
    std::vector<uint8_t> o8((size_t)n);
    for (int i = 0; i < n; ++i) o8[i] = occ[i] ? 1 : 0;
    HIP_TRY(hipMemcpy(occluded, o8.data(), (size_t)n, mem == PG_MEM_DEVICE ? hipMemcpyHostToDevice : hipMemcpyHostToHost));
    TraceCounters tc[2];
    HIP_TRY(hipMemcpy(tc, s->traceCn.p, sizeof(tc), hipMemcpyDeviceToHost));
    PgCounters &c = s->counters;
    c.shadow_rays += (uint64_t)n;
    c.shadow_node_visits = tc[1].node_visits; c.shadow_tri_tests = tc[1].tri_tests;
    c.node_visits = tc[0].node_visits + tc[1].node_visits;
    c.tri_tests = tc[0].tri_tests + tc[1].tri_tests + c.light_tri_tests;
    c.shadow_launches += 1;
    float ms = 0;
    if (hipEventElapsedTime(&ms, a, b) == hipSuccess) c.shadow_ms += ms;
    return PG_OK;
}

int pg_counters(PgScene *s, PgCounters *out) {
    if (!s || !out) return setError(PG_ERR_INVALID, "pg_counters: null argument");
    *out = s->counters;
    return PG_OK;
}
int pg_scene_set_option(PgScene *s, int32_t option, int32_t value) {
    if (!s) return setError(PG_ERR_INVALID, "pg_scene_set_option: null scene");
    switch (option) {
    case PG_OPT_OVERLAP_SHADOW: s->overlapShadow = value != 0; return PG_OK;
    default: return setError(PG_ERR_INVALID, "pg_scene_set_option: unknown option %d", option);
    }
}
int pg_counters_reset(PgScene *s) {
    if (!s) return setError(PG_ERR_INVALID, "pg_counters_reset: null argument");
    HIP_TRY(hipSetDevice(s->device));
    memset(&s->counters, 0, sizeof(s->counters));
    HIP_TRY(hipMemset(s->traceCn.p, 0, s->traceCn.bytes));
    HIP_TRY(hipMemset(s->lightTests.p, 0, s->lightTests.bytes));
    return PG_OK;
}
}  // extern "C"
