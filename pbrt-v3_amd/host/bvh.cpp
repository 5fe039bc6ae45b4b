// BVHAccel construction on the host (the build stays on the CPU; SURVEY.md
// section 8 row a7).  Restates src/accelerators/bvh.cpp:183-402 (recursive SAH /
// Middle / EqualCounts build with 12 buckets) and :640-658 (flattenBVHTree)
// with the same float arithmetic and the same std::partition/std::nth_element
// calls, so the node array and leaf contents match the reference's tree.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <future>
#include <thread>
#include "scene.h"
#include "error.h"
#include "api.h"

namespace pbrt {

Bounds3f Triangle::WorldBound() const {  // triangle.cpp:180-186
    const int *vi = v();
    return Union(Bounds3f(mesh->p[vi[0]], mesh->p[vi[1]]), mesh->p[vi[2]]);
}
Float Triangle::Area() const {  // triangle.cpp:574-580; 0.5 is a double literal there
    const int *vi = v();
    const Point3f &p0 = mesh->p[vi[0]], &p1 = mesh->p[vi[1]], &p2 = mesh->p[vi[2]];
    return 0.5 * Cross(p1 - p0, p2 - p0).Length();
}

Sphere::Sphere(const Transform &o2w, const Transform &w2o, bool ro, Float r, Float z0, Float z1, Float pm)  // sphere.h:50-60
    : ObjectToWorld(o2w), WorldToObject(w2o), reverseOrientation(ro), transformSwapsHandedness(o2w.SwapsHandedness()), radius(r),
      zMin(Clamp(std::min(z0, z1), -r, r)), zMax(Clamp(std::max(z0, z1), -r, r)),
      thetaMin(std::acos(Clamp(std::min(z0, z1) / r, -1, 1))), thetaMax(std::acos(Clamp(std::max(z0, z1) / r, -1, 1))),
      phiMax(Radians(Clamp(pm, 0, 360))) {}
std::shared_ptr<Sphere> Sphere::Cylinder(const Transform &o2w, const Transform &w2o, bool ro, Float r, Float z0, Float z1, Float pm) {  // cylinder.h:50-57
    auto c = std::make_shared<Sphere>(o2w, w2o, ro, r, -r, r, pm);
    c->shape = PG_SHAPE_CYLINDER;
    c->zMin = std::min(z0, z1); c->zMax = std::max(z0, z1);
    c->thetaMin = c->thetaMax = 0;
    return c;
}
std::shared_ptr<Sphere> Sphere::Disk(const Transform &o2w, const Transform &w2o, bool ro, Float h, Float r, Float ri, Float pm) {  // disk.h:50-57
    auto d = std::make_shared<Sphere>(o2w, w2o, ro, r, -r, r, pm);
    d->shape = PG_SHAPE_DISK;
    d->height = h; d->innerRadius = ri;
    d->zMin = d->zMax = h;  // ObjectBound: z = height on both corners (disk.cpp:43-46)
    d->thetaMin = d->thetaMax = 0;
    return d;
}
std::shared_ptr<Sphere> Sphere::Cone(const Transform &o2w, const Transform &w2o, bool ro, Float h, Float r, Float pm) {  // cone.cpp:43-53
    auto c = std::make_shared<Sphere>(o2w, w2o, ro, r, -r, r, pm);
    c->shape = PG_SHAPE_CONE;
    c->height = h; c->zMin = 0; c->zMax = h;  // ObjectBound: (-r, -r, 0) .. (r, r, height)
    c->thetaMin = c->thetaMax = 0;
    return c;
}
std::shared_ptr<Sphere> Sphere::Paraboloid(const Transform &o2w, const Transform &w2o, bool ro, Float r, Float z0, Float z1, Float pm) {  // paraboloid.cpp:43-54
    auto c = std::make_shared<Sphere>(o2w, w2o, ro, r, -r, r, pm);
    c->shape = PG_SHAPE_PARABOLOID;
    c->zMin = std::min(z0, z1); c->zMax = std::max(z0, z1);
    c->thetaMin = c->thetaMax = 0;
    return c;
}
std::shared_ptr<Sphere> Sphere::Hyperboloid(const Transform &o2w, const Transform &w2o, bool ro, Point3f p1, Point3f p2, Float pm) {  // hyperboloid.cpp:43-71
    const Float radius1 = std::sqrt(p1.x * p1.x + p1.y * p1.y), radius2 = std::sqrt(p2.x * p2.x + p2.y * p2.y);
    auto c = std::make_shared<Sphere>(o2w, w2o, ro, std::max(radius1, radius2), -1, 1, pm);  // radius = rMax
    c->shape = PG_SHAPE_HYPERBOLOID;
    c->zMin = std::min(p1.z, p2.z); c->zMax = std::max(p1.z, p2.z);
    c->thetaMin = c->thetaMax = 0;
    // the implicit form ah (x^2 + y^2) - ch z^2 = 1 through both end points, found from a point further along the line
    if (p2.z == 0.f) std::swap(p1, p2);
    Point3f pp = p1;
    Float ah, ch;
    int guard = 0;
    do {
        pp = pp + (p2 - p1) * (Float)2.;
        const Float xy1 = pp.x * pp.x + pp.y * pp.y, xy2 = p2.x * p2.x + p2.y * p2.y;
        ah = (1.f / xy1 - (pp.z * pp.z) / (xy1 * p2.z * p2.z)) / (1 - (xy2 * pp.z * pp.z) / (xy1 * p2.z * p2.z));
        ch = (ah * xy2 - 1) / (p2.z * p2.z);
    } while ((std::isinf(ah) || std::isnan(ah)) && ++guard < 1000);  // the reference loops for ever on degenerate end points
    c->p1 = p1; c->p2 = p2; c->ah = ah; c->ch = ch;
    return c;
}
Float Sphere::Area() const {
    if (shape == PG_SHAPE_CONE) return radius * std::sqrt((height * height) + (radius * radius)) * phiMax / 2;  // cone.cpp:200-203
    if (shape == PG_SHAPE_PARABOLOID) {  // paraboloid.cpp:204-209
        const Float radius2 = radius * radius, k = 4 * zMax / radius2;
        return (radius2 * radius2 * phiMax / (12 * zMax * zMax)) * (std::pow(k * zMax + 1, 1.5f) - std::pow(k * zMin + 1, 1.5f));
    }
    if (shape == PG_SHAPE_HYPERBOLOID) return 0;  // only an emitter would read it, and these shapes cannot emit
    if (shape == PG_SHAPE_CYLINDER) return (zMax - zMin) * radius * phiMax;
    if (shape == PG_SHAPE_DISK) return phiMax * 0.5 * (radius * radius - innerRadius * innerRadius);  // 0.5 is a double literal there
    return phiMax * radius * (zMax - zMin);
}
Bounds3f Sphere::WorldBound() const {  // Transform::operator()(Bounds3f), transform.cpp:237-249
    const Float lo[3] = {-radius, -radius, zMin}, hi[3] = {radius, radius, zMax};  // ObjectBound
    Bounds3f ret;
    for (int corner = 0; corner < 8; ++corner) {  // min/max over the eight transformed corners: the order does not matter
        Point3f c((corner & 1) ? hi[0] : lo[0], (corner & 2) ? hi[1] : lo[1], (corner & 4) ? hi[2] : lo[2]);
        Point3f w = ObjectToWorld.Pt(c);
        ret = corner == 0 ? Bounds3f(w) : Union(ret, w);
    }
    return ret;
}

static Bounds3f TransformBounds(const Transform &t, const Bounds3f &b) {  // Transform::operator()(Bounds3f), transform.cpp:237-249
    Bounds3f ret;
    for (int corner = 0; corner < 8; ++corner) {
        Point3f c((corner & 1) ? b.pMax.x : b.pMin.x, (corner & 2) ? b.pMax.y : b.pMin.y, (corner & 4) ? b.pMax.z : b.pMin.z);
        Point3f w = t.Pt(c);
        ret = corner == 0 ? Bounds3f(w) : Union(ret, w);
    }
    return ret;
}
Bounds3f ObjectDefinition::WorldBound() const {  // the BVHAccel's root bounds, or the lone primitive's
    if (accel) return accel->WorldBound();
    return prims.empty() ? Bounds3f() : prims[0].WorldBound();
}
Bounds3f GeometricPrimitive::WorldBound() const {
    // TransformedPrimitive::WorldBound (primitive.h:107-109) -> AnimatedTransform::MotionBounds (transform.cpp:1215-1224): the start
    // transform's box, the union of the two ends' boxes, or -- when the motion rotates -- the corners' paths bounded at the zeros of their
    // derivatives (host/motion_bounds.cpp)
    if (object) {
        const Bounds3f b = object->WorldBound();
        if (xf->animated) return MotionBounds(xf->InstanceToWorld, xf->time[0], xf->InstanceToWorldEnd, xf->time[1], b);
        return TransformBounds(xf->InstanceToWorld, b);
    }
    return sphere ? sphere->WorldBound() : shape.WorldBound();
}

struct BVHAccel::PrimInfo {  // what the build knows of a primitive (BVHPrimitiveInfo, bvh.cpp:49-59)
    size_t primitiveNumber = 0;
    Bounds3f bounds;
    Point3f centroid;  // [order] .5f * pMin + .5f * pMax, not (pMin + pMax) / 2
    PrimInfo() {}
    PrimInfo(size_t index, const Bounds3f &b) : primitiveNumber(index), bounds(b) {
        for (int k = 0; k < 3; ++k) centroid[k] = .5f * b.pMin[k] + .5f * b.pMax[k];
    }
};
struct BVHAccel::BuildNode {  // BVHBuildNode, bvh.cpp:61-83
    void InitLeaf(int first, int n, const Bounds3f &b) {
        firstPrimOffset = first; nPrimitives = n; bounds = b; children[0] = children[1] = nullptr;
        nNodes = 1;
    }
    void InitInterior(int axis, BuildNode *c0, BuildNode *c1) {
        children[0] = c0; children[1] = c1;
        bounds = Union(c0->bounds, c1->bounds);
        splitAxis = axis; nPrimitives = 0;
        nNodes = 1 + c0->nNodes + c1->nNodes;
    }
    Bounds3f bounds;
    BuildNode *children[2];
    int splitAxis, firstPrimOffset, nPrimitives;
    int nNodes;  // nodes of this subtree: where its second child lands in the depth-first array is known before it is written
};

BVHAccel::BuildNode *BVHAccel::allocNode() {
    // every thread carves nodes out of its own chunk; only fetching a new chunk takes the lock
    const size_t chunk = 1 << 14;
    thread_local uint64_t owner = 0;
    thread_local BuildNode *cur = nullptr, *chunkEnd = nullptr;
    if (owner != buildId || cur == chunkEnd) {
        std::lock_guard<std::mutex> lock(arenaMutex);
        arena.emplace_back(new BuildNode[chunk]);
        owner = buildId; cur = arena.back().get(); chunkEnd = cur + chunk;
    }
    return cur++;
}

static std::atomic<uint64_t> g_nextBuildId(1);
BVHAccel::BVHAccel(std::vector<GeometricPrimitive> p, int maxPrims, SplitMethod sm)
    : primitives(std::move(p)), maxPrimsInNode(std::min(255, maxPrims)), splitMethod(sm), buildId(g_nextBuildId++) {
    if (primitives.empty()) return;
    const bool timing = getenv("PBRT_HOST_TIMING") != nullptr;
    auto tick = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        auto now = std::chrono::steady_clock::now();
        if (timing) fprintf(stderr, "pbrt host: BVHAccel %s %.3f s\n", what, std::chrono::duration<double>(now - tick).count());
        tick = now;
    };
    std::vector<PrimInfo> primitiveInfo(primitives.size());
    {   // every primitive's bound on its own: split over the build's threads
        int nt = PbrtOptions.nThreads > 0 ? PbrtOptions.nThreads : (int)std::thread::hardware_concurrency();
        if (const char *e = getenv("PBRT_NTHREADS")) { if (atoi(e) > 0) nt = atoi(e); }
        nt = std::max(1, std::min(nt, 64));
        if (primitives.size() < (size_t)(1 << 16)) nt = 1;
        auto fill = [&](size_t b, size_t e) { for (size_t i = b; i < e; ++i) primitiveInfo[i] = {i, primitives[i].WorldBound()}; };
        std::vector<std::thread> pool;
        const size_t per = (primitives.size() + nt - 1) / nt;
        for (int t = 1; t < nt; ++t) pool.emplace_back(fill, std::min(primitives.size(), t * per), std::min(primitives.size(), (t + 1) * per));
        fill(0, std::min(primitives.size(), per));
        for (auto &th : pool) th.join();
    }
    lap("primitive bounds");
    int totalNodes = 0;
    std::vector<GeometricPrimitive> orderedPrims;
    if (splitMethod == SplitMethod::HLBVH) {  // bvh.cpp:207-212
        std::vector<int> order;
        if (PbrtOptions.deviceBVH) {
            std::vector<float> b(6 * primitives.size());
            for (size_t i = 0; i < primitives.size(); ++i)
                for (int k = 0; k < 3; ++k) { b[6 * i + k] = primitiveInfo[i].bounds.pMin[k]; b[6 * i + 3 + k] = primitiveInfo[i].bounds.pMax[k]; }
            if (!DeviceHLBVHBuild((int)primitives.size(), b.data(), maxPrimsInNode, &nodes, &order)) { Error("HLBVH: the device build failed (there is no silent host fallback)."); Fatal(); }
        } else {
            BuildNode *root = HLBVHBuild(primitiveInfo, &totalNodes, order);
            nodes.resize(totalNodes);
            flattenBVHTree(root, 0, 0);
        }
        orderedPrims.resize(primitives.size());
        for (size_t i = 0; i < primitives.size(); ++i) orderedPrims[i] = primitives[order[i]];
        primitives.swap(orderedPrims);
        arena.clear();
        return;
    }
    orderedPrims.resize(primitives.size());
    std::atomic<int> nodeCount(0);
    // --nthreads (0 = all cores): 2^spawnDepth subtrees are built concurrently
    int threads = PbrtOptions.nThreads > 0 ? PbrtOptions.nThreads : (int)std::thread::hardware_concurrency();
    if (const char *e = getenv("PBRT_NTHREADS")) { if (atoi(e) > 0) threads = atoi(e); }  // for hosts that drive the C API (no --nthreads there)
    int spawnDepth = 0;
    while ((2 << spawnDepth) <= std::max(1, threads) && spawnDepth < 6) ++spawnDepth;
    lap("set-up");
    BuildNode *root = recursiveBuild(primitiveInfo, 0, (int)primitives.size(), &nodeCount, orderedPrims, spawnDepth);
    if (timing) fprintf(stderr, "pbrt host: recursiveBuild of %zu primitives on up to %d threads\n", primitives.size(), 1 << spawnDepth);
    lap("recursiveBuild");
    totalNodes = root->nNodes;
    primitives.swap(orderedPrims);
    nodes.resize(totalNodes);
    flattenBVHTree(root, 0, spawnDepth);
    lap("flatten");
    arena.clear();
    lap("free build nodes");
}
BVHAccel::BVHAccel(int maxPrims, SplitMethod sm) : maxPrimsInNode(std::min(255, maxPrims)), splitMethod(sm), buildId(g_nextBuildId++) {}
void BVHAccel::HLBVHFromBounds(int n, const float *bounds, int maxPrimsInNode, std::vector<PgBVHNode> *nodes, std::vector<int> *order) {
    nodes->clear(); order->clear();
    if (n <= 0) return;
    BVHAccel a(maxPrimsInNode, SplitMethod::HLBVH);
    std::vector<PrimInfo> primitiveInfo((size_t)n);
    for (int i = 0; i < n; ++i) {
        Bounds3f b;
        b.pMin = Point3f(bounds[6 * i], bounds[6 * i + 1], bounds[6 * i + 2]); b.pMax = Point3f(bounds[6 * i + 3], bounds[6 * i + 4], bounds[6 * i + 5]);
        primitiveInfo[i] = {(size_t)i, b};
    }
    int totalNodes = 0;
    BuildNode *root = a.HLBVHBuild(primitiveInfo, &totalNodes, *order);
    a.nodes.resize(totalNodes);
    a.flattenBVHTree(root, 0, 0);
    nodes->swap(a.nodes);
}

Bounds3f BVHAccel::WorldBound() const {  // bvh.cpp:228-230
    if (nodes.empty()) return Bounds3f();
    Bounds3f b;
    b.pMin = Point3f(nodes[0].bmin[0], nodes[0].bmin[1], nodes[0].bmin[2]);
    b.pMax = Point3f(nodes[0].bmax[0], nodes[0].bmax[1], nodes[0].bmax[2]);
    return b;
}

// The 12-bucket surface-area-heuristic sweep shared by recursiveBuild and buildUpperSAH.  For every boundary i between
// buckets the reference re-unions buckets 0..i and i+1..11 (bvh.cpp:329-345, :599-614); here one forward and one backward
// pass over the buckets produce the same eleven (bounds, count) pairs -- unions are min / max of floats and the counts are
// integers, so both are exact and the costs come out bit-identical: base + (n0 * area(b0) + n1 * area(b1)) / area(all).
struct SahBuckets {
    static constexpr int N = 12;
    int count[N] = {};
    Bounds3f box[N];
    void Add(int b, const Bounds3f &bounds) { ++count[b]; box[b] = Union(box[b], bounds); }
    // bucket index after which to split (the FIRST of the cheapest boundaries) and its cost
    int CheapestBoundary(Float base, const Bounds3f &all, Float *costOut) const {
        Bounds3f above[N];  // above[i] = union of buckets i+1 .. N-1
        int nAbove[N];
        Bounds3f acc;
        int n = 0;
        for (int i = N - 1; i >= 0; --i) { above[i] = acc; nAbove[i] = n; acc = Union(acc, box[i]); n += count[i]; }
        const Float totalArea = all.SurfaceArea();
        acc = Bounds3f(); n = 0;
        int best = 0;
        Float bestCost = 0;
        for (int i = 0; i < N - 1; ++i) {
            acc = Union(acc, box[i]); n += count[i];
            const Float c = base + (n * acc.SurfaceArea() + nAbove[i] * above[i].SurfaceArea()) / totalArea;
            if (i == 0 || c < bestCost) { bestCost = c; best = i; }
        }
        *costOut = bestCost;
        return best;
    }
};

BVHAccel::BuildNode *BVHAccel::recursiveBuild(std::vector<PrimInfo> &primitiveInfo, int start, int end, std::atomic<int> *totalNodes,
                                              std::vector<GeometricPrimitive> &orderedPrims, int spawnDepth) {
    BuildNode *node = allocNode();  // (counted through BuildNode::nNodes: an atomic counter here is bumped by every thread for every node)
    Bounds3f bounds;
    for (int i = start; i < end; ++i) bounds = Union(bounds, primitiveInfo[i].bounds);
    int nPrimitives = end - start;
    auto makeLeaf = [&]() {
        int firstPrimOffset = start;  // == orderedPrims.size() at this point of the reference's depth-first build
        // moved, not copied: every primitive lands in exactly one leaf, and a copy would bump the mesh's shared reference count from
        // all threads at once
        for (int i = start; i < end; ++i) orderedPrims[i] = std::move(primitives[primitiveInfo[i].primitiveNumber]);
        node->InitLeaf(firstPrimOffset, nPrimitives, bounds);
        return node;
    };
    if (nPrimitives == 1) return makeLeaf();
    Bounds3f centroidBounds;
    for (int i = start; i < end; ++i) centroidBounds = Union(centroidBounds, primitiveInfo[i].centroid);
    int dim = centroidBounds.MaximumExtent();
    int mid = (start + end) / 2;
    if (centroidBounds.pMax[dim] == centroidBounds.pMin[dim]) return makeLeaf();

    auto equalCounts = [&]() {
        mid = (start + end) / 2;
        std::nth_element(&primitiveInfo[start], &primitiveInfo[mid], &primitiveInfo[end - 1] + 1,
                         [dim](const PrimInfo &a, const PrimInfo &b) { return a.centroid[dim] < b.centroid[dim]; });
    };
    switch (splitMethod) {
    case SplitMethod::Middle: {
        Float pmid = (centroidBounds.pMin[dim] + centroidBounds.pMax[dim]) / 2;
        PrimInfo *midPtr = std::partition(&primitiveInfo[start], &primitiveInfo[end - 1] + 1,
                                          [dim, pmid](const PrimInfo &pi) { return pi.centroid[dim] < pmid; });
        mid = int(midPtr - &primitiveInfo[0]);
        if (mid != start && mid != end) break;
        equalCounts();  // fall through to EqualCounts (bvh.cpp:292-296)
        break;
    }
    case SplitMethod::EqualCounts: equalCounts(); break;
    case SplitMethod::SAH:
    default: {
        if (nPrimitives <= 2) { equalCounts(); break; }
        constexpr int nBuckets = SahBuckets::N;
        auto bucketOf = [&centroidBounds, dim](const PrimInfo &pi) {  // bvh.cpp:318-322: the centroid's twelfth along the split axis
            int b = nBuckets * centroidBounds.Offset(pi.centroid)[dim];
            if (b == nBuckets) b = nBuckets - 1;
            // bvh.cpp:323-324 CHECK_GE(b, 0), CHECK_LT(b, nBuckets): non-finite or overflowing vertex coordinates end here in the
            // reference too (it aborts); the library reports the scene instead of indexing out of bounds
            if (b < 0 || b >= nBuckets) { Error("BVHAccel: a primitive's centroid falls into no SAH bucket (non-finite or overflowing bounds)"); Fatal(); }
            return b;
        };
        SahBuckets buckets;
        for (int i = start; i < end; ++i) buckets.Add(bucketOf(primitiveInfo[i]), primitiveInfo[i].bounds);
        Float minCost;
        const int minCostSplitBucket = buckets.CheapestBoundary(1, bounds, &minCost);
        Float leafCost = nPrimitives;
        if (nPrimitives > maxPrimsInNode || minCost < leafCost) {
            PrimInfo *pmid = std::partition(&primitiveInfo[start], &primitiveInfo[end - 1] + 1,
                                            [&](const PrimInfo &pi) { return bucketOf(pi) <= minCostSplitBucket; });
            mid = int(pmid - &primitiveInfo[0]);
        } else
            return makeLeaf();
        break;
    }
    }
    // The reference passes both recursive calls as function arguments
    // (bvh.cpp:393-397); their evaluation order only permutes whole leaf
    // blocks inside orderedPrims, never a leaf's contents or the tree shape.
    BuildNode *c0, *c1;
    if (spawnDepth > 0 && nPrimitives >= (1 << 15)) {  // the two halves touch disjoint ranges of primitiveInfo / orderedPrims
        std::future<BuildNode *> first = std::async(std::launch::async, [&]() { return recursiveBuild(primitiveInfo, start, mid, totalNodes, orderedPrims, spawnDepth - 1); });
        c1 = recursiveBuild(primitiveInfo, mid, end, totalNodes, orderedPrims, spawnDepth - 1);
        c0 = first.get();
    } else {
        c0 = recursiveBuild(primitiveInfo, start, mid, totalNodes, orderedPrims, 0);
        c1 = recursiveBuild(primitiveInfo, mid, end, totalNodes, orderedPrims, 0);
    }
    node->InitInterior(dim, c0, c1);
    return node;
}

// flattenBVHTree (bvh.cpp:640-658): the same depth-first layout, written at offsets computed from the subtree sizes -- a node at
// `offset` has its first child at offset + 1 and its second at offset + 1 + nodes(first child) -- so subtrees do not wait for
// one another and the large ones go to their own threads (the serial recursion was half of a 5 M-triangle scene's load time
// on the GPU box's 256 threads).
void BVHAccel::flattenBVHTree(BuildNode *node, int offset, int spawnDepth) {
    for (;;) {
        PgBVHNode *linearNode = &nodes[offset];
        for (int i = 0; i < 3; ++i) { linearNode->bmin[i] = node->bounds.pMin[i]; linearNode->bmax[i] = node->bounds.pMax[i]; }
        linearNode->pad = 0;
        if (node->nPrimitives > 0) {
            linearNode->offset = node->firstPrimOffset;
            linearNode->nprims = (uint16_t)node->nPrimitives;
            linearNode->axis = 0;
            return;
        }
        linearNode->axis = (uint8_t)node->splitAxis;
        linearNode->nprims = 0;
        const int second = offset + 1 + node->children[0]->nNodes;
        linearNode->offset = second;
        if (spawnDepth > 0 && node->nNodes >= (1 << 16)) {
            BuildNode *c0 = node->children[0];
            std::future<void> first = std::async(std::launch::async, [this, c0, offset, spawnDepth]() { flattenBVHTree(c0, offset + 1, spawnDepth - 1); });
            flattenBVHTree(node->children[1], second, spawnDepth - 1);
            first.get();
            return;
        }
        flattenBVHTree(node->children[0], offset + 1, 0);
        node = node->children[1];  // (tail call)
        offset = second;
        spawnDepth = 0;
    }
}

// ---- HLBVH (bvh.cpp:107-180, :404-638): Morton codes of the centroids, a stable radix sort, one LBVH treelet per run of
// equal top-12 Morton bits, an SAH tree over the treelet roots.  The reference builds the treelets on several threads and
// hands out leaf ranges of orderedPrims with an atomic counter; this is its single-thread order (treelets in index order),
// in which orderedPrims is simply the Morton-sorted primitive list.  Ray results do not depend on that order.
struct BVHAccel::MortonPrim { int primitiveIndex; uint32_t mortonCode; };
static uint32_t LeftShift3(uint32_t x) {  // bvh.cpp:107-131: spread the low 10 bits to every third position
    if (x == (1 << 10)) --x;
    x = (x | (x << 16)) & 0x30000ff;
    x = (x | (x << 8)) & 0x300f00f;
    x = (x | (x << 4)) & 0x30c30c3;
    x = (x | (x << 2)) & 0x9249249;
    return x;
}
BVHAccel::BuildNode *BVHAccel::HLBVHBuild(const std::vector<PrimInfo> &primitiveInfo, int *totalNodes, std::vector<int> &order) {
    Bounds3f bounds;
    for (const PrimInfo &pi : primitiveInfo) bounds = Union(bounds, pi.centroid);
    std::vector<MortonPrim> mortonPrims(primitiveInfo.size());
    const int mortonScale = 1 << 10;
    for (size_t i = 0; i < primitiveInfo.size(); ++i) {
        mortonPrims[i].primitiveIndex = (int)primitiveInfo[i].primitiveNumber;
        Vector3f o = bounds.Offset(primitiveInfo[i].centroid);
        Vector3f v(o.x * mortonScale, o.y * mortonScale, o.z * mortonScale);  // centroidOffset * mortonScale
        mortonPrims[i].mortonCode = (LeftShift3((uint32_t)v.z) << 2) | (LeftShift3((uint32_t)v.y) << 1) | LeftShift3((uint32_t)v.x);
    }
    // RadixSort (bvh.cpp:140-180) is a stable LSD sort of the 30-bit codes
    std::stable_sort(mortonPrims.begin(), mortonPrims.end(), [](const MortonPrim &a, const MortonPrim &b) { return a.mortonCode < b.mortonCode; });
    struct Treelet { int startIndex, nPrimitives; BuildNode *root; };
    std::vector<Treelet> treelets;
    const uint32_t mask = 0x3ffc0000;  // the top 12 of the 30 bits
    for (int start = 0, end = 1; end <= (int)mortonPrims.size(); ++end)
        if (end == (int)mortonPrims.size() || ((mortonPrims[start].mortonCode & mask) != (mortonPrims[end].mortonCode & mask))) {
            treelets.push_back({start, end - start, nullptr});
            start = end;
        }
    int orderedPrimsOffset = 0;
    order.resize(primitiveInfo.size());
    for (Treelet &tr : treelets)
        tr.root = emitLBVH(primitiveInfo, &mortonPrims[tr.startIndex], tr.nPrimitives, totalNodes, order, &orderedPrimsOffset, 29 - 12);
    std::vector<BuildNode *> finishedTreelets;
    for (Treelet &tr : treelets) finishedTreelets.push_back(tr.root);
    return buildUpperSAH(finishedTreelets, 0, (int)finishedTreelets.size(), totalNodes);
}
BVHAccel::BuildNode *BVHAccel::emitLBVH(const std::vector<PrimInfo> &primitiveInfo, const MortonPrim *mortonPrims, int nPrimitives, int *totalNodes,
                                        std::vector<int> &order, int *orderedPrimsOffset, int bitIndex) {
    if (bitIndex == -1 || nPrimitives < maxPrimsInNode) {  // a leaf (note: strictly fewer than maxPrimsInNode)
        (*totalNodes)++;
        BuildNode *node = allocNode();
        Bounds3f bounds;
        int firstPrimOffset = *orderedPrimsOffset;
        *orderedPrimsOffset += nPrimitives;
        for (int i = 0; i < nPrimitives; ++i) {
            int primitiveIndex = mortonPrims[i].primitiveIndex;
            order[firstPrimOffset + i] = primitiveIndex;
            bounds = Union(bounds, primitiveInfo[primitiveIndex].bounds);
        }
        node->InitLeaf(firstPrimOffset, nPrimitives, bounds);
        return node;
    }
    const uint32_t mask = 1u << bitIndex;
    if ((mortonPrims[0].mortonCode & mask) == (mortonPrims[nPrimitives - 1].mortonCode & mask))  // no split on this bit
        return emitLBVH(primitiveInfo, mortonPrims, nPrimitives, totalNodes, order, orderedPrimsOffset, bitIndex - 1);
    int searchStart = 0, searchEnd = nPrimitives - 1;  // first primitive whose bit differs from the run's first
    while (searchStart + 1 != searchEnd) {
        int mid = (searchStart + searchEnd) / 2;
        if ((mortonPrims[searchStart].mortonCode & mask) == (mortonPrims[mid].mortonCode & mask)) searchStart = mid;
        else searchEnd = mid;
    }
    const int splitOffset = searchEnd;
    (*totalNodes)++;
    BuildNode *node = allocNode();
    BuildNode *c0 = emitLBVH(primitiveInfo, mortonPrims, splitOffset, totalNodes, order, orderedPrimsOffset, bitIndex - 1);
    BuildNode *c1 = emitLBVH(primitiveInfo, &mortonPrims[splitOffset], nPrimitives - splitOffset, totalNodes, order, orderedPrimsOffset, bitIndex - 1);
    node->InitInterior(bitIndex % 3, c0, c1);
    return node;
}
BVHAccel::BuildNode *BVHAccel::buildUpperSAH(std::vector<BuildNode *> &treeletRoots, int start, int end, int *totalNodes) {  // bvh.cpp:537-638
    int nNodes = end - start;
    if (nNodes == 1) return treeletRoots[start];
    (*totalNodes)++;
    BuildNode *node = allocNode();
    Bounds3f bounds;
    for (int i = start; i < end; ++i) bounds = Union(bounds, treeletRoots[i]->bounds);
    Bounds3f centroidBounds;
    for (int i = start; i < end; ++i) {
        Point3f centroid = (treeletRoots[i]->bounds.pMin + treeletRoots[i]->bounds.pMax) * 0.5f;
        centroidBounds = Union(centroidBounds, centroid);
    }
    const int dim = centroidBounds.MaximumExtent();
    const int nBuckets = SahBuckets::N;
    auto bucketOf = [&](const BuildNode *n) {
        Float centroid = (n->bounds.pMin[dim] + n->bounds.pMax[dim]) * 0.5f;
        int b = nBuckets * ((centroid - centroidBounds.pMin[dim]) / (centroidBounds.pMax[dim] - centroidBounds.pMin[dim]));
        if (b == nBuckets) b = nBuckets - 1;
        if (b < 0 || b >= nBuckets) { Error("HLBVH: a treelet's centroid falls into no SAH bucket (non-finite or overflowing bounds)"); Fatal(); }  // bvh.cpp:583-584 CHECKs
        return b;
    };
    SahBuckets buckets;
    for (int i = start; i < end; ++i) buckets.Add(bucketOf(treeletRoots[i]), treeletRoots[i]->bounds);
    Float minCost;
    const int minCostSplitBucket = buckets.CheapestBoundary(.125f, bounds, &minCost);  // traversal cost 1/8 up here (bvh.cpp:611)
    BuildNode **pmid = std::partition(&treeletRoots[start], &treeletRoots[end - 1] + 1, [&](const BuildNode *n) { return bucketOf(n) <= minCostSplitBucket; });
    int mid = (int)(pmid - &treeletRoots[0]);
    if (mid <= start || mid >= end) {  // the reference CHECK-fails here (degenerate centroid bounds)
        Error("HLBVH: degenerate split of %d treelets; falling back to a median split.", nNodes);
        mid = (start + end) / 2;
    }
    BuildNode *c0 = buildUpperSAH(treeletRoots, start, mid, totalNodes);
    BuildNode *c1 = buildUpperSAH(treeletRoots, mid, end, totalNodes);
    node->InitInterior(dim, c0, c1);
    return node;
}

std::shared_ptr<BVHAccel> CreateDefaultBVHAccel(std::vector<GeometricPrimitive> prims) { return std::make_shared<BVHAccel>(std::move(prims)); }
std::shared_ptr<BVHAccel> CreateBVHAccelerator(std::vector<GeometricPrimitive> prims, const ParamSet &ps) {
    // Accelerator "bvh": "string splitmethod" sah | hlbvh | middle | equal (default sah), "integer maxnodeprims" (default 4)
    static const struct { const char *name; BVHAccel::SplitMethod method; } kMethods[] = {
        {"sah", BVHAccel::SplitMethod::SAH}, {"hlbvh", BVHAccel::SplitMethod::HLBVH},
        {"middle", BVHAccel::SplitMethod::Middle}, {"equal", BVHAccel::SplitMethod::EqualCounts}};
    const std::string wanted = ps.FindOneString("splitmethod", "sah");
    const BVHAccel::SplitMethod *found = nullptr;
    for (const auto &m : kMethods) if (wanted == m.name) found = &m.method;
    if (!found) Warning("BVH split method \"%s\" unknown.  Using \"sah\".", wanted.c_str());
    return std::make_shared<BVHAccel>(std::move(prims), ps.FindOneInt("maxnodeprims", 4), found ? *found : BVHAccel::SplitMethod::SAH);
}
}  // namespace pbrt
