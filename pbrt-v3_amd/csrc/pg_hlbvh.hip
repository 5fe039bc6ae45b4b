// pg_hlbvh.hip -- BVHAccel::HLBVHBuild on the device (accelerators/bvh.cpp:404-658), behind pg_hlbvh_build.
//
//   k_centroid_bounds  Union of the primitives' centroids                      bvh.cpp:408-410
//   k_morton           Morton code of every centroid                           bvh.cpp:415-429, :107-131
//   k_radix_*          stable LSD radix sort of (code, index) on the 30 code bits  bvh.cpp:432, :140-180
//   k_treelet_starts   runs of equal top-12 Morton bits                        bvh.cpp:437-456
//   k_emit_lbvh        one lane per treelet: emitLBVH in pre-order             bvh.cpp:485-535
//   (host)             buildUpperSAH over <= 4096 treelet roots + the layout   bvh.cpp:537-638, :640-658
//   k_flatten          treelet nodes -> LinearBVHNode at their final offsets   bvh.cpp:640-658
//
// Everything a float result depends on (centroids, Bounds3f::Offset, the unions, the SAH costs) is evaluated with the
// reference's operations in the reference's order, so nodes and primitive order equal the reference's single-thread build.
// HBM-bound integer / min-max work: no MFMA.
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include "pg_kernels.h"

namespace {
#define HB_TRY(expr)                                                                                                  \
    do {                                                                                                              \
        hipError_t e_ = (expr);                                                                                       \
        if (e_ != hipSuccess) return pgSetError(PG_ERR_DEVICE, (std::string(#expr) + ": " + hipGetErrorString(e_)).c_str()); \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    hipError_t alloc(size_t n) { return n ? hipMalloc(&p, n) : hipSuccess; }
    ~DevBuf() { if (p) (void)hipFree(p); }
};
constexpr float kMaxF = 3.40282346638528859811704183484516925e+38f;
constexpr int kBoundsBlocks = 1024;

// Treelet build node: pre-order inside its treelet, so the first child of node j is j + 1
struct TNode { float b[6]; int child1, nPrims, firstPrim, axis; };

__device__ __forceinline__ float fminp(float a, float b) { return (b < a) ? b : a; }  // std::min
__device__ __forceinline__ float fmaxp(float a, float b) { return (a < b) ? b : a; }  // std::max

// BVHPrimitiveInfo::centroid = .5f * pMin + .5f * pMax (bvh.cpp:55), unioned over all primitives
__global__ __launch_bounds__(256) void k_centroid_bounds(int n, const float *__restrict__ bounds, float *__restrict__ partial) {
    float lo[3] = {kMaxF, kMaxF, kMaxF}, hi[3] = {-kMaxF, -kMaxF, -kMaxF};
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        for (int k = 0; k < 3; ++k) {
            const float c = .5f * bounds[6 * (size_t)i + k] + .5f * bounds[6 * (size_t)i + 3 + k];
            lo[k] = fminp(lo[k], c); hi[k] = fmaxp(hi[k], c);
        }
    __shared__ float s[6][256];
    for (int k = 0; k < 3; ++k) { s[k][threadIdx.x] = lo[k]; s[3 + k][threadIdx.x] = hi[k]; }
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w)
            for (int k = 0; k < 3; ++k) {
                s[k][threadIdx.x] = fminp(s[k][threadIdx.x], s[k][threadIdx.x + w]);
                s[3 + k][threadIdx.x] = fmaxp(s[3 + k][threadIdx.x], s[3 + k][threadIdx.x + w]);
            }
        __syncthreads();
    }
    if (threadIdx.x < 6) partial[6 * blockIdx.x + threadIdx.x] = s[threadIdx.x][0];
}
__device__ __forceinline__ uint32_t left_shift3(uint32_t x) {  // bvh.cpp:107-131
    if (x == (1u << 10)) --x;
    x = (x | (x << 16)) & 0x30000ffu;
    x = (x | (x << 8)) & 0x300f00fu;
    x = (x | (x << 4)) & 0x30c30c3u;
    x = (x | (x << 2)) & 0x9249249u;
    return x;
}
struct CB { float lo[3], hi[3]; };
__global__ __launch_bounds__(256) void k_morton(int n, const float *__restrict__ bounds, CB cb, uint32_t *__restrict__ codes, int *__restrict__ idx) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t q[3];
    for (int k = 0; k < 3; ++k) {
        const float c = .5f * bounds[6 * (size_t)i + k] + .5f * bounds[6 * (size_t)i + 3 + k];
        float o = c - cb.lo[k];  // Bounds3::Offset, geometry.h:745-751
        if (cb.hi[k] > cb.lo[k]) o /= cb.hi[k] - cb.lo[k];
        q[k] = (uint32_t)(o * (float)(1 << 10));  // centroidOffset * mortonScale, EncodeMorton3 (bvh.cpp:133-138)
    }
    codes[i] = (left_shift3(q[2]) << 2) | (left_shift3(q[1]) << 1) | left_shift3(q[0]);
    idx[i] = i;
}
// ---- RadixSort(&mortonPrims), bvh.cpp:140-180: a stable sort of the (Morton code, primitive index) pairs by the 30 code bits.  The
// reference takes 5 passes of 6 bits; any stable sort gives the same order, here 3 passes of 8 bits and one of 6.  Per pass: every
// wave counts the digits of its own contiguous segment (k_radix_hist), one block turns the counts -- digit-major, segment-minor,
// which is the output order -- into offsets (k_radix_scan), and every wave walks its segment again in order, 64 keys at a time:
// a key's place is its segment's running offset for the digit plus the number of lower lanes holding the same digit, found with
// eight ballots (k_radix_scatter).  HBM-bound integer work, off the render path (scene set-up).
constexpr int RS_DIGITS = 256;
__global__ __launch_bounds__(64) void k_radix_hist(int n, int seg, int nUnits, const uint32_t *__restrict__ keys, int shift, int *__restrict__ counts) {
    __shared__ int h[RS_DIGITS];
    const int u = blockIdx.x, lane = threadIdx.x;
    for (int d = lane; d < RS_DIGITS; d += 64) h[d] = 0;
    __syncthreads();
    const int lo = u * seg, hi = min(n, lo + seg);
    for (int i = lo + lane; i < hi; i += 64) atomicAdd(&h[(keys[i] >> shift) & (RS_DIGITS - 1)], 1);
    __syncthreads();
    for (int d = lane; d < RS_DIGITS; d += 64) counts[d * nUnits + u] = h[d];
}
__global__ __launch_bounds__(1024) void k_radix_scan(int total, int *__restrict__ counts) {  // exclusive prefix sum, in place, one block
    __shared__ int s[1024];
    const int t = threadIdx.x, len = (total + 1023) / 1024, lo = min(total, t * len), hi = min(total, lo + len);
    int sum = 0;
    for (int i = lo; i < hi; ++i) sum += counts[i];
    s[t] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const int v = t >= off ? s[t - off] : 0;
        __syncthreads();
        s[t] += v;
        __syncthreads();
    }
    int run = s[t] - sum;
    for (int i = lo; i < hi; ++i) { const int c = counts[i]; counts[i] = run; run += c; }
}
__global__ __launch_bounds__(64) void k_radix_scatter(int n, int seg, int nUnits, const uint32_t *__restrict__ keysIn, const int *__restrict__ valsIn,
                                                      uint32_t *__restrict__ keysOut, int *__restrict__ valsOut, int shift, const int *__restrict__ offsets) {
    __shared__ int offs[RS_DIGITS];
    const int u = blockIdx.x, lane = threadIdx.x;
    for (int d = lane; d < RS_DIGITS; d += 64) offs[d] = offsets[d * nUnits + u];
    __syncthreads();
    const int lo = u * seg, hi = min(n, lo + seg);
    for (int base = lo; base < hi; base += 64) {
        const int i = base + lane;
        const bool in = i < hi;
        const uint32_t key = in ? keysIn[i] : 0u;
        const int val = in ? valsIn[i] : 0;
        const int d = (int)((key >> shift) & (RS_DIGITS - 1));
        unsigned long long same = __ballot(in);  // the lanes of this step that hold my digit
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1;
            const unsigned long long m = __ballot(bit);
            same &= bit ? m : ~m;
        }
        const int rank = __popcll(same & ((1ull << lane) - 1ull));
        const int pos = in ? offs[d] + rank : 0;
        __syncthreads();  // every lane has read its offset
        if (in && rank == 0) offs[d] += __popcll(same);
        __syncthreads();
        if (in) { keysOut[pos] = key; valsOut[pos] = val; }
    }
}
__global__ __launch_bounds__(256) void k_treelet_starts(int n, const uint32_t *__restrict__ codes, int *__restrict__ starts, int *__restrict__ count, int cap) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t mask = 0x3ffc0000u;  // bvh.cpp:442
    if (i == 0 || ((codes[i - 1] & mask) != (codes[i] & mask))) {
        const int k = atomicAdd(count, 1);
        if (k < cap) starts[k] = i;
    }
}
// emitLBVH (bvh.cpp:485-535) of one treelet, recursion unrolled onto a stack; nodes are numbered in the order the recursion
// creates them (pre-order).  Leaves take their primitives in Morton order, so -- treelets being laid out in index order --
// a leaf's firstPrimOffset is its run's position in the sorted array.
__global__ __launch_bounds__(64) void k_emit_lbvh(int nTreelets, const int *__restrict__ starts, const int *__restrict__ counts,
                                                  const uint32_t *__restrict__ codes, const int *__restrict__ idx, const float *__restrict__ bounds,
                                                  int maxPrimsInNode, TNode *__restrict__ tnodes, int *__restrict__ tcount) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= nTreelets) return;
    const int base = starts[t];
    const uint32_t *mc = codes + base;
    TNode *tn = tnodes + 2 * (size_t)base;
    int stS[40], stC[40], stB[40], stP[40];  // start, count, bitIndex, parent awaiting its second child (-1: none)
    int sp = 0, cnt = 0;
    stS[0] = 0; stC[0] = counts[t]; stB[0] = 29 - 12; stP[0] = -1; sp = 1;
    while (sp > 0) {
        --sp;
        const int s = stS[sp], c = stC[sp], parent = stP[sp];
        int bit = stB[sp];
        bool leaf = c < maxPrimsInNode;  // strictly fewer (bvh.cpp:491)
        if (!leaf) {
            while (bit >= 0 && ((mc[s] >> bit) & 1u) == ((mc[s + c - 1] >> bit) & 1u)) --bit;  // bvh.cpp:506-510
            leaf = bit == -1;
        }
        const int my = cnt++;
        if (parent >= 0) tn[parent].child1 = my;
        TNode nd;
        if (leaf) {
            float lo[3] = {kMaxF, kMaxF, kMaxF}, hi[3] = {-kMaxF, -kMaxF, -kMaxF};
            for (int i = 0; i < c; ++i) {
                const float *b = bounds + 6 * (size_t)idx[base + s + i];
                for (int k = 0; k < 3; ++k) { lo[k] = fminp(lo[k], b[k]); hi[k] = fmaxp(hi[k], b[3 + k]); }
            }
            for (int k = 0; k < 3; ++k) { nd.b[k] = lo[k]; nd.b[3 + k] = hi[k]; }
            nd.child1 = -1; nd.nPrims = c; nd.firstPrim = base + s; nd.axis = 0;
            tn[my] = nd;
        } else {
            int searchStart = s, searchEnd = s + c - 1;  // bvh.cpp:513-523
            while (searchStart + 1 != searchEnd) {
                const int mid = (searchStart + searchEnd) / 2;
                if (((mc[searchStart] >> bit) & 1u) == ((mc[mid] >> bit) & 1u)) searchStart = mid;
                else searchEnd = mid;
            }
            const int split = searchEnd - s;
            for (int k = 0; k < 6; ++k) nd.b[k] = 0;
            nd.child1 = -1; nd.nPrims = 0; nd.firstPrim = 0; nd.axis = bit % 3;
            tn[my] = nd;
            // second child below the first on the stack, so the first is numbered next
            stS[sp] = s + split; stC[sp] = c - split; stB[sp] = bit - 1; stP[sp] = my; ++sp;
            stS[sp] = s; stC[sp] = split; stB[sp] = bit - 1; stP[sp] = -1; ++sp;
        }
    }
    // interior bounds = Union(children) (bvh.cpp:74-82): children have larger pre-order numbers than their parent
    for (int j = cnt - 1; j >= 0; --j) {
        if (tn[j].nPrims > 0) continue;
        const TNode &a = tn[j + 1], &b = tn[tn[j].child1];
        for (int k = 0; k < 3; ++k) { tn[j].b[k] = fminp(a.b[k], b.b[k]); tn[j].b[3 + k] = fmaxp(a.b[3 + k], b.b[3 + k]); }
    }
    tcount[t] = cnt;
}
__global__ __launch_bounds__(256) void k_gather_roots(int nTreelets, const int *__restrict__ starts, const TNode *__restrict__ tnodes, float *__restrict__ roots) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= nTreelets) return;
    for (int k = 0; k < 6; ++k) roots[6 * t + k] = tnodes[2 * (size_t)starts[t]].b[k];
}
// flattenBVHTree (bvh.cpp:640-658) of the treelets: one block per treelet
__global__ __launch_bounds__(256) void k_flatten(const int *__restrict__ starts, const int *__restrict__ tcount, const int *__restrict__ tbase,
                                                 const TNode *__restrict__ tnodes, PgBVHNode *__restrict__ out) {
    const int t = blockIdx.x;
    const TNode *tn = tnodes + 2 * (size_t)starts[t];
    const int base = tbase[t];
    for (int j = threadIdx.x; j < tcount[t]; j += 256) {
        const TNode nd = tn[j];
        PgBVHNode o;
        for (int k = 0; k < 3; ++k) { o.bmin[k] = nd.b[k]; o.bmax[k] = nd.b[3 + k]; }
        o.offset = nd.nPrims > 0 ? nd.firstPrim : base + nd.child1;
        o.nprims = (uint16_t)nd.nPrims; o.axis = (uint8_t)nd.axis; o.pad = 0;
        out[base + j] = o;
    }
}
__global__ __launch_bounds__(256) void k_scatter_nodes(int n, const int *__restrict__ where, const PgBVHNode *__restrict__ src, PgBVHNode *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[where[i]] = src[i];
}

// ---- buildUpperSAH (bvh.cpp:537-638) on the host over the treelet roots: at most 4096 of them --------------------------
struct B3 {
    float lo[3] = {kMaxF, kMaxF, kMaxF}, hi[3] = {-kMaxF, -kMaxF, -kMaxF};  // Bounds3f(), geometry.h:663-667
};
B3 unionB(const B3 &a, const B3 &b) { B3 r; for (int k = 0; k < 3; ++k) { r.lo[k] = std::min(a.lo[k], b.lo[k]); r.hi[k] = std::max(a.hi[k], b.hi[k]); } return r; }
B3 unionP(const B3 &a, const float p[3]) { B3 r; for (int k = 0; k < 3; ++k) { r.lo[k] = std::min(a.lo[k], p[k]); r.hi[k] = std::max(a.hi[k], p[k]); } return r; }
float surfaceArea(const B3 &b) {  // geometry.h:722-725
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return 2 * (dx * dy + dx * dz + dy * dz);
}
int maximumExtent(const B3 &b) {  // geometry.h:730-738
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    if (dx > dy && dx > dz) return 0;
    else if (dy > dz) return 1;
    else return 2;
}
struct UNode { B3 bounds; int child[2] = {-1, -1}; int axis = 0; int treelet = -1; };  // treelet >= 0: a treelet root
struct Upper {
    std::vector<UNode> nodes;
    std::vector<std::string> errors;
    int build(std::vector<int> &roots, int start, int end) {  // roots: indices into nodes (the treelet roots, permuted in place)
        const int nNodes = end - start;
        if (nNodes == 1) return roots[start];
        B3 bounds;
        for (int i = start; i < end; ++i) bounds = unionB(bounds, nodes[roots[i]].bounds);
        B3 centroidBounds;
        for (int i = start; i < end; ++i) {
            const B3 &b = nodes[roots[i]].bounds;
            const float c[3] = {(b.lo[0] + b.hi[0]) * 0.5f, (b.lo[1] + b.hi[1]) * 0.5f, (b.lo[2] + b.hi[2]) * 0.5f};
            centroidBounds = unionP(centroidBounds, c);
        }
        const int dim = maximumExtent(centroidBounds);
        constexpr int nBuckets = 12;
        struct Bucket { int count = 0; B3 bounds; } buckets[nBuckets];
        auto bucketOf = [&](int r) {
            const B3 &nb = nodes[r].bounds;
            const float centroid = (nb.lo[dim] + nb.hi[dim]) * 0.5f;
            int b = nBuckets * ((centroid - centroidBounds.lo[dim]) / (centroidBounds.hi[dim] - centroidBounds.lo[dim]));
            if (b == nBuckets) b = nBuckets - 1;
            return b;
        };
        for (int i = start; i < end; ++i) {
            const int b = bucketOf(roots[i]);
            if (b < 0 || b >= nBuckets) return -1;  // the reference CHECK-fails here
            buckets[b].count++;
            buckets[b].bounds = unionB(buckets[b].bounds, nodes[roots[i]].bounds);
        }
        float cost[nBuckets - 1];
        for (int i = 0; i < nBuckets - 1; ++i) {
            B3 b0, b1;
            int count0 = 0, count1 = 0;
            for (int j = 0; j <= i; ++j) { b0 = unionB(b0, buckets[j].bounds); count0 += buckets[j].count; }
            for (int j = i + 1; j < nBuckets; ++j) { b1 = unionB(b1, buckets[j].bounds); count1 += buckets[j].count; }
            cost[i] = .125f + (count0 * surfaceArea(b0) + count1 * surfaceArea(b1)) / surfaceArea(bounds);
        }
        float minCost = cost[0];
        int minCostSplitBucket = 0;
        for (int i = 1; i < nBuckets - 1; ++i)
            if (cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; }
        int *pmid = std::partition(&roots[start], &roots[end - 1] + 1, [&](int r) { return bucketOf(r) <= minCostSplitBucket; });
        int mid = (int)(pmid - &roots[0]);
        if (mid <= start || mid >= end) mid = (start + end) / 2;  // the reference CHECK-fails; the host builder's fallback
        const int me = (int)nodes.size();
        nodes.emplace_back();
        const int c0 = build(roots, start, mid);
        if (c0 < 0) return -1;
        const int c1 = build(roots, mid, end);
        if (c1 < 0) return -1;
        nodes[me].child[0] = c0; nodes[me].child[1] = c1; nodes[me].axis = dim;
        nodes[me].bounds = unionB(nodes[c0].bounds, nodes[c1].bounds);
        return me;
    }
};
}  // namespace

extern "C" int pg_hlbvh_build(int32_t n, const float *bounds, int32_t maxPrimsInNode, PgBVHNode *nodesOut, int32_t *nNodesOut, int32_t *orderedOut) {
    if (n < 0 || !nNodesOut || (n > 0 && (!bounds || !nodesOut || !orderedOut))) return pgSetError(PG_ERR_INVALID, "pg_hlbvh_build: null argument");
    *nNodesOut = 0;
    if (n == 0) return PG_OK;
    if (maxPrimsInNode > 255) maxPrimsInNode = 255;  // BVHAccel ctor, bvh.cpp:185
    hipStream_t stream = nullptr;
    DevBuf dBounds, dPartial, dCodes[2], dIdx[2], dTemp, dStarts, dCount, dCounts, dTNodes, dTCount, dRoots, dBase, dOut, dUpWhere, dUpNodes;
    HB_TRY(dBounds.alloc(sizeof(float) * 6 * (size_t)n));
    HB_TRY(hipMemcpyAsync(dBounds.p, bounds, sizeof(float) * 6 * (size_t)n, hipMemcpyHostToDevice, stream));
    // 1. centroid bounds
    HB_TRY(dPartial.alloc(sizeof(float) * 6 * kBoundsBlocks));
    const int nbBlocks = std::min(kBoundsBlocks, (n + 255) / 256);
    hipLaunchKernelGGL(k_centroid_bounds, dim3(nbBlocks), dim3(256), 0, stream, n, (const float *)dBounds.p, (float *)dPartial.p);
    std::vector<float> partial(6 * (size_t)nbBlocks);
    HB_TRY(hipMemcpyAsync(partial.data(), dPartial.p, sizeof(float) * partial.size(), hipMemcpyDeviceToHost, stream));
    HB_TRY(hipStreamSynchronize(stream));
    CB cb;
    for (int k = 0; k < 3; ++k) { cb.lo[k] = kMaxF; cb.hi[k] = -kMaxF; }
    for (int b = 0; b < nbBlocks; ++b)
        for (int k = 0; k < 3; ++k) { cb.lo[k] = std::min(cb.lo[k], partial[6 * b + k]); cb.hi[k] = std::max(cb.hi[k], partial[6 * b + 3 + k]); }
    // 2. Morton codes, 3. stable radix sort on the 30 code bits
    for (int i = 0; i < 2; ++i) { HB_TRY(dCodes[i].alloc(sizeof(uint32_t) * (size_t)n)); HB_TRY(dIdx[i].alloc(sizeof(int) * (size_t)n)); }
    const int nBlocks = (n + 255) / 256;
    hipLaunchKernelGGL(k_morton, dim3(nBlocks), dim3(256), 0, stream, n, (const float *)dBounds.p, cb, (uint32_t *)dCodes[0].p, (int *)dIdx[0].p);
    {
        const int nUnits = std::max(1, std::min(2048, (n + 1023) / 1024));
        const int seg = ((n + nUnits - 1) / nUnits + 63) / 64 * 64;
        HB_TRY(dTemp.alloc(sizeof(int) * (size_t)RS_DIGITS * nUnits));
        int src = 0;
        for (int shift = 0; shift < 30; shift += 8, src ^= 1) {  // 8 + 8 + 8 + 6 bits (the codes have 30)
            hipLaunchKernelGGL(k_radix_hist, dim3(nUnits), dim3(64), 0, stream, n, seg, nUnits, (const uint32_t *)dCodes[src].p, shift, (int *)dTemp.p);
            hipLaunchKernelGGL(k_radix_scan, dim3(1), dim3(1024), 0, stream, RS_DIGITS * nUnits, (int *)dTemp.p);
            hipLaunchKernelGGL(k_radix_scatter, dim3(nUnits), dim3(64), 0, stream, n, seg, nUnits, (const uint32_t *)dCodes[src].p, (const int *)dIdx[src].p,
                               (uint32_t *)dCodes[src ^ 1].p, (int *)dIdx[src ^ 1].p, shift, (const int *)dTemp.p);
        }
        // four passes: the sorted pairs are back in buffer 0
    }
    const uint32_t *codes = (const uint32_t *)dCodes[0].p;
    const int *idx = (const int *)dIdx[0].p;
    // 4. treelets
    const int cap = 4096;  // 12 bits
    HB_TRY(dStarts.alloc(sizeof(int) * cap));
    HB_TRY(dCount.alloc(sizeof(int)));
    HB_TRY(hipMemsetAsync(dCount.p, 0, sizeof(int), stream));
    hipLaunchKernelGGL(k_treelet_starts, dim3(nBlocks), dim3(256), 0, stream, n, codes, (int *)dStarts.p, (int *)dCount.p, cap);
    int nTreelets = 0;
    HB_TRY(hipMemcpyAsync(&nTreelets, dCount.p, sizeof(int), hipMemcpyDeviceToHost, stream));
    HB_TRY(hipStreamSynchronize(stream));
    if (nTreelets < 1 || nTreelets > cap) return pgSetError(PG_ERR_DEVICE, "pg_hlbvh_build: impossible treelet count");
    std::vector<int> starts((size_t)nTreelets), counts((size_t)nTreelets);
    HB_TRY(hipMemcpy(starts.data(), dStarts.p, sizeof(int) * (size_t)nTreelets, hipMemcpyDeviceToHost));
    std::sort(starts.begin(), starts.end());
    for (int t = 0; t < nTreelets; ++t) counts[t] = (t + 1 < nTreelets ? starts[t + 1] : n) - starts[t];
    HB_TRY(dCounts.alloc(sizeof(int) * (size_t)nTreelets));
    HB_TRY(hipMemcpyAsync(dStarts.p, starts.data(), sizeof(int) * (size_t)nTreelets, hipMemcpyHostToDevice, stream));
    HB_TRY(hipMemcpyAsync(dCounts.p, counts.data(), sizeof(int) * (size_t)nTreelets, hipMemcpyHostToDevice, stream));
    // 5. one LBVH per treelet
    HB_TRY(dTNodes.alloc(sizeof(TNode) * 2 * (size_t)n));
    HB_TRY(dTCount.alloc(sizeof(int) * (size_t)nTreelets));
    HB_TRY(dRoots.alloc(sizeof(float) * 6 * (size_t)nTreelets));
    hipLaunchKernelGGL(k_emit_lbvh, dim3((nTreelets + 63) / 64), dim3(64), 0, stream, nTreelets, (const int *)dStarts.p, (const int *)dCounts.p, codes, idx,
                       (const float *)dBounds.p, (int)maxPrimsInNode, (TNode *)dTNodes.p, (int *)dTCount.p);
    hipLaunchKernelGGL(k_gather_roots, dim3((nTreelets + 255) / 256), dim3(256), 0, stream, nTreelets, (const int *)dStarts.p, (const TNode *)dTNodes.p, (float *)dRoots.p);
    std::vector<int> tcount((size_t)nTreelets);
    std::vector<float> roots(6 * (size_t)nTreelets);
    HB_TRY(hipMemcpyAsync(tcount.data(), dTCount.p, sizeof(int) * (size_t)nTreelets, hipMemcpyDeviceToHost, stream));
    HB_TRY(hipMemcpyAsync(roots.data(), dRoots.p, sizeof(float) * roots.size(), hipMemcpyDeviceToHost, stream));
    HB_TRY(hipStreamSynchronize(stream));
    // 6. the SAH tree over the treelet roots and the final node layout (host; tiny)
    Upper up;
    up.nodes.resize((size_t)nTreelets);
    std::vector<int> rootIdx((size_t)nTreelets);
    for (int t = 0; t < nTreelets; ++t) {
        for (int k = 0; k < 3; ++k) { up.nodes[t].bounds.lo[k] = roots[6 * t + k]; up.nodes[t].bounds.hi[k] = roots[6 * t + 3 + k]; }
        up.nodes[t].treelet = t;
        rootIdx[t] = t;
    }
    const int root = up.build(rootIdx, 0, nTreelets);
    if (root < 0) return pgSetError(PG_ERR_INVALID, "pg_hlbvh_build: degenerate treelet bounds (the reference aborts on this input)");
    std::vector<int> tbase((size_t)nTreelets, 0), upWhere;
    std::vector<PgBVHNode> upNodes;
    int offset = 0;
    // flattenBVHTree over the upper tree; a treelet occupies tcount[t] consecutive entries
    struct Frame { int node; int slot; int stage; };
    std::vector<Frame> stack;
    std::vector<int> myOffset(up.nodes.size(), -1);
    stack.push_back({root, -1, 0});
    while (!stack.empty()) {
        Frame f = stack.back();
        stack.pop_back();
        const UNode &u = up.nodes[f.node];
        if (f.stage == 0) {
            if (u.treelet >= 0) { tbase[u.treelet] = offset; myOffset[f.node] = offset; offset += tcount[u.treelet]; continue; }
            myOffset[f.node] = offset++;
            PgBVHNode ln;
            memset(&ln, 0, sizeof(ln));
            for (int k = 0; k < 3; ++k) { ln.bmin[k] = u.bounds.lo[k]; ln.bmax[k] = u.bounds.hi[k]; }
            ln.axis = (uint8_t)u.axis; ln.nprims = 0;
            upWhere.push_back(myOffset[f.node]);
            upNodes.push_back(ln);
            stack.push_back({f.node, (int)upNodes.size() - 1, 1});  // after both children: fill in the second child's offset
            stack.push_back({u.child[1], -1, 0});
            stack.push_back({u.child[0], -1, 0});
        } else upNodes[f.slot].offset = myOffset[u.child[1]];
    }
    const int totalNodes = offset;
    if (totalNodes > 2 * n) return pgSetError(PG_ERR_DEVICE, "pg_hlbvh_build: node count exceeds 2n");
    // 7. write the linear nodes
    HB_TRY(dBase.alloc(sizeof(int) * (size_t)nTreelets));
    HB_TRY(dOut.alloc(sizeof(PgBVHNode) * (size_t)totalNodes));
    HB_TRY(hipMemcpyAsync(dBase.p, tbase.data(), sizeof(int) * (size_t)nTreelets, hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(k_flatten, dim3(nTreelets), dim3(256), 0, stream, (const int *)dStarts.p, (const int *)dTCount.p, (const int *)dBase.p, (const TNode *)dTNodes.p,
                       (PgBVHNode *)dOut.p);
    if (!upNodes.empty()) {
        HB_TRY(dUpWhere.alloc(sizeof(int) * upWhere.size()));
        HB_TRY(dUpNodes.alloc(sizeof(PgBVHNode) * upNodes.size()));
        HB_TRY(hipMemcpyAsync(dUpWhere.p, upWhere.data(), sizeof(int) * upWhere.size(), hipMemcpyHostToDevice, stream));
        HB_TRY(hipMemcpyAsync(dUpNodes.p, upNodes.data(), sizeof(PgBVHNode) * upNodes.size(), hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_scatter_nodes, dim3(((int)upNodes.size() + 255) / 256), dim3(256), 0, stream, (int)upNodes.size(), (const int *)dUpWhere.p,
                           (const PgBVHNode *)dUpNodes.p, (PgBVHNode *)dOut.p);
    }
    HB_TRY(hipMemcpyAsync(nodesOut, dOut.p, sizeof(PgBVHNode) * (size_t)totalNodes, hipMemcpyDeviceToHost, stream));
    HB_TRY(hipMemcpyAsync(orderedOut, idx, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, stream));
    HB_TRY(hipStreamSynchronize(stream));
    HB_TRY(hipGetLastError());
    *nNodesOut = totalNodes;
    return PG_OK;
}
