// pg_sortexp.hip -- EXPERIMENT build only (make GPUEXTRA=-DPG_EXPERIMENT_SORT, tools/experiments/r02s_exp4_*.sh): what would k_trace gain if the
// rays of a queue were handed to its waves in an order that keeps neighbours together (origin cell + direction octant)?
// The rays stay where they are; a permutation (radix sort of a 27-bit key per ray, hipCUB) tells k_trace which entry to
// take for position i of a region, and results go to the entry's own index, so nothing else of the pipeline changes.
// The sort here is NOT the product's: it only answers whether a fast binning pass would be worth building.
#ifdef PG_EXPERIMENT_SORT
#include "pg_kernels.h"
#include <hipcub/hipcub.hpp>

__device__ inline unsigned sx_spread3(unsigned v) {  // 8 bits -> every third bit
    v &= 0xffu;
    v = (v | (v << 8)) & 0x0000f00fu;
    v = (v | (v << 4)) & 0x000c30c3u;
    v = (v | (v << 2)) & 0x00249249u;
    return v;
}
// key of entry idx of a queue with PG_REGIONS x regionCap entries: region in bits 28-30 (so that the sorted array keeps each
// region's regionCap entries in the region's own range), then cell / octant; entries beyond the region's count sort last
__global__ void k_sort_keys(RayQueue q, float bx, float by, float bz, float sx, float sy, float sz, unsigned *keys, int *vals, int n, int mode,
                            int cellBits) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int r = idx / q.regionCap, j = idx - r * q.regionCap;
    unsigned key = 0x0fffffffu;
    if (j < q.counts[r * PG_COUNT_STRIDE]) {
        const float4 o = q.o[idx], d = q.d[idx];
        const unsigned oct = (d.x < 0 ? 1u : 0u) | (d.y < 0 ? 2u : 0u) | (d.z < 0 ? 4u : 0u);
        const float fx = fminf(fmaxf((o.x - bx) * sx, 0.f), 255.f), fy = fminf(fmaxf((o.y - by) * sy, 0.f), 255.f),
                    fz = fminf(fmaxf((o.z - bz) * sz, 0.f), 255.f);
        const unsigned shift = 8 - cellBits;
        const unsigned m = sx_spread3((unsigned)fx >> shift) | (sx_spread3((unsigned)fy >> shift) << 1) | (sx_spread3((unsigned)fz >> shift) << 2);
        key = mode == 1 ? ((oct << 24) | m) : ((m << 3) | oct);
        if (key >= 0x0fffffffu) key = 0x0ffffffeu;
    }
    // mode 3: no region bits -- the sorted order is cut into PG_REGIONS equal shares afterwards (sortexp_spatial)
    keys[idx] = mode == 3 ? (key == 0x0fffffffu ? 0xffffffffu : key) : (((unsigned)r << 28) | key);
    vals[idx] = idx;
}
// Spatial partition across the XCDs: the globally sorted rays (origin cell, octant) are dealt out in PG_REGIONS contiguous,
// equally long shares, region r = share r, so that an XCD's L2 sees the rays -- and mostly the BVH nodes -- of one part of space.
__global__ void k_spatial_counts(RayQueue q, int *balanced) {  // one thread
    int n = 0;
    for (int r = 0; r < PG_REGIONS; ++r) n += q.counts[r * PG_COUNT_STRIDE];
    const int share = (n + PG_REGIONS - 1) / PG_REGIONS;
    for (int r = 0; r < PG_REGIONS; ++r) { int c = n - r * share; c = c < 0 ? 0 : (c > share ? share : c); balanced[r * PG_COUNT_STRIDE] = c; }
    balanced[PG_REGIONS * PG_COUNT_STRIDE] = share;
}
__global__ void k_spatial_perm(const int *sortedVals, const int *balanced, int regionCap, int *perm, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int r = i / regionCap, j = i - r * regionCap;
    if (j < balanced[r * PG_COUNT_STRIDE]) perm[i] = sortedVals[r * balanced[PG_REGIONS * PG_COUNT_STRIDE] + j];
}

struct SortScratch { unsigned *k0 = nullptr, *k1 = nullptr; int *v0 = nullptr, *v1 = nullptr; void *temp = nullptr; size_t tempBytes = 0; int n = 0; int *balanced = nullptr; };
static SortScratch g_ss[2];

// permutation of queue q's entries into scratch slot `which`; returns the device pointer (nullptr on failure)
const int *sortexp_permutation(const DScene &sc, RayQueue q, int which, int mode, int cellBits, hipStream_t s) {
    SortScratch &ss = g_ss[which];
    const int n = q.regionCap * PG_REGIONS;
    if (n <= 0) return nullptr;
    if (ss.n < n) {
        hipFree(ss.k0); hipFree(ss.k1); hipFree(ss.v0); hipFree(ss.v1); hipFree(ss.temp);
        ss = SortScratch();
        if (hipMalloc(&ss.k0, 4 * (size_t)n) || hipMalloc(&ss.k1, 4 * (size_t)n) || hipMalloc(&ss.v0, 4 * (size_t)n) || hipMalloc(&ss.v1, 4 * (size_t)n)) return nullptr;
        size_t tb = 0;
        hipcub::DeviceRadixSort::SortPairs(nullptr, tb, ss.k0, ss.k1, ss.v0, ss.v1, n, 0, 31, s);
        if (hipMalloc(&ss.temp, tb)) return nullptr;
        ss.tempBytes = tb; ss.n = n;
    }
    const float ex = sc.rootBox[3] - sc.rootBox[0], ey = sc.rootBox[4] - sc.rootBox[1], ez = sc.rootBox[5] - sc.rootBox[2];
    hipLaunchKernelGGL(k_sort_keys, dim3((n + 255) / 256), dim3(256), 0, s, q, sc.rootBox[0], sc.rootBox[1], sc.rootBox[2], ex > 0 ? 256.f / ex : 0.f,
                       ey > 0 ? 256.f / ey : 0.f, ez > 0 ? 256.f / ez : 0.f, ss.k0, ss.v0, n, mode, cellBits);
    size_t tb = ss.tempBytes;
    hipcub::DeviceRadixSort::SortPairs(ss.temp, tb, ss.k0, ss.k1, ss.v0, ss.v1, n, 0, 31, s);
    return ss.v1;
}
// mode 3: permutation + balanced region counts (countsOut: what k_trace should read instead of q.counts)
const int *sortexp_spatial(const DScene &sc, RayQueue q, int which, int cellBits, hipStream_t s, int **countsOut) {
    const int *sorted = sortexp_permutation(sc, q, which, 3, cellBits, s);
    if (!sorted) return nullptr;
    SortScratch &ss = g_ss[which];
    const int n = q.regionCap * PG_REGIONS;
    if (!ss.balanced && hipMalloc(&ss.balanced, sizeof(int) * (PG_REGIONS * PG_COUNT_STRIDE + 1))) return nullptr;
    hipLaunchKernelGGL(k_spatial_counts, dim3(1), dim3(1), 0, s, q, ss.balanced);
    hipLaunchKernelGGL(k_spatial_perm, dim3((n + 255) / 256), dim3(256), 0, s, ss.v1, ss.balanced, q.regionCap, ss.v0, n);
    *countsOut = ss.balanced;
    return ss.v0;  // (v0 is free again after the sort)
}
#endif
