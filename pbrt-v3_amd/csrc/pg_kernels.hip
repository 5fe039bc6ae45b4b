// pg_kernels.hip -- hand-written HIP kernels (gfx950 / CDNA4) of the wavefront
// PathIntegrator.  One process per GPU; every kernel works on SoA ray queues
// (32 B/ray as two float4) instead of pbrt's per-ray recursion:
//
//   k_generate   HaltonSampler + PerspectiveCamera   (sampler.cpp:46-52, halton.cpp:96-127, perspective.cpp:95-144)
//   k_trace      BVHAccel::Intersect / IntersectP + Triangle::Intersect[P]  (pg_traverse.hip)
//   k_shade      PathIntegrator::Li loop body + EstimateDirect set-up      (path.cpp:81-187, integrator.cpp:85-215)
//   k_resolve    EstimateDirect's visibility / MIS terms once rays are back (integrator.cpp:143-212)
//   k_film       SamplerIntegrator::Render's scrub + FilmTile::AddSample    (integrator.cpp:294-320, film.h:121-161)
//   k_light_tables  SpatialLightDistribution::ComputeDistribution per voxel (lightdistrib.cpp:232-300)
//
// No MFMA: there is no dense contraction on this path; traversal is a
// latency/bandwidth-bound gather over the node and triangle arrays.
#include "pg_device.h"
#include "pg_sphere.h"
#include "pg_kernels.h"
#include "pg_texture.h"
#include "pg_motion.h"
#include "pg_bssrdf.h"
#include "pg_grid.h"

#define PG_BLOCK 256
// threads per block of the shading kernel: its blocks meet at two barriers around the queue append, so a block is only as fast
// as its slowest wave -- measured on the GPU
#ifndef PG_SHADE_BLOCK
#define PG_SHADE_BLOCK 128
#endif
PG_DEV int lane_id() { return __lane_id(); }

// Queue append, aggregated per block: every wave ballots its pushes, the block sums them through LDS and ONE lane per
// queue does the atomicAdd on the block's region counter (region = blockIdx.x % 8); lanes get consecutive entries in
// (wave, lane) order.  NQ queues are appended to in one exchange (two barriers).  Must be reached by all threads.
// Consumers and producers share one mapping: block b owns entries [(b>>3)*256, +256) of region b & 7.
// With BINNED, queue 0's entries are additionally grouped by `bin0` (0..7) inside the block's range: k_shade bins the next
// bounce's rays by direction octant, so the 64 consecutive rays a traversal wave picks up come from one tile AND mostly
// one octant (same near/far child order, same subtrees) instead of four to eight.
template <int NQ, bool BINNED, int BLOCK = PG_BLOCK>
PG_DEV void block_push(const RayQueue *q, const bool *pred, int *pos, int bin0 = 0) {
    constexpr int NW = BLOCK / 64;
    __shared__ int s_cnt[NQ][NW];
    __shared__ int s_base[NQ];
    __shared__ int s_bin[BINNED ? 8 : 1][NW];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    unsigned long long mask[NQ];
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        mask[k] = __ballot(pred[k]);
        if (lane == 0) s_cnt[k][wave] = __popcll(mask[k]);
    }
    unsigned long long binMask = 0;
    if (BINNED) {
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot(pred[0] && bin0 == b);
            if (lane == 0) s_bin[b][wave] = __popcll(m);
            if (bin0 == b) binMask = m;
        }
    }
    __syncthreads();
    // lanes 0 .. NQ-1 reserve the NQ queues' entries with ONE atomic instruction (one round trip to the counters instead of
    // NQ dependent ones); the queue's fields are picked by compare-and-select on named copies so that they stay in registers
    if (threadIdx.x < NQ) {
        const int k = threadIdx.x;
        int total = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) total += s_cnt[k][w];
        static_assert(NQ <= 3, "block_push: at most three queues");
        int *const c0 = q[0].counts, *const c1 = q[NQ > 1 ? 1 : 0].counts, *const c2 = q[NQ > 2 ? 2 : 0].counts;
        const int cap0 = q[0].regionCap, cap1 = q[NQ > 1 ? 1 : 0].regionCap, cap2 = q[NQ > 2 ? 2 : 0].regionCap;
        int *const cnt = k == 0 ? c0 : (k == 1 ? c1 : c2);
        const int cap = k == 0 ? cap0 : (k == 1 ? cap1 : cap2);
        const int r = blockIdx.x & (PG_REGIONS - 1);
        s_base[k] = total ? r * cap + atomicAdd(&cnt[r * PG_COUNT_STRIDE], total) : 0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        int off = s_base[k];
        if (BINNED && k == 0) {
            for (int b = 0; b < 8; ++b)
                for (int w = 0; w < NW; ++w)
                    if (b < bin0 || (b == bin0 && w < wave)) off += s_bin[b][w];
            pos[k] = pred[k] ? off + __popcll(binMask & ((1ull << lane) - 1ull)) : -1;
        } else {
            for (int w = 0; w < wave; ++w) off += s_cnt[k][w];
            pos[k] = pred[k] ? off + __popcll(mask[k] & ((1ull << lane) - 1ull)) : -1;
        }
    }
}
// The queue entry this thread consumes (or -1): block b walks region b & 7.
template <int BLOCK = PG_BLOCK>
PG_DEV int queue_item(const RayQueue &q) {
    const int r = blockIdx.x & (PG_REGIONS - 1);
    const int j = (blockIdx.x >> 3) * BLOCK + threadIdx.x;
    return j < q.counts[r * PG_COUNT_STRIDE] ? r * q.regionCap + j : -1;
}

// A shading thread's POSITION in its launch's order (the index queue_item gave it, before RenderParams::order maps it to an entry; a
// retry launch reads it back from the retry list): recomputed from the block and thread indices where it is needed, not kept in a register.
PG_DEV int shade_position(const RenderParams &rp, const RayQueue &q) {
    if (rp.retryCount > 0) return rp.retryList[blockIdx.x * PG_SHADE_BLOCK + threadIdx.x];
    return (blockIdx.x & (PG_REGIONS - 1)) * q.regionCap + (blockIdx.x >> 3) * PG_SHADE_BLOCK + threadIdx.x;
}

PG_DEV unsigned long long wave_sum(unsigned long long v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

// ReportValue(pathLength, bounces) of the Li calls that end in this wave (path.cpp:186, volpath.cpp:187): every lane of the block calls it
PG_DEV void stats_path_end(unsigned long long *shards, bool ended, int len) {
    const unsigned long long m = __ballot(ended);
    if (m == 0) return;
    const unsigned long long sum = wave_sum(ended ? (unsigned long long)len : 0ull);
    int mn = ended ? len : 0xffff, mx = ended ? len : -1;
    for (int off = 32; off > 0; off >>= 1) { const int a = __shfl_down(mn, off), b = __shfl_down(mx, off); mn = a < mn ? a : mn; mx = b > mx ? b : mx; }
    if (lane_id() == 0) {
        unsigned long long *sh = shards + (blockIdx.x & (PG_LIGHT_TEST_SHARDS - 1)) * PG_LIGHT_TEST_STRIDE;
        atomicAdd(sh + PG_STAT_LEN_SUM, sum); atomicAdd(sh + PG_STAT_LEN_COUNT, (unsigned long long)__popcll(m));
        atomicMax(sh + PG_STAT_LEN_MINC, (unsigned long long)(0xffff - mn)); atomicMax(sh + PG_STAT_LEN_MAXP, (unsigned long long)(mx + 1));
    }
}
// a counter of the same shards raised by the number of lanes for which `pred` holds
PG_DEV void stats_count(unsigned long long *shards, int word, bool pred) {  // (the lanes that are active call it: the first of them for which pred holds adds)
    const unsigned long long m = __ballot(pred);
    if (m != 0 && lane_id() == __ffsll(m) - 1) atomicAdd(shards + (blockIdx.x & (PG_LIGHT_TEST_SHARDS - 1)) * PG_LIGHT_TEST_STRIDE + word, (unsigned long long)__popcll(m));
}

struct Tri { V3 p0, p1, p2; uint32_t flags; int material, light; };
PG_DEV Tri load_tri(const DScene &sc, int prim) {
    float4 a = sc.tris[PG_TRI_STRIDE * prim], b = sc.tris[PG_TRI_STRIDE * prim + 1], c = sc.tris[PG_TRI_STRIDE * prim + 2];
    Tri t;
    t.p0 = mk(a.x, a.y, a.z); t.p1 = mk(b.x, b.y, b.z); t.p2 = mk(c.x, c.y, c.z);
    t.flags = __float_as_uint(a.w); t.material = __float_as_int(b.w); t.light = __float_as_int(c.w);
    return t;
}

// ===========================================================================
// Sampler
// ===========================================================================
PG_DEV uint64_t inverse_radical_inverse(uint32_t base, uint64_t inverse, int nDigits) {  // lowdiscrepancy.h:80-91
    uint64_t index = 0;
    for (int i = 0; i < nDigits; ++i) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}
PG_DEV int pmod(int a, int b) { int r = a - (a / b) * b; return (r < 0) ? r + b : r; }  // pbrt.h:314-317

// HaltonSampler::GetIndexForSample, halton.cpp:96-117
PG_DEV uint64_t halton_index(const PgRenderDesc &rd, int px, int py, uint64_t sampleNum) {
    uint64_t offsetForCurrentPixel = 0;
    if (rd.sample_stride > 1) {
        int pm0 = pmod(px, 128), pm1 = pmod(py, 128);
        uint64_t dimOffset0 = inverse_radical_inverse(2, pm0, rd.base_exponents[0]);
        uint64_t dimOffset1 = inverse_radical_inverse(3, pm1, rd.base_exponents[1]);
        offsetForCurrentPixel += dimOffset0 * (uint64_t)(rd.sample_stride / rd.base_scales[0]) * (uint64_t)rd.mult_inverse[0];
        offsetForCurrentPixel += dimOffset1 * (uint64_t)(rd.sample_stride / rd.base_scales[1]) * (uint64_t)rd.mult_inverse[1];
        offsetForCurrentPixel %= (uint64_t)rd.sample_stride;
    }
    return offsetForCurrentPixel + sampleNum * (uint64_t)rd.sample_stride;
}
// HaltonSampler::SampleDimension, halton.cpp:119-127
// SobolSampler (samplers/sobol.cpp:41-59): SobolIntervalToIndex and SobolSampleFloat, lowdiscrepancy.h:229-273
PG_DEV uint64_t sobol_interval_to_index(const DScene &sc, uint32_t m, uint64_t frame, int px, int py) {
    if (m == 0) return 0;
    const uint32_t m2 = m << 1;
    uint64_t index = frame << m2;
    uint64_t delta = 0;
    for (int c = 0; frame; frame >>= 1, ++c)
        if (frame & 1) delta ^= sc.vdcSobol[(m - 1) * 52 + c];
    uint64_t b = (((uint64_t)((uint32_t)px) << m) | ((uint32_t)py)) ^ delta;
    for (int c = 0; b; b >>= 1, ++c)
        if (b & 1) index ^= sc.vdcSobolInv[(m - 1) * 52 + c];
    return index;
}
PG_DEV float sobol_sample(const DScene &sc, uint64_t a, int dim) {
    if (dim > 1023) dim = 1023;  // NumSobolDimensions; the reference aborts beyond (sobol.cpp:47-50)
    uint32_t v = 0;
    for (int i = dim * 52; a != 0; a >>= 1, i++)
        if (a & 1) v ^= sc.sobolMatrices[i];
    return pmin(v * 0x1p-32f, PG_ONE_MINUS_EPS);
}
// GlobalSampler::GetIndexForSample of the configured sampler (halton.cpp:96-117, sobol.cpp:41-44)
PG_DEV uint64_t sampler_index(const DScene &sc, const PgRenderDesc &rd, int px, int py, uint64_t sampleNum) {
    if (rd.sampler == 1) return sobol_interval_to_index(sc, (uint32_t)rd.sobol_log2_resolution, sampleNum, px - rd.sample_bounds[0], py - rd.sample_bounds[1]);
    return halton_index(rd, px, py, sampleNum);
}
// SampleDimension of the configured sampler for any dimension >= 2 (dimensions 0 and 1, the film position, are drawn by
// k_generate alone: SobolSampler remaps them with the current pixel, sobol.cpp:53-56)
PG_DEV float halton_sample(const DScene &sc, const PgRenderDesc &rd, uint64_t index, int dim) {
    if (rd.sampler == 1) return sobol_sample(sc, index, dim);
    if (rd.sample_at_pixel_center && (dim == 0 || dim == 1)) return 0.5f;
    if (dim == 0) return radical_inverse_base2(index >> rd.base_exponents[0]);
    if (dim == 1) return radical_inverse(3, index / (uint64_t)rd.base_scales[1]);
    if (dim >= sc.nPermDims) dim = sc.nPermDims - 1;  // host sizes the table from maxdepth; halton.h:71-76 aborts here
    if ((index >> 32) != 0) return scrambled_radical_inverse((uint32_t)sc.primes[dim], sc.perms + sc.permSums[dim], index);
    // 32-bit index: scrambled_radical_inverse's loop with the division done by multiplication (DScene::haltonDims)
    const int4 hd = sc.haltonDims[2 * dim], hc = sc.haltonDims[2 * dim + 1];
    const uint32_t base = (uint32_t)hd.x, mul = (uint32_t)hd.z, sh = (uint32_t)hd.w - 1u;
    const uint16_t *perm = sc.perms + hd.y;
    const float invBase = __int_as_float(hc.x);  // 1.f / (float)base
    uint64_t reversedDigits = 0;
    float invBaseN = 1;
    uint32_t a32 = (uint32_t)index;
    while (a32) {
        const uint32_t t = __umulhi(mul, a32);
        const uint32_t next = (t + ((a32 - t) >> 1)) >> sh;
        reversedDigits = reversedDigits * base + perm[a32 - next * base];
        invBaseN *= invBase;
        a32 = next;
    }
    return pmin(invBaseN * ((float)reversedDigits + __int_as_float(hc.y) /* invBase * perm[0] / (1 - invBase) */), PG_ONE_MINUS_EPS);
}

// The N Halton dimensions dim0 .. dim0+N-1 of one sample index at once (HaltonSampler::SampleDimension for each, halton.cpp:119-127
// + ScrambledRadicalInverse, lowdiscrepancy.cpp:405-424): the digit loops of the N dimensions run side by side, so the N
// permutation-table loads of a digit position are in flight together and the wave waits once per digit position instead of
// once per digit of every dimension (a shading wave spent a third of its time in these loops, profiles/r02s_*).  Per
// dimension the operations and their order are those of scrambled_radical_inverse().  dim0 is the same in every lane
// (the caller checks), so bases and table offsets are scalar; the index must fit 32 bits.
template <int N>
PG_DEV void halton_batch(const DScene &sc, uint32_t a0, int dim0, float *out) {
    uint32_t base[N], off[N], mul[N], sh[N], a[N];
    uint64_t rev[N];
    float invBase[N], tail[N], invBaseN[N];
#pragma unroll
    for (int j = 0; j < N; ++j) {
        int d = dim0 + j;
        if (d >= sc.nPermDims) d = sc.nPermDims - 1;
        const int4 hd = sc.haltonDims[2 * d], hc = sc.haltonDims[2 * d + 1];
        base[j] = (uint32_t)hd.x; off[j] = (uint32_t)hd.y; mul[j] = (uint32_t)hd.z; sh[j] = (uint32_t)hd.w - 1u;
        invBase[j] = __int_as_float(hc.x); tail[j] = __int_as_float(hc.y);
        a[j] = a0; rev[j] = 0; invBaseN[j] = 1;
    }
    uint32_t any = a0;
    while (any) {
        any = 0;
        uint32_t next[N], p[N];
#pragma unroll
        for (int j = 0; j < N; ++j) {  // the digit of every dimension, and its table entry on the way
            const uint32_t t = __umulhi(mul[j], a[j]);
            next[j] = (t + ((a[j] - t) >> 1)) >> sh[j];  // = a[j] / base[j] (DScene::haltonDims)
            p[j] = sc.perms[off[j] + (a[j] - next[j] * base[j])];
        }
        // all N loads are issued before the first is used (left alone, the compiler sinks each into the branch it makes of the
        // select below and waits for them one by one)
        static_assert(N == 7, "halton_batch: the barrier below names seven values");
        asm volatile("" : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5]), "+v"(p[6]));
#pragma unroll
        for (int j = 0; j < N; ++j) {
            // a dimension whose digits are used up stays as it is: it is multiplied by one and gets zero added (exact), which
            // keeps the loop body free of branches
            const bool act = a[j] != 0;
            rev[j] = rev[j] * (act ? base[j] : 1u) + (act ? p[j] : 0u);
            invBaseN[j] *= act ? invBase[j] : 1.f;
            a[j] = next[j];
            any |= next[j];
        }
    }
#pragma unroll
    for (int j = 0; j < N; ++j)
        out[j] = pmin(invBaseN[j] * ((float)rev[j] + tail[j]), PG_ONE_MINUS_EPS);
}

// ---- the samplers that draw from one RNG stream per tile (RandomSampler; the PixelSamplers, sampler.cpp:100-134) ------------
PG_DEV uint32_t rng_u32(TileSamplerState &t) {  // RNG::UniformUInt32, rng.h:137-143
    const unsigned long long oldstate = t.state;
    ++t.draws;
    t.state = oldstate * 0x5851f42d4c957f2dULL + t.inc;
    const uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    const uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}
// the state after `delta` more UniformUInt32 calls (each is one step of the 64-bit LCG): O(log delta), as PCG's advance()
PG_DEV void rng_advance(TileSamplerState &t, unsigned long long delta) {
    unsigned long long accMult = 1, accPlus = 0, curMult = 0x5851f42d4c957f2dULL, curPlus = t.inc;
    for (; delta > 0; delta >>= 1) {
        if (delta & 1) { accMult *= curMult; accPlus = accPlus * curMult + curPlus; }
        curPlus = (curMult + 1) * curPlus;
        curMult *= curMult;
    }
    t.state = accMult * t.state + accPlus;
}
PG_DEV uint32_t rng_u32b(TileSamplerState &t, uint32_t b) {  // rng.h:68-74
    const uint32_t threshold = (~b + 1u) % b;
    for (;;) { const uint32_t r = rng_u32(t); if (r >= threshold) return r % b; }
}
PG_DEV float rng_float(TileSamplerState &t) { return pmin(PG_ONE_MINUS_EPS, (float)rng_u32(t) * 0x1p-32f); }  // rng.h:75-82
PG_DEV float ts_get1d(const DScene &sc, int tile) {  // PixelSampler::Get1D (RandomSampler: tsDims = 0)
    TileSamplerState &t = sc.ts[tile];
    if (t.cur1D < sc.tsDims) return sc.ts1[((size_t)tile * sc.tsDims + t.cur1D++) * sc.tsSpp + t.sampleIndex];
    return rng_float(t);
}
PG_DEV void ts_get2d(const DScene &sc, int kind, int tile, float &a, float &b) {  // PixelSampler::Get2D
    TileSamplerState &t = sc.ts[tile];
    if (t.cur2D < sc.tsDims) {
        const float *p = sc.ts2 + (((size_t)tile * sc.tsDims + t.cur2D++) * sc.tsSpp + t.sampleIndex) * 2;
        a = p[0]; b = p[1];
    } else if (kind == PG_SAMPLER_RANDOM) { a = rng_float(t); b = rng_float(t); }  // random.cpp:50-54: braced list, left to right
    else { b = rng_float(t); a = rng_float(t); }  // `Point2f(rng.UniformFloat(), rng.UniformFloat())` as the reference build (g++) evaluates it: y first
}
// tsBatched (DScene): PixelSampler::Get1D / Get2D of sample `sample` of pixel `pixel`; dim = current2DDimension << 6 | current1DDimension
PG_DEV float tsb_get1d(const DScene &sc, int pixel, int sample, int &dim) {
    const int c = dim & 63;
    if (c >= sc.tsDims) { atomicOr(sc.tsOverflow, 1); return 0.5f; }
    dim += 1;
    return sc.ts1[((size_t)pixel * sc.tsDims + c) * sc.tsSpp + sample];
}
PG_DEV void tsb_get2d(const DScene &sc, int pixel, int sample, int &dim, float &a, float &b) {
    const int c = dim >> 6;
    if (c >= sc.tsDims) { atomicOr(sc.tsOverflow, 1); a = b = 0.5f; return; }
    dim += 64;
    const float *p = sc.ts2 + (((size_t)pixel * sc.tsDims + c) * sc.tsSpp + sample) * 2;
    a = p[0]; b = p[1];
}
PG_DEV void ts_shuffle(float *samp, int count, int width, TileSamplerState &t) {  // Shuffle, sampling.h:151-157
    for (int i = 0; i < count; ++i) {
        const int other = i + (int)rng_u32b(t, (uint32_t)(count - i));
        for (int j = 0; j < width; ++j) { const float v = samp[width * i + j]; samp[width * i + j] = samp[width * other + j]; samp[width * other + j] = v; }
    }
}
PG_DEV void ts_van_der_corput(int n, float *samples, TileSamplerState &t) {  // VanDerCorput(1, n, ...), lowdiscrepancy.h:154-207
    uint32_t v = rng_u32(t);
    for (uint32_t i = 0; i < (uint32_t)n; ++i) {  // GrayCodeSample, C[k] = 1 << (31 - k)
        samples[i] = pmin((float)v * 0x1p-32f, PG_ONE_MINUS_EPS);
        v ^= 0x80000000u >> __builtin_ctz(i + 1);
    }
    for (int i = 0; i < n; ++i) ts_shuffle(samples + i, 1, 1, t);  // a one-element shuffle still draws a number
    ts_shuffle(samples, n, 1, t);
}
PG_DEV void ts_sobol2d(int n, float *samples, TileSamplerState &t) {  // Sobol2D(1, n, ...), lowdiscrepancy.h:209-236
    uint32_t v0 = rng_u32(t), v1 = rng_u32(t);
    for (uint32_t i = 0; i < (uint32_t)n; ++i) {
        samples[2 * i] = pmin((float)v0 * 0x1p-32f, PG_ONE_MINUS_EPS);
        samples[2 * i + 1] = pmin((float)v1 * 0x1p-32f, PG_ONE_MINUS_EPS);
        const int k = __builtin_ctz(i + 1);
        v0 ^= 0x80000000u >> k;
        uint32_t c = 0x80000000u;  // CSobol[1][k]: c_0 = 2^31, c_j = c_(j-1) ^ (c_(j-1) >> 1)
        for (int j = 0; j < k; ++j) c ^= c >> 1;
        v1 ^= c;
    }
    for (int i = 0; i < n; ++i) ts_shuffle(samples + 2 * i, 1, 2, t);
    ts_shuffle(samples, n, 2, t);
}
// a tile's global index, pixel (lx, ly) of it and whether that pixel exists (tiles at the image edge are clipped)
PG_DEV bool ts_tile_pixel(const RenderParams &rp, int local, int lx, int ly, int &t, int &px, int &py) {
    t = rp.rd.tile_first + local * rp.rd.tile_step;
    const int tx = t % rp.nTilesX, ty = t / rp.nTilesX;
    px = rp.rd.sample_bounds[0] + tx * 16 + lx;
    py = rp.rd.sample_bounds[1] + ty * 16 + ly;
    return px < rp.rd.sample_bounds[2] && py < rp.rd.sample_bounds[3];
}
// tileSampler = sampler->Clone(seed), seed = tile.y * nTiles.x + tile.x (integrator.cpp:247-248); RNG::SetSequence, rng.h:129-135
__global__ void k_ts_init(DScene sc, RenderParams rp) {
    const int local = blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= rp.nTilesBatch) return;
    TileSamplerState t;
    t.state = 0u; t.inc = ((unsigned long long)(rp.rd.tile_first + local * rp.rd.tile_step) << 1u) | 1u;
    rng_u32(t);
    t.state += 0x853c49e6748fea9bULL;
    rng_u32(t);
    t.cur1D = t.cur2D = t.sampleIndex = t.active = 0; t.lens0 = t.lens1 = 0; t.px = t.py = 0; t.draws = 0; t.time = 0;
    sc.ts[local] = t;
}
// <Sampler>::StartPixel for pixel (lx, ly) of every tile: the pixel's sample arrays from the tile's stream -- also for pixels
// outside the integrator's pixel bounds, which are then skipped (integrator.cpp:264-273)
PG_DEV void ts_start_pixel(const DScene &sc, const PgRenderDesc &rd, TileSamplerState &t, float *s1, float *s2) {
    {
        const int n = sc.tsSpp, nd = sc.tsDims;
        if (rd.sampler == PG_SAMPLER_STRATIFIED) {  // stratified.cpp:43-70, sampling.cpp:42-60
            const int nx = rd.strat_samples[0], ny = rd.strat_samples[1];
            for (int i = 0; i < nd; ++i) {
                float *p = s1 + (size_t)i * n;
                const float invNSamples = 1.f / (float)n;
                for (int k = 0; k < n; ++k) { const float delta = rd.strat_jitter ? rng_float(t) : 0.5f; p[k] = pmin(((float)k + delta) * invNSamples, PG_ONE_MINUS_EPS); }
                ts_shuffle(p, n, 1, t);
            }
            for (int i = 0; i < nd; ++i) {
                float *p = s2 + (size_t)i * n * 2;
                const float dx = 1.f / (float)nx, dy = 1.f / (float)ny;
                for (int y = 0; y < ny; ++y)
                    for (int x = 0; x < nx; ++x) {
                        const float jx = rd.strat_jitter ? rng_float(t) : 0.5f;
                        const float jy = rd.strat_jitter ? rng_float(t) : 0.5f;
                        p[2 * (y * nx + x)] = pmin(((float)x + jx) * dx, PG_ONE_MINUS_EPS);
                        p[2 * (y * nx + x) + 1] = pmin(((float)y + jy) * dy, PG_ONE_MINUS_EPS);
                    }
                ts_shuffle(p, n, 2, t);
            }
        } else if (rd.sampler == PG_SAMPLER_ZEROTWO) {  // zerotwosequence.cpp:53-69
            for (int i = 0; i < nd; ++i) ts_van_der_corput(n, s1 + (size_t)i * n, t);
            for (int i = 0; i < nd; ++i) ts_sobol2d(n, s2 + (size_t)i * n * 2, t);
        } else if (rd.sampler == PG_SAMPLER_MAXMINDIST) {  // maxmin.cpp:43-69
            int cIndex = 0;
            while ((1 << (cIndex + 1)) <= n) ++cIndex;  // Log2Int(samplesPerPixel)
            const uint32_t *CPixel = sc.cmaxmin + 32 * cIndex;
            const float invSPP = 1.f / (float)n;
            for (int i = 0; i < n; ++i) {
                uint32_t v = 0, a = (uint32_t)i;
                for (int k = 0; a != 0; ++k, a >>= 1) if (a & 1) v ^= CPixel[k];  // MultiplyGenerator, lowdiscrepancy.h:93-98
                s2[2 * i] = (float)i * invSPP;
                s2[2 * i + 1] = pmin((float)v * 0x1p-32f, PG_ONE_MINUS_EPS);
            }
            ts_shuffle(s2, n, 2, t);
            for (int i = 0; i < nd; ++i) ts_van_der_corput(n, s1 + (size_t)i * n, t);
            for (int i = 1; i < nd; ++i) ts_sobol2d(n, s2 + (size_t)i * n * 2, t);
        }  // RandomSampler::StartPixel only fills requested sample arrays: none
    }
}
__global__ void k_ts_start_pixel(DScene sc, RenderParams rp, int lx, int ly) {
    const int local = blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= rp.nTilesBatch) return;
    TileSamplerState t = sc.ts[local];
    int tile, px, py;
    const bool exists = ts_tile_pixel(rp, local, lx, ly, tile, px, py);
    t.active = 0;
    if (exists) {
        const PgRenderDesc &rd = rp.rd;
        const int n = sc.tsSpp, nd = sc.tsDims;
        ts_start_pixel(sc, rd, t, sc.ts1 + (size_t)local * nd * n, sc.ts2 + (size_t)local * nd * n * 2);
        t.px = px; t.py = py;
        t.active = px >= rd.pixel_bounds[0] && px < rd.pixel_bounds[2] && py >= rd.pixel_bounds[1] && py < rd.pixel_bounds[3];
    }
    sc.ts[local] = t;
}
// tsBatched: <Sampler>::StartPixel for ALL pixels of every tile (integrator.cpp:264-273; the tile's stream is consumed by StartPixel
// alone, pixel after pixel).  One block per tile, one thread per pixel: a StartPixel consumes a fixed count of numbers unless one of
// RNG::UniformUInt32(b)'s rejection loops repeats (rng.h:68-74: probability < b / 2^32 per call), so thread p starts from the tile's
// state advanced by (pixels before it) x (that count), generates its pixel's arrays, and reports what it really consumed; a prefix
// sum over the pixels gives the true starting points, and the pixels whose starting point was wrong run again -- until none is
// (one pass nearly always; a pass per rejection otherwise; the result is the sequential one whatever the first guess was).
__global__ __launch_bounds__(256) void k_ts_start_tile(DScene sc, RenderParams rp) {
    const int local = blockIdx.x, p = threadIdx.x;
    __shared__ unsigned int s_cnt[256];
    __shared__ unsigned long long s_start[256];
    __shared__ unsigned long long s_total;
    __shared__ int s_dirty;
    int tile, px, py;
    const bool exists = ts_tile_pixel(rp, local, p & 15, p >> 4, tile, px, py);
    const TileSamplerState t0 = sc.ts[local];
    const PgRenderDesc &rd = rp.rd;
    const unsigned long long n = (unsigned long long)sc.tsSpp, nd = (unsigned long long)sc.tsDims;
    // numbers one StartPixel takes when no rejection loop repeats (a first guess only)
    unsigned long long nominal = 0;
    if (rd.sampler == PG_SAMPLER_STRATIFIED) nominal = nd * ((rd.strat_jitter ? n : 0) + n) + nd * ((rd.strat_jitter ? 2 * n : 0) + n);
    else if (rd.sampler == PG_SAMPLER_ZEROTWO) nominal = nd * (2 * n + 1) + nd * (2 * n + 2);
    else if (rd.sampler == PG_SAMPLER_MAXMINDIST) nominal = n + nd * (2 * n + 1) + (nd - 1) * (2 * n + 2);
    nominal += (unsigned long long)(long long)rp.tsGuessSkew;
    const int x0 = rd.sample_bounds[0] + (tile % rp.nTilesX) * 16;
    const int wE = rd.sample_bounds[2] - x0 < 16 ? rd.sample_bounds[2] - x0 : 16;  // the tile's existing pixels: wE columns
    unsigned long long start = exists ? (unsigned long long)((p >> 4) * wE + (p & 15)) * nominal : 0;
    const size_t pixel = (size_t)local * 256 + p, len = (size_t)n * nd;
    bool dirty = exists;
    unsigned int cnt = 0;
    for (int pass = 0; pass < 300; ++pass) {
        if (dirty) {
            TileSamplerState t = t0;
            rng_advance(t, start);
            t.draws = 0;
            ts_start_pixel(sc, rd, t, sc.ts1 + pixel * len, sc.ts2 + pixel * len * 2);
            cnt = t.draws;
        }
        s_cnt[p] = exists ? cnt : 0;
        if (p == 0) s_dirty = 0;
        __syncthreads();
        if (p == 0) {
            unsigned long long acc = 0;
            for (int k = 0; k < 256; ++k) { s_start[k] = acc; acc += s_cnt[k]; }
            s_total = acc;
        }
        __syncthreads();
        dirty = exists && s_start[p] != start;
        start = s_start[p];
        if (dirty) s_dirty = 1;
        __syncthreads();
        if (!s_dirty) break;
        __syncthreads();
    }
    // 300 passes without a fixed point would need ~300 rejection-loop repeats inside one tile (probability < 1e-1000): said, not assumed
    if (p == 0 && s_dirty && sc.tsOverflow) atomicOr(sc.tsOverflow, 2);
    if (p == 0) { TileSamplerState t = t0; rng_advance(t, s_total); sc.ts[local] = t; }
}
void launch_ts_start_tile(const DScene &sc, const RenderParams &rp, hipStream_t s) {
    hipLaunchKernelGGL(k_ts_start_tile, dim3(rp.nTilesBatch), dim3(256), 0, s, sc, rp);
}
void launch_ts_init(const DScene &sc, const RenderParams &rp, hipStream_t s) {
    hipLaunchKernelGGL(k_ts_init, dim3((rp.nTilesBatch + 63) / 64), dim3(64), 0, s, sc, rp);
}
void launch_ts_start_pixel(const DScene &sc, const RenderParams &rp, int lx, int ly, hipStream_t s) {
    hipLaunchKernelGGL(k_ts_start_pixel, dim3((rp.nTilesBatch + 63) / 64), dim3(64), 0, s, sc, rp, lx, ly);
}

// ===========================================================================
// Camera ray generation: one lane per (pixel, sample) slot
// ===========================================================================
PG_DEV V3 xform_point(const float *m, V3 p) {  // transform.h:219-231
    float x = p.x, y = p.y, z = p.z;
    float xp = m[0] * x + m[1] * y + m[2] * z + m[3];
    float yp = m[4] * x + m[5] * y + m[6] * z + m[7];
    float zp = m[8] * x + m[9] * y + m[10] * z + m[11];
    float wp = m[12] * x + m[13] * y + m[14] * z + m[15];
    if (wp == 1) return mk(xp, yp, zp);
    float inv = 1.f / wp;
    return mk(inv * xp, inv * yp, inv * zp);
}
PG_DEV void xform_ray(const float *m, V3 &o, V3 &d, float &tMax) {  // transform.h:249-262,277-300
    float x = o.x, y = o.y, z = o.z;
    float xp = (m[0] * x + m[1] * y) + (m[2] * z + m[3]);
    float yp = (m[4] * x + m[5] * y) + (m[6] * z + m[7]);
    float zp = (m[8] * x + m[9] * y) + (m[10] * z + m[11]);
    float wp = (m[12] * x + m[13] * y) + (m[14] * z + m[15]);
    float xAbsSum = (fabsf(m[0] * x) + fabsf(m[1] * y) + fabsf(m[2] * z) + fabsf(m[3]));
    float yAbsSum = (fabsf(m[4] * x) + fabsf(m[5] * y) + fabsf(m[6] * z) + fabsf(m[7]));
    float zAbsSum = (fabsf(m[8] * x) + fabsf(m[9] * y) + fabsf(m[10] * z) + fabsf(m[11]));
    V3 oError = mk(xAbsSum, yAbsSum, zAbsSum) * pgamma(3);
    V3 on;
    if (wp == 1) on = mk(xp, yp, zp);
    else { float inv = 1.f / wp; on = mk(inv * xp, inv * yp, inv * zp); }
    float dx = d.x, dy = d.y, dz = d.z;
    V3 dn = mk(m[0] * dx + m[1] * dy + m[2] * dz, m[4] * dx + m[5] * dy + m[6] * dz, m[8] * dx + m[9] * dy + m[10] * dz);
    float lengthSquared = lensq(dn);
    if (lengthSquared > 0) {
        float dt = dot(vabs(dn), oError) / lengthSquared;
        on = on + dn * dt;
        tMax -= dt;
    }
    o = on; d = dn;
}

// slot -> (local tile, sample-in-batch, pixel-in-tile); a wave is 64 pixels of one tile for one sample
PG_DEV bool slot_to_pixel(const RenderParams &rp, int slot, int &px, int &py, int &sn) {
    int pix = slot & 255;
    int rest = slot >> 8;
    int sIdx = rest % rp.sCount;
    int tileInBatch = rest / rp.sCount;
    int local = rp.tileLocal0 + tileInBatch;
    int t = rp.rd.tile_first + local * rp.rd.tile_step;
    int tx = t % rp.nTilesX, ty = t / rp.nTilesX;
    px = rp.rd.sample_bounds[0] + tx * 16 + (pix & 15);
    py = rp.rd.sample_bounds[1] + ty * 16 + (pix >> 4);
    sn = rp.s0 + sIdx;
    if (px >= rp.rd.sample_bounds[2] || py >= rp.rd.sample_bounds[3]) return false;
    // InsideExclusive(pixel, pixelBounds), integrator.cpp:273
    return px >= rp.rd.pixel_bounds[0] && px < rp.rd.pixel_bounds[2] && py >= rp.rd.pixel_bounds[1] && py < rp.rd.pixel_bounds[3];
}

// The matrix AnimatedTransform CameraToWorld carries a ray of time `time` to world space with (AnimatedTransform::operator()(Ray),
// transform.cpp:1171-1181): the start transform up to startTime, the end transform from endTime on, in between Interpolate (pg_motion.h)
// -- only .m is formed: a ray is transformed by m alone (transform.h:249-262), so Transform(scale)'s inverse is never looked at.
PG_DEV void camera_matrix_at(const PgRenderDesc &rd, float time, float *m) {
    if (!rd.camera_animated || time <= rd.camera_time[0]) { for (int k = 0; k < 16; ++k) m[k] = rd.camera_to_world[k]; return; }
    if (time >= rd.camera_time[1]) { for (int k = 0; k < 16; ++k) m[k] = rd.camera_to_world_end[k]; return; }
    const float dt = (time - rd.camera_time[0]) / (rd.camera_time[1] - rd.camera_time[0]);
    interpolate_trs<false>(rd.camera_T, rd.camera_R, rd.camera_S, dt, m, nullptr);
}
// CameraSample::time -> the ray's time, perspective.cpp:90 / :121: Lerp(sample.time, shutterOpen, shutterClose)
PG_DEV float camera_time(const PgRenderDesc &rd, float u) { return (1 - u) * rd.shutter_open + u * rd.shutter_close; }
// Camera::GenerateRay: PerspectiveCamera (perspective.cpp:95-115,141), OrthographicCamera (orthographic.cpp:68-93,115) and
// EnvironmentCamera (environment.cpp:43-56), ending in CameraToWorld(ray) -- c2w: the matrix of the ray's time (camera_matrix_at).
// (l0, l1) = CameraSample::pLens.
PG_DEV void camera_ray(const PgRenderDesc &rd, const float *c2w, float pFilmX, float pFilmY, float l0, float l1, V3 &o, V3 &d, float &tMax) {
    V3 pCamera = xform_point(rd.raster_to_camera, mk(pFilmX, pFilmY, 0));
    o = mk(0, 0, 0);
    d = normalize(mk(pCamera.x, pCamera.y, pCamera.z));
    tMax = PG_INF;
    if (rd.camera_type == 1) { o = pCamera; d = mk(0, 0, 1); }  // OrthographicCamera, orthographic.cpp:76-78
    if (rd.camera_type == 2) {  // EnvironmentCamera::GenerateRay, environment.cpp:43-56
        const float theta = PG_PI * pFilmY / rd.full_res[1];
        const float phi = 2 * PG_PI * pFilmX / rd.full_res[0];
        float sT, cT, sP, cP;
        pg_sincosf(theta, &sT, &cT);
        pg_sincosf(phi, &sP, &cP);
        o = mk(0, 0, 0);
        d = mk((float)sT * (float)cP, (float)cT, (float)sT * (float)sP);
    }
    if (rd.lens_radius > 0) {
        float lx, ly;
        concentric_sample_disk(l0, l1, lx, ly);
        lx = rd.lens_radius * lx; ly = rd.lens_radius * ly;
        float ft = rd.focal_distance / d.z;
        V3 pFocus = o + d * ft;
        o = mk(lx, ly, 0);
        d = normalize(pFocus - o);
    }
    xform_ray(c2w, o, d, tMax);
}
// The camera ray's differentials (GenerateRayDifferential: perspective.cpp:117-140, orthographic.cpp:95-113, the finite
// difference of camera.cpp:52-93 for the environment camera), carried to world space (transform.h:264-273) and scaled by
// 1 / sqrt(spp) (integrator.cpp:282-283, geometry.h:908-913).  (o, d) = the world-space camera ray.
PG_DEV void camera_differentials(const PgRenderDesc &rd, const float *c2w, float pFilmX, float pFilmY, float l0, float l1, V3 o, V3 d, V3 &rxO, V3 &rxD, V3 &ryO, V3 &ryD) {
    if (rd.camera_type == 2) {
        const float eps = .05f;
        V3 xo, xd, yo, yd;
        float tm;
        camera_ray(rd, c2w, pFilmX + eps, pFilmY, l0, l1, xo, xd, tm);
        camera_ray(rd, c2w, pFilmX, pFilmY + eps, l0, l1, yo, yd, tm);
        rxO = o + vdiv(xo - o, eps); rxD = d + vdiv(xd - d, eps);
        ryO = o + vdiv(yo - o, eps); ryD = d + vdiv(yd - d, eps);
    } else {
        const V3 dxCamera = mk(rd.dx_camera[0], rd.dx_camera[1], rd.dx_camera[2]), dyCamera = mk(rd.dy_camera[0], rd.dy_camera[1], rd.dy_camera[2]);
        const V3 pCamera = xform_point(rd.raster_to_camera, mk(pFilmX, pFilmY, 0));
        V3 co = mk(0, 0, 0), cd = normalize(mk(pCamera.x, pCamera.y, pCamera.z));  // the camera-space main ray again
        if (rd.camera_type == 1) { co = pCamera; cd = mk(0, 0, 1); }
        float lx = 0, ly = 0;
        if (rd.lens_radius > 0) {
            concentric_sample_disk(l0, l1, lx, ly);
            lx = rd.lens_radius * lx; ly = rd.lens_radius * ly;
            const float ft = rd.focal_distance / cd.z;
            const V3 pFocus = co + cd * ft;
            co = mk(lx, ly, 0);
            cd = normalize(pFocus - co);
        }
        if (rd.camera_type == 0) {
            if (rd.lens_radius > 0) {
                const V3 dx = normalize(pCamera + dxCamera);
                float ft = rd.focal_distance / dx.z;
                V3 pFocus = mk(0, 0, 0) + dx * ft;
                rxO = mk(lx, ly, 0);
                rxD = normalize(pFocus - rxO);
                const V3 dy = normalize(pCamera + dyCamera);
                ft = rd.focal_distance / dy.z;
                pFocus = mk(0, 0, 0) + dy * ft;
                ryO = mk(lx, ly, 0);
                ryD = normalize(pFocus - ryO);
            } else {
                rxO = ryO = co;
                rxD = normalize(pCamera + dxCamera);
                ryD = normalize(pCamera + dyCamera);
            }
        } else {
            if (rd.lens_radius > 0) {
                const float ft = rd.focal_distance / cd.z;
                V3 pFocus = (pCamera + dxCamera) + mk(0, 0, 1) * ft;
                rxO = mk(lx, ly, 0);
                rxD = normalize(pFocus - rxO);
                pFocus = (pCamera + dyCamera) + mk(0, 0, 1) * ft;
                ryO = mk(lx, ly, 0);
                ryD = normalize(pFocus - ryO);
            } else {
                rxO = co + dxCamera;
                ryO = co + dyCamera;
                rxD = ryD = cd;
            }
        }
        rxO = xform_point(c2w, rxO); ryO = xform_point(c2w, ryO);
        rxD = m4_vec(c2w, rxD); ryD = m4_vec(c2w, ryD);
    }
    const float sc = 1 / sqrtf((float)rd.spp);
    rxO = o + (rxO - o) * sc; ryO = o + (ryO - o) * sc;
    rxD = d + (rxD - d) * sc; ryD = d + (ryD - d) * sc;
}
// ANIM: the camera moves (PgRenderDesc::camera_animated) -- every sample's camera-to-world matrix is interpolated at its time; the still
// camera's kernel does not carry that code (64 instead of 39 registers, 113 spilled scalars)
template <bool ANIM>
__global__ __launch_bounds__(PG_BLOCK) void k_generate(DScene sc, RenderParams rp, PathState st, RayQueue q) {
    int slot = blockIdx.x * PG_BLOCK + threadIdx.x;
    bool valid = slot < rp.capacity;
    int px = 0, py = 0, sn = 0;
    if (valid) valid = slot_to_pixel(rp, slot, px, py, sn);
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    float tMax = PG_INF, rayTime = 0;
    if (valid) {
        const PgRenderDesc &rd = rp.rd;
        uint64_t index = sc.tsBatched ? 0 : sampler_index(sc, rd, px, py, (uint64_t)sn);
        // GetCameraSample, sampler.cpp:46-52: dims 0,1 film; 2 time; 3,4 lens
        float u0 = 0, u1 = 0, tsl0 = 0, tsl1 = 0, uTime = 0;
        int tsDim = 0;
        if (sc.tsBatched) {  // a PixelSampler's arrays: film 2D, time 1D, lens 2D; the path's state names its pixel and sample
            const int pixel = (rp.tileLocal0 + (slot >> 8) / rp.sCount) * 256 + (slot & 255);
            index = (uint64_t)(uint32_t)pixel | ((uint64_t)(uint32_t)sn << 32);
            tsb_get2d(sc, pixel, sn, tsDim, u0, u1);
            uTime = tsb_get1d(sc, pixel, sn, tsDim);
            tsb_get2d(sc, pixel, sn, tsDim, tsl0, tsl1);
        } else { u0 = halton_sample(sc, rd, index, 0); u1 = halton_sample(sc, rd, index, 1); if (ANIM) uTime = halton_sample(sc, rd, index, 2); }
        if (rd.sampler == 1) {  // SobolSampler::SampleDimension, sobol.cpp:53-56
            u0 = u0 * rd.sobol_resolution + rd.sample_bounds[0];
            u0 = u0 - px;
            u0 = u0 < 0.f ? 0.f : (u0 > PG_ONE_MINUS_EPS ? PG_ONE_MINUS_EPS : u0);
            u1 = u1 * rd.sobol_resolution + rd.sample_bounds[1];
            u1 = u1 - py;
            u1 = u1 < 0.f ? 0.f : (u1 > PG_ONE_MINUS_EPS ? PG_ONE_MINUS_EPS : u1);
        }
        float pFilmX = (float)px + u0, pFilmY = (float)py + u1;
        float l0 = 0, l1 = 0;
        if (sc.tsBatched) { l0 = tsl0; l1 = tsl1; }
        else if (rd.lens_radius > 0) { l0 = halton_sample(sc, rd, index, 3); l1 = halton_sample(sc, rd, index, 4); }
        if constexpr (ANIM) {
            float c2w[16];
            rayTime = camera_time(rd, uTime);
            camera_matrix_at(rd, rayTime, c2w);
            camera_ray(rd, c2w, pFilmX, pFilmY, l0, l1, o, d, tMax);
        } else camera_ray(rd, rd.camera_to_world, pFilmX, pFilmY, l0, l1, o, d, tMax);
        st.L[slot] = make_float4(0, 0, 0, pFilmX);
        st.beta[slot] = make_float4(1, 1, 1, pFilmY);
        st.meta[slot] = make_int4((int)(uint32_t)index, (int)(uint32_t)(index >> 32), __float_as_int(1.f), ((sc.tsBatched ? tsDim : 5) << 20) | PG_META_HASDIFF);
    } else if (slot < rp.capacity) {
        st.L[slot] = make_float4(0, 0, 0, 0);
        st.meta[slot] = make_int4(0, 0, 0, PG_META_DONE | 0x40000);  // 0x40000: slot holds no sample
    }
    int pos;
    block_push<1, false>(&q, &valid, &pos);
    if (valid) {
        q.o[pos] = make_float4(o.x, o.y, o.z, tMax);
        q.d[pos] = make_float4(d.x, d.y, d.z, __int_as_float(slot));
        if (float *qt = PG_QUEUE_TIMES(sc, q)) qt[pos] = rayTime;  // Ray::time (perspective.cpp:90 / :121), read by k_trace at moving instances
        if (st.qs[0].L) { st.qs[0].L[pos] = st.L[slot]; st.qs[0].beta[pos] = st.beta[slot]; st.qs[0].meta[pos] = st.meta[slot]; }  // just written by this thread
        if (st.qs[0].medium) st.qs[0].medium[pos] = rp.rd.camera_medium + 1;  // volpath: camera rays start in the camera's medium (camera.h:78)
    }
}
void launch_generate(const DScene &sc, const RenderParams &rp, PathState st, RayQueue q, hipStream_t s) {
    int nblk = (rp.capacity + PG_BLOCK - 1) / PG_BLOCK;
    if (rp.rd.camera_animated || sc.hasMotion) hipLaunchKernelGGL(k_generate<true>  /* moving instances need the samples' times too */, dim3(nblk), dim3(PG_BLOCK), 0, s, sc, rp, st, q);
    else hipLaunchKernelGGL(k_generate<false>, dim3(nblk), dim3(PG_BLOCK), 0, s, sc, rp, st, q);
}

// Tile-serial samplers: sample `sampleIndex` of every tile's current pixel -- StartNextSample's reset, GetCameraSample's draws
// (film 2D, time 1D, lens 2D: always drawn, they advance the stream), the camera ray.  Slot = the tile's local index.
__global__ __launch_bounds__(PG_BLOCK) void k_ts_generate(DScene sc, RenderParams rp, PathState st, RayQueue q, int sampleIndex) {
    const int local = blockIdx.x * PG_BLOCK + threadIdx.x;
    bool valid = local < rp.nTilesBatch && sc.ts[local].active;
    V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
    float tMax = PG_INF, rayTime = 0;
    if (valid) {
        const PgRenderDesc &rd = rp.rd;
        TileSamplerState &t = sc.ts[local];
        t.sampleIndex = sampleIndex; t.cur1D = t.cur2D = 0;
        float u0, u1, l0, l1;
        ts_get2d(sc, rd.sampler, local, u0, u1);
        const float uTime = ts_get1d(sc, local);  // time
        ts_get2d(sc, rd.sampler, local, l0, l1);
        t.lens0 = l0; t.lens1 = l1; t.time = uTime;
        const float pFilmX = (float)t.px + u0, pFilmY = (float)t.py + u1;
        float c2w[16];
        rayTime = camera_time(rd, uTime);
        camera_matrix_at(rd, rayTime, c2w);
        camera_ray(rd, c2w, pFilmX, pFilmY, l0, l1, o, d, tMax);
        st.L[local] = make_float4(0, 0, 0, pFilmX);
        st.beta[local] = make_float4(1, 1, 1, pFilmY);
        st.meta[local] = make_int4(0, 0, __float_as_int(1.f), PG_META_HASDIFF);
    }
    int pos;
    block_push<1, false>(&q, &valid, &pos);
    if (valid) {
        q.o[pos] = make_float4(o.x, o.y, o.z, tMax);
        q.d[pos] = make_float4(d.x, d.y, d.z, __int_as_float(local));
        if (float *qt = PG_QUEUE_TIMES(sc, q)) qt[pos] = rayTime;
        if (st.qs[0].L) { st.qs[0].L[pos] = st.L[local]; st.qs[0].beta[pos] = st.beta[local]; st.qs[0].meta[pos] = st.meta[local]; }
        if (st.qs[0].medium) st.qs[0].medium[pos] = rp.rd.camera_medium + 1;
    }
}
void launch_ts_generate(const DScene &sc, const RenderParams &rp, PathState st, RayQueue q, int sampleIndex, hipStream_t s) {
    hipLaunchKernelGGL(k_ts_generate, dim3((rp.nTilesBatch + PG_BLOCK - 1) / PG_BLOCK), dim3(PG_BLOCK), 0, s, sc, rp, st, q, sampleIndex);
}

// ===========================================================================
// Shading
// ===========================================================================
struct Isect { V3 p, pError, wo, n, ns, sdpdu, sdpdv, sdndu, sdndv; };  // n: geometric normal; ns, sd*: shading.n, shading.dpdu/dpdv/dndu/dndv (the last three feed bump mapping only)

PG_DEV V3 tri_normal(const Tri &t) {  // triangle.cpp:346-348
    V3 n = normalize(cross(t.p0 - t.p2, t.p1 - t.p2));
    if (t.flags & PG_TRI_FLIP_NORMAL) n = -n;
    return n;
}
PG_DEV void load_uv(const DScene &sc, int prim, uint32_t flags, float uv[6]) {  // triangle.h:98-108
    if (sc.attrUV && (flags & PG_TRI_HAS_UV)) tri_attr_uv(sc, prim, uv);
    else { uv[0] = 0; uv[1] = 0; uv[2] = 1; uv[3] = 0; uv[4] = 1; uv[5] = 1; }
}
// The tail of Triangle::Intersect (triangle.cpp:293-348) for the surviving hit.
// Per-vertex attribute a (N or S) of triangle prim interpolated with weights (w0, w1, w2): w0*a0 + w1*a1 + w2*a2.
PG_DEV V3 tri_interp(const float4 *attr, int prim, float w0, float w1, float w2) {
    const float4 a = attr[3 * (size_t)prim], b = attr[3 * (size_t)prim + 1], c = attr[3 * (size_t)prim + 2];
    return mk(a.x, a.y, a.z) * w0 + mk(b.x, b.y, b.z) * w1 + mk(c.x, c.y, c.z) * w2;
}
PG_DEV V3 tri_interp_normal(const DScene &sc, int prim, float w0, float w1, float w2) {  // the same with the per-vertex normals of DScene::triAttr
    float n[9];
    tri_attr_normals(sc, prim, n);
    return mk(n[0], n[1], n[2]) * w0 + mk(n[3], n[4], n[5]) * w1 + mk(n[6], n[7], n[8]) * w2;
}
// Triangle::Intersect's geometric normal for a hit with barycentrics (b0,b1,b2): Normalize(Cross(dp02, dp12)), flipped by
// reverseOrientation ^ transformSwapsHandedness (triangle.cpp:346-348), then made to face the interpolated shading normal
// when the mesh has per-vertex normals (SetShadingGeometry's Faceforward, interaction.cpp:79-81) -- full version below.
// The tail of Triangle::Intersect (triangle.cpp:293-419) for the surviving hit.
PG_DEV Isect make_isect(const DScene &sc, int prim, const Tri &t, float b0, float b1, float b2, V3 rayD) {
    Isect is;
    float uv[6];
    load_uv(sc, prim, t.flags, uv);
    V3 dpdu, dpdv;
    tri_dpdu_dpdv(t.p0, t.p1, t.p2, uv, dpdu, dpdv);
    float xAbsSum = (fabsf(b0 * t.p0.x) + fabsf(b1 * t.p1.x) + fabsf(b2 * t.p2.x));
    float yAbsSum = (fabsf(b0 * t.p0.y) + fabsf(b1 * t.p1.y) + fabsf(b2 * t.p2.y));
    float zAbsSum = (fabsf(b0 * t.p0.z) + fabsf(b1 * t.p1.z) + fabsf(b2 * t.p2.z));
    is.pError = mk(xAbsSum, yAbsSum, zAbsSum) * pgamma(7);
    is.p = t.p0 * b0 + t.p1 * b1 + t.p2 * b2;
    is.wo = normalize(-rayD);  // Interaction ctor, interaction.h:60
    is.n = tri_normal(t);
    is.ns = is.n;
    is.sdpdu = dpdu; is.sdpdv = dpdv;
    is.sdndu = is.sdndv = mk(0, 0, 0);
    const bool hasN = sc.attrN && (t.flags & PG_TRI_HAS_N), hasS = sc.triS && (t.flags & PG_TRI_HAS_S);
    if (hasN || hasS) {  // shading geometry, triangle.cpp:350-419 (dndu/dndv only feed ray differentials)
        V3 ns = is.n, ss = normalize(dpdu), ts;
        if (hasN) {
            V3 v = tri_interp_normal(sc, prim, b0, b1, b2);
            if (lensq(v) > 0) ns = normalize(v);
        }
        if (hasS) {
            V3 v = tri_interp(sc.triS, prim, b0, b1, b2);
            if (lensq(v) > 0) ss = normalize(v);
        }
        ts = cross(ss, ns);
        if (lensq(ts) > 0.f) { ts = normalize(ts); ss = cross(ts, ns); }
        else coordinate_system(ns, ss, ts);
        if (hasN) {  // dndu, dndv of the shading geometry, triangle.cpp:383-416
            const float d02x = uv[0] - uv[4], d02y = uv[1] - uv[5], d12x = uv[2] - uv[4], d12y = uv[3] - uv[5];
            float nv[9];
            tri_attr_normals(sc, prim, nv);
            const V3 n0 = mk(nv[0], nv[1], nv[2]), n1 = mk(nv[3], nv[4], nv[5]), n2 = mk(nv[6], nv[7], nv[8]);
            const V3 dn1 = n0 - n2, dn2 = n1 - n2;
            const float determinant = d02x * d12y - d02y * d12x;
            if ((double)fabsf(determinant) < 1e-8) {
                const V3 dn = cross(n2 - n0, n1 - n0);
                if (lensq(dn) != 0) coordinate_system(dn, is.sdndu, is.sdndv);
            } else {
                const float invDet = 1 / determinant;
                is.sdndu = (dn1 * d12y - dn2 * d02y) * invDet;
                is.sdndv = (dn1 * (-d12x) + dn2 * d02x) * invDet;
            }
        }
        if (t.flags & PG_TRI_REVERSE_ORIENTATION) ts = -ts;
        // SetShadingGeometry(ss, ts, ..., orientationIsAuthoritative = true), interaction.cpp:74-90
        is.ns = normalize(cross(ss, ts));
        if (dot(is.n, is.ns) < 0.f) is.n = -is.n;
        is.sdpdu = ss; is.sdpdv = ts;
    }
    return is;
}
// The geometric normal alone, as make_isect leaves it in isect.n (for Le() at a hit that is not shaded further).
PG_DEV V3 hit_normal(const DScene &sc, int prim, const Tri &t, float b0, float b1, float b2) {
    if ((sc.attrN && (t.flags & PG_TRI_HAS_N)) || (sc.triS && (t.flags & PG_TRI_HAS_S))) return make_isect(sc, prim, t, b0, b1, b2, mk(0, 0, 1)).n;
    return tri_normal(t);
}

// BSDF: LambertianReflection and/or MicrofacetReflection(TrowbridgeReitz, FresnelDielectric(1.5, 1)) lobes in the order
// the materials add them (matte.cpp:45-62, plastic.cpp:45-70); reflection.h:164-213, reflection.cpp:680-796.
struct Bsdf {
    V3 ns, ng, ss, ts; Spec R, Ks; float alpha; int nBxDFs; bool hasDiff, hasSpec;
    bool orenNayar; float onA, onB;  // the diffuse lobe is OrenNayar(R, sigma) (reflection.h:425-431) instead of LambertianReflection
    int specular;  // 0; 1 = SpecularReflection(Kr, FresnelNoOp) (mirror.cpp:44-56); 2 = FresnelSpecular(Kr, Kt, 1, eta) (glass.cpp:45-65)
    Spec Kr, Kt; float eta;
};
#define PG_BSDF_REFLECTION 1
#define PG_BSDF_TRANSMISSION 2
#define PG_BSDF_SPECULAR 16
PG_DEV V3 world_to_local(const Bsdf &b, V3 v) { return mk(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
PG_DEV V3 local_to_world(const Bsdf &b, V3 v) {
    return mk(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z,
              b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
// reflection.h:52-83
PG_DEV float cos2_theta(V3 w) { return w.z * w.z; }
PG_DEV float sin2_theta(V3 w) { return pmax(0.f, 1.f - cos2_theta(w)); }
PG_DEV float sin_theta(V3 w) { return sqrtf(sin2_theta(w)); }
PG_DEV float tan_theta(V3 w) { return sin_theta(w) / w.z; }
PG_DEV float tan2_theta(V3 w) { return sin2_theta(w) / cos2_theta(w); }
PG_DEV float cos_phi(V3 w) { float st = sin_theta(w); return (st == 0) ? 1 : clampf(w.x / st, -1, 1); }
PG_DEV float sin_phi(V3 w) { float st = sin_theta(w); return (st == 0) ? 0 : clampf(w.y / st, -1, 1); }
PG_DEV float cos2_phi(V3 w) { return cos_phi(w) * cos_phi(w); }
PG_DEV float sin2_phi(V3 w) { return sin_phi(w) * sin_phi(w); }
PG_DEV float fr_dielectric(float cosThetaI, float etaI, float etaT) {  // reflection.cpp:47-68
    cosThetaI = clampf(cosThetaI, -1, 1);
    if (!(cosThetaI > 0.f)) { float t = etaI; etaI = etaT; etaT = t; cosThetaI = fabsf(cosThetaI); }
    float sinThetaI = sqrtf(pmax(0.f, 1 - cosThetaI * cosThetaI));
    float sinThetaT = etaI / etaT * sinThetaI;
    if (sinThetaT >= 1) return 1;
    float cosThetaT = sqrtf(pmax(0.f, 1 - sinThetaT * sinThetaT));
    float Rparl = ((etaT * cosThetaI) - (etaI * cosThetaT)) / ((etaT * cosThetaI) + (etaI * cosThetaT));
    float Rperp = ((etaI * cosThetaI) - (etaT * cosThetaT)) / ((etaI * cosThetaI) + (etaT * cosThetaT));
    return (Rparl * Rparl + Rperp * Rperp) / 2;
}
// TrowbridgeReitzDistribution with alphax == alphay == a, microfacet.cpp:155-184, microfacet.h:57-63
PG_DEV float tr_D(float a, V3 wh) {
    float tan2Theta = tan2_theta(wh);
    if (isinf(tan2Theta)) return 0.f;
    const float cos4Theta = cos2_theta(wh) * cos2_theta(wh);
    float e = (cos2_phi(wh) / (a * a) + sin2_phi(wh) / (a * a)) * tan2Theta;
    return 1 / (PG_PI * a * a * cos4Theta * (1 + e) * (1 + e));
}
PG_DEV float tr_lambda(float a, V3 w) {
    float absTanTheta = fabsf(tan_theta(w));
    if (isinf(absTanTheta)) return 0.f;
    float alpha = sqrtf(cos2_phi(w) * a * a + sin2_phi(w) * a * a);
    float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + sqrtf(1.f + alpha2Tan2Theta)) / 2;
}
PG_DEV float tr_G1(float a, V3 w) { return 1 / (1 + tr_lambda(a, w)); }
PG_DEV float tr_G(float a, V3 wo, V3 wi) { return 1 / (1 + tr_lambda(a, wo) + tr_lambda(a, wi)); }
PG_DEV float tr_pdf(float a, V3 wo, V3 wh) { return tr_D(a, wh) * tr_G1(a, wo) * absdot(wo, wh) / fabsf(wo.z); }  // microfacet.cpp:339-345
// TrowbridgeReitzSample11 / TrowbridgeReitzSample / Sample_wh (visible area), microfacet.cpp:238-336.  The
// normal-incidence branch evaluates sqrt/cos/sin in double on float arguments, as the reference's source does.
PG_DEV void tr_sample11(float cosTheta, float U1, float U2, float &slope_x, float &slope_y) {
    if ((double)cosTheta > .9999) {
        float r = (float)sqrt((double)(U1 / (1 - U1)));
        float phi = (float)(6.28318530718 * (double)U2);
        double s, c;
        sincos((double)phi, &s, &c);
        slope_x = (float)((double)r * c);
        slope_y = (float)((double)r * s);
        return;
    }
    float sinTheta = sqrtf(pmax(0.f, 1.f - cosTheta * cosTheta));
    float tanTheta = sinTheta / cosTheta;
    float a = 1 / tanTheta;
    float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
    float A = 2 * U1 / G1 - 1;
    float tmp = 1.f / (A * A - 1.f);
    if (tmp > 1e10f) tmp = 1e10f;
    float B = tanTheta;
    float D = sqrtf(pmax(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.f));
    float slope_x_1 = B * tmp - D;
    float slope_x_2 = B * tmp + D;
    slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
    float S;
    if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
    else { S = -1.f; U2 = 2.f * (.5f - U2); }
    float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
              (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
    slope_y = S * z * sqrtf(1.f + slope_x * slope_x);
}
PG_DEV V3 tr_sample_wh(float a, V3 wo, float u0, float u1) {
    const bool flip = wo.z < 0;
    V3 wi = flip ? -wo : wo;
    V3 wiStretched = normalize(mk(a * wi.x, a * wi.y, wi.z));
    float slope_x, slope_y;
    tr_sample11(wiStretched.z, u0, u1, slope_x, slope_y);
    float tmp = cos_phi(wiStretched) * slope_x - sin_phi(wiStretched) * slope_y;
    slope_y = sin_phi(wiStretched) * slope_x + cos_phi(wiStretched) * slope_y;
    slope_x = tmp;
    slope_x = a * slope_x;
    slope_y = a * slope_y;
    V3 wh = normalize(mk(-slope_x, -slope_y, 1.f));
    return flip ? -wh : wh;
}
// MicrofacetReflection::f / Pdf / Sample_f, reflection.cpp:226-238, :410-429 (local coordinates)
PG_DEV Spec mf_f(const Bsdf &b, V3 wo, V3 wi) {
    float cosThetaO = fabsf(wo.z), cosThetaI = fabsf(wi.z);
    V3 wh = wi + wo;
    if (cosThetaI == 0 || cosThetaO == 0) return sp(0);
    if (wh.x == 0 && wh.y == 0 && wh.z == 0) return sp(0);
    wh = normalize(wh);
    V3 whf = (dot(wh, mk(0, 0, 1)) < 0.f) ? -wh : wh;  // Faceforward
    Spec F = sp(fr_dielectric(dot(wi, whf), 1.5f, 1.f));  // FresnelDielectric(1.5f, 1.f)
    return (((b.Ks * tr_D(b.alpha, wh)) * tr_G(b.alpha, wo, wi)) * F) / (4 * cosThetaI * cosThetaO);
}
PG_DEV float mf_pdf(const Bsdf &b, V3 wo, V3 wi) {
    if (!(wo.z * wi.z > 0)) return 0;
    V3 wh = normalize(wo + wi);
    return tr_pdf(b.alpha, wo, wh) / (4 * dot(wo, wh));
}
PG_DEV void mf_sample(const Bsdf &b, V3 wo, V3 &wi, float u0, float u1, float &pdf) {  // leaves pdf = 0 when no sample
    if (wo.z == 0) return;
    V3 wh = tr_sample_wh(b.alpha, wo, u0, u1);
    if (dot(wo, wh) < 0) return;
    wi = -wo + wh * (2 * dot(wo, wh));  // Reflect, reflection.h:93-95
    if (!(wo.z * wi.z > 0)) return;
    pdf = tr_pdf(b.alpha, wo, wh) / (4 * dot(wo, wh));
}
PG_DEV float lambert_pdf(V3 wo, V3 wi) { return (wo.z * wi.z > 0) ? fabsf(wi.z) * PG_INVPI : 0; }  // reflection.cpp:392-394
// the diffuse lobe: LambertianReflection::f (reflection.cpp:178-180) or OrenNayar::f (:197-219)
PG_DEV Spec diffuse_f(const Bsdf &b, V3 wo, V3 wi) {
    if (!b.orenNayar) return b.R * PG_INVPI;
    float sinThetaI = sin_theta(wi), sinThetaO = sin_theta(wo);
    float maxCos = 0;
    if ((double)sinThetaI > 1e-4 && (double)sinThetaO > 1e-4) {
        float sinPhiI = sin_phi(wi), cosPhiI = cos_phi(wi);
        float sinPhiO = sin_phi(wo), cosPhiO = cos_phi(wo);
        float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
        maxCos = pmax(0.f, dCos);
    }
    float sinAlpha, tanBeta;
    if (fabsf(wi.z) > fabsf(wo.z)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / fabsf(wi.z); }
    else { sinAlpha = sinThetaI; tanBeta = sinThetaO / fabsf(wo.z); }
    return (b.R * PG_INVPI) * (b.onA + b.onB * maxCos * sinAlpha * tanBeta);
}
PG_DEV Spec bsdf_f_local(const Bsdf &b, V3 wo, V3 wi, bool reflect) {
    Spec f = sp(0);
    if (b.hasDiff && reflect) f = f + diffuse_f(b, wo, wi);
    if (b.hasSpec && reflect) f = f + mf_f(b, wo, wi);
    return f;
}
PG_DEV Spec bsdf_f(const Bsdf &b, V3 woW, V3 wiW) {  // reflection.cpp:680-693
    V3 wi = world_to_local(b, wiW), wo = world_to_local(b, woW);
    if (wo.z == 0) return sp(0);
    bool reflect = dot(wiW, b.ng) * dot(woW, b.ng) > 0;
    return bsdf_f_local(b, wo, wi, reflect);
}
PG_DEV float bsdf_pdf(const Bsdf &b, V3 woW, V3 wiW) {  // reflection.cpp:781-796
    if (b.nBxDFs == 0) return 0.f;
    V3 wo = world_to_local(b, woW), wi = world_to_local(b, wiW);
    if (wo.z == 0) return 0.f;
    float pdf = 0.f;
    if (b.hasDiff) pdf += lambert_pdf(wo, wi);
    if (b.hasSpec) pdf += mf_pdf(b, wo, wi);
    return pdf / b.nBxDFs;
}
// A BSDF whose only lobe is specular: SpecularReflection::Sample_f (reflection.cpp:136-143) or FresnelSpecular::Sample_f
// (:487-521) inside BSDF::Sample_f (:714-779).
PG_DEV Spec bsdf_sample_specular(const Bsdf &b, V3 woWorld, V3 &wiWorld, float u0, float &pdf, int &sampledType) {
    V3 wo = world_to_local(b, woWorld), wi;
    Spec f;
    pdf = 0; sampledType = 0;
    if (wo.z == 0) return sp(0);
    if (b.specular == 1) {
        wi = mk(-wo.x, -wo.y, wo.z);
        pdf = 1;
        f = (sp(1.f) * b.Kr) / fabsf(wi.z);
        sampledType = PG_BSDF_SPECULAR | PG_BSDF_REFLECTION;
    } else {
        const float ur = pmin(u0 * 1 - 0, PG_ONE_MINUS_EPS);
        const float F = fr_dielectric(wo.z, 1.f, b.eta);
        if (ur < F) {
            wi = mk(-wo.x, -wo.y, wo.z);
            sampledType = PG_BSDF_SPECULAR | PG_BSDF_REFLECTION;
            pdf = F;
            f = (b.Kr * F) / fabsf(wi.z);
        } else {
            const bool entering = wo.z > 0;
            const float etaI = entering ? 1.f : b.eta, etaT = entering ? b.eta : 1.f;
            const V3 n = (wo.z < 0.f) ? mk(0, 0, -1) : mk(0, 0, 1);  // Faceforward(Normal3f(0, 0, 1), wo)
            const float eta = etaI / etaT;
            // Refract, reflection.h:97-109
            const float cosThetaI = dot(n, wo);
            const float sin2ThetaI = pmax(0.f, 1 - cosThetaI * cosThetaI);
            const float sin2ThetaT = eta * eta * sin2ThetaI;
            if (sin2ThetaT >= 1) return sp(0);
            const float cosThetaT = sqrtf(1 - sin2ThetaT);
            wi = (-wo) * eta + n * (eta * cosThetaI - cosThetaT);
            Spec ft = b.Kt * (1 - F);
            ft = ft * ((etaI * etaI) / (etaT * etaT));  // TransportMode::Radiance
            sampledType = PG_BSDF_SPECULAR | PG_BSDF_TRANSMISSION;
            pdf = 1 - F;
            f = ft / fabsf(wi.z);
        }
    }
    if (pdf == 0) { sampledType = 0; return sp(0); }
    wiWorld = local_to_world(b, wi);
    return f;
}
PG_DEV Spec bsdf_sample_f(const Bsdf &b, V3 woWorld, V3 &wiWorld, float u0, float u1, float &pdf) {  // reflection.cpp:714-779
    int matchingComps = b.nBxDFs;
    pdf = 0;
    if (matchingComps == 0) return sp(0);
    int comp = (int)floorf(u0 * matchingComps);
    if (comp > matchingComps - 1) comp = matchingComps - 1;
    const bool useSpec = b.hasSpec && (comp == 1 || !b.hasDiff);  // lobe order: Lambertian, then microfacet
    float ur0 = pmin(u0 * matchingComps - comp, PG_ONE_MINUS_EPS);
    V3 wo = world_to_local(b, woWorld), wi = mk(0, 0, 0);
    if (wo.z == 0) return sp(0);
    if (useSpec) mf_sample(b, wo, wi, ur0, u1, pdf);
    else {  // BxDF::Sample_f, :383-390
        wi = cosine_sample_hemisphere(ur0, u1);
        if (wo.z < 0) wi.z *= -1;
        pdf = lambert_pdf(wo, wi);
    }
    if (pdf == 0) return sp(0);
    wiWorld = local_to_world(b, wi);
    if (matchingComps > 1) {
        pdf += useSpec ? lambert_pdf(wo, wi) : mf_pdf(b, wo, wi);
        pdf /= matchingComps;
    }
    bool reflect = dot(wiWorld, b.ng) * dot(woWorld, b.ng) > 0;
    return bsdf_f_local(b, wo, wi, reflect);
}

// Triangle::Sample(u, pdf) + Shape::Sample(ref, u, pdf) + DiffuseAreaLight::Sample_Li
// (triangle.cpp:582-607, shape.cpp:56-70, diffuse.cpp:68-81).
// ===========================================================================
// BSDF over a material's BxDF list (PgBxDF, include/pbrt_gpu.h): every BxDF of core/reflection.cpp the closed set of
// materials can add, and BSDF::f / Pdf / Sample_f over the list (reflection.cpp:680-796).  Used by the EXT kernels for
// every material; the specialised Bsdf above is the same arithmetic with the list shape of matte/plastic/mirror/glass
// baked in.
// ===========================================================================
#define PG_BSDF_DIFFUSE 4
#define PG_BSDF_GLOSSY 8
#define PG_BSDF_ALL 31
// `types`: the PgBxDFType of lobes[i] in bits 4i .. 4i+3, `scaled`: bit i = lobes[i] has ScaledBxDF wrappers -- read once when the
// list is bound (lbsdf_bind), so that BSDF::NumComponents / the component choice of Sample_f / the matching tests of f and Pdf are
// register arithmetic instead of a chain of dependent loads through the lane's list pointer
// LB: what a list is made of.  PgBxDF (120 B, the ABI's record: the scene's constant lists, the shading kernel's own copies) or PkLobe,
// the 48-B record k_material writes and k_shade<3> reads (pg_kernels.h) -- the same member NAMES over a type-dependent layout, so that
// lobe_f / lobe_pdf / lobe_sample_f / lobe_fresnel below are ONE text for both and read each field where it is used.  A MixMaterial's
// ScaledBxDF factors take a record of their own behind a PkLobe; all lobes of a hit have them or none has (the hit's material is a mix or
// it is not): recStride = 2 or 1 records per lobe.  (Round 5: the lists were up to 5 x 120 B + 32 B per textured hit, written by k_material
// and read back line by line -- 4.9 x the slot's algorithmic bytes at the L2s' memory side.)
template <class LB> struct LobeBsdfT { V3 ns, ng, ss, ts; const LB *lobes; int recStride; int n; float eta; unsigned types, scaled; };
typedef LobeBsdfT<PgBxDF> LobeBsdf;
PG_DEV int lobe_fresnel_kind(const PgBxDF &b) { return b.fresnel; }
PG_DEV int lobe_fresnel_kind(const PkLobe &b) { return (int)((b.hdr >> 4) & 3u); }
PG_DEV void lbsdf_bind(LobeBsdfT<PgBxDF> &b, const PgBxDF *lobes, int n, float eta) {
    b.lobes = lobes; b.recStride = 1; b.n = n; b.eta = eta; b.types = 0; b.scaled = 0;
    for (int i = 0; i < n; ++i) {
        b.types |= (unsigned)lobes[i].type << (4 * i);
        b.scaled |= (lobes[i].n_scales > 0 ? 1u : 0u) << i;
    }
}
PG_DEV void lbsdf_bind(LobeBsdfT<PkLobe> &b, const PkLobe *rec, int n, float eta, int recStride) {
    b.lobes = rec; b.recStride = recStride; b.n = n; b.eta = eta; b.types = 0; b.scaled = 0;
    for (int i = 0; i < n; ++i) {
        const unsigned hdr = rec[i * recStride].hdr;
        b.types |= (hdr & 15u) << (4 * i);
        b.scaled |= (((hdr >> 6) & 3u) > 0 ? 1u : 0u) << i;
    }
}
template <class LB> PG_DEV const LB &lobe_at(const LobeBsdfT<LB> &b, int i) { return b.lobes[i * b.recStride]; }
template <class LB> PG_DEV int lbsdf_lobe(const LobeBsdfT<LB> &b, int i) { return (int)((b.types >> (4 * i)) & 15u); }
template <class LB> PG_DEV V3 world_to_local(const LobeBsdfT<LB> &b, V3 v) { return mk(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
template <class LB> PG_DEV V3 local_to_world(const LobeBsdfT<LB> &b, V3 v) {
    return mk(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z,
              b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
PG_DEV Spec operator-(Spec a, Spec b) { return sp3(a.r - b.r, a.g - b.g, a.b - b.b); }
PG_DEV Spec operator/(Spec a, Spec b) { return sp3(a.r / b.r, a.g / b.g, a.b / b.b); }
PG_DEV Spec sp_sqrt(Spec a) { return sp3(sqrtf(a.r), sqrtf(a.g), sqrtf(a.b)); }
PG_DEV Spec fr_conductor(float cosThetaI, Spec etai, Spec etat, Spec k) {  // reflection.cpp:71-96
    cosThetaI = clampf(cosThetaI, -1, 1);
    const Spec eta = etat / etai;
    const Spec etak = k / etai;
    const float cosThetaI2 = cosThetaI * cosThetaI;
    const float sinThetaI2 = 1.f - cosThetaI2;
    const Spec eta2 = eta * eta;
    const Spec etak2 = etak * etak;
    const Spec t0 = (eta2 - etak2) - sp(sinThetaI2);
    const Spec a2plusb2 = sp_sqrt(t0 * t0 + (eta2 * 4.f) * etak2);
    const Spec t1 = a2plusb2 + sp(cosThetaI2);
    const Spec a = sp_sqrt((a2plusb2 + t0) * 0.5f);
    const Spec t2 = a * (2.f * cosThetaI);
    const Spec Rs = (t1 - t2) / (t1 + t2);
    const Spec t3 = a2plusb2 * cosThetaI2 + sp(sinThetaI2 * sinThetaI2);
    const Spec t4 = t2 * sinThetaI2;
    const Spec Rp = (Rs * (t3 - t4)) / (t3 + t4);
    return (Rp + Rs) * 0.5f;
}
// TrowbridgeReitzDistribution(alphax, alphay), microfacet.cpp:155-184, :310-345
PG_DEV float tr2_D(float ax, float ay, V3 wh) {
    float tan2Theta = tan2_theta(wh);
    if (isinf(tan2Theta)) return 0.f;
    const float cos4Theta = cos2_theta(wh) * cos2_theta(wh);
    float e = (cos2_phi(wh) / (ax * ax) + sin2_phi(wh) / (ay * ay)) * tan2Theta;
    return 1 / (PG_PI * ax * ay * cos4Theta * (1 + e) * (1 + e));
}
PG_DEV float tr2_lambda(float ax, float ay, V3 w) {
    float absTanTheta = fabsf(tan_theta(w));
    if (isinf(absTanTheta)) return 0.f;
    float alpha = sqrtf(cos2_phi(w) * ax * ax + sin2_phi(w) * ay * ay);
    float alpha2Tan2Theta = (alpha * absTanTheta) * (alpha * absTanTheta);
    return (-1 + sqrtf(1.f + alpha2Tan2Theta)) / 2;
}
PG_DEV float tr2_G1(float ax, float ay, V3 w) { return 1 / (1 + tr2_lambda(ax, ay, w)); }
PG_DEV float tr2_G(float ax, float ay, V3 wo, V3 wi) { return 1 / (1 + tr2_lambda(ax, ay, wo) + tr2_lambda(ax, ay, wi)); }
PG_DEV float tr2_pdf(float ax, float ay, V3 wo, V3 wh) { return tr2_D(ax, ay, wh) * tr2_G1(ax, ay, wo) * absdot(wo, wh) / fabsf(wo.z); }
PG_DEV V3 tr2_sample_wh(float ax, float ay, V3 wo, float u0, float u1) {
    const bool flip = wo.z < 0;
    V3 wi = flip ? -wo : wo;
    V3 wiStretched = normalize(mk(ax * wi.x, ay * wi.y, wi.z));
    float slope_x, slope_y;
    tr_sample11(wiStretched.z, u0, u1, slope_x, slope_y);
    float tmp = cos_phi(wiStretched) * slope_x - sin_phi(wiStretched) * slope_y;
    slope_y = sin_phi(wiStretched) * slope_x + cos_phi(wiStretched) * slope_y;
    slope_x = tmp;
    slope_x = ax * slope_x;
    slope_y = ay * slope_y;
    V3 wh = normalize(mk(-slope_x, -slope_y, 1.f));
    return flip ? -wh : wh;
}
PG_DEV bool same_hemisphere(V3 w, V3 wp) { return w.z * wp.z > 0; }  // reflection.h:111-113
PG_DEV V3 reflect_about(V3 wo, V3 n) { return -wo + n * (2 * dot(wo, n)); }  // reflection.h:93-95
PG_DEV bool refract_dir(V3 wi, V3 n, float eta, V3 &wt) {  // reflection.h:97-109
    float cosThetaI = dot(n, wi);
    float sin2ThetaI = pmax(0.f, 1 - cosThetaI * cosThetaI);
    float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    float cosThetaT = sqrtf(1 - sin2ThetaT);
    wt = (-wi) * eta + n * (eta * cosThetaI - cosThetaT);
    return true;
}
PG_DEV int lobe_type(int type) {  // the BxDFType each constructor passes to BxDF()
    switch (type) {
    case PG_BXDF_LAMBERT_R: case PG_BXDF_OREN_NAYAR: return PG_BSDF_REFLECTION | PG_BSDF_DIFFUSE;
    case PG_BXDF_LAMBERT_T: return PG_BSDF_TRANSMISSION | PG_BSDF_DIFFUSE;
    case PG_BXDF_SPECULAR_R: return PG_BSDF_REFLECTION | PG_BSDF_SPECULAR;
    case PG_BXDF_SPECULAR_T: return PG_BSDF_TRANSMISSION | PG_BSDF_SPECULAR;
    case PG_BXDF_FRESNEL_SPECULAR: return PG_BSDF_REFLECTION | PG_BSDF_TRANSMISSION | PG_BSDF_SPECULAR;
    case PG_BXDF_MICROFACET_R: case PG_BXDF_FRESNEL_BLEND: return PG_BSDF_REFLECTION | PG_BSDF_GLOSSY;
    case PG_BXDF_MICROFACET_T: return PG_BSDF_TRANSMISSION | PG_BSDF_GLOSSY;
    }
    return 0;
}
PG_DEV bool lobe_matches(int lobe, int t) { const int type = lobe_type(lobe); return (type & t) == type; }  // reflection.h:225
template <class LB> PG_DEV Spec lobe_fresnel(const LB &b, float cosI) {  // Fresnel::Evaluate, reflection.cpp:120-135
    const int kind = lobe_fresnel_kind(b);
    if (kind == PG_FRESNEL_DIELECTRIC) return sp(fr_dielectric(cosI, b.eta_a, b.eta_b));
    if (kind == PG_FRESNEL_CONDUCTOR) return fr_conductor(fabsf(cosI), sp(1.f), sp_of(b.cond_eta), sp_of(b.cond_k));
    return sp(1.f);
}
PG_DEV float pow5f(float v) { return (v * v) * (v * v) * v; }
template <class LB> PG_DEV Spec lobe_f(const LB &b, int lobe, V3 wo, V3 wi) {  // BxDF::f of the wrapped BxDF (lobe = b.type), local frame
    const float ax = b.alpha_x, ay = b.alpha_y;
    switch (lobe) {
    case PG_BXDF_LAMBERT_R: return sp_of(b.R) * PG_INVPI;  // reflection.cpp:178-180
    case PG_BXDF_LAMBERT_T: return sp_of(b.T) * PG_INVPI;  // :188-190
    case PG_BXDF_OREN_NAYAR: {  // :197-219
        float sinThetaI = sin_theta(wi), sinThetaO = sin_theta(wo);
        float maxCos = 0;
        if (sinThetaI > 1e-4f && sinThetaO > 1e-4f) {
            float sinPhiI = sin_phi(wi), cosPhiI = cos_phi(wi);
            float sinPhiO = sin_phi(wo), cosPhiO = cos_phi(wo);
            float dCos = cosPhiI * cosPhiO + sinPhiI * sinPhiO;
            maxCos = pmax(0.f, dCos);
        }
        float sinAlpha, tanBeta;
        if (fabsf(wi.z) > fabsf(wo.z)) { sinAlpha = sinThetaO; tanBeta = sinThetaI / fabsf(wi.z); }
        else { sinAlpha = sinThetaI; tanBeta = sinThetaO / fabsf(wo.z); }
        return (sp_of(b.R) * PG_INVPI) * (b.on_a + b.on_b * maxCos * sinAlpha * tanBeta);
    }
    case PG_BXDF_MICROFACET_R: {  // :226-238
        float cosThetaO = fabsf(wo.z), cosThetaI = fabsf(wi.z);
        V3 wh = wi + wo;
        if (cosThetaI == 0 || cosThetaO == 0) return sp(0);
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return sp(0);
        wh = normalize(wh);
        V3 whf = (wh.z < 0.f) ? -wh : wh;  // Faceforward(wh, (0,0,1))
        Spec F = lobe_fresnel(b, dot(wi, whf));
        return (((sp_of(b.R) * tr2_D(ax, ay, wh)) * tr2_G(ax, ay, wo, wi)) * F) / (4 * cosThetaI * cosThetaO);
    }
    case PG_BXDF_MICROFACET_T: {  // :247-274, TransportMode::Radiance
        if (same_hemisphere(wo, wi)) return sp(0);
        float cosThetaO = wo.z, cosThetaI = wi.z;
        if (cosThetaI == 0 || cosThetaO == 0) return sp(0);
        float eta = wo.z > 0 ? (b.eta_b / b.eta_a) : (b.eta_a / b.eta_b);
        V3 wh = normalize(wo + wi * eta);
        if (wh.z < 0) wh = -wh;
        if (dot(wo, wh) * dot(wi, wh) > 0) return sp(0);
        Spec F = sp(fr_dielectric(dot(wo, wh), b.eta_a, b.eta_b));
        float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
        float factor = 1 / eta;
        return ((sp(1.f) - F) * sp_of(b.T)) *
               fabsf(tr2_D(ax, ay, wh) * tr2_G(ax, ay, wo, wi) * eta * eta * absdot(wi, wh) * absdot(wo, wh) * factor * factor /
                     (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
    }
    case PG_BXDF_FRESNEL_BLEND: {  // :295-310; Rd = R, Rs = T
        const Spec Rd = sp_of(b.R), Rs = sp_of(b.T);
        Spec diffuse = (((Rd * (28.f / (23.f * PG_PI))) * (sp(1.f) - Rs)) * (1 - pow5f(1 - .5f * fabsf(wi.z)))) * (1 - pow5f(1 - .5f * fabsf(wo.z)));
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return sp(0);
        wh = normalize(wh);
        Spec schlick = Rs + (sp(1.f) - Rs) * pow5f(1 - dot(wi, wh));  // SchlickFresnel, reflection.h:496-499
        Spec specular = schlick * (tr2_D(ax, ay, wh) / (4 * absdot(wi, wh) * pmax(fabsf(wi.z), fabsf(wo.z))));
        return diffuse + specular;
    }
    }
    return sp(0);  // specular BxDFs: f() = 0
}
template <class LB> PG_DEV float lobe_pdf(const LB &b, int lobe, V3 wo, V3 wi) {
    const float ax = b.alpha_x, ay = b.alpha_y;
    switch (lobe) {
    case PG_BXDF_LAMBERT_R: case PG_BXDF_OREN_NAYAR: return same_hemisphere(wo, wi) ? fabsf(wi.z) * PG_INVPI : 0;  // :392-394
    case PG_BXDF_LAMBERT_T: return !same_hemisphere(wo, wi) ? fabsf(wi.z) * PG_INVPI : 0;  // :405-408
    case PG_BXDF_MICROFACET_R: {  // :425-429
        if (!same_hemisphere(wo, wi)) return 0;
        V3 wh = normalize(wo + wi);
        return tr2_pdf(ax, ay, wo, wh) / (4 * dot(wo, wh));
    }
    case PG_BXDF_MICROFACET_T: {  // :444-458
        if (same_hemisphere(wo, wi)) return 0;
        float eta = wo.z > 0 ? (b.eta_b / b.eta_a) : (b.eta_a / b.eta_b);
        V3 wh = normalize(wo + wi * eta);
        if (dot(wo, wh) * dot(wi, wh) > 0) return 0;
        float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
        float dwh_dwi = fabsf((eta * eta * dot(wi, wh)) / (sqrtDenom * sqrtDenom));
        return tr2_pdf(ax, ay, wo, wh) * dwh_dwi;
    }
    case PG_BXDF_FRESNEL_BLEND: {  // :480-485
        if (!same_hemisphere(wo, wi)) return 0;
        V3 wh = normalize(wo + wi);
        float pdf_wh = tr2_pdf(ax, ay, wo, wh);
        return .5f * (fabsf(wi.z) * PG_INVPI + pdf_wh / (4 * dot(wo, wh)));
    }
    }
    return 0;
}
// BxDF::Sample_f of the wrapped BxDF; pdf keeps the caller's 0 on the reference's early `return 0` paths.  wantF = false: the
// value of a NON-specular BxDF is not computed (0 is returned) -- BSDF::Sample_f throws it away and sums f() over all matching
// BxDFs instead (reflection.cpp:768-775); direction, pdf and the specular BxDFs' values are what they always are
template <class LB> PG_DEV Spec lobe_sample_f(const LB &b, int lobe, V3 wo, V3 &wi, float u0, float u1, float &pdf, int &sampledType, bool wantF = true) {
    const float ax = b.alpha_x, ay = b.alpha_y;
    switch (lobe) {
    case PG_BXDF_LAMBERT_R: case PG_BXDF_OREN_NAYAR:  // BxDF::Sample_f, :383-390
        wi = cosine_sample_hemisphere(u0, u1);
        if (wo.z < 0) wi.z *= -1;
        pdf = lobe_pdf(b, lobe, wo, wi);
        return wantF ? lobe_f(b, lobe, wo, wi) : sp(0);
    case PG_BXDF_LAMBERT_T:  // :396-403
        wi = cosine_sample_hemisphere(u0, u1);
        if (wo.z > 0) wi.z *= -1;
        pdf = lobe_pdf(b, lobe, wo, wi);
        return wantF ? lobe_f(b, lobe, wo, wi) : sp(0);
    case PG_BXDF_SPECULAR_R:  // :136-143
        wi = mk(-wo.x, -wo.y, wo.z);
        pdf = 1;
        return (lobe_fresnel(b, wi.z) * sp_of(b.R)) / fabsf(wi.z);
    case PG_BXDF_SPECULAR_T: {  // :151-167
        const bool entering = wo.z > 0;
        const float etaI = entering ? b.eta_a : b.eta_b, etaT = entering ? b.eta_b : b.eta_a;
        const V3 n = (wo.z < 0.f) ? mk(0, 0, -1) : mk(0, 0, 1);
        if (!refract_dir(wo, n, etaI / etaT, wi)) return sp(0);
        pdf = 1;
        Spec ft = sp_of(b.T) * (sp(1.f) - sp(fr_dielectric(wi.z, b.eta_a, b.eta_b)));
        ft = ft * ((etaI * etaI) / (etaT * etaT));
        return ft / fabsf(wi.z);
    }
    case PG_BXDF_FRESNEL_SPECULAR: {  // :487-521
        const float F = fr_dielectric(wo.z, b.eta_a, b.eta_b);
        if (u0 < F) {
            wi = mk(-wo.x, -wo.y, wo.z);
            sampledType = PG_BSDF_SPECULAR | PG_BSDF_REFLECTION;
            pdf = F;
            return (sp_of(b.R) * F) / fabsf(wi.z);
        }
        const bool entering = wo.z > 0;
        const float etaI = entering ? b.eta_a : b.eta_b, etaT = entering ? b.eta_b : b.eta_a;
        const V3 n = (wo.z < 0.f) ? mk(0, 0, -1) : mk(0, 0, 1);
        if (!refract_dir(wo, n, etaI / etaT, wi)) return sp(0);
        Spec ft = sp_of(b.T) * (1 - F);
        ft = ft * ((etaI * etaI) / (etaT * etaT));
        sampledType = PG_BSDF_SPECULAR | PG_BSDF_TRANSMISSION;
        pdf = 1 - F;
        return ft / fabsf(wi.z);
    }
    case PG_BXDF_MICROFACET_R: {  // :410-423
        if (wo.z == 0) return sp(0);
        V3 wh = tr2_sample_wh(ax, ay, wo, u0, u1);
        if (dot(wo, wh) < 0) return sp(0);
        wi = reflect_about(wo, wh);
        if (!same_hemisphere(wo, wi)) return sp(0);
        pdf = tr2_pdf(ax, ay, wo, wh) / (4 * dot(wo, wh));
        return wantF ? lobe_f(b, lobe, wo, wi) : sp(0);
    }
    case PG_BXDF_MICROFACET_T: {  // :431-442
        if (wo.z == 0) return sp(0);
        V3 wh = tr2_sample_wh(ax, ay, wo, u0, u1);
        if (dot(wo, wh) < 0) return sp(0);
        float eta = wo.z > 0 ? (b.eta_a / b.eta_b) : (b.eta_b / b.eta_a);
        if (!refract_dir(wo, wh, eta, wi)) return sp(0);
        pdf = lobe_pdf(b, lobe, wo, wi);
        return wantF ? lobe_f(b, lobe, wo, wi) : sp(0);
    }
    case PG_BXDF_FRESNEL_BLEND: {  // :460-478
        if ((double)u0 < .5) {
            u0 = pmin(2 * u0, PG_ONE_MINUS_EPS);
            wi = cosine_sample_hemisphere(u0, u1);
            if (wo.z < 0) wi.z *= -1;
        } else {
            u0 = pmin(2 * (u0 - .5f), PG_ONE_MINUS_EPS);
            V3 wh = tr2_sample_wh(ax, ay, wo, u0, u1);
            wi = reflect_about(wo, wh);
            if (!same_hemisphere(wo, wi)) return sp(0);
        }
        pdf = lobe_pdf(b, lobe, wo, wi);
        return wantF ? lobe_f(b, lobe, wo, wi) : sp(0);
    }
    }
    return sp(0);
}
PG_DEV Spec lobe_scale(const PgBxDF &b, Spec f, int) {  // ScaledBxDF, reflection.cpp:98-107: innermost wrapper first
    for (int i = 0; i < b.n_scales; ++i) f = sp_of(b.scale[i]) * f;
    return f;
}
PG_DEV Spec lobe_scale(const PkLobe &b, Spec f, int recStride) {  // the factors: nine floats in the record behind the lobe's (half a lobe's stride on: a scaled list has two records per lobe)
    const float *sc = reinterpret_cast<const float *>(&b + (recStride >> 1));
    const int n = (int)((b.hdr >> 6) & 3u);
    for (int i = 0; i < n; ++i) f = sp_of(sc + 3 * i) * f;
    return f;
}
template <class LB> PG_DEV int lbsdf_num_components(const LobeBsdfT<LB> &b, int flags) {  // reflection.cpp:672-678
    int num = 0;
    for (int i = 0; i < b.n; ++i) if (lobe_matches(lbsdf_lobe(b, i), flags)) ++num;
    return num;
}
template <class LB> PG_DEV Spec lbsdf_scaled(const LobeBsdfT<LB> &b, int i, Spec f) {
    if (!((b.scaled >> i) & 1u)) return f;
    const auto &L = lobe_at(b, i);
    return lobe_scale(L, f, b.recStride);
}
// BxDF::f and BxDF::Pdf of one BxDF for the same pair of directions.  Each value is what lobe_f / lobe_pdf compute -- the same
// operations in the same order --, but the microfacet BxDFs' half vector, D(wh) and Lambda(wo) are computed once for both (a path
// vertex asks for f AND pdf of the other matching BxDFs inside both of its BSDF::Sample_f calls.  For the light's direction BSDF::f and
// BSDF::Pdf stay two walks: fused there as well the kernel gained nothing on microfacet scenes and lost 20 % on an all-Lambert one --
// a matter of what the register allocator makes of it, profiles/r04l_*).
template <class LB> PG_DEV void lobe_f_pdf(const LB &b, int lobe, V3 wo, V3 wi, bool wantF, bool wantPdf, Spec &f, float &pdf) {
    const float ax = b.alpha_x, ay = b.alpha_y;
    f = sp(0); pdf = 0;
    switch (lobe) {
    case PG_BXDF_MICROFACET_R: {  // reflection.cpp:226-238, :425-429
        V3 wh = wi + wo;  // (= wo + wi of Pdf: the same three sums)
        const bool whZero = wh.x == 0 && wh.y == 0 && wh.z == 0;
        wh = normalize(wh);
        const float D = tr2_D(ax, ay, wh), lambdaO = tr2_lambda(ax, ay, wo);
        if (wantPdf && same_hemisphere(wo, wi)) pdf = (D * (1 / (1 + lambdaO)) * absdot(wo, wh) / fabsf(wo.z)) / (4 * dot(wo, wh));
        if (wantF) {
            const float cosThetaO = fabsf(wo.z), cosThetaI = fabsf(wi.z);
            if (!(cosThetaI == 0 || cosThetaO == 0 || whZero)) {
                const V3 whf = (wh.z < 0.f) ? -wh : wh;  // Faceforward(wh, (0,0,1))
                const Spec F = lobe_fresnel(b, dot(wi, whf));
                f = (((sp_of(b.R) * D) * (1 / (1 + lambdaO + tr2_lambda(ax, ay, wi)))) * F) / (4 * cosThetaI * cosThetaO);
            }
        }
        return;
    }
    case PG_BXDF_FRESNEL_BLEND: {  // :295-310, :480-485
        V3 wh = wi + wo;
        const bool whZero = wh.x == 0 && wh.y == 0 && wh.z == 0;
        wh = normalize(wh);
        const float D = tr2_D(ax, ay, wh);
        if (wantPdf && same_hemisphere(wo, wi)) {
            const float pdf_wh = D * tr2_G1(ax, ay, wo) * absdot(wo, wh) / fabsf(wo.z);
            pdf = .5f * (fabsf(wi.z) * PG_INVPI + pdf_wh / (4 * dot(wo, wh)));
        }
        if (wantF && !whZero) {
            const Spec Rd = sp_of(b.R), Rs = sp_of(b.T);
            const Spec diffuse = (((Rd * (28.f / (23.f * PG_PI))) * (sp(1.f) - Rs)) * (1 - pow5f(1 - .5f * fabsf(wi.z)))) * (1 - pow5f(1 - .5f * fabsf(wo.z)));
            const Spec schlick = Rs + (sp(1.f) - Rs) * pow5f(1 - dot(wi, wh));
            f = diffuse + schlick * (D / (4 * absdot(wi, wh) * pmax(fabsf(wi.z), fabsf(wo.z))));
        }
        return;
    }
    }
    if (wantF) f = lobe_f(b, lobe, wo, wi);
    if (wantPdf) pdf = lobe_pdf(b, lobe, wo, wi);
}
template <class LB> PG_DEV Spec lbsdf_f_local(const LobeBsdfT<LB> &b, V3 wo, V3 wi, bool reflect, int flags) {
    Spec f = sp(0);
    for (int i = 0; i < b.n; ++i) {
        const int lobe = lbsdf_lobe(b, i), type = lobe_type(lobe);
        if ((type & flags) == type && ((reflect && (type & PG_BSDF_REFLECTION)) || (!reflect && (type & PG_BSDF_TRANSMISSION))))
            { const auto &L = lobe_at(b, i); f = f + lbsdf_scaled(b, i, lobe_f(L, lobe, wo, wi)); }
    }
    return f;
}
template <class LB> PG_DEV Spec lbsdf_f(const LobeBsdfT<LB> &b, V3 woW, V3 wiW, int flags) {  // reflection.cpp:680-693
    V3 wi = world_to_local(b, wiW), wo = world_to_local(b, woW);
    if (wo.z == 0) return sp(0);
    return lbsdf_f_local(b, wo, wi, dot(wiW, b.ng) * dot(woW, b.ng) > 0, flags);
}
template <class LB> PG_DEV float lbsdf_pdf(const LobeBsdfT<LB> &b, V3 woW, V3 wiW, int flags) {  // reflection.cpp:781-796
    if (b.n == 0) return 0.f;
    V3 wo = world_to_local(b, woW), wi = world_to_local(b, wiW);
    if (wo.z == 0) return 0.f;
    float pdf = 0.f;
    int matchingComps = 0;
    for (int i = 0; i < b.n; ++i) {
        const int lobe = lbsdf_lobe(b, i);
        if (lobe_matches(lobe, flags)) { ++matchingComps; const auto &L = lobe_at(b, i); pdf += lobe_pdf(L, lobe, wo, wi); }
    }
    return matchingComps > 0 ? pdf / matchingComps : 0.f;
}
// BSDF::Sample_f, reflection.cpp:714-779: returns f, pdf = 0 when there is no sample
template <class LB> PG_DEV Spec lbsdf_sample_f(const LobeBsdfT<LB> &b, V3 woWorld, V3 &wiWorld, float u0, float u1, float &pdf, int flags, int &sampledType) {
    const int matchingComps = lbsdf_num_components(b, flags);
    sampledType = 0;
    pdf = 0;
    if (matchingComps == 0) return sp(0);
    int comp = (int)floorf(u0 * matchingComps);
    if (comp > matchingComps - 1) comp = matchingComps - 1;
    int chosen = 0, count = comp;
    for (int i = 0; i < b.n; ++i)
        if (lobe_matches(lbsdf_lobe(b, i), flags) && count-- == 0) { chosen = i; break; }
    const auto &bxdf = lobe_at(b, chosen);
    const int chosenLobe = lbsdf_lobe(b, chosen);
    const float uR0 = pmin(u0 * matchingComps - comp, PG_ONE_MINUS_EPS);
    V3 wi = mk(0, 0, 0), wo = world_to_local(b, woWorld);
    if (wo.z == 0) return sp(0);
    sampledType = lobe_type(chosenLobe);
    const bool bxdfSpecular = (sampledType & PG_BSDF_SPECULAR) != 0;
    Spec f = lbsdf_scaled(b, chosen, lobe_sample_f(bxdf, chosenLobe, wo, wi, uR0, u1, pdf, sampledType, bxdfSpecular));
    if (pdf == 0) { sampledType = 0; return sp(0); }
    wiWorld = local_to_world(b, wi);
    if (!bxdfSpecular) {  // :760-775: the other matching BxDFs' pdfs, and f over all matching BxDFs of the hemisphere the pair is in
        const bool reflect = dot(wiWorld, b.ng) * dot(woWorld, b.ng) > 0;
        f = sp(0);
        for (int i = 0; i < b.n; ++i) {
            const int lobe = lbsdf_lobe(b, i), type = lobe_type(lobe);
            if ((type & flags) != type) continue;
            const bool wantPdf = matchingComps > 1 && i != chosen;
            const bool wantF = (reflect && (type & PG_BSDF_REFLECTION)) || (!reflect && (type & PG_BSDF_TRANSMISSION));
            if (!(wantPdf || wantF)) continue;
            Spec fi;
            float pi;
            { const auto &L = lobe_at(b, i); lobe_f_pdf(L, lobe, wo, wi, wantF, wantPdf, fi, pi); }
            if (wantPdf) pdf += pi;
            if (wantF) f = f + lbsdf_scaled(b, i, fi);
        }
    }
    if (matchingComps > 1) pdf /= matchingComps;
    return f;
}

PG_DEV Spec sp_clamp0(Spec v) { return sp3(v.r < 0 ? 0 : v.r, v.g < 0 ? 0 : v.g, v.b < 0 ? 0 : v.b); }  // Spectrum::Clamp()
PG_DEV void lobe_init(PgBxDF &b, int type) {
    b.type = type; b.fresnel = PG_FRESNEL_NOOP; b.eta_a = b.eta_b = 1; b.alpha_x = b.alpha_y = 1; b.on_a = 1; b.on_b = 0; b.n_scales = 0;
    for (int i = 0; i < 3; ++i) { b.R[i] = b.T[i] = b.cond_eta[i] = b.cond_k[i] = 0; }
}
PG_DEV void lobe_set(float *dst, Spec v) { dst[0] = v.r; dst[1] = v.g; dst[2] = v.b; }
PG_DEV float roughness_to_alpha(float roughness) {  // microfacet.h:127-132; std::log evaluated in double, rounded once
    roughness = pmax(roughness, 1e-3f);
    const float x = pg_logf(roughness);
    return 1.62142f + 0.819955f * x + 0.1734f * x * x + 0.0171201f * x * x * x + 0.000640711f * x * x * x * x;
}
PG_DEV void lobe_tr(PgBxDF &b, float ax, float ay) { b.alpha_x = pmax(0.001f, ax); b.alpha_y = pmax(0.001f, ay); }  // microfacet.h:109-113
// Material::ComputeScatteringFunctions of material `mat` at this hit (materials/): appends its BxDFs to out[n...]; eta receives
// BSDF::eta.  D bounds the nesting of mix materials; cap = the room in out[] (PG_MAX_BXDFS, or the scene's largest list when the
// list goes to k_material's buffer).
// Where MatEval puts a BxDF list: an array of PgBxDFs (the shading kernel's own, MODE 2) or k_material's packed records (LobeBsdfT)
struct LobeOutRaw {
    PgBxDF *p;
    PG_DEV void put(int i, const PgBxDF &b) const { p[i] = b; }
    PG_DEV void add_scale(int i, Spec s) const { if (p[i].n_scales < PG_MAX_BXDF_SCALES) { lobe_set(p[i].scale[p[i].n_scales], s); p[i].n_scales++; } }
};
struct LobeOutPacked {
    float4 *rec;    // record 0 of this thread's list; record r lies r planes on (MatPre)
    int recStride;  // 2: the hit's material is a mix, every lobe is followed by the record of its scale factors
    int plane;      // records per plane = the queue's entries
    PG_DEV void put(int i, const PgBxDF &b) const {
        float4 *q = rec + 3 * ((size_t)(i * recStride) * plane);
        pg_pack_lobe(b, reinterpret_cast<float *>(q));
        if (recStride == 2) pg_pack_lobe_scales(b, reinterpret_cast<float *>(q + 3 * (size_t)plane));
    }
    PG_DEV void add_scale(int i, Spec s) const {  // ScaledBxDF around lobe i (mixmat.cpp:60-65): the next free factor of its scale record
        float4 *q = rec + 3 * ((size_t)(i * recStride) * plane);
        const unsigned hdr = __float_as_uint(q[0].x), ns = (hdr >> 6) & 3u;
        if (ns < PG_MAX_BXDF_SCALES) {
            float *f = reinterpret_cast<float *>(q + 3 * (size_t)plane) + 3 * ns;
            f[0] = s.r; f[1] = s.g; f[2] = s.b;
            q[0].x = __uint_as_float(hdr + (1u << 6));
        }
    }
};
template <int D, int W = 0, class OUT = LobeOutRaw> struct MatEval {
    static PG_DEV_CALL void run(const DScene &sc, int mat, const TexHit &h, OUT out, int &n, float &eta, int capArg) {
        const int cap = W == 0 ? PG_MAX_BXDFS : capArg;  // (the shading kernel's own copies: a constant, no register held for it)
        const PgMaterial &m = sc.materials[mat];
        if (m.type != PG_MAT_TEXTURED) {  // constant parameters: the list the host built
            for (int i = 0; i < m.n_bxdfs && n < cap; ++i) out.put(n++, sc.bxdfs[m.first_bxdf + i]);
            eta = m.bsdf_eta;
            return;
        }
        const PgTexturedMaterial &tm = sc.textured[m.textured_index];
        auto TS = [&](int i) { return TexEval<PG_TEX_DEPTH, W>::s(sc, tm.s[i], h); };
        auto TF = [&](int i) { return TexEval<PG_TEX_DEPTH, W>::f(sc, tm.f[i], h); };
        auto push = [&](const PgBxDF &b) { if (n < cap) out.put(n++, b); };
        PgBxDF b;
        eta = 1;
        switch (tm.kind) {
        case PG_KIND_MATTE: {  // matte.cpp:45-62
            const Spec r = sp_clamp0(TS(0));
            const float sig = clampf(TF(0), 0, 90);
            if (!is_black(r)) {
                lobe_init(b, sig == 0 ? PG_BXDF_LAMBERT_R : PG_BXDF_OREN_NAYAR);
                lobe_set(b.R, r);
                if (sig != 0) {
                    const float sigma = (PG_PI / 180) * sig, sigma2 = sigma * sigma;
                    b.on_a = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                    b.on_b = 0.45f * sigma2 / (sigma2 + 0.09f);
                }
                push(b);
            }
            break;
        }
        case PG_KIND_PLASTIC: {  // plastic.cpp:45-70
            const Spec kd = sp_clamp0(TS(0));
            if (!is_black(kd)) { lobe_init(b, PG_BXDF_LAMBERT_R); lobe_set(b.R, kd); push(b); }
            const Spec ks = sp_clamp0(TS(1));
            if (!is_black(ks)) {
                lobe_init(b, PG_BXDF_MICROFACET_R); lobe_set(b.R, ks);
                b.fresnel = PG_FRESNEL_DIELECTRIC; b.eta_a = 1.5f; b.eta_b = 1.f;
                float rough = TF(0);
                if (tm.remap_roughness) rough = roughness_to_alpha(rough);
                lobe_tr(b, rough, rough);
                push(b);
            }
            break;
        }
        case PG_KIND_MIRROR: {  // mirror.cpp:44-56
            const Spec R = sp_clamp0(TS(0));
            if (!is_black(R)) { lobe_init(b, PG_BXDF_SPECULAR_R); lobe_set(b.R, R); push(b); }
            break;
        }
        case PG_KIND_GLASS: {  // glass.cpp:45-96
            const float e = TF(2);
            float urough = TF(0), vrough = TF(1);
            const Spec R = sp_clamp0(TS(0)), T = sp_clamp0(TS(1));
            eta = e;
            if (is_black(R) && is_black(T)) break;
            if (urough == 0 && vrough == 0) { lobe_init(b, PG_BXDF_FRESNEL_SPECULAR); lobe_set(b.R, R); lobe_set(b.T, T); b.eta_a = 1.f; b.eta_b = e; push(b); }
            else {
                if (tm.remap_roughness) { urough = roughness_to_alpha(urough); vrough = roughness_to_alpha(vrough); }
                if (!is_black(R)) { lobe_init(b, PG_BXDF_MICROFACET_R); lobe_set(b.R, R); b.fresnel = PG_FRESNEL_DIELECTRIC; b.eta_a = 1.f; b.eta_b = e; lobe_tr(b, urough, vrough); push(b); }
                if (!is_black(T)) { lobe_init(b, PG_BXDF_MICROFACET_T); lobe_set(b.T, T); b.eta_a = 1.f; b.eta_b = e; lobe_tr(b, urough, vrough); push(b); }
            }
            break;
        }
        case PG_KIND_UBER: {  // uber.cpp:45-101
            const float e = TF(3);
            const Spec op = sp_clamp0(TS(4));
            const Spec t = sp_clamp0(op * -1.f + sp(1.f));
            if (!is_black(t)) { lobe_init(b, PG_BXDF_SPECULAR_T); lobe_set(b.T, t); push(b); eta = 1.f; }
            else eta = e;
            const Spec kd = op * sp_clamp0(TS(0));
            if (!is_black(kd)) { lobe_init(b, PG_BXDF_LAMBERT_R); lobe_set(b.R, kd); push(b); }
            const Spec ks = op * sp_clamp0(TS(1));
            if (!is_black(ks)) {
                float roughu = tm.has_u ? TF(1) : TF(0);
                float roughv = tm.has_v ? TF(2) : roughu;
                if (tm.remap_roughness) { roughu = roughness_to_alpha(roughu); roughv = roughness_to_alpha(roughv); }
                lobe_init(b, PG_BXDF_MICROFACET_R); lobe_set(b.R, ks); b.fresnel = PG_FRESNEL_DIELECTRIC; b.eta_a = 1.f; b.eta_b = e; lobe_tr(b, roughu, roughv);
                push(b);
            }
            const Spec kr = op * sp_clamp0(TS(2));
            if (!is_black(kr)) { lobe_init(b, PG_BXDF_SPECULAR_R); lobe_set(b.R, kr); b.fresnel = PG_FRESNEL_DIELECTRIC; b.eta_a = 1.f; b.eta_b = e; push(b); }
            const Spec kt = op * sp_clamp0(TS(3));
            if (!is_black(kt)) { lobe_init(b, PG_BXDF_SPECULAR_T); lobe_set(b.T, kt); b.eta_a = 1.f; b.eta_b = e; push(b); }
            break;
        }
        case PG_KIND_METAL: {  // metal.cpp:61-82
            float uRough = tm.has_u ? TF(1) : TF(0);
            float vRough = tm.has_v ? TF(2) : TF(0);
            if (tm.remap_roughness) { uRough = roughness_to_alpha(uRough); vRough = roughness_to_alpha(vRough); }
            lobe_init(b, PG_BXDF_MICROFACET_R);
            lobe_set(b.R, sp(1.f));
            b.fresnel = PG_FRESNEL_CONDUCTOR;
            lobe_set(b.cond_eta, TS(0)); lobe_set(b.cond_k, TS(1));
            lobe_tr(b, uRough, vRough);
            push(b);
            break;
        }
        case PG_KIND_SUBSTRATE: {  // substrate.cpp:45-65
            const Spec d = sp_clamp0(TS(0)), s2 = sp_clamp0(TS(1));
            float roughu = TF(0), roughv = TF(1);
            if (!is_black(d) || !is_black(s2)) {
                if (tm.remap_roughness) { roughu = roughness_to_alpha(roughu); roughv = roughness_to_alpha(roughv); }
                lobe_init(b, PG_BXDF_FRESNEL_BLEND); lobe_set(b.R, d); lobe_set(b.T, s2); lobe_tr(b, roughu, roughv); push(b);
            }
            break;
        }
        case PG_KIND_TRANSLUCENT: {  // translucent.cpp:45-80
            const float e = 1.5f;
            eta = e;
            const Spec r = sp_clamp0(TS(2)), t = sp_clamp0(TS(3));
            if (is_black(r) && is_black(t)) break;
            const Spec kd = sp_clamp0(TS(0));
            if (!is_black(kd)) {
                if (!is_black(r)) { lobe_init(b, PG_BXDF_LAMBERT_R); lobe_set(b.R, r * kd); push(b); }
                if (!is_black(t)) { lobe_init(b, PG_BXDF_LAMBERT_T); lobe_set(b.T, t * kd); push(b); }
            }
            const Spec ks = sp_clamp0(TS(1));
            if (!is_black(ks) && (!is_black(r) || !is_black(t))) {
                float rough = TF(0);
                if (tm.remap_roughness) rough = roughness_to_alpha(rough);
                if (!is_black(r)) { lobe_init(b, PG_BXDF_MICROFACET_R); lobe_set(b.R, r * ks); b.fresnel = PG_FRESNEL_DIELECTRIC; b.eta_a = 1.f; b.eta_b = e; lobe_tr(b, rough, rough); push(b); }
                if (!is_black(t)) { lobe_init(b, PG_BXDF_MICROFACET_T); lobe_set(b.T, t * ks); b.eta_a = 1.f; b.eta_b = e; lobe_tr(b, rough, rough); push(b); }
            }
            break;
        }
        case PG_KIND_MIX: {  // mixmat.cpp:45-65
            const Spec s1 = sp_clamp0(TS(0));
            const Spec s2 = sp_clamp0(sp(1.f) - s1);
            for (int j = 0; j < 2; ++j) {
                const int first = n;
                float subEta = 1;
                MatEval<D - 1, W, OUT>::run(sc, tm.sub[j], h, out, n, subEta, cap);
                if (j == 0) eta = subEta;  // si->bsdf stays m1's
                const Spec scl = j == 0 ? s1 : s2;
                for (int i = first; i < n; ++i) out.add_scale(i, scl);
            }
            break;
        }
        }
    }
};
template <int W, class OUT> struct MatEval<0, W, OUT> {  // below the deepest mix the host allows: only constant-parameter materials
    static PG_DEV void run(const DScene &sc, int mat, const TexHit &, OUT out, int &n, float &eta, int capArg) {
        const int cap = W == 0 ? PG_MAX_BXDFS : capArg;
        const PgMaterial &m = sc.materials[mat];
        for (int i = 0; i < m.n_bxdfs && n < cap; ++i) out.put(n++, sc.bxdfs[m.first_bxdf + i]);
        eta = m.bsdf_eta;
    }
};

struct LightSample { V3 p, n, pError; };
// SpotLight::Falloff, spot.cpp:62-72
PG_DEV float spot_falloff(const PgLight &l, V3 w) {
    V3 wl = normalize(mk(l.w2l[0] * w.x + l.w2l[1] * w.y + l.w2l[2] * w.z, l.w2l[3] * w.x + l.w2l[4] * w.y + l.w2l[5] * w.z,
                         l.w2l[6] * w.x + l.w2l[7] * w.y + l.w2l[8] * w.z));  // Transform::operator()(Vector3), transform.h:233-239
    float cosTheta = wl.z;
    if (cosTheta < l.cos_total_width) return 0;
    if (cosTheta >= l.cos_falloff_start) return 1;
    float delta = (cosTheta - l.cos_total_width) / (l.cos_falloff_start - l.cos_total_width);
    return (delta * delta) * (delta * delta);
}
// ---- Sphere as the shape of a DiffuseAreaLight: Sphere::Sample / ::Pdf, shapes/sphere.cpp:205-305
PG_DEV bool sphere_ref_inside(const PgSphere &s, V3 refp, V3 refErr, V3 refn) {  // sphere.cpp:223-226, :294-297
    const V3 pCenter = m4_point(s.o2w, mk(0, 0, 0));
    const V3 pOrigin = offset_ray_origin(refp, refErr, refn, pCenter - refp);
    return lensq(pOrigin - pCenter) <= s.radius * s.radius;
}
PG_DEV float sphere_cone_pdf(const PgSphere &s, V3 refp) {  // sphere.cpp:301-304, UniformConePdf sampling.cpp:132-134
    const V3 pCenter = m4_point(s.o2w, mk(0, 0, 0));
    const float sinThetaMax2 = s.radius * s.radius / lensq(refp - pCenter);
    const float cosThetaMax = sqrtf(pmax(0.f, 1 - sinThetaMax2));
    return 1 / (2 * PG_PI * (1 - cosThetaMax));
}
PG_DEV LightSample sphere_sample(const PgSphere &s, V3 refp, V3 refErr, V3 refn, float u0, float u1, float &pdf) {
    LightSample it;
    const float radius = s.radius;
    const V3 pCenter = m4_point(s.o2w, mk(0, 0, 0));
    if (sphere_ref_inside(s, refp, refErr, refn)) {  // uniform over the area (sphere.cpp:205-217), then to solid angle (:227-239)
        const float z = 1 - 2 * u0;  // UniformSampleSphere, sampling.cpp:98-103
        const float r = sqrtf(pmax(0.f, 1.f - z * z));
        const float phi = 2 * PG_PI * u1;
        float sP, cP;
        pg_sincosf(phi, &sP, &cP);
        V3 pObj = mk(r * (float)cP, r * (float)sP, z) * radius;
        it.n = normalize(m4_normal(s.w2o, pObj));
        if (s.reverse_orientation) it.n = it.n * -1.f;
        pObj = pObj * (radius / sqrtf(lensq(pObj)));
        it.p = m4_point_err2(s.o2w, pObj, vabs(pObj) * pgamma(5), it.pError);
        pdf = 1 / (s.phi_max * radius * (s.z_max - s.z_min));
        V3 wi = it.p - refp;
        if (lensq(wi) == 0) pdf = 0;
        else {
            wi = normalize(wi);
            pdf *= lensq(refp - it.p) / absdot(it.n, -wi);
        }
        if (isinf(pdf)) pdf = 0.f;
        return it;
    }
    // uniformly inside the subtended cone, sphere.cpp:241-289
    const float dc = sqrtf(lensq(refp - pCenter));
    const float invDc = 1 / dc;
    const V3 wc = (pCenter - refp) * invDc;
    V3 wcX, wcY;
    coordinate_system(wc, wcX, wcY);
    const float sinThetaMax = radius * invDc;
    const float sinThetaMax2 = sinThetaMax * sinThetaMax;
    const float invSinThetaMax = 1 / sinThetaMax;
    const float cosThetaMax = sqrtf(pmax(0.f, 1 - sinThetaMax2));
    float cosTheta = (cosThetaMax - 1) * u0 + 1;
    float sinTheta2 = 1 - cosTheta * cosTheta;
    if (sinThetaMax2 < 0.00068523f) {  // sin^2(1.5 deg): the reference's small-angle series
        sinTheta2 = sinThetaMax2 * u0;
        cosTheta = sqrtf(1 - sinTheta2);
    }
    const float cosAlpha = sinTheta2 * invSinThetaMax + cosTheta * sqrtf(pmax(0.f, 1.f - sinTheta2 * invSinThetaMax * invSinThetaMax));
    const float sinAlpha = sqrtf(pmax(0.f, 1.f - cosAlpha * cosAlpha));
    const float phi = u1 * 2 * PG_PI;
    float sP, cP;
    pg_sincosf(phi, &sP, &cP);
    // SphericalDirection(sinAlpha, cosAlpha, phi, -wcX, -wcY, -wc), geometry.h:1461-1466
    const V3 nWorld = (-wcX) * (sinAlpha * (float)cP) + (-wcY) * (sinAlpha * (float)sP) + (-wc) * cosAlpha;
    const V3 pWorld = pCenter + nWorld * radius;
    it.p = pWorld;
    it.pError = vabs(pWorld) * pgamma(5);
    it.n = nWorld;
    if (s.reverse_orientation) it.n = it.n * -1.f;
    pdf = 1 / (2 * PG_PI * (1 - cosThetaMax));
    return it;
}

// Cylinder::Sample(u, pdf), cylinder.cpp:208-223; Disk::Sample(u, pdf), disk.cpp:128-137
PG_DEV LightSample quadric_sample_area(const PgSphere &s, float u0, float u1, float &pdf) {
    LightSample it;
    if (s.shape == PG_SHAPE_CYLINDER) {
        const float z = plerp(u0, s.z_min, s.z_max);
        const float phi = u1 * s.phi_max;
        float sP, cP;
        pg_sincosf(phi, &sP, &cP);
        V3 pObj = mk(s.radius * (float)cP, s.radius * (float)sP, z);
        it.n = normalize(m4_normal(s.w2o, mk(pObj.x, pObj.y, 0)));
        if (s.reverse_orientation) it.n = it.n * -1.f;
        const float hitRad = sqrtf(pObj.x * pObj.x + pObj.y * pObj.y);
        pObj.x *= s.radius / hitRad;
        pObj.y *= s.radius / hitRad;
        it.p = m4_point_err2(s.o2w, pObj, vabs(mk(pObj.x, pObj.y, 0)) * pgamma(3), it.pError);
    } else {
        float px, py;
        concentric_sample_disk(u0, u1, px, py);
        const V3 pObj = mk(px * s.radius, py * s.radius, s.height);
        it.n = normalize(m4_normal(s.w2o, mk(0, 0, 1)));
        if (s.reverse_orientation) it.n = it.n * -1.f;
        it.p = m4_point_err2(s.o2w, pObj, mk(0, 0, 0), it.pError);
    }
    pdf = 1 / s.area;
    return it;
}

// Triangle::Sample(u, pdf), triangle.cpp:582-607 with UniformSampleTriangle, sampling.cpp:154-157
PG_DEV LightSample tri_sample_area(const DScene &sc, const Tri &t, int prim, float area, float u0, float u1, float &pdf) {
    LightSample ls;
    float su0 = sqrtf(u0);
    float b0 = 1 - su0, b1 = u1 * su0;
    float b2 = (1 - b0 - b1);
    ls.p = t.p0 * b0 + t.p1 * b1 + t.p2 * b2;
    ls.n = normalize(cross(t.p1 - t.p0, t.p2 - t.p0));
    if (sc.attrN && (t.flags & PG_TRI_HAS_N)) {  // triangle.cpp:593-597: orientation follows the shading normal
        V3 ns = tri_interp_normal(sc, prim, b0, b1, b2);
        if (dot(ls.n, ns) < 0.f) ls.n = -ls.n;
    } else if (t.flags & PG_TRI_FLIP_NORMAL) ls.n = ls.n * -1.f;
    V3 pAbsSum = vabs(t.p0 * b0) + vabs(t.p1 * b1) + vabs(t.p2 * b2);
    ls.pError = pAbsSum * pgamma(6);
    pdf = 1 / area;
    return ls;
}
// Shape::Sample(ref, u, pdf), shape.cpp:56-70: the area sample's pdf with respect to solid angle at refp
PG_DEV void area_to_solid_angle(V3 refp, const LightSample &ls, float &pdf) {
    V3 w = ls.p - refp;
    if (lensq(w) == 0) pdf = 0;
    else {
        w = normalize(w);
        pdf *= lensq(refp - ls.p) / absdot(ls.n, -w);
        if (isinf(pdf)) pdf = 0.f;
    }
}
// ---- InfiniteAreaLight with constant radiance: Lmap is a 1x1 MIPMap (lights/infinite.cpp, core/mipmap.h:245-274).
// sinf/cosf/acosf/atan2f of the reference (glibc) are matched by evaluating in double and rounding once.
PG_DEV V3 mat3_mul(const float *m, V3 w) {  // Transform::operator()(Vector3), transform.h:233-239
    return mk(m[0] * w.x + m[1] * w.y + m[2] * w.z, m[3] * w.x + m[4] * w.y + m[5] * w.z, m[6] * w.x + m[7] * w.y + m[8] * w.z);
}
PG_DEV Spec env_lookup(const DScene &sc, const PgLight &l, float s0_, float t0_) {  // Lmap->Lookup(st): width 0 -> triangle(0, st), mipmap.h:214-243
    return mip_triangle(sc, sc.images[l.env_image], 0, s0_, t0_);
}
PG_DEV Spec env_le(const DScene &sc, const PgLight &l, V3 rayD) {  // InfiniteAreaLight::Le, infinite.cpp:93-97
    V3 w = normalize(mat3_mul(l.w2l, rayD));
    return env_lookup(sc, l, spherical_phi(w) * PG_INV2PI, spherical_theta(w) * PG_INVPI);
}
// ProjectionLight::Projection (projection.cpp:79-91) and GonioPhotometricLight::Scale (goniometric.h:67-76); Lookup(st) of
// their maps is triangle(0, st) (mipmap.h:214-243 with width 0)
PG_DEV Spec light_projection(const DScene &sc, const PgLight &l, V3 w) {
    const V3 wl = mat3_mul(l.w2l, w);
    if (wl.z < l.hither) return sp(0);
    const V3 p = m4_point(l.proj, wl);
    if (!(p.x >= l.screen[0] && p.x <= l.screen[2] && p.y >= l.screen[1] && p.y <= l.screen[3])) return sp(0);
    if (l.env_image < 0) return sp(1);
    float ox = p.x - l.screen[0], oy = p.y - l.screen[1];  // Bounds2::Offset
    if (l.screen[2] > l.screen[0]) ox /= l.screen[2] - l.screen[0];
    if (l.screen[3] > l.screen[1]) oy /= l.screen[3] - l.screen[1];
    return env_lookup(sc, l, ox, oy);
}
PG_DEV Spec light_gonio_scale(const DScene &sc, const PgLight &l, V3 w) {
    V3 wp = normalize(mat3_mul(l.w2l, w));
    const float t = wp.y; wp.y = wp.z; wp.z = t;
    const float theta = spherical_theta(wp), phi = spherical_phi(wp);
    if (l.env_image < 0) return sp(1);
    return env_lookup(sc, l, phi * PG_INV2PI, theta * PG_INVPI);
}
PG_DEV float env_sample_1d(const float *func, const float *cdf, float funcInt, int n, float u, float &pdf, int *off) {  // sampling.h:72-89
    int size = n + 1, first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; } else len = half;
    }
    int offset = first - 1; offset = offset < 0 ? 0 : (offset > size - 2 ? size - 2 : offset);
    if (off) *off = offset;
    float du = u - cdf[offset];
    if ((cdf[offset + 1] - cdf[offset]) > 0) du /= (cdf[offset + 1] - cdf[offset]);
    pdf = (funcInt > 0) ? func[offset] / funcInt : 0;
    return (offset + du) / n;
}
PG_DEV float env_pdf_li(const DScene &sc, const PgLight &l, V3 w) {  // InfiniteAreaLight::Pdf_Li, infinite.cpp:127-135; Distribution2D::Pdf, sampling.h:135-141
    V3 wi = mat3_mul(l.w2l, w);
    float theta = spherical_theta(wi), phi = spherical_phi(wi);
    float sinTheta = pg_sinf(theta);
    if (sinTheta == 0) return 0;
    float p0 = phi * PG_INV2PI, p1 = theta * PG_INVPI;
    const int nu = l.env_nu, nv = l.env_nv;
    const float *tab = sc.envTables + l.env_table;
    int iu = (int)(p0 * nu); iu = iu < 0 ? 0 : (iu > nu - 1 ? nu - 1 : iu);
    int iv = (int)(p1 * nv); iv = iv < 0 ? 0 : (iv > nv - 1 ? nv - 1 : iv);
    const float *row = tab + (size_t)(2 * nu + 2) * iv, *marg = tab + (size_t)(2 * nu + 2) * nv;
    return (row[iu] / marg[2 * nv + 1]) / (2 * PG_PI * PG_PI * sinTheta);
}
// EXT: the scene has primitives or lights beyond triangles + area / delta lights (spheres, infinite lights); the plain
// instantiation keeps the common kernels at their register count
template <bool EXT>
PG_DEV Spec light_sample_li(const DScene &sc, const PgLight &light, V3 refp, V3 refErr, V3 refn, float u0, float u1, V3 &wi, float &pdf,
                            LightSample &ls) {
    if (EXT && light.type == PG_LIGHT_INFINITE) {  // InfiniteAreaLight::Sample_Li, infinite.cpp:99-125
        float pdf0, pdf1;
        int v;
        const int nu = light.env_nu, nv = light.env_nv;
        const float *tab = sc.envTables + light.env_table, *marg = tab + (size_t)(2 * nu + 2) * nv;
        float d1 = env_sample_1d(marg, marg + nv, marg[2 * nv + 1], nv, u1, pdf1, &v);  // Distribution2D::SampleContinuous, sampling.h:127-134
        const float *row = tab + (size_t)(2 * nu + 2) * v;
        float d0 = env_sample_1d(row, row + nu, row[2 * nu + 1], nu, u0, pdf0, nullptr);
        float mapPdf = pdf0 * pdf1;
        ls.n = mk(0, 0, 0); ls.pError = mk(0, 0, 0); ls.p = refp;
        pdf = 0;
        if (mapPdf == 0) return sp(0);
        float theta = d1 * PG_PI, phi = d0 * 2 * PG_PI;
        float sT, cT, sP, cP;
        pg_sincosf(theta, &sT, &cT);
        pg_sincosf(phi, &sP, &cP);
        float cosTheta = (float)cT, sinTheta = (float)sT, sinPhi = (float)sP, cosPhi = (float)cP;
        wi = mat3_mul(light.l2w, mk(sinTheta * cosPhi, sinTheta * sinPhi, cosTheta));
        pdf = mapPdf / (2 * PG_PI * PG_PI * sinTheta);
        if (sinTheta == 0) pdf = 0;
        ls.p = refp + wi * (2 * light.world_radius);
        return env_lookup(sc, light, d0, d1);
    }
    if (light.type != PG_LIGHT_AREA) {  // delta lights: point.cpp:43-52, spot.cpp:51-60, distant.cpp:50-60
        const Spec I = sp3(light.L[0], light.L[1], light.L[2]);
        const V3 pos = mk(light.pos[0], light.pos[1], light.pos[2]);
        ls.n = mk(0, 0, 0); ls.pError = mk(0, 0, 0);  // VisibilityTester end point: Interaction(p, time, mediumInterface)
        pdf = 1.f;
        if (light.type == PG_LIGHT_DISTANT) {
            wi = pos;  // wLight
            ls.p = refp + pos * (2 * light.world_radius);  // pOutside
            return I;
        }
        wi = normalize(pos - refp);
        ls.p = pos;
        const float d2 = lensq(pos - refp);
        if (light.type == PG_LIGHT_SPOT) return (I * spot_falloff(light, -wi)) / d2;
        if (EXT && light.type == PG_LIGHT_PROJECTION) return (I * light_projection(sc, light, -wi)) / d2;  // projection.cpp:71-77
        if (EXT && light.type == PG_LIGHT_GONIO) return (I * light_gonio_scale(sc, light, -wi)) / d2;     // goniometric.cpp:43-52
        return I / d2;
    }
    Tri t = load_tri(sc, light.prim);
    const bool quadric = EXT && (t.flags & PG_PRIM_SPHERE);
    if (quadric && sc.spheres[__float_as_int(t.p0.x)].shape == PG_SHAPE_SPHERE)
        ls = sphere_sample(sc.spheres[__float_as_int(t.p0.x)], refp, refErr, refn, u0, u1, pdf);
    else {
        if (quadric) ls = quadric_sample_area(sc.spheres[__float_as_int(t.p0.x)], u0, u1, pdf);
        else ls = tri_sample_area(sc, t, light.prim, light.area, u0, u1, pdf);
        area_to_solid_angle(refp, ls, pdf);
    }
    if (pdf == 0 || lensq(ls.p - refp) == 0) { pdf = 0; return sp(0); }
    wi = normalize(ls.p - refp);
    return (light.two_sided || dot(ls.n, -wi) > 0) ? sp3(light.L[0], light.L[1], light.L[2]) : sp(0);
}
// The light as the shading kernel first meets it: DScene::lightHot's record, fetched in one round trip.
struct LightHot { int type, prim, two_sided; float area; Spec L; Tri tri; };
PG_DEV LightHot load_light_hot(const DScene &sc, int lightNum) {
    const float4 *h = sc.lightHot + 5 * (size_t)lightNum;
    const float4 h0 = h[0], h1 = h[1], a = h[2], b = h[3], c = h[4];
    LightHot lh;
    lh.type = __float_as_int(h0.x); lh.prim = __float_as_int(h0.y); lh.two_sided = __float_as_int(h0.z); lh.area = h0.w;
    lh.L = sp3(h1.x, h1.y, h1.z);
    lh.tri.p0 = mk(a.x, a.y, a.z); lh.tri.p1 = mk(b.x, b.y, b.z); lh.tri.p2 = mk(c.x, c.y, c.z);
    lh.tri.flags = __float_as_uint(a.w); lh.tri.material = __float_as_int(b.w); lh.tri.light = __float_as_int(c.w);
    return lh;
}
// light_sample_li for a light whose hot record is at hand: a diffuse area light on a triangle -- the common case -- is sampled
// from the record alone; everything else goes through the general routine (same arithmetic either way)
template <bool EXT>
PG_DEV Spec light_sample_li_hot(const DScene &sc, const LightHot &lh, const PgLight &light, V3 refp, V3 refErr, V3 refn, float u0, float u1, V3 &wi,
                                float &pdf, LightSample &ls) {
    if (lh.type != PG_LIGHT_AREA || (lh.tri.flags & PG_PRIM_SPHERE)) return light_sample_li<EXT>(sc, light, refp, refErr, refn, u0, u1, wi, pdf, ls);
    ls = tri_sample_area(sc, lh.tri, lh.prim, lh.area, u0, u1, pdf);
    area_to_solid_angle(refp, ls, pdf);
    if (pdf == 0 || lensq(ls.p - refp) == 0) { pdf = 0; return sp(0); }
    wi = normalize(ls.p - refp);
    return (lh.two_sided || dot(ls.n, -wi) > 0) ? lh.L : sp(0);
}

// LightDistribution::Lookup (lightdistrib.cpp:68-82,135-149) -> table of one Distribution1D
PG_DEV const float *light_distribution(const DScene &sc, V3 p) {
    if (sc.lightStrategy != PG_LIGHTS_SPATIAL) return sc.distTable;
    V3 o = p - mk(sc.bmin[0], sc.bmin[1], sc.bmin[2]);  // Bounds3::Offset, geometry.h:801-807
    if (sc.bmax[0] > sc.bmin[0]) o.x /= sc.bmax[0] - sc.bmin[0];
    if (sc.bmax[1] > sc.bmin[1]) o.y /= sc.bmax[1] - sc.bmin[1];
    if (sc.bmax[2] > sc.bmin[2]) o.z /= sc.bmax[2] - sc.bmin[2];
    int pi0 = (int)(o.x * sc.nVoxels[0]), pi1 = (int)(o.y * sc.nVoxels[1]), pi2 = (int)(o.z * sc.nVoxels[2]);
    pi0 = pi0 < 0 ? 0 : (pi0 > sc.nVoxels[0] - 1 ? sc.nVoxels[0] - 1 : pi0);
    pi1 = pi1 < 0 ? 0 : (pi1 > sc.nVoxels[1] - 1 ? sc.nVoxels[1] - 1 : pi1);
    pi2 = pi2 < 0 ? 0 : (pi2 > sc.nVoxels[2] - 1 ? sc.nVoxels[2] - 1 : pi2);
    size_t idx = ((size_t)pi2 * sc.nVoxels[1] + pi1) * sc.nVoxels[0] + pi0;
    if (sc.sparseLights) {  // computed on first touch, like the reference's hash table: ask for it and let the caller retry
        const int slot = sc.voxelSlot[idx];
        if (slot >= 0) return sc.distTable + (size_t)slot * (size_t)(2 * sc.nLights + 2);
        if (slot == -1 && atomicCAS(&sc.voxelSlot[idx], -1, -2) == -1) sc.voxelRequests[atomicAdd(&sc.voxelCounters[0], 1)] = (int)idx;
        return nullptr;
    }
    return sc.distTable + idx * (size_t)(2 * sc.nLights + 2);
}
// Distribution1D::SampleDiscrete, sampling.h:90-100 + FindInterval pbrt.h:403-415
PG_DEV int sample_discrete(const float *tab, int n, float u, float &pdf) {
    if (n <= 3) {
        // a table of at most 8 floats (func[n], cdf[n+1], funcInt) is fetched in ONE round trip and searched in registers; the
        // general path below meets it with three or four dependent loads (funcInt, the cdf entries of the search, func)
        float row[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) row[k] = k < 2 * n + 2 ? tab[k] : 0.f;
        asm volatile("" : "+v"(row[0]), "+v"(row[1]), "+v"(row[2]), "+v"(row[3]), "+v"(row[4]), "+v"(row[5]), "+v"(row[6]), "+v"(row[7]));
        auto at = [&](int k) { return k == 0 ? row[0] : (k == 1 ? row[1] : (k == 2 ? row[2] : (k == 3 ? row[3] : (k == 4 ? row[4] : (k == 5 ? row[5] : (k == 6 ? row[6] : row[7])))))); };
        const float funcInt = at(2 * n + 1);
        int size = n + 1, first = 0, len = size;
        while (len > 0) {
            int half = len >> 1, middle = first + half;
            if (at(n + middle) <= u) { first = middle + 1; len -= half + 1; }
            else len = half;
        }
        int offset = first - 1;
        offset = offset < 0 ? 0 : (offset > size - 2 ? size - 2 : offset);
        pdf = (funcInt > 0) ? at(offset) / (funcInt * n) : 0;
        return offset;
    }
    const float *func = tab, *cdf = tab + n;
    float funcInt = tab[2 * n + 1];
    int size = n + 1, first = 0, len = size;
    while (len > 0) {
        int half = len >> 1, middle = first + half;
        if (cdf[middle] <= u) { first = middle + 1; len -= half + 1; }
        else len = half;
    }
    int offset = first - 1;
    offset = offset < 0 ? 0 : (offset > size - 2 ? size - 2 : offset);
    pdf = (funcInt > 0) ? func[offset] / (funcInt * n) : 0;
    return offset;
}

PG_DEV void spawn_ray(const Isect &is, V3 d, V3 &o) { o = offset_ray_origin(is.p, is.pError, is.n, d); }  // interaction.h:64-67

// MODE 0: triangles + matte/plastic/mirror/glass + area and delta lights (specialised BSDF).  MODE 1 (EXT): any material as
// its BxDF list, quadrics, object instances, infinite lights.  MODE 2: MODE 1 + textured materials, evaluated per hit.

// ===========================================================================
// Participating media (VolPathIntegrator): HomogeneousMedium and the Henyey-Greenstein phase function.  expf / logf /
// sinf / cosf are evaluated in double and rounded once, like every libm call of this file (DESIGN.md "libm").
// ===========================================================================
#define PG_MAX_FLOAT 3.40282346638528859811704183484516925e+38f
PG_DEV Spec sp_exp(Spec a) { return sp3(pg_expf(a.r), pg_expf(a.g), pg_expf(a.b)); }  // Exp(), spectrum.h:414-419
// HomogeneousMedium::Tr, homogeneous.cpp:44-47, for a ray with the given tMax and |d|
PG_DEV Spec medium_tr(const PgMedium &m, float tMax, float dLen) {
    const Spec negSt = sp3(-m.sigma_t[0], -m.sigma_t[1], -m.sigma_t[2]);
    return sp_exp(negSt * pmin(tMax * dLen, PG_MAX_FLOAT));
}
PG_DEV float phase_hg(float cosTheta, float g) {  // medium.h:69-72
    const float denom = 1 + g * g + 2 * g * cosTheta;
    return 0.07957747154594766788f * (1 - g * g) / (denom * sqrtf(denom));
}
PG_DEV float hg_sample_p(float g, V3 wo, V3 &wi, float u0, float u1) {  // HenyeyGreenstein::Sample_p, medium.cpp:194-213
    float cosTheta;
    if ((double)fabsf(g) < 1e-3) cosTheta = 1 - 2 * u0;
    else {
        const float sqrTerm = (1 - g * g) / (1 + g - 2 * g * u0);
        cosTheta = -(1 + g * g - sqrTerm * sqrTerm) / (2 * g);
    }
    const float sinTheta = sqrtf(pmax(0.f, 1 - cosTheta * cosTheta));
    const float phi = 2 * PG_PI * u1;
    V3 v1, v2;
    coordinate_system(wo, v1, v2);
    float sP, cP;
    pg_sincosf(phi, &sP, &cP);
    wi = (v1 * (sinTheta * (float)cP) + v2 * (sinTheta * (float)sP)) + wo * cosTheta;  // SphericalDirection, geometry.h:1461-1466
    return phase_hg(cosTheta, g);
}
// GeometricPrimitive::Intersect's mediumInterface (primitive.cpp:121-125): the primitive's own when it marks a transition,
// else the ray's medium on both sides.  Media are index + 1, 0 = none.
PG_DEV void prim_interface(const DScene &sc, int prim, int rayMedium, int &mIn, int &mOut) {
    const int in1 = sc.triMediumIn ? sc.triMediumIn[prim] + 1 : 0, out1 = sc.triMediumOut ? sc.triMediumOut[prim] + 1 : 0;
    if (in1 != out1) { mIn = in1; mOut = out1; }
    else mIn = mOut = rayMedium;
}

// InstanceToWorld, its inverse and IsIdentity of the instance closest-hit result `i` was reached through: the instance's own or -- a moving one --
// PrimitiveToWorld.Interpolate(r.time), which k_trace<.., XP_ANIM> computed for the accepted hit's ray and left at animXf[i] (pg_motion.h)
// (inner: the TransformedPrimitive INSIDE the object of a hit two levels deep, DScene::hasNest: its matrices wait in the buffer's second half)
PG_DEV const float *inst_i2w(const DScene &sc, int inst, int i, bool inner = false) { const PgInstance &in = sc.instances[inst]; return in.animated ? sc.animXf + (size_t)PG_XF_STRIDE * ((inner ? (size_t)sc.nestXfOff : 0) + i) : in.i2w; }
PG_DEV const float *inst_w2i(const DScene &sc, int inst, int i, bool inner = false) { const PgInstance &in = sc.instances[inst]; return in.animated ? sc.animXf + (size_t)PG_XF_STRIDE * ((inner ? (size_t)sc.nestXfOff : 0) + i) + 16 : in.w2i; }
PG_DEV bool inst_identity(const DScene &sc, int inst, int i, bool inner = false) { const PgInstance &in = sc.instances[inst]; return in.animated ? sc.animXf[(size_t)PG_XF_STRIDE * ((inner ? (size_t)sc.nestXfOff : 0) + i) + 32] != 0.f : in.identity != 0; }
// DScene::hasNest: a closest hit's instance word is outer + nInstances * (inner + 1); inner = -1 for a hit one level deep
PG_DEV void nest_decode(const DScene &sc, int &inst, int &inst2) { inst2 = -1; if (sc.hasNest && inst >= 0) { inst2 = inst / sc.nInstances - 1; inst = inst % sc.nInstances; } }
// InterpolatedPrimToWorld(*isect) of a hit reached through an object instance, transform.cpp:262-297
PG_DEV void isect_to_world(const float *i2w, const float *w2i, Isect &is) {
    Isect w;
    w.p = m4_point_err2(i2w, is.p, is.pError, w.pError);
    w.n = normalize(m4_normal(w2i, is.n));
    w.wo = normalize(m4_vec(i2w, is.wo));
    w.sdpdu = m4_vec(i2w, is.sdpdu);
    w.sdpdv = m4_vec(i2w, is.sdpdv);
    w.sdndu = m4_normal(w2i, is.sdndu); w.sdndv = m4_normal(w2i, is.sdndv);
    w.ns = normalize(m4_normal(w2i, is.ns));
    if (dot(w.ns, w.n) < 0.f) w.ns = -w.ns;  // Faceforward(shading.n, n)
    is = w;
}
// What textures read of the SurfaceInteraction at main-queue entry i: (u, v), p and ComputeDifferentials' outputs
// (interaction.cpp:101-147).  sph*: a quadric hit's (u, v) and geometric dpdu / dpdv; filmX / filmY: the camera sample's pFilm
PG_DEV void tex_hit_setup(const DScene &sc, const PgRenderDesc &rd, const RayQueue &qin, int i, int slot, int prim, const Tri &tri, float4 h4, V3 rayD, int inst, int inst2,
                          bool onSphere, float sphU, float sphV, V3 sphDpdu, V3 sphDpdv, const Isect &is, int4 meta, float filmX, float filmY,
                          bool tileSerial, bool pixelArrays, uint64_t index, TexHit &th) {
    th.p = is.p;
    V3 gdpdu, gdpdv;  // the geometric dpdu / dpdv (not the shading ones)
    if (onSphere) { th.u = sphU; th.v = sphV; gdpdu = sphDpdu; gdpdv = sphDpdv; }
    else {
        float uv[6];
        load_uv(sc, prim, tri.flags, uv);
        tri_dpdu_dpdv(tri.p0, tri.p1, tri.p2, uv, gdpdu, gdpdv);
        th.u = h4.y * uv[0] + h4.z * uv[2] + h4.w * uv[4];  // uvHit, triangle.cpp:332
        th.v = h4.y * uv[1] + h4.z * uv[3] + h4.w * uv[5];
    }
    if (inst2 >= 0 && !inst_identity(sc, inst2, i, true)) { gdpdu = m4_vec(inst_i2w(sc, inst2, i, true), gdpdu); gdpdv = m4_vec(inst_i2w(sc, inst2, i, true), gdpdv); }  // (the inner transform first)
    if (inst >= 0 && !inst_identity(sc, inst, i)) { gdpdu = m4_vec(inst_i2w(sc, inst, i), gdpdu); gdpdv = m4_vec(inst_i2w(sc, inst, i), gdpdv); }
    th.dpdx = th.dpdy = mk(0, 0, 0);
    th.dudx = th.dvdx = th.dudy = th.dvdy = 0;
    if (meta.w & PG_META_HASDIFF) {  // SurfaceInteraction::ComputeDifferentials, interaction.cpp:101-147
        const float4 o4 = qin.o[i];
        const V3 rayO = mk(o4.x, o4.y, o4.z);
        float l0 = 0, l1 = 0;
        if (tileSerial) { l0 = sc.ts[slot].lens0; l1 = sc.ts[slot].lens1; }  // the camera sample's pLens, kept by k_ts_generate
        else if (pixelArrays) { int d2 = 1 << 6; tsb_get2d(sc, meta.x, meta.y, d2, l0, l1); }  // (its second 2D dimension)
        else if (rd.lens_radius > 0) { l0 = halton_sample(sc, rd, index, 3); l1 = halton_sample(sc, rd, index, 4); }
        // the camera sample's time again (its third number), for the camera-to-world transform of that moment
        float uTime = 0;
        if (rd.camera_animated) {
            if (tileSerial) uTime = sc.ts[slot].time;
            else if (pixelArrays) { int d1 = 0; uTime = tsb_get1d(sc, meta.x, meta.y, d1); }
            else uTime = halton_sample(sc, rd, index, 2);
        }
        V3 rxO, rxD, ryO, ryD;
        if (rd.camera_animated) {  // (two calls: one pointer that is either the description's matrix or a private array would be a generic one)
            float c2w[16];
            camera_matrix_at(rd, camera_time(rd, uTime), c2w);
            camera_differentials(rd, c2w, filmX, filmY, l0, l1, rayO, rayD, rxO, rxD, ryO, ryD);
        } else camera_differentials(rd, rd.camera_to_world, filmX, filmY, l0, l1, rayO, rayD, rxO, rxD, ryO, ryD);
        const V3 n = is.n, p = is.p;
        const float dd = dot(n, p);
        const float tx = -(dot(n, rxO) - dd) / dot(n, rxD);
        const float ty = -(dot(n, ryO) - dd) / dot(n, ryD);
        if (!(isinf(tx) || isnan(tx)) && !(isinf(ty) || isnan(ty))) {
            const V3 px = rxO + rxD * tx, py = ryO + ryD * ty;
            th.dpdx = px - p;
            th.dpdy = py - p;
            int d0, d1;
            if (fabsf(n.x) > fabsf(n.y) && fabsf(n.x) > fabsf(n.z)) { d0 = 1; d1 = 2; }
            else if (fabsf(n.y) > fabsf(n.z)) { d0 = 0; d1 = 2; }
            else { d0 = 0; d1 = 1; }
            auto comp = [](V3 v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); };
            const float A00 = comp(gdpdu, d0), A01 = comp(gdpdv, d0), A10 = comp(gdpdu, d1), A11 = comp(gdpdv, d1);
            const float Bx0 = comp(px, d0) - comp(p, d0), Bx1 = comp(px, d1) - comp(p, d1);
            const float By0 = comp(py, d0) - comp(p, d0), By1 = comp(py, d1) - comp(p, d1);
            const float det = A00 * A11 - A01 * A10;  // SolveLinearSystem2x2, transform.cpp:41-49
            if (!(fabsf(det) < 1e-10f)) {
                th.dudx = (A11 * Bx0 - A01 * Bx1) / det; th.dvdx = (A00 * Bx1 - A10 * Bx0) / det;
                if (isnan(th.dudx) || isnan(th.dvdx)) th.dudx = th.dvdx = 0;
                th.dudy = (A11 * By0 - A01 * By1) / det; th.dvdy = (A00 * By1 - A10 * By0) / det;
                if (isnan(th.dudy) || isnan(th.dvdy)) th.dudy = th.dvdy = 0;
            }
        }
    }
}
// Material::Bump (material.cpp:46-85) of the material whose BSDF this is: its own bump map, or -- through mix materials, whose
// BSDF is their first material's -- the first material's
template <int W = 0>
PG_DEV void material_bump(const DScene &sc, int material, const TexHit &th, Isect &is) {
    int bm = material;
    for (int lvl = 0; lvl < 4; ++lvl) {
        const PgMaterial &mm = sc.materials[bm];
        if (mm.type != PG_MAT_TEXTURED) break;
        const PgTexturedMaterial &tmm = sc.textured[mm.textured_index];
        if (tmm.kind == PG_KIND_MIX) { if (tmm.sub[0] < 0) break; bm = tmm.sub[0]; continue; }
        if (tmm.has_bump) {
            TexHit ev = th;
            float du = .5f * (fabsf(th.dudx) + fabsf(th.dudy));
            if (du == 0) du = .0005f;
            ev.p = th.p + is.sdpdu * du; ev.u = th.u + du; ev.v = th.v + 0.f;
            const float uDisplace = TexEval<PG_TEX_DEPTH, W>::f(*sc.self, tmm.bump, ev);
            float dv = .5f * (fabsf(th.dvdx) + fabsf(th.dvdy));
            if (dv == 0) dv = .0005f;
            ev.p = th.p + is.sdpdv * dv; ev.u = th.u + 0.f; ev.v = th.v + dv;
            const float vDisplace = TexEval<PG_TEX_DEPTH, W>::f(*sc.self, tmm.bump, ev);
            const float displace = TexEval<PG_TEX_DEPTH, W>::f(*sc.self, tmm.bump, th);
            const V3 bdpdu = (is.sdpdu + is.ns * ((uDisplace - displace) / du)) + is.sdndu * displace;
            const V3 bdpdv = (is.sdpdv + is.ns * ((vDisplace - displace) / dv)) + is.sdndv * displace;
            is.ns = normalize(cross(bdpdu, bdpdv));  // SetShadingGeometry(..., false), interaction.cpp:72-89
            if (dot(is.ns, is.n) < 0.f) is.ns = -is.ns;
            is.sdpdu = bdpdu; is.sdpdv = bdpdv;
        }
        break;
    }
}

// PG_SHADE_MIN_WAVES: waves per SIMD the BxDF-list variants (MODE 1) are compiled for.  Left alone the volumetric one takes 177
// VGPRs = 2 waves; capped at 168 (3 waves, 20 B of scratch) it is 31 % faster (150 -> 104 ms per frame on the 1 M-triangle volpath
// workload, profiles/r02p_*); 4 waves (128 VGPRs, 200 B of scratch) lose most of that again.  The surface-only variant is at 3 already.
#ifndef PG_SHADE_MIN_WAVES
#define PG_SHADE_MIN_WAVES 3
#endif
// the specialised surface kernel (MODE 0): 125 VGPRs = four waves per SIMD without the cap; the cap keeps later edits there
// the per-hit material evaluation (textured materials), path / volpath: the volpath kernel comes out at 259 registers -- three over the
// edge of 2 waves per SIMD -- and gains 36 % from being held to 248 (10 M-triangle divergent stand-in: 556 -> 357 ms per frame); the
// path kernel does not answer to 2 or 3 waves (253 / 258 / 268 ms; profiles/r03e_shade2_occupancy_anyhit_order.txt)
#ifndef PG_SHADE2_WAVES
#define PG_SHADE2_WAVES 1
#endif
#ifndef PG_SHADE2V_WAVES
#define PG_SHADE2V_WAVES 2
#endif
#ifndef PG_SHADE0_WAVES
#define PG_SHADE0_WAVES 4
#endif
// SSS: the scene has materials with a BSSRDF (subsurface / kdsubsurface): a vertex whose sampled direction is a transmission leaves
// through the BSSRDF branch of Li (path.cpp:152-174) -- the lane draws Sample_S's numbers, builds the probe segment of Sample_Sp and
// hands the path over to the probe / exit kernels below instead of pushing its next ray.  A separate instantiation: scenes without
// such materials run the code they ran before.
// GRID: the scene has a GridDensityMedium ("heterogeneous" medium, media/grid.cpp).  Its transmittance is estimated by ratio tracking
// with numbers from the PATH's sampler, drawn inside VisibilityTester::Tr / Scene::IntersectTr -- i.e. after a vertex's uLight /
// uScattering and before its next-direction sample -- so how many the two transmittance rays take decides which dimension (Halton,
// Sobol') or stream position (tile-serial samplers) the next direction is drawn at.  Such scenes shade every vertex in two launches:
// phase 1 = everything up to the MIS candidate (emission, medium sampling incl. delta tracking, the light sample), then the
// transmittance rays (k_through draws and counts), k_resolve_vol, and phase 2 = the vertex rebuilt, its next direction and the
// roulette.  Scenes without a grid medium run the single-pass kernels they ran before (phase 0).
// HomogeneousMedium::Sample's channel and distance from its two numbers, homogeneous.cpp:51-54
PG_DEV void homogeneous_sample_distance(const PgMedium &mm, float uChannel, float uDist, int &channel, float &dist) {
    channel = (int)(uChannel * 3);
    if (channel > 2) channel = 2;
    dist = -pg_logf((1 - uDist)) / mm.sigma_t[channel];
}
struct GridShade { float4 *vertex; int phase; };  // vertex[slot] = (medium interaction point, kind: 0 none, 1 medium vertex, 2 surface vertex)
template <int MODE, bool VOL, bool SSS = false, bool GRID = false>
__global__ __launch_bounds__(PG_SHADE_BLOCK, MODE == 0 ? PG_SHADE0_WAVES : (((MODE == 1 || MODE == 3) && PG_SHADE_MIN_WAVES > 0) ? PG_SHADE_MIN_WAVES : (MODE == 2 ? (VOL ? PG_SHADE2V_WAVES : PG_SHADE2_WAVES) : 1))) void k_shade(DScene sc, RenderParams rp, PathState st, RayQueue qin, const float4 *__restrict__ hits,
                                                     RayQueue qnext, RayQueue qshadow, RayQueue qmis, unsigned long long *lightTriTests, VolState vs,
                                                     const float *__restrict__ hitT, QueueState qsIn, QueueState qsOut, SssState sss, GridShade gsh) {
    // MODE 3: MODE 1 over the lists k_material left for the hits on materials with textured parameters (rp.matPre)
    constexpr bool EXT = MODE >= 1, TEX = MODE == 2, PRE = MODE == 3;
    static_assert(!PRE || (!SSS && !GRID), "materials evaluated ahead: the plain path / volpath kernels");
    static_assert(!GRID || VOL, "grid media: volpath");
    const bool phaseA = GRID && gsh.phase == 1, phaseB = GRID && gsh.phase == 2;
    int vertexKind = 0;  // GRID: what phase 1 found at this entry (phase 2 reads it back)
    static_assert(!SSS || EXT, "materials with a BSSRDF are BxDF-list materials");
    bool pushJob = false;  // SSS: this lane's path goes on through the BSSRDF (its probe ray waits in s_ray[0])
    // path state and pending terms in queue order (see PathState); volpath: unless the scene has BSSRDF materials or a grid medium, whose
    // kernels (the probe chains, the two shading phases) find a path's state by its slot
    constexpr bool QSTATE = !VOL || (!SSS && !GRID);
    int i;
    if (rp.retryCount > 0) {  // second pass over the entries that waited for a light-distribution voxel (the list holds their POSITIONS in the shading order)
        const int j = blockIdx.x * PG_SHADE_BLOCK + threadIdx.x;
        i = j < rp.retryCount ? rp.retryList[j] : -1;
    } else i = queue_item<PG_SHADE_BLOCK>(qin);
    if (rp.order && i >= 0) i = rp.order[i];  // k_shade_order: the entries of a window grouped by material class
    bool deferred = false;  // sparse light tables: this vertex met a voxel without a distribution; nothing is committed
    unsigned long long tsState0 = 0;  // tile-serial samplers: the tile's stream position and dimension counters on entry
    int tsCur1D0 = 0, tsCur2D0 = 0;
    const bool valid = i >= 0;
    // Output rays are staged in LDS ([queue][o|d][thread]) the moment they are known and copied to their queues after the
    // block-wide append at the end: holding three rays plus the pending direct-light terms in registers until then had
    // pushed the kernel to 130 VGPRs with scratch spills (3 waves/SIMD).
    __shared__ float4 s_ray[3][2][PG_SHADE_BLOCK];
    __shared__ float4 s_state[QSTATE ? 3 : (MODE == 2 ? 1 : 2)][PG_SHADE_BLOCK];  // QSTATE: (L, beta, meta) until the entry's place in the next queue is known; by slot (volpath): L and the film position wait here
    // EXT: of the seven Halton numbers a vertex computes ahead (halton_batch) the last four -- uScattering and the next direction's pair,
    // drawn late -- wait in LDS: they were what the register allocator put into scratch when the BxDF-list body grew; with them (and L,
    // see below) the kernel stays at 159 - 161 registers without scratch (profiles/r04j_bxdf_list_ab.txt).  (Not MODE 0: 2 KB more LDS
    // per block would cost it a resident block)
    __shared__ float s_pre[EXT ? 4 : 1][PG_SHADE_BLOCK];
    // for the integrator's statistics at the end of the kernel (nothing of them is held in a register across it): the bounce count the path arrived
    // with, | 0x4000: this vertex is one of volpath's volumeInteractions, | 0x8000: of its surfaceInteractions
    __shared__ unsigned short s_stat[PG_SHADE_BLOCK];
    const int tid = threadIdx.x;
    bool pushNext = false, pushShadow = false, pushMis = false;
    int slot = 0;
    int pdi = 0;  // index of this vertex's pending direct-light terms: the ray's queue position (QSTATE), else the slot
    unsigned int nLightTests = 0;
    int lightNum = -1;
    int nextBin = 0;  // direction octant of the continuation ray (groups the next queue, see block_push)
    // candidate of the BSDF-sampling half of MIS, tested against the light's triangle at the end
    bool misCand = false;
    V3 misRo = mk(0, 0, 0), misWi = mk(0, 0, 1), misP = mk(0, 0, 0);
    Spec misF = sp(0);
    float misPdf = 0, misLightArea = 1;
    int misLightPrim = 0;
    bool misInside = false;  // sphere light whose sphere contains the shaded point: Sphere::Pdf falls back to Shape::Pdf
    int misMedium = 0;       // VOL: medium of the BSDF-sampled ray
    int newMed = 0;          // VOL, queue-order state: the medium of the path's next ray
    float volWeight = 0;     // VOL: MIS weight of the light sample (-1: delta light)
    if (valid) {
        const float4 d4 = qin.d[i], h4 = hits[i];
        slot = __float_as_int(d4.w);
        pdi = QSTATE ? i : slot;
        const V3 rayD = mk(d4.x, d4.y, d4.z);
        const int prim = __float_as_int(h4.x);
        const PgRenderDesc &rd = rp.rd;
        float4 L4, B4;
        int4 meta;
        if constexpr (QSTATE) { L4 = qsIn.L[i]; B4 = qsIn.beta[i]; meta = qsIn.meta[i]; }
        else { L4 = st.L[slot]; B4 = st.beta[slot]; meta = st.meta[slot]; }
        Spec L = sp3(L4.x, L4.y, L4.z), beta = sp3(B4.x, B4.y, B4.z);
        const uint64_t index = (uint64_t)(uint32_t)meta.x | ((uint64_t)(uint32_t)meta.y << 32);
        int dim = (int)((uint32_t)meta.w >> 20);
        // Sampler::Get1D / Get2D in the order the reference calls them: a GlobalSampler (halton, sobol) is indexed by
        // (sample index, dimension); the tile-serial samplers advance their tile's state (slot = tile, one path per tile)
        const bool tileSerial = rd.sampler >= PG_SAMPLER_RANDOM && !sc.tsBatched;
        const bool pixelArrays = sc.tsBatched != 0;  // a PixelSampler's arrays for this path's pixel: index = (pixel, sample), dim = the two dimension counters
        if (tileSerial) { tsState0 = sc.ts[slot].state; tsCur1D0 = sc.ts[slot].cur1D; tsCur2D0 = sc.ts[slot].cur2D; }  // restored if this vertex is deferred (sparse light tables)
        // Halton: the PG_NPRE dimensions a surface vertex usually draws (light choice, uLight, uScattering, the
        // next direction) are computed together before the first draw (halton_batch); preDim0 < 0: not computed
        constexpr int PG_NPRE = 7;
        float pre[PG_NPRE];
        int preDim0 = -1;
        auto sample_dim = [&](int d) -> float {
            const int k = d - preDim0;
            if (preDim0 >= 0 && k >= 0 && k < PG_NPRE) {
                if constexpr (EXT) return k == 0 ? pre[0] : (k == 1 ? pre[1] : (k == 2 ? pre[2] : s_pre[k - 3][tid]));
                return k == 0 ? pre[0] : (k == 1 ? pre[1] : (k == 2 ? pre[2] : (k == 3 ? pre[3] : (k == 4 ? pre[4] : (k == 5 ? pre[5] : pre[6])))));
            }
            return halton_sample(sc, rd, index, d);
        };
        auto draw1 = [&]() -> float { return tileSerial ? ts_get1d(sc, slot) : (pixelArrays ? tsb_get1d(sc, meta.x, meta.y, dim) : sample_dim(dim++)); };
        auto draw2 = [&](float &a, float &b) {
            if (tileSerial) ts_get2d(sc, rd.sampler, slot, a, b);
            else if (pixelArrays) tsb_get2d(sc, meta.x, meta.y, dim, a, b);
            else { a = sample_dim(dim); b = sample_dim(dim + 1); dim += 2; }
        };
        float etaScale = __int_as_float(meta.z);  // path.cpp:79
        int bounces = meta.w & 0xffff;
        s_stat[tid] = (unsigned short)bounces;
        const bool specularBounce = (meta.w & PG_META_SPECULAR) != 0;
        const bool found = prim >= 0;
        // VOL: the ray's medium (index + 1); volpath.cpp:76-78 samples it before anything else happens at the vertex
        int med = 0;
        bool volDead = false, inMedium = false;
        V3 mediumP = mk(0, 0, 0);
        if constexpr (VOL) {
            if constexpr (QSTATE) med = qsIn.medium[i]; else med = vs.medium[slot];
            newMed = med;
            if (phaseB) {  // the medium was sampled in phase 1: beta carries its weight, the vertex record says what came of it
                const float4 v4 = gsh.vertex[slot];
                vertexKind = __float_as_int(v4.w);
                inMedium = vertexKind == 1;
                mediumP = mk(v4.x, v4.y, v4.z);
            } else if (GRID && med && sc.mediaGrid[med - 1] >= 0) {  // GridDensityMedium::Sample, grid.cpp:61-86: delta tracking
                const PgMedium &mm = sc.media[med - 1];
                const PgDensityGrid &gd = sc.grids[sc.mediaGrid[med - 1]];
                const float4 o4 = qin.o[i];
                const float tMaxRay = found ? hitT[i] : o4.w;
                float tg = 0;
                inMedium = grid_sample(gd, sc.gridDensity + gd.density_offset, mk(o4.x, o4.y, o4.z), rayD, tMaxRay, draw1, tg);
                if (inMedium) {
                    mediumP = mk(o4.x, o4.y, o4.z) + rayD * tg;
                    beta = beta * (sp3(mm.sigma_s[0], mm.sigma_s[1], mm.sigma_s[2]) / gd.sigma_t);
                }
            } else if (med) {  // HomogeneousMedium::Sample, homogeneous.cpp:49-74
                const PgMedium &mm = sc.media[med - 1];
                const float4 o4 = qin.o[i];
                int channel;
                float dist;
                if (!GRID && rp.volPre) {  // drawn by k_shade_order (same dimensions, same arithmetic)
                    const float2 pv = rp.volPre[i];
                    channel = __float_as_int(pv.x); dist = pv.y;
                    dim += 2;
                } else {
                    const float uc = draw1();
                    homogeneous_sample_distance(mm, uc, draw1(), channel, dist);
                }
                const float dLen = sqrtf(lensq(rayD));
                const float tMaxRay = found ? hitT[i] : o4.w;  // ray.tMax after Scene::Intersect
                const float t = pmin(dist / dLen, tMaxRay);
                inMedium = t < tMaxRay;
                if (inMedium) mediumP = mk(o4.x, o4.y, o4.z) + rayD * t;
                const Spec sigT = sp3(mm.sigma_t[0], mm.sigma_t[1], mm.sigma_t[2]), sigS = sp3(mm.sigma_s[0], mm.sigma_s[1], mm.sigma_s[2]);
                const Spec Tr = sp_exp((sp3(-sigT.r, -sigT.g, -sigT.b) * pmin(t, PG_MAX_FLOAT)) * dLen);
                const Spec density = inMedium ? sigT * Tr : Tr;
                float pdf = 0;
                pdf += density.r; pdf += density.g; pdf += density.b;
                pdf *= 1 / (float)3;
                if (pdf == 0) pdf = 1;
                beta = beta * (inMedium ? (Tr * sigS) / pdf : Tr / pdf);
            }
            volDead = is_black(beta);  // volpath.cpp:78
        }
        Tri tri;
        if (found) tri = load_tri(sc, prim);
        // the hit's material record, fetched as soon as its index is known (one round trip, overlapped with the interaction's
        // arithmetic) instead of field by field where each is used
        const PgMaterial mtl = sc.materials[found ? tri.material : 0];
        Isect is;
        float sphU = 0, sphV = 0;  // MODE 2: a quadric hit's (u, v) and geometric dpdu / dpdv, for textures
        V3 sphDpdu = mk(0, 0, 0), sphDpdv = mk(0, 0, 0);
        const bool onSphere = EXT && found && (tri.flags & PG_PRIM_SPHERE);
        // a hit reached through an object instance was computed on the instance-space ray (primitive.cpp:80-82)
        int inst = (EXT && found && sc.hitInst) ? sc.hitInst[i] : -1;
        // MODE 2 also shades the scenes whose hits can lie under TWO transforms (DScene::hasNest: a moving shape inside an object definition): the
        // outer instance's, then the inner TransformedPrimitive's on the way in; InterpolatedPrimToWorld of the inner, then of the outer, on the way out
        int inst2 = -1;
        if constexpr (TEX) nest_decode(sc, inst, inst2);
        V3 shapeRayD = rayD;
        if (inst >= 0) shapeRayD = m4_vec(inst_w2i(sc, inst, i), rayD);
        if (TEX && inst2 >= 0) shapeRayD = m4_vec(inst_w2i(sc, inst2, i, true), shapeRayD);
        if (onSphere) {  // the hit record of a sphere carries tHit: Sphere::Intersect's interaction from the ray and the root
            const float4 o4 = qin.o[i];
            V3 shapeRayO = mk(o4.x, o4.y, o4.z);
            if (inst >= 0) { float dt; instance_ray(inst_w2i(sc, inst, i), shapeRayO, rayD, shapeRayO, shapeRayD, dt); }
            if (TEX && inst2 >= 0) { float dt; instance_ray(inst_w2i(sc, inst2, i, true), shapeRayO, shapeRayD, shapeRayO, shapeRayD, dt); }
            const SphereHit sh = sphere_interaction(sc.spheres[__float_as_int(tri.p0.x)], shapeRayO, shapeRayD, h4.y);
            is.p = sh.p; is.pError = sh.pError; is.wo = sh.wo; is.n = sh.n; is.ns = sh.n; is.sdpdu = sh.dpdu;
            is.sdpdv = sh.dpdv; is.sdndu = sh.dndu; is.sdndv = sh.dndv;
            if (TEX) { sphU = sh.u; sphV = sh.v; sphDpdu = sh.dpdu; sphDpdv = sh.dpdv; }
        }
        // path.cpp:91-102 emitted light at the vertex (volpath.cpp:103-110: only when no medium interaction was sampled)
        if (phaseB || (VOL && (volDead || inMedium))) {
        } else if ((bounces == 0 || specularBounce) && found && tri.light >= 0) {
            const PgLight &l = sc.lights[tri.light];
            V3 nrm = onSphere ? is.n : hit_normal(sc, prim, tri, h4.y, h4.z, h4.w);
            Spec Le = (l.two_sided || dot(nrm, -rayD) > 0) ? sp3(l.L[0], l.L[1], l.L[2]) : sp(0);
            L = L + beta * Le;
        } else if ((bounces == 0 || specularBounce) && found) L = L + beta * sp(0);
        else if (EXT && (bounces == 0 || specularBounce) && !found && sc.hasInfinite) {  // path.cpp:96-100: every infinite light's Le(ray)
            for (int li = 0; li < sc.nLights; ++li)
                if (sc.lights[li].type == PG_LIGHT_INFINITE) L = L + beta * env_le(sc, sc.lights[li], rayD);
        }
        // QSTATE: L is final for this launch here (k_resolve adds the vertex's direct lighting), and the camera sample's film position in
        // L.w / beta.w is only carried along: both wait in LDS from here on instead of in five registers across the whole BSDF part --
        // with them the BxDF-list kernels stay under the 168 registers of three waves per SIMD without scratch
        if constexpr (!TEX) {
            s_state[0][tid] = make_float4(L.r, L.g, L.b, L4.w);
            reinterpret_cast<float *>(&s_state[1][tid])[3] = B4.w;
        }
        bool alive = found && bounces < rd.max_depth;  // path.cpp:104
        int newFlags = 0;
        bool handled = false;
        if constexpr (VOL) {
            if (volDead) alive = false;
            else if (inMedium) alive = bounces < rd.max_depth;  // volpath.cpp:83
            // ++volumeInteractions / ++surfaceInteractions (volpath.cpp:87, :99): noted in LDS, counted at the end of the kernel
            if (!phaseB && !volDead) s_stat[tid] = (unsigned short)(bounces | (inMedium ? (alive ? 0x4000 : 0) : 0x8000));
            if (phaseB) alive = vertexKind == 1 || vertexKind == 2;
            if (alive && inMedium) {
                vertexKind = 1;
                // ---- scattering at a point in the medium, volpath.cpp:80-96: MediumInteraction(p, -ray.d, ..., medium, phase)
                handled = true;
                {   // the dimensions of this vertex's draws side by side (as at a surface vertex, below)
                    const int dimU = __builtin_amdgcn_readfirstlane(dim);
                    const bool can = !tileSerial && !pixelArrays && rd.sampler == 0 && (index >> 32) == 0 && dim >= 2;
                    if (__ballot(!can || dim != dimU) == 0) { halton_batch<PG_NPRE>(sc, (uint32_t)index, dimU, pre); preDim0 = dimU; if constexpr (EXT) { s_pre[0][tid] = pre[3]; s_pre[1][tid] = pre[4]; s_pre[2][tid] = pre[5]; s_pre[3][tid] = pre[6]; } }
                }
                const float g = sc.media[med - 1].g;
                const V3 zero = mk(0, 0, 0), wo = -rayD;
                const float *tab = (sc.nLights > 0 && !phaseB) ? light_distribution(sc, mediumP) : nullptr;
                if (sc.nLights > 0 && !phaseB && !tab) deferred = true;
                if (tab) {  // UniformSampleOneLight(mi, ..., handleMedia = true), integrator.cpp:85-106
                    float lightSelPdf;
                    lightNum = sample_discrete(tab, sc.nLights, draw1(), lightSelPdf);
                    if (lightSelPdf != 0) {
                        float uL0, uL1, uS0, uS1;  // uLight, uScattering (integrator.cpp:101-102)
                        draw2(uL0, uL1);
                        draw2(uS0, uS1);
                        const PgLight &light = sc.lights[lightNum];
                        V3 wi = zero;
                        float lightPdf = 0;
                        float4 pdLight = make_float4(0, 0, 0, 0);
                        LightSample ls;
                        Spec Li = light_sample_li<true>(sc, light, mediumP, zero, zero, uL0, uL1, wi, lightPdf, ls);
                        if (lightPdf > 0 && !is_black(Li)) {
                            const float ph = phase_hg(dot(wo, wi), g);  // integrator.cpp:135-141
                            if (ph != 0) {
                                V3 origin = offset_ray_origin(mediumP, zero, zero, ls.p - mediumP);
                                V3 target = offset_ray_origin(ls.p, ls.pError, ls.n, origin - ls.p);
                                const V3 shD = target - origin;
                                s_ray[1][0][tid] = make_float4(origin.x, origin.y, origin.z, 1 - PG_SHADOW_EPS);
                                s_ray[1][1][tid] = make_float4(shD.x, shD.y, shD.z, __int_as_float(pdi));  // (transmittance rays find their terms by pdi)
                                pushShadow = true;
                                const bool isDelta = PG_LIGHT_IS_DELTA(light.type);
                                volWeight = isDelta ? -1.f : power_heuristic(1, lightPdf, 1, ph);
                                pdLight = make_float4(ph, ph, ph, 0);
                                vs.pdLi[pdi] = make_float4(Li.r, Li.g, Li.b, lightPdf);
                                vs.p1[0][pdi] = make_float4(ls.p.x, ls.p.y, ls.p.z, volWeight); vs.p1[1][pdi] = make_float4(ls.pError.x, ls.pError.y, ls.pError.z, 0);
                                vs.p1[2][pdi] = make_float4(ls.n.x, ls.n.y, ls.n.z, 0);
                                vs.trAcc[0][pdi] = make_float4(1, 1, 1, __int_as_float(med));  // MediumInteraction::GetMedium(): the medium itself
                            }
                        }
                        if (light.type == PG_LIGHT_AREA || light.type == PG_LIGHT_INFINITE) {  // integrator.cpp:164-212 with the phase function
                            V3 wi2;
                            const float ph2 = hg_sample_p(g, wo, wi2, uS0, uS1);
                            if (ph2 != 0 && ph2 > 0) {
                                misCand = true;
                                misRo = mediumP; misWi = wi2; misF = sp(ph2); misPdf = ph2; misP = mediumP; misMedium = med;
                                misLightPrim = light.type == PG_LIGHT_INFINITE ? -1 - lightNum : light.prim; misLightArea = light.area;
                                if (light.type == PG_LIGHT_AREA) {
                                    const float4 la = sc.tris[PG_TRI_STRIDE * light.prim];
                                    if (__float_as_uint(la.w) & PG_PRIM_SPHERE)
                                        misInside = sc.spheres[__float_as_int(la.x)].shape != PG_SHAPE_SPHERE ||
                                                    sphere_ref_inside(sc.spheres[__float_as_int(la.x)], mediumP, zero, zero);
                                }
                            }
                        }
                        pdLight.w = lightSelPdf;
                        st.pdLight[pdi] = pdLight;
                        st.pdBeta[pdi] = make_float4(beta.r, beta.g, beta.b, 0.f);
                    }
                }
                if (!phaseA) {
                // mi.phase->Sample_p for the next direction, volpath.cpp:92-95
                V3 wi;
                float u0, u1;
                draw2(u0, u1);
                hg_sample_p(g, wo, wi, u0, u1);
                s_ray[0][0][tid] = make_float4(mediumP.x, mediumP.y, mediumP.z, PG_INF);
                s_ray[0][1][tid] = make_float4(wi.x, wi.y, wi.z, __int_as_float(slot));
                nextBin = (wi.x < 0 ? 1 : 0) | (wi.y < 0 ? 2 : 0) | (wi.z < 0 ? 4 : 0);
                pushNext = true;  // the path stays in `med`
                Spec rrBeta = beta * etaScale;  // volpath.cpp:178-184
                if (max_component(rrBeta) < rd.rr_threshold && bounces > 3) {
                    float qq = pmax(.05f, 1 - max_component(rrBeta));
                    if (draw1() < qq) pushNext = false;
                    else beta = beta / (1 - qq);
                }
                bounces += 1;
                }
            }
        }
        if (alive && !handled) {
            if (!onSphere) is = make_isect(sc, prim, tri, h4.y, h4.z, h4.w, shapeRayD);
            if (TEX && inst2 >= 0 && !inst_identity(sc, inst2, i, true)) isect_to_world(inst_i2w(sc, inst2, i, true), inst_w2i(sc, inst2, i, true), is);
            if (inst >= 0 && !inst_identity(sc, inst, i)) isect_to_world(inst_i2w(sc, inst, i), inst_w2i(sc, inst, i), is);
            const PgMaterial &m = mtl;
            int mIn = 0, mOut = 0;  // VOL: isect.mediumInterface
            if constexpr (VOL) prim_interface(sc, prim, med, mIn, mOut);
            if (m.type == PG_MAT_NONE) {  // path.cpp:107-113: skip over medium boundaries
              if (!phaseB) {  // (no draws at such a vertex: phase 1 finishes it)
                V3 nextO;
                spawn_ray(is, rayD, nextO);
                s_ray[0][0][tid] = make_float4(nextO.x, nextO.y, nextO.z, PG_INF);
                s_ray[0][1][tid] = make_float4(rayD.x, rayD.y, rayD.z, __int_as_float(slot));
                pushNext = true;
                newFlags = meta.w & PG_META_SPECULAR;  // `continue` leaves specularBounce as it was
                if constexpr (VOL) { newMed = dot(rayD, is.n) > 0 ? mOut : mIn; if constexpr (!QSTATE) vs.medium[slot] = newMed; }  // Interaction::GetMedium(w), interaction.h:86-88
              }
            } else {
                vertexKind = 2;
                // MatteMaterial::ComputeScatteringFunctions (matte.cpp:45-62), BSDF ctor (reflection.h:167-172)
                // BSDF: the EXT kernel evaluates the material's BxDF list (any material); the plain kernel has the list
                // shapes of matte / plastic / mirror / glass baked in (same arithmetic, fewer registers)
                {
                    // every lane here draws at least the next direction; the batch needs one dimension for the whole wave
                    const int dimU = __builtin_amdgcn_readfirstlane(dim);
                    const bool can = !tileSerial && !pixelArrays && rd.sampler == 0 && (index >> 32) == 0 && dim >= 2;
                    if (__ballot(!can || dim != dimU) == 0) { halton_batch<PG_NPRE>(sc, (uint32_t)index, dimU, pre); preDim0 = dimU; if constexpr (EXT) { s_pre[0][tid] = pre[3]; s_pre[1][tid] = pre[4]; s_pre[2][tid] = pre[5]; s_pre[3][tid] = pre[6]; } }
                }
                Bsdf bsdf;
                LobeBsdfT<typename std::conditional<PRE, PkLobe, PgBxDF>::type> lb;
                int sssIdx = -1;  // SSS: index of the hit's BSSRDF, or none
                PgBxDF lobeStore[TEX ? PG_MAX_BXDFS : 1];  // MODE 2: this hit's BxDF list (ComputeScatteringFunctions with textures)
                if constexpr (EXT) {
                    TexHit th;
                    if constexpr (TEX) {
                        tex_hit_setup(sc, rd, qin, i, slot, prim, tri, h4, rayD, inst, inst2, onSphere, sphU, sphV, sphDpdu, sphDpdv, is, meta, L4.w, B4.w, tileSerial,
                                      pixelArrays, index, th);
                        material_bump(sc, tri.material, th, is);
                    }
                    const bool preEvaluated = PRE && m.type == PG_MAT_TEXTURED;
                    float4 head0 = make_float4(0, 0, 0, 0), head1 = head0;
                    const int plane = qin.regionCap * PG_REGIONS;  // PRE: k_material's records lie by POSITION in the shading order, one plane per record (MatPre)
                    if (preEvaluated) {  // the shading frame after Material::Bump, as k_material left it
                        const int p = shade_position(rp, qin);
                        head0 = rp.matPre.head[p]; head1 = rp.matPre.head[(size_t)plane + p];
                        is.ns = mk(head0.x, head0.y, head0.z);
                        is.sdpdu = mk(head1.x, head1.y, head1.z);
                    }
                    lb.ns = is.ns; lb.ng = is.n;
                    lb.ss = normalize(is.sdpdu);
                    lb.ts = cross(lb.ns, lb.ss);
                    if constexpr (PRE) {
                        if (preEvaluated)  // k_material's list: head1.w = the number of BxDFs, bit 8: a mix (scale records)
                            lbsdf_bind(lb, reinterpret_cast<const PkLobe *>(rp.matPre.lobes) + shade_position(rp, qin), __float_as_int(head1.w) & 0xff, head0.w, ((__float_as_int(head1.w) & 0x100) ? 2 : 1) * plane);
                        else {  // a material with constant parameters: its list in the same records, packed once at pg_scene_create
                            const int2 pk = sc.matPk[tri.material];
                            lbsdf_bind(lb, reinterpret_cast<const PkLobe *>(sc.bxdfsPk) + pk.x, m.n_bxdfs, m.bsdf_eta, pk.y);
                        }
                    } else if constexpr (TEX) {
                        int nl = 0;
                        float etaL = 1;
                        MatEval<2>::run(*sc.self, tri.material, th, LobeOutRaw{lobeStore}, nl, etaL, PG_MAX_BXDFS);
                        lbsdf_bind(lb, lobeStore, nl, etaL);
                    } else lbsdf_bind(lb, sc.bxdfs + m.first_bxdf, m.n_bxdfs, m.bsdf_eta);
                    if constexpr (SSS) {  // si->bssrdf = TabulatedBSSRDF(...): subsurface.cpp:87-90, kdsubsurface.cpp:88-93
                        sssIdx = sc.materialBssrdf[tri.material];
                        if (sssIdx >= 0) {
                            const PgBSSRDF &bd = sc.bssrdfs[sssIdx];
                            float sigT[3] = {bd.sigma_t[0], bd.sigma_t[1], bd.sigma_t[2]}, rho[3] = {bd.rho[0], bd.rho[1], bd.rho[2]};
                            if (bd.textured) {
                                if (lb.n == 0) sssIdx = -1;  // ComputeScatteringFunctions returned before it set the BSSRDF (R and T both black)
                                else if constexpr (TEX) {
                                    const Spec ta = sp_clamp0(TexEval<PG_TEX_DEPTH>::s(*sc.self, bd.a, th)), tb = sp_clamp0(TexEval<PG_TEX_DEPTH>::s(*sc.self, bd.b, th));
                                    const float a3[3] = {ta.r, ta.g, ta.b}, b3[3] = {tb.r, tb.g, tb.b};
                                    const DBssrdf tbl = bssrdf_bind(bd, sc.bssrdfTables);
                                    for (int c = 0; c < 3; ++c) {
                                        float sa, ssc;
                                        if (bd.textured == 1) { sa = a3[c] * bd.scale; ssc = b3[c] * bd.scale; }  // subsurface.cpp:87-88
                                        else {  // SubsurfaceFromDiffuse(table, Kd, scale * mfp), bssrdf.cpp:182-191
                                            const float mfree = b3[c] * bd.scale;
                                            const float r = invert_catmull_rom(tbl.nRho, tbl.rhoSamples, tbl.rhoEff, a3[c]);
                                            ssc = r / mfree; sa = (1 - r) / mfree;
                                        }
                                        sigT[c] = sa + ssc;  // TabulatedBSSRDF's constructor, bssrdf.h:146-150
                                        rho[c] = sigT[c] != 0 ? (ssc / sigT[c]) : 0;
                                    }
                                }
                            }
                            if (sssIdx >= 0) {
                                sss.coef[0][slot] = make_float4(sigT[0], sigT[1], sigT[2], 0);
                                sss.coef[1][slot] = make_float4(rho[0], rho[1], rho[2], 0);
                            }
                        }
                    }
                } else {
                    bsdf.ns = is.ns; bsdf.ng = is.n;
                    bsdf.ss = normalize(is.sdpdu);
                    bsdf.ts = cross(bsdf.ns, bsdf.ss);
                    bsdf.R = sp3(m.kd[0] < 0 ? 0 : m.kd[0], m.kd[1] < 0 ? 0 : m.kd[1], m.kd[2] < 0 ? 0 : m.kd[2]);
                    bsdf.hasDiff = !is_black(bsdf.R);
                    bsdf.orenNayar = false; bsdf.onA = 1; bsdf.onB = 0;
                    if (m.type == PG_MAT_MATTE) {  // matte.cpp:55-61; OrenNayar ctor, reflection.h:425-431
                        const float sig = clampf(m.sigma, 0, 90);
                        if (sig != 0) {
                            const float sigma = (PG_PI / 180) * sig, sigma2 = sigma * sigma;
                            bsdf.orenNayar = true;
                            bsdf.onA = 1.f - (sigma2 / (2.f * (sigma2 + 0.33f)));
                            bsdf.onB = 0.45f * sigma2 / (sigma2 + 0.09f);
                        }
                    }
                    // PlasticMaterial (plastic.cpp:57-69): m.roughness already holds the distribution's alpha (host: RoughnessToAlpha)
                    bsdf.Ks = sp3(m.ks[0] < 0 ? 0 : m.ks[0], m.ks[1] < 0 ? 0 : m.ks[1], m.ks[2] < 0 ? 0 : m.ks[2]);
                    bsdf.hasSpec = m.type == PG_MAT_PLASTIC && !is_black(bsdf.Ks);
                    bsdf.alpha = m.roughness;
                    bsdf.nBxDFs = (bsdf.hasDiff ? 1 : 0) + (bsdf.hasSpec ? 1 : 0);
                    bsdf.specular = 0; bsdf.eta = 1;
                    bsdf.Kr = sp3(m.kr[0] < 0 ? 0 : m.kr[0], m.kr[1] < 0 ? 0 : m.kr[1], m.kr[2] < 0 ? 0 : m.kr[2]);
                    bsdf.Kt = sp3(m.kt[0] < 0 ? 0 : m.kt[0], m.kt[1] < 0 ? 0 : m.kt[1], m.kt[2] < 0 ? 0 : m.kt[2]);
                    if (m.type == PG_MAT_MIRROR || m.type == PG_MAT_GLASS) {  // mirror.cpp:44-56, glass.cpp:45-65 (smooth)
                        bsdf.hasDiff = false;
                        if (m.type == PG_MAT_MIRROR) bsdf.specular = is_black(bsdf.Kr) ? 0 : 1;
                        else { bsdf.eta = m.eta; bsdf.specular = (is_black(bsdf.Kr) && is_black(bsdf.Kt)) ? 0 : 2; }
                        bsdf.nBxDFs = bsdf.specular ? 1 : 0;
                    }
                }
                const V3 shNs = EXT ? lb.ns : bsdf.ns;
                const int nonSpecular = PG_BSDF_ALL & ~PG_BSDF_SPECULAR;
                bool hasNonSpecular;
                if constexpr (EXT) hasNonSpecular = lbsdf_num_components(lb, nonSpecular) > 0;
                else hasNonSpecular = bsdf.nBxDFs > 0 && !bsdf.specular;
                // ---- direct lighting: UniformSampleOneLight (integrator.cpp:85-106) + EstimateDirect set-up
                if (!VOL && hasNonSpecular && sc.nLights == 0) lightNum = -2;  // (still one of path.cpp:121's totalPaths -- and a zero-radiance one: k_resolve counts pdInfo.z != -1)
                const bool wantLight = (VOL || hasNonSpecular) && sc.nLights > 0 && !phaseB;  // path.cpp:119: only with non-specular lobes (volpath.cpp:124-127: always)
                const float *tab = wantLight ? light_distribution(sc, is.p) : nullptr;
                if (wantLight && !tab) deferred = true;
                if (tab) {
                    float lightSelPdf;
                    lightNum = sample_discrete(tab, sc.nLights, draw1(), lightSelPdf);
                    if (lightSelPdf != 0) {
                        float uL0, uL1, uS0, uS1;  // uLight, uScattering (integrator.cpp:101-102)
                        draw2(uL0, uL1);
                        draw2(uS0, uS1);
                        const LightHot lh = load_light_hot(sc, lightNum);
                        const PgLight &light = sc.lights[lightNum];
                        V3 wi = mk(0, 0, 0);
                        float lightPdf = 0, scatteringPdf = 0;
                        float4 pdLight = make_float4(0, 0, 0, 0);
                        LightSample ls;
                        Spec Li = light_sample_li_hot<EXT>(sc, lh, light, is.p, is.pError, is.n, uL0, uL1, wi, lightPdf, ls);
                        if (lightPdf > 0 && !is_black(Li)) {
                            Spec f;
                            if constexpr (EXT) { f = lbsdf_f(lb, is.wo, wi, nonSpecular) * absdot(wi, shNs); scatteringPdf = lbsdf_pdf(lb, is.wo, wi, nonSpecular); }
                            else { f = bsdf_f(bsdf, is.wo, wi) * absdot(wi, shNs); scatteringPdf = bsdf_pdf(bsdf, is.wo, wi); }
                            if (!is_black(f)) {
                                // VisibilityTester: p0.SpawnRayTo(p1), interaction.h:73-78
                                V3 origin = offset_ray_origin(is.p, is.pError, is.n, ls.p - is.p);
                                V3 target = offset_ray_origin(ls.p, ls.pError, ls.n, origin - ls.p);
                                const V3 shD = target - origin;
                                s_ray[1][0][tid] = make_float4(origin.x, origin.y, origin.z, 1 - PG_SHADOW_EPS);
                                s_ray[1][1][tid] = make_float4(shD.x, shD.y, shD.z, __int_as_float(VOL ? pdi : slot));
                                pushShadow = true;
                                // delta lights take no MIS weight (integrator.cpp:155-160)
                                const bool isDelta = PG_LIGHT_IS_DELTA(lh.type);
                                if constexpr (VOL) {  // Li still is to be multiplied by VisibilityTester::Tr (integrator.cpp:146-150): keep the factors apart
                                    volWeight = isDelta ? -1.f : power_heuristic(1, lightPdf, 1, scatteringPdf);
                                    pdLight = make_float4(f.r, f.g, f.b, 0);
                                    vs.pdLi[pdi] = make_float4(Li.r, Li.g, Li.b, lightPdf);
                                    vs.p1[0][pdi] = make_float4(ls.p.x, ls.p.y, ls.p.z, volWeight); vs.p1[1][pdi] = make_float4(ls.pError.x, ls.pError.y, ls.pError.z, 0);
                                    vs.p1[2][pdi] = make_float4(ls.n.x, ls.n.y, ls.n.z, 0);
                                    vs.trAcc[0][pdi] = make_float4(1, 1, 1, __int_as_float(dot(shD, is.n) > 0 ? mOut : mIn));
                                } else {
                                    Spec c = isDelta ? (f * Li) / lightPdf : ((f * Li) * power_heuristic(1, lightPdf, 1, scatteringPdf)) / lightPdf;
                                    pdLight = make_float4(c.r, c.g, c.b, 0);
                                }
                            }
                        }
                        // pending terms of this vertex, consumed by k_resolve (pdMis and the MIS weight follow below) -- stored HERE: held until after the
                        // BSDF sample below they were five registers at the kernel's peak (tools/reg_pressure.py: lbsdf_sample_f's microfacet terms)
                        pdLight.w = lightSelPdf;
                        st.pdLight[pdi] = pdLight;
                        st.pdBeta[pdi] = make_float4(beta.r, beta.g, beta.b, 0.f);
                        // (beta waits in LDS, where it goes at the end anyway, while the BSDF is sampled: three more registers off the same peak)
                        if constexpr (!TEX) { float *sb = reinterpret_cast<float *>(&s_state[1][tid]); sb[0] = beta.r; sb[1] = beta.g; sb[2] = beta.b; }
                        // BSDF sampling half of MIS (integrator.cpp:164-212): sample now, while the BSDF is live; the
                        // light.Pdf_Li triangle test runs at the end of the kernel, when little else is (register pressure)
                        V3 wi2 = wi;
                        float sPdf2 = 0;
                        Spec f2 = sp(0);
                        if (lh.type == PG_LIGHT_AREA || (EXT && lh.type == PG_LIGHT_INFINITE)) {  // integrator.cpp:164: if (!IsDeltaLight(light.flags))
                            if constexpr (EXT) { int st2; f2 = lbsdf_sample_f(lb, is.wo, wi2, uS0, uS1, sPdf2, nonSpecular, st2); }
                            else f2 = bsdf_sample_f(bsdf, is.wo, wi2, uS0, uS1, sPdf2);
                            f2 = f2 * absdot(wi2, shNs);
                        }
                        if (!is_black(f2) && sPdf2 > 0) {
                            misCand = true;
                            spawn_ray(is, wi2, misRo);
                            misWi = wi2; misF = f2; misPdf = sPdf2; misP = is.p;
                            if constexpr (VOL) misMedium = dot(wi2, is.n) > 0 ? mOut : mIn;
                            misLightPrim = (EXT && lh.type == PG_LIGHT_INFINITE) ? -1 - lightNum : lh.prim; misLightArea = lh.area;
                            if (EXT && lh.type == PG_LIGHT_AREA && (lh.tri.flags & PG_PRIM_SPHERE)) {  // only the sphere overrides Shape::Pdf (with its cone pdf, from outside)
                                const PgSphere &lsph = sc.spheres[__float_as_int(lh.tri.p0.x)];
                                misInside = lsph.shape != PG_SHAPE_SPHERE || sphere_ref_inside(lsph, is.p, is.pError, is.n);
                            }
                        }
                        if constexpr (!TEX) { const volatile float *sb = reinterpret_cast<const volatile float *>(&s_state[1][tid]); beta = sp3(sb[0], sb[1], sb[2]); }
                    }
                }
                // ---- sample the BSDF for the next direction (path.cpp:130-150)
                if (!phaseA) {
                // (the ray's direction once more from its queue entry: kept from the top of the kernel it was three registers at the peak above)
                const float4 dNow = qin.d[i];
                V3 wo = -mk(dNow.x, dNow.y, dNow.z), wi;
                float pdf;
                float u0, u1;
                draw2(u0, u1);
                int sampledType = 0;
                Spec f;
                float bsdfEta;
                if constexpr (EXT) { f = lbsdf_sample_f(lb, wo, wi, u0, u1, pdf, PG_BSDF_ALL, sampledType); bsdfEta = lb.eta; }
                else { f = bsdf.specular ? bsdf_sample_specular(bsdf, wo, wi, u0, pdf, sampledType) : bsdf_sample_f(bsdf, wo, wi, u0, u1, pdf); bsdfEta = bsdf.eta; }
                if (!(is_black(f) || pdf == 0.f)) {
                    beta = beta * ((f * absdot(wi, shNs)) / pdf);
                    if (sampledType & PG_BSDF_SPECULAR) newFlags |= PG_META_SPECULAR;  // path.cpp:142
                    if ((sampledType & PG_BSDF_SPECULAR) && (sampledType & PG_BSDF_TRANSMISSION))  // path.cpp:143-149
                        etaScale *= (dot(wo, is.n) > 0) ? (bsdfEta * bsdfEta) : 1 / (bsdfEta * bsdfEta);
                    V3 nextO;
                    spawn_ray(is, wi, nextO);
                    s_ray[0][0][tid] = make_float4(nextO.x, nextO.y, nextO.z, PG_INF);
                    s_ray[0][1][tid] = make_float4(wi.x, wi.y, wi.z, __int_as_float(slot));
                    nextBin = (wi.x < 0 ? 1 : 0) | (wi.y < 0 ? 2 : 0) | (wi.z < 0 ? 4 : 0);
                    pushNext = true;
                    if constexpr (VOL) { newMed = dot(wi, is.n) > 0 ? mOut : mIn; if constexpr (!QSTATE) { if (!deferred) vs.medium[slot] = newMed; } }
                    bool throughBssrdf = false;
                    if constexpr (SSS) {
                        if (sssIdx >= 0 && (sampledType & PG_BSDF_TRANSMISSION)) {
                            // path.cpp:152-156: bssrdf->Sample_S(scene, sampler.Get1D(), sampler.Get2D(), ...) -- the reference build
                            // evaluates the arguments right to left: the 2D sample is drawn first.  Then the first half of
                            // SeparableBSSRDF::Sample_Sp (bssrdf.cpp:257-292): the probe segment; the path ends here when there is none
                            throughBssrdf = true;
                            pushNext = false;
                            float u2x, u2y;
                            draw2(u2x, u2y);
                            float u1 = draw1();
                            DBssrdf b = bssrdf_bind(sc.bssrdfs[sssIdx], sc.bssrdfTables);
                            const float4 cs = sss.coef[0][slot], cr = sss.coef[1][slot];
                            b.sigma_t[0] = cs.x; b.sigma_t[1] = cs.y; b.sigma_t[2] = cs.z; b.rho[0] = cr.x; b.rho[1] = cr.y; b.rho[2] = cr.z;
                            V3 baseP, pTarget;
                            if (bssrdf_probe_segment(b, lb.ss, lb.ts, lb.ns, is.p, u1, u2x, u2y, baseP, pTarget)) {
                                const V3 pd = pTarget - baseP;  // base.SpawnRayTo(pTarget) of an Interaction without normal or error: from baseP itself
                                if (!(pd.x == 0 && pd.y == 0 && pd.z == 0)) {  // bssrdf.cpp:306: a zero direction ends the chain before it starts
                                    s_ray[0][0][tid] = make_float4(baseP.x, baseP.y, baseP.z, 1 - PG_SHADOW_EPS);
                                    s_ray[0][1][tid] = make_float4(pd.x, pd.y, pd.z, __int_as_float(slot));
                                    pushJob = true;
                                    sss.po[slot] = make_float4(is.p.x, is.p.y, is.p.z, u1);
                                    sss.frame[0][slot] = make_float4(lb.ns.x, lb.ns.y, lb.ns.z, b.eta);
                                    sss.frame[1][slot] = make_float4(lb.ss.x, lb.ss.y, lb.ss.z, __int_as_float(sssIdx));
                                    sss.frame[2][slot] = make_float4(lb.ts.x, lb.ts.y, lb.ts.z, __int_as_float(sc.bssrdfs[sssIdx].match_material));
                                    sss.target[slot] = make_float4(pTarget.x, pTarget.y, pTarget.z, 0);
                                    sss.count[slot] = make_int2(0, 0);
                                    if constexpr (VOL) sss.medium[slot] = make_int2(0, 0);  // Sample_Sp's `base` has no MediumInterface
                                }
                            }
                        }
                    }
                    // Russian roulette, path.cpp:176-184 (a path inside the BSSRDF branch: at its exit vertex, k_sss_exit)
                    Spec rrBeta = beta * etaScale;
                    if (!throughBssrdf && max_component(rrBeta) < rd.rr_threshold && bounces > 3) {
                        float qq = pmax(.05f, 1 - max_component(rrBeta));
                        if (draw1() < qq) pushNext = false;
                        else beta = beta / (1 - qq);
                    }
                }
                bounces += 1;
                }
            }
        }
        if (deferred) {
        } else if constexpr (QSTATE) {  // written after the append: to the ray's entry of the next queue, or (path over) L to its slot
            if constexpr (TEX) {
                s_state[0][tid] = make_float4(L.r, L.g, L.b, L4.w);
                s_state[1][tid] = make_float4(beta.r, beta.g, beta.b, B4.w);
            } else {  // (L and beta.w are there already, see above)
                float *sb = reinterpret_cast<float *>(&s_state[1][tid]);
                sb[0] = beta.r; sb[1] = beta.g; sb[2] = beta.b;
            }
            s_state[2][tid] = make_float4(__int_as_float(meta.x), __int_as_float(meta.y), etaScale, __int_as_float((dim << 20) | bounces | newFlags));
        } else {
            // (GRID: phase 1 leaves the incoming flags for phase 2's rebuild of the vertex -- a null-material surface, finished in
            // phase 1, sets its own --; phase 2 leaves L alone: k_resolve_vol has added this vertex's direct lighting to it)
            if (phaseA && vertexKind != 0) newFlags = meta.w & 0xf0000;
            if (!phaseB) { if constexpr (TEX) st.L[slot] = make_float4(L.r, L.g, L.b, L4.w); else st.L[slot] = s_state[0][tid]; }
            // (phase 2 has nothing to do for a vertex that phase 1 finished -- a surface without a material, the end of a path -- and must
            // leave its state alone: writing it back would clear the flags phase 1 set, e.g. "the last bounce was specular", which a
            // path keeps across a material-less surface, path.cpp:107-113)
            if (!(phaseB && vertexKind == 0)) {
                st.beta[slot] = make_float4(beta.r, beta.g, beta.b, TEX ? B4.w : reinterpret_cast<float *>(&s_state[1][tid])[3]);
                st.meta[slot] = make_int4(meta.x, meta.y, __float_as_int(etaScale), (dim << 20) | bounces | newFlags);
            }
            if (phaseA) gsh.vertex[slot] = make_float4(mediumP.x, mediumP.y, mediumP.z, __int_as_float(alive ? vertexKind : 0));
        }
    }
    if (deferred) {  // retried once the voxel's distribution exists: no ray, no state, no pending term leaves this launch
        pushNext = pushShadow = misCand = false;
        if (rp.rd.sampler >= PG_SAMPLER_RANDOM) { sc.ts[slot].state = tsState0; sc.ts[slot].cur1D = tsCur1D0; sc.ts[slot].cur2D = tsCur2D0; }  // nor a draw from the tile's stream
        rp.retryList[atomicAdd(&sc.voxelCounters[1], 1)] = shade_position(rp, qin);
    }
    if (misCand) {
        // light.Pdf_Li -> Shape::Pdf(ref, wi): intersect the light's own triangle (shape.cpp:72-87, diffuse.cpp:83-87)
        float lightPdf2 = 0;
        if (EXT && misLightPrim < 0) lightPdf2 = env_pdf_li(sc, sc.lights[-1 - misLightPrim], misWi);  // infinite light: no geometry to test
        Tri lt = load_tri(sc, misLightPrim < 0 ? 0 : misLightPrim);
        float t, lb0, lb1, lb2;
        if (EXT && misLightPrim >= 0 && (lt.flags & PG_PRIM_SPHERE)) {  // Sphere::Pdf, sphere.cpp:292-305
            const PgSphere &ls = sc.spheres[__float_as_int(lt.p0.x)];
            if (!misInside) lightPdf2 = sphere_cone_pdf(ls, misP);
            else if (sphere_test(ls, misRo, misWi, PG_INF, t)) {  // Shape::Pdf, shape.cpp:72-87
                const SphereHit sh = sphere_interaction(ls, misRo, misWi, t);
                float pdf = lensq(misP - sh.p) / (absdot(sh.n, -misWi) * misLightArea);
                if (isinf(pdf)) pdf = 0.f;
                lightPdf2 = pdf;
            }
        } else {
            if (misLightPrim >= 0) ++nLightTests;
            if (misLightPrim >= 0 && tri_test(lt.p0, lt.p1, lt.p2, misRo, misWi, PG_INF, t, lb0, lb1, lb2) && !(lt.flags & PG_TRI_BOGUS)) {
                V3 lp = lt.p0 * lb0 + lt.p1 * lb1 + lt.p2 * lb2;
                V3 ln = normalize(cross(lt.p0 - lt.p2, lt.p1 - lt.p2));
                float pdf = lensq(misP - lp) / (absdot(ln, -misWi) * misLightArea);
                if (isinf(pdf)) pdf = 0.f;
                lightPdf2 = pdf;
            }
        }
        if (lightPdf2 != 0) {
            s_ray[2][0][tid] = make_float4(misRo.x, misRo.y, misRo.z, PG_INF);
            s_ray[2][1][tid] = make_float4(misWi.x, misWi.y, misWi.z, __int_as_float(VOL ? pdi : slot));
            pushMis = true;
            st.pdMis[pdi] = make_float4(misF.r, misF.g, misF.b, misPdf);
            st.pdBeta[pdi].w = power_heuristic(1, misPdf, 1, lightPdf2);
            if constexpr (VOL) vs.trAcc[1][pdi] = make_float4(1, 1, 1, __int_as_float(misMedium));
        }
    }
    const RayQueue outQ[3] = {qnext, qshadow, qmis};
    const bool outPred[3] = {pushNext, pushShadow, pushMis};
    int outPos[3];
    block_push<3, true, PG_SHADE_BLOCK>(outQ, outPred, outPos, nextBin);
    const int posNext = outPos[0], posShadow = outPos[1], posMis = outPos[2];
    if (pushNext) { qnext.o[posNext] = s_ray[0][0][tid]; qnext.d[posNext] = s_ray[0][1][tid]; }
    if (pushShadow) { qshadow.o[posShadow] = s_ray[1][0][tid]; qshadow.d[posShadow] = s_ray[1][1][tid]; }
    if (pushMis) { qmis.o[posMis] = s_ray[2][0][tid]; qmis.d[posMis] = s_ray[2][1][tid]; }
    if (sc.rayTimes) {  // scenes with moving shapes / instances: the spawned rays carry the interaction's time = the ray's (interaction.h:77-95)
        const float t = valid ? PG_QUEUE_TIMES(sc, qin)[i] : 0.f;
        if (pushNext) PG_QUEUE_TIMES(sc, qnext)[posNext] = t;
        if (pushShadow) PG_QUEUE_TIMES(sc, qshadow)[posShadow] = t;
        if (pushMis) PG_QUEUE_TIMES(sc, qmis)[posMis] = t;
    }
    if constexpr (QSTATE) {
        if (valid && !deferred) {
            if (pushNext) {
                const float4 m4 = s_state[2][tid];
                qsOut.L[posNext] = s_state[0][tid]; qsOut.beta[posNext] = s_state[1][tid];
                qsOut.meta[posNext] = make_int4(__float_as_int(m4.x), __float_as_int(m4.y), __float_as_int(m4.z), __float_as_int(m4.w));
                if constexpr (VOL) qsOut.medium[posNext] = newMed;
            } else st.L[slot] = s_state[0][tid];
            st.pdInfo[pdi] = make_int4(posShadow, posMis, lightNum, pushNext ? posNext : ~slot);
        }
    } else if (valid && !deferred && !phaseB) st.pdInfo[slot] = make_int4(posShadow, posMis, lightNum, VOL ? __float_as_int(volWeight) : 0);
    if constexpr (SSS) {
        // the probe rays of the paths that go on through a BSSRDF, region by region like every other queue; such a path's L went to
        // its slot above (where k_resolve adds this vertex's direct lighting), beta and meta follow it there
        if (deferred) pushJob = false;
        int posJob;
        block_push<1, false, PG_SHADE_BLOCK>(&sss.qjob, &pushJob, &posJob);
        if (pushJob) {
            sss.qjob.o[posJob] = s_ray[0][0][tid]; sss.qjob.d[posJob] = s_ray[0][1][tid];
            if (sc.rayTimes) PG_QUEUE_TIMES(sc, sss.qjob)[posJob] = PG_QUEUE_TIMES(sc, qin)[i];  // the probe rays' time: po's = the ray's (bssrdf.cpp:290-296)
            if constexpr (QSTATE) {
                const float4 m4 = s_state[2][tid];
                st.beta[slot] = s_state[1][tid];
                st.meta[slot] = make_int4(__float_as_int(m4.x), __float_as_int(m4.y), __float_as_int(m4.z), __float_as_int(m4.w));
            }
        }
    }
    unsigned long long nl = wave_sum(nLightTests);
    if (lane_id() == 0 && nl) atomicAdd(lightTriTests + (blockIdx.x & (PG_LIGHT_TEST_SHARDS - 1)) * PG_LIGHT_TEST_STRIDE, nl);
    {   // the integrator's statistics (PgCounters, ABI 28).  "Path length" of the paths that end at this vertex: ReportValue(pathLength, bounces), path.cpp:186,
        // with `bounces` of the reference's loop = the count the path arrived with (a scattering vertex has raised the stored one since).  A path that goes on
        // through a BSSRDF reports in k_sss_exit; a grid scene's vertex reports in the phase that finishes it
        bool ended = valid && !deferred && !pushNext && !pushJob;
        if constexpr (GRID) ended = ended && (phaseA ? vertexKind == 0 : vertexKind != 0);
        const int code = valid ? s_stat[tid] : 0;
        stats_path_end(lightTriTests, ended, code & 0x3fff);
        if constexpr (VOL) {  // (a deferred vertex is shaded again by the retry launch)
            stats_count(lightTriTests, PG_STAT_VOLUME, !deferred && (code & 0x4000));
            stats_count(lightTriTests, PG_STAT_SURFACE, !deferred && (code & 0x8000));
        }
    }
}
// Material::ComputeScatteringFunctions ahead of the shading launch, for the main-queue entries whose hit has a material with
// textured parameters: the interaction (as k_shade builds it), what textures read of it (tex_hit_setup), Material::Bump and the
// material's BxDF list, written to rp.matPre at the entry's index; k_shade<3, .> takes them from there and is the BxDF-list kernel
// otherwise.  Why two launches: evaluated inside the shading kernel (k_shade<2, .>) the texture / material evaluators' registers and
// call frames come on top of a path vertex's whole state -- 251 VGPRs, two waves per SIMD, the list of up to 8 x 120 B in scratch --
// and the kernel waits on dependent fetches (material -> texture node -> MIP level -> texels) with little else resident to run.
// Here only the interaction is live across the evaluators, the list is written where it is read from, and the threads take the
// entries in the same material-class order (rp.order).  An entry the shading kernel will not shade as a surface vertex (the path
// is over, a null material, volpath: the ray scattered in its medium first) is skipped when that is known without drawing.
#ifndef PG_MATERIAL_WAVES
#define PG_MATERIAL_WAVES 3
#endif
template <bool VOL>
__global__ __launch_bounds__(PG_SHADE_BLOCK, PG_MATERIAL_WAVES) void k_material(DScene sc, RenderParams rp, PathState st, RayQueue qin, const float4 *__restrict__ hits,
                                                                               VolState vs, const float *__restrict__ hitT, QueueState qsIn) {
    const int p = queue_item<PG_SHADE_BLOCK>(qin);  // the thread's position in the shading order: where its list goes (MatPre)
    if (p < 0) return;
    const int i = rp.order ? rp.order[p] : p;
    const float4 h4 = hits[i];
    const int prim = __float_as_int(h4.x);
    if (prim < 0) return;
    const Tri tri = load_tri(sc, prim);
    if (sc.materials[tri.material].type != PG_MAT_TEXTURED) return;
    const float4 d4 = qin.d[i];
    const int slot = __float_as_int(d4.w);
    const V3 rayD = mk(d4.x, d4.y, d4.z);
    const PgRenderDesc &rd = rp.rd;
    const bool bySlot = VOL && qsIn.L == nullptr;  // volpath scenes with BSSRDF materials or grid media keep their path state by slot
    const int4 meta = bySlot ? st.meta[slot] : qsIn.meta[i];
    if ((meta.w & 0xffff) >= rd.max_depth) return;  // path.cpp:104
    if constexpr (VOL) {
        if (rp.volPre) {  // the medium sample k_shade_order drew: a ray that scatters before the surface never reaches it (volpath.cpp:76-96)
            const int med = bySlot ? vs.medium[slot] : qsIn.medium[i];
            if (med && !(sc.mediaGrid && sc.mediaGrid[med - 1] >= 0) && rp.volPre[i].y / sqrtf(lensq(rayD)) < hitT[i]) return;
        }
    }
    const bool tileSerial = rd.sampler >= PG_SAMPLER_RANDOM && !sc.tsBatched, pixelArrays = sc.tsBatched != 0;
    const uint64_t index = (uint64_t)(uint32_t)meta.x | ((uint64_t)(uint32_t)meta.y << 32);
    Isect is;
    float sphU = 0, sphV = 0;
    V3 sphDpdu = mk(0, 0, 0), sphDpdv = mk(0, 0, 0);
    const bool onSphere = (tri.flags & PG_PRIM_SPHERE) != 0;
    const int inst = sc.hitInst ? sc.hitInst[i] : -1;
    V3 shapeRayD = rayD;
    if (inst >= 0) shapeRayD = m4_vec(inst_w2i(sc, inst, i), rayD);
    if (onSphere) {
        const float4 o4 = qin.o[i];
        V3 shapeRayO = mk(o4.x, o4.y, o4.z);
        if (inst >= 0) { float dt; instance_ray(inst_w2i(sc, inst, i), shapeRayO, rayD, shapeRayO, shapeRayD, dt); }
        const SphereHit sh = sphere_interaction(sc.spheres[__float_as_int(tri.p0.x)], shapeRayO, shapeRayD, h4.y);
        is.p = sh.p; is.pError = sh.pError; is.wo = sh.wo; is.n = sh.n; is.ns = sh.n; is.sdpdu = sh.dpdu;
        is.sdpdv = sh.dpdv; is.sdndu = sh.dndu; is.sdndv = sh.dndv;
        sphU = sh.u; sphV = sh.v; sphDpdu = sh.dpdu; sphDpdv = sh.dpdv;
    } else is = make_isect(sc, prim, tri, h4.y, h4.z, h4.w, shapeRayD);
    if (inst >= 0 && !inst_identity(sc, inst, i)) isect_to_world(inst_i2w(sc, inst, i), inst_w2i(sc, inst, i), is);
    float filmX = 0, filmY = 0;  // the camera sample's pFilm, for the camera ray's differentials
    if (meta.w & PG_META_HASDIFF) { filmX = bySlot ? st.L[slot].w : qsIn.L[i].w; filmY = bySlot ? st.beta[slot].w : qsIn.beta[i].w; }
    TexHit th;
    tex_hit_setup(sc, rd, qin, i, slot, prim, tri, h4, rayD, inst, -1, onSphere, sphU, sphV, sphDpdu, sphDpdv, is, meta, filmX, filmY, tileSerial, pixelArrays, index, th);
    material_bump<1>(sc, tri.material, th, is);
    int nl = 0;
    float etaL = 1;
    // the list in packed records (LobeBsdfT): stride = the records per entry; a mix's lobes take two each
    const bool mix = sc.textured[sc.materials[tri.material].textured_index].kind == PG_KIND_MIX;
    const int plane = qin.regionCap * PG_REGIONS;
    const LobeOutPacked out = {rp.matPre.lobes + (size_t)p * 3, mix ? 2 : 1, plane};
    MatEval<2, 1, LobeOutPacked>::run(*sc.self, tri.material, th, out, nl, etaL, mix ? rp.matPre.stride / 2 : rp.matPre.stride);
    rp.matPre.head[p] = make_float4(is.ns.x, is.ns.y, is.ns.z, etaL);
    rp.matPre.head[(size_t)plane + p] = make_float4(is.sdpdu.x, is.sdpdu.y, is.sdpdu.z, __int_as_float(nl | (mix ? 0x100 : 0)));
}
static void launch_material(const DScene &sc, const RenderParams &rp, PathState st, VolState vs, RayQueue qin, const float4 *hits, const float *hitT, QueueState qi,
                            bool vol, hipStream_t s) {
    const int nblk = PG_REGIONS * (qin.regionCap / PG_SHADE_BLOCK);
    if (nblk == 0) return;
    if (vol) hipLaunchKernelGGL(k_material<true>, dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, vs, hitT, qi);
    else hipLaunchKernelGGL(k_material<false>, dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, vs, hitT, qi);
}
// The order in which a shading launch takes the entries of its queue (RenderParams::order): block b sorts the entry indices of
// window b >> 3 of region b & 7 by the material class of the entry's hit -- counting sort, stable, so that the entries of a class stay
// in queue order and neighbouring lanes still read neighbouring entries.  Window and region are those of the consumer's
// blocks (queue_item): the shading blocks of a window run side by side on the region's XCD and find each other's lines in its L2.
// VOL (volpath, GlobalSamplers, no grid medium): the kernel also draws the medium sample of every entry whose ray is in a medium
// (HomogeneousMedium::Sample's two numbers at the path's current dimension; the shading kernel reads channel and distance from
// volPre instead of drawing them) and gives the entries that scatter inside the medium a class of their own.
template <bool VOL>
__global__ __launch_bounds__(1024) void k_shade_order(DScene sc, RayQueue q, const float4 *__restrict__ hits, int *__restrict__ order,
                                                      PgRenderDesc rd, PathState st, VolState vs, const float *__restrict__ hitT, float2 *__restrict__ volPre, QueueState qs) {
    constexpr int NCHUNK = PG_ORDER_WINDOW / 64, ROUNDS = PG_ORDER_WINDOW / 1024, NW = 1024 / 64;
    static_assert(PG_ORDER_WINDOW % 1024 == 0 && PG_ORDER_CLASSES == 16, "k_shade_order: window of whole blocks, 16 classes");
    const int r = blockIdx.x & (PG_REGIONS - 1), base = (blockIdx.x >> 3) * PG_ORDER_WINDOW;
    const int count = q.counts[r * PG_COUNT_STRIDE];
    if (base >= count) return;
    __shared__ int s_cnt[NCHUNK][PG_ORDER_CLASSES + 1];  // entries of a class in a 64-entry chunk, then their first place in the window
    __shared__ int s_total[PG_ORDER_CLASSES];
    const int lane = lane_id(), wave = threadIdx.x >> 6;
    int cls[ROUNDS], rank[ROUNDS];
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int chunk = k * NW + wave, j = base + chunk * 64 + lane;
        int c = PG_ORDER_CLASSES - 1;
        if (j < count) {
            const size_t e = (size_t)r * q.regionCap + j;
            const int prim = __float_as_int(hits[e].x);
            c = prim < 0 ? PG_ORDER_CLASSES - 2 : (sc.primClass ? sc.primClass[prim] : 0);
            if constexpr (VOL) {
                const float4 d4 = q.d[e];
                const int slot = __float_as_int(d4.w), med = qs.L ? qs.medium[e] : vs.medium[slot];  // (path state in queue order, or by slot)
                if (med && !(sc.mediaGrid && sc.mediaGrid[med - 1] >= 0)) {
                    const int4 meta = qs.L ? qs.meta[e] : st.meta[slot];
                    const uint64_t index = (uint64_t)(uint32_t)meta.x | ((uint64_t)(uint32_t)meta.y << 32);
                    const int dim = (int)((uint32_t)meta.w >> 20);
                    const float uc = halton_sample(sc, rd, index, dim);
                    int channel;
                    float dist;
                    homogeneous_sample_distance(sc.media[med - 1], uc, halton_sample(sc, rd, index, dim + 1), channel, dist);
                    const float tMaxRay = prim >= 0 ? hitT[e] : q.o[e].w;
                    volPre[e] = make_float2(__int_as_float(channel), dist);
                    if (dist / sqrtf(lensq(mk(d4.x, d4.y, d4.z))) < tMaxRay) c = PG_ORDER_CLASSES - 3;  // (only the grouping depends on this)
                }
            }
        }
        unsigned long long mine = 0;
        for (int b = 0; b < PG_ORDER_CLASSES; ++b) {
            const unsigned long long m = __ballot(c == b);
            if (lane == 0) s_cnt[chunk][b] = __popcll(m);
            if (c == b) mine = m;
        }
        cls[k] = c;
        rank[k] = __popcll(mine & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < PG_ORDER_CLASSES) {  // per class: where each chunk's entries start among the class's
        int sum = 0;
        for (int ch = 0; ch < NCHUNK; ++ch) { const int v = s_cnt[ch][threadIdx.x]; s_cnt[ch][threadIdx.x] = sum; sum += v; }
        s_total[threadIdx.x] = sum;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ROUNDS; ++k) {
        const int chunk = k * NW + wave, j = base + chunk * 64 + lane;
        if (j >= count) continue;
        int off = s_cnt[chunk][cls[k]] + rank[k];
        for (int b = 0; b < PG_ORDER_CLASSES; ++b) if (b < cls[k]) off += s_total[b];
        order[(size_t)r * q.regionCap + base + off] = r * q.regionCap + j;
    }
}
void launch_shade_order(const DScene &sc, RayQueue qin, const float4 *hits, int *order, hipStream_t s) {
    const int nblk = PG_REGIONS * ((qin.regionCap + PG_ORDER_WINDOW - 1) / PG_ORDER_WINDOW);
    if (nblk == 0 || !sc.primClass || !order) return;
    hipLaunchKernelGGL(k_shade_order<false>, dim3(nblk), dim3(1024), 0, s, sc, qin, hits, order, PgRenderDesc{}, PathState{}, VolState{}, (const float *)nullptr, (float2 *)nullptr, QueueState{});
}
void launch_shade_order_vol(const DScene &sc, const RenderParams &rp, PathState st, VolState vs, RayQueue qin, const float4 *hits, const float *hitT,
                            int *order, float2 *volPre, hipStream_t s, int cur) {
    const int nblk = PG_REGIONS * ((qin.regionCap + PG_ORDER_WINDOW - 1) / PG_ORDER_WINDOW);
    if (nblk == 0 || !order) return;
    if (volPre) hipLaunchKernelGGL(k_shade_order<true>, dim3(nblk), dim3(1024), 0, s, sc, qin, hits, order, rp.rd, st, vs, hitT, volPre, st.qs[cur]);
    else launch_shade_order(sc, qin, hits, order, s);
}
void launch_shade(const DScene &sc, const RenderParams &rp, PathState st, RayQueue qin, const float4 *hits, RayQueue qnext,
                  RayQueue qshadow, RayQueue qmis, unsigned long long *lightTriTests, hipStream_t s, int cur, const SssState *sss) {
    int nblk = rp.retryCount > 0 ? (rp.retryCount + PG_SHADE_BLOCK - 1) / PG_SHADE_BLOCK : PG_REGIONS * (qin.regionCap / PG_SHADE_BLOCK);
    if (nblk == 0) return;
    const VolState vs = {};
    const float *noT = nullptr;
    const QueueState qi = st.qs[cur], qo = st.qs[cur ^ 1];
    const SssState none = {};
    const GridShade gsh = {nullptr, 0};
    const int mode = pg_shade_mode(sc, rp, false, sss != nullptr, false);  // (the branches below are that function's cases)
    if (sss && sc.nBssrdfs > 0) {  // materials with a BSSRDF are BxDF-list materials: the general kernels
        if (mode == 2) hipLaunchKernelGGL((k_shade<2, false, true>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, noT, qi, qo, *sss, gsh);
        else hipLaunchKernelGGL((k_shade<1, false, true>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, noT, qi, qo, *sss, gsh);
    } else if (mode == 3) {  // materials first (not again for the entries a sparse light table sent back: theirs are there)
        if (rp.retryCount == 0) launch_material(sc, rp, st, vs, qin, hits, noT, qi, false, s);
        hipLaunchKernelGGL((k_shade<3, false>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, noT, qi, qo, none, gsh);
    } else if (mode == 2) hipLaunchKernelGGL((k_shade<2, false>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, noT, qi, qo, none, gsh);
    else if (mode == 1) hipLaunchKernelGGL((k_shade<1, false>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, noT, qi, qo, none, gsh);
    else hipLaunchKernelGGL((k_shade<0, false>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, noT, qi, qo, none, gsh);
}
void launch_shade_vol(const DScene &sc, const RenderParams &rp, PathState st, VolState vs, RayQueue qin, const float4 *hits, const float *hitT,
                      RayQueue qnext, RayQueue qshadow, RayQueue qmis, unsigned long long *lightTriTests, hipStream_t s, const SssState *sss,
                      float4 *gridVertex, int phase, int cur) {
    int nblk = rp.retryCount > 0 ? (rp.retryCount + PG_SHADE_BLOCK - 1) / PG_SHADE_BLOCK : PG_REGIONS * (qin.regionCap / PG_SHADE_BLOCK);
    if (nblk == 0) return;
    const QueueState none = {nullptr, nullptr, nullptr, nullptr};
    const QueueState qi = st.qs[cur], qo = st.qs[cur ^ 1];  // the plain kernels: path state in queue order
    const SssState nosss = {};
    const GridShade gsh = {gridVertex, phase};
    const int mode = pg_shade_mode(sc, rp, true, sss != nullptr, phase != 0);
    if (phase != 0 && sss && sc.nBssrdfs > 0) {  // a grid medium AND BSSRDF materials: the second phase hands a path that goes on through a BSSRDF to the probe chains
        if (mode == 2) hipLaunchKernelGGL((k_shade<2, true, true, true>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, hitT, none, none, *sss, gsh);
        else hipLaunchKernelGGL((k_shade<1, true, true, true>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, hitT, none, none, *sss, gsh);
    } else if (phase != 0) {  // a scene with a grid medium: the two-phase kernels
        if (mode == 2) hipLaunchKernelGGL((k_shade<2, true, false, true>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, hitT, none, none, nosss, gsh);
        else hipLaunchKernelGGL((k_shade<1, true, false, true>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, hitT, none, none, nosss, gsh);
    } else if (sss && sc.nBssrdfs > 0) {
        if (mode == 2) hipLaunchKernelGGL((k_shade<2, true, true>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, hitT, none, none, *sss, gsh);
        else hipLaunchKernelGGL((k_shade<1, true, true>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, hitT, none, none, *sss, gsh);
    } else {
#define PG_LAUNCH_VOL(MODE_) hipLaunchKernelGGL((k_shade<MODE_, true>), dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, qin, hits, qnext, qshadow, qmis, lightTriTests, vs, hitT, qi, qo, nosss, gsh)
        if (mode == 3) {
            if (rp.retryCount == 0) launch_material(sc, rp, st, vs, qin, hits, hitT, qi, true, s);
            PG_LAUNCH_VOL(3);
        } else if (mode == 2) PG_LAUNCH_VOL(2);
        else PG_LAUNCH_VOL(1);
#undef PG_LAUNCH_VOL
    }
}

// EstimateDirect's two "Add ... contribution" steps (integrator.cpp:143-161, 196-212) and
// L += beta * Ld / lightPdf (integrator.cpp:104, path.cpp:122-126), once both rays are back.
template <bool EXT>
__global__ __launch_bounds__(PG_BLOCK) void k_resolve(DScene sc, PathState st, RayQueue qin, RayQueue qmis, const int *__restrict__ occluded,
                                                       const float4 *__restrict__ misHits, QueueState qsNext, unsigned long long *stats) {
    // stats != nullptr: the launch over a bounce's main queue counts "Zero-radiance paths" (path.cpp:119-126): a vertex whose pdInfo names a light
    // (k_shade chose one: its BSDF has non-specular BxDFs) is one of totalPaths, and one of zeroRadiancePaths when its Ld comes out black
    const int i = queue_item<>(qin);
    int4 info = make_int4(-1, -1, -1, 0);
    if (i >= 0) info = st.pdInfo[i];  // the pending terms lie in queue order (PathState)
    const bool counted = stats && i >= 0 && info.z != -1;
    bool black = true;
    if (i >= 0 && !(info.x < 0 && info.y < 0)) {  // (else Ld == 0: L += beta * 0 leaves L unchanged)
    const float4 pl = st.pdLight[i], pm = st.pdMis[i], pb = st.pdBeta[i];
    Spec Ld = sp(0);
    if (info.x >= 0 && !occluded[info.x]) Ld = Ld + sp3(pl.x, pl.y, pl.z);
    if (info.y >= 0) {
        const float4 h = misHits[info.y];
        const int prim = __float_as_int(h.x);
        if (prim >= 0) {
            Tri t = load_tri(sc, prim);
            if (t.light == info.z) {  // lightIsect.primitive->GetAreaLight() == &light
                const PgLight &l = sc.lights[t.light];
                const float4 d4 = qmis.d[info.y];
                V3 wi = mk(d4.x, d4.y, d4.z);
                V3 nrm;
                if (EXT && (t.flags & PG_PRIM_SPHERE)) {
                    const float4 o4 = qmis.o[info.y];
                    nrm = sphere_interaction(sc.spheres[__float_as_int(t.p0.x)], mk(o4.x, o4.y, o4.z), wi, h.y).n;
                } else nrm = hit_normal(sc, prim, t, h.y, h.z, h.w);
                Spec Li = (l.two_sided || dot(nrm, -wi) > 0) ? sp3(l.L[0], l.L[1], l.L[2]) : sp(0);
                if (!is_black(Li)) Ld = Ld + ((((sp3(pm.x, pm.y, pm.z) * Li) * sp(1.f)) * pb.w) / pm.w);
            }
        } else if (EXT && sc.lights[info.z].type == PG_LIGHT_INFINITE) {  // integrator.cpp:207-208: no surface hit: Li = light.Le(ray)
            const float4 d4 = qmis.d[info.y];
            Spec Li = env_le(sc, sc.lights[info.z], mk(d4.x, d4.y, d4.z));
            if (!is_black(Li)) Ld = Ld + ((((sp3(pm.x, pm.y, pm.z) * Li) * sp(1.f)) * pb.w) / pm.w);
        }
    }
    // the path's L: with its ray in the next queue's state, or -- the path is over -- in its slot
    float4 *Lp = info.w >= 0 ? &qsNext.L[info.w] : &st.L[~info.w];
    const float4 L4 = *Lp;
    const Spec add = sp3(pb.x, pb.y, pb.z) * (Ld / pl.w);  // beta * UniformSampleOneLight(...)
    black = is_black(add);
    const Spec L = sp3(L4.x, L4.y, L4.z) + add;
    *Lp = make_float4(L.r, L.g, L.b, L4.w);
    }
    if (stats) {  // (uniform over the launch)
        stats_count(stats, PG_STAT_PATHS, counted);
        stats_count(stats, PG_STAT_PATHS_ZERO, counted && black);
    }
}
void launch_resolve(const DScene &sc, PathState st, RayQueue qin, RayQueue qmis, const int *occluded, const float4 *misHits, hipStream_t s,
                    int cur, unsigned long long *stats) {
    int nblk = PG_REGIONS * (qin.regionCap / PG_BLOCK);
    if (nblk == 0) return;
    if (sc.ext) hipLaunchKernelGGL(k_resolve<true>, dim3(nblk), dim3(PG_BLOCK), 0, s, sc, st, qin, qmis, occluded, misHits, st.qs[cur ^ 1], stats);
    else hipLaunchKernelGGL(k_resolve<false>, dim3(nblk), dim3(PG_BLOCK), 0, s, sc, st, qin, qmis, occluded, misHits, st.qs[cur ^ 1], stats);
}

// ===========================================================================
// VolPathIntegrator's transmittance rays.  VisibilityTester::Tr (light.cpp:63-81) and Scene::IntersectTr (scene.cpp:57-70)
// are loops of Scene::Intersect calls that step through surfaces without a material; here every loop iteration is one
// closest-hit launch over the queue of rays still under way followed by this kernel, which finishes a ray or re-spawns it.
// ===========================================================================
// p, pError and n of the SurfaceInteraction of a closest hit (Triangle::Intersect / Sphere::Intersect, then the instance's
// InterpolatedPrimToWorld), as k_shade builds them
// NEST: the caller can meet hits under two transforms (k_through; the probe chains cannot: scenes with BSSRDF materials and such hits are refused)
template <bool NEST>
PG_DEV void through_point(const DScene &sc, int ri, float4 o4, V3 rayD, float4 h4, int prim, const Tri &tri, V3 &p, V3 &pError, V3 &n) {
    int inst = sc.hitInst ? sc.hitInst[ri] : -1, inst2 = -1;
    if constexpr (NEST) nest_decode(sc, inst, inst2);
    V3 shapeRayD = rayD;
    if (inst >= 0) shapeRayD = m4_vec(inst_w2i(sc, inst, ri), rayD);
    if (inst2 >= 0) shapeRayD = m4_vec(inst_w2i(sc, inst2, ri, true), shapeRayD);
    if (tri.flags & PG_PRIM_SPHERE) {
        V3 shapeRayO = mk(o4.x, o4.y, o4.z);
        if (inst >= 0) { float dt; instance_ray(inst_w2i(sc, inst, ri), shapeRayO, rayD, shapeRayO, shapeRayD, dt); }
        if (inst2 >= 0) { float dt; instance_ray(inst_w2i(sc, inst2, ri, true), shapeRayO, shapeRayD, shapeRayO, shapeRayD, dt); }
        const SphereHit sh = sphere_interaction(sc.spheres[__float_as_int(tri.p0.x)], shapeRayO, shapeRayD, h4.y);
        p = sh.p; pError = sh.pError; n = sh.n;
    } else {
        const Isect is = make_isect(sc, prim, tri, h4.y, h4.z, h4.w, shapeRayD);
        p = is.p; pError = is.pError; n = is.n;
    }
    if (inst2 >= 0 && !inst_identity(sc, inst2, ri, true)) {
        V3 pe;
        p = m4_point_err2(inst_i2w(sc, inst2, ri, true), p, pError, pe);
        pError = pe;
        n = normalize(m4_normal(inst_w2i(sc, inst2, ri, true), n));
    }
    if (inst >= 0 && !inst_identity(sc, inst, ri)) {
        V3 pe;
        p = m4_point_err2(inst_i2w(sc, inst, ri), p, pError, pe);
        pError = pe;
        n = normalize(m4_normal(inst_w2i(sc, inst, ri), n));
    }
}
// GRID: a segment inside a GridDensityMedium is attenuated by ratio tracking (grid.cpp:88-120), whose numbers come from the path's
// sampler: they are drawn here, at the path's current dimension / stream position, which st.meta[slot] keeps for the shading
// kernel's second phase.  The host runs the kind-0 rays of a bounce to their ends before the kind-1 rays start (the reference
// evaluates visibility.Tr before it samples the BSDF: integrator.cpp:146-150, then :164-212).
template <int KIND, bool GRID>
__global__ __launch_bounds__(PG_BLOCK) void k_through(DScene sc, PathState st, VolState vs, RayQueue qin, const float4 *__restrict__ hits,
                                                       const float *__restrict__ hitT, int hitBase, RayQueue qout, RenderParams rp) {
    const int i = queue_item<>(qin);
    bool push = false;
    float4 no = make_float4(0, 0, 0, 0), nd = make_float4(0, 0, 0, 0);
    if (i >= 0) {
        const float4 o4 = qin.o[i], d4 = qin.d[i];
        const int slot = __float_as_int(d4.w);
        const V3 rayD = mk(d4.x, d4.y, d4.z);
        const int ri = hitBase + i;
        const float4 h4 = hits[ri];
        const int prim = __float_as_int(h4.x);
        const bool surface = prim >= 0;
        const float4 acc = vs.trAcc[KIND][slot];
        Spec Tr = sp3(acc.x, acc.y, acc.z);
        int med = __float_as_int(acc.w);
        Tri tri;
        if (surface) tri = load_tri(sc, prim);
        const bool opaque = surface && sc.materials[tri.material].type != PG_MAT_NONE;  // isect.primitive->GetMaterial() != nullptr
        if (KIND == 0 && opaque) Tr = sp(0.f);  // light.cpp:70-72: blocked
        else {
            if (GRID && med && sc.mediaGrid[med - 1] >= 0) {
                const PgDensityGrid &gd = sc.grids[sc.mediaGrid[med - 1]];
                int4 meta = st.meta[slot];
                const uint64_t index = (uint64_t)(uint32_t)meta.x | ((uint64_t)(uint32_t)meta.y << 32);
                int dim = (int)((uint32_t)meta.w >> 20);
                const bool tileSerial = rp.rd.sampler >= PG_SAMPLER_RANDOM;
                auto draw = [&]() -> float { return tileSerial ? ts_get1d(sc, slot) : halton_sample(sc, rp.rd, index, dim++); };
                Tr = Tr * sp(grid_tr(gd, sc.gridDensity + gd.density_offset, mk(o4.x, o4.y, o4.z), rayD, surface ? hitT[ri] : o4.w, draw));
                meta.w = (int)(((uint32_t)dim << 20) | ((uint32_t)meta.w & 0xfffffu));
                st.meta[slot] = meta;
            } else if (med) Tr = Tr * medium_tr(sc.media[med - 1], surface ? hitT[ri] : o4.w, sqrtf(lensq(rayD)));
            if (KIND == 1 && !(surface && !opaque)) {
                // IntersectTr is over (scene.cpp:64-67): the sampled light's radiance along the ray (integrator.cpp:199-208)
                const int lightNum = st.pdInfo[slot].z;
                Spec Li = sp(0);
                if (surface) {
                    if (tri.light == lightNum) {
                        const PgLight &l = sc.lights[tri.light];
                        V3 nrm;
                        if (tri.flags & PG_PRIM_SPHERE) nrm = sphere_interaction(sc.spheres[__float_as_int(tri.p0.x)], mk(o4.x, o4.y, o4.z), rayD, h4.y).n;
                        else nrm = hit_normal(sc, prim, tri, h4.y, h4.z, h4.w);
                        if (l.two_sided || dot(nrm, -rayD) > 0) Li = sp3(l.L[0], l.L[1], l.L[2]);
                    }
                } else if (sc.lights[lightNum].type == PG_LIGHT_INFINITE) Li = env_le(sc, sc.lights[lightNum], rayD);
                vs.misLi[slot] = make_float4(Li.r, Li.g, Li.b, 0);
            } else if (surface) {
                // a surface without a material: step over it (light.cpp:79 isect.SpawnRayTo(p1), scene.cpp:68 isect.SpawnRay(ray.d))
                V3 p, pError, n;
                through_point<true>(sc, ri, o4, rayD, h4, prim, tri, p, pError, n);
                int mIn, mOut;
                prim_interface(sc, prim, med, mIn, mOut);
                V3 origin, d;
                float tMax;
                if (KIND == 0) {
                    const float4 a = vs.p1[0][slot], b = vs.p1[1][slot], c = vs.p1[2][slot];
                    const V3 lp = mk(a.x, a.y, a.z), lpe = mk(b.x, b.y, b.z), ln = mk(c.x, c.y, c.z);
                    origin = offset_ray_origin(p, pError, n, lp - p);  // interaction.h:73-78
                    const V3 target = offset_ray_origin(lp, lpe, ln, origin - lp);
                    d = target - origin;
                    tMax = 1 - PG_SHADOW_EPS;
                } else {
                    origin = offset_ray_origin(p, pError, n, rayD);
                    d = rayD;
                    tMax = PG_INF;
                }
                med = dot(d, n) > 0 ? mOut : mIn;
                no = make_float4(origin.x, origin.y, origin.z, tMax);
                nd = make_float4(d.x, d.y, d.z, __int_as_float(slot));
                push = true;
            }
        }
        vs.trAcc[KIND][slot] = make_float4(Tr.r, Tr.g, Tr.b, __int_as_float(med));
    }
    int pos;
    block_push<1, false>(&qout, &push, &pos);
    if (push) { qout.o[pos] = no; qout.d[pos] = nd; }
    if (push && sc.rayTimes) PG_QUEUE_TIMES(sc, qout)[pos] = PG_QUEUE_TIMES(sc, qin)[i];
}
void launch_through(const DScene &sc, PathState st, VolState vs, int kind, RayQueue qin, const float4 *hits, const float *hitT, int hitBase,
                    RayQueue qout, hipStream_t s, const RenderParams *rpGrid) {
    int nblk = PG_REGIONS * (qin.regionCap / PG_BLOCK);
    if (nblk == 0) return;
    if (rpGrid) {
        if (kind == 0) hipLaunchKernelGGL((k_through<0, true>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, st, vs, qin, hits, hitT, hitBase, qout, *rpGrid);
        else hipLaunchKernelGGL((k_through<1, true>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, st, vs, qin, hits, hitT, hitBase, qout, *rpGrid);
        return;
    }
    const RenderParams none = {};
    if (kind == 0) hipLaunchKernelGGL((k_through<0, false>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, st, vs, qin, hits, hitT, hitBase, qout, none);
    else hipLaunchKernelGGL((k_through<1, false>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, st, vs, qin, hits, hitT, hitBase, qout, none);
}
// EstimateDirect's sums with handleMedia = true (integrator.cpp:143-161, 196-212), once the through rays are finished
// QS: the pending terms lie at the ray's queue position and the path's L with its next ray (queue-order state, PathState) -- else by slot
template <bool QS>
__global__ __launch_bounds__(PG_BLOCK) void k_resolve_vol(DScene sc, PathState st, VolState vs, RayQueue qin, QueueState qsNext) {
    const int i = queue_item<>(qin);
    if (i < 0) return;
    const int slot = QS ? i : __float_as_int(qin.d[i].w);
    const int4 info = st.pdInfo[slot];
    if (info.x < 0 && info.y < 0) return;
    const float4 pl = st.pdLight[slot], pm = st.pdMis[slot], pb = st.pdBeta[slot];
    Spec Ld = sp(0);
    if (info.x >= 0) {
        const float4 li = vs.pdLi[slot], t0 = vs.trAcc[0][slot];
        const Spec Li = sp3(li.x, li.y, li.z) * sp3(t0.x, t0.y, t0.z);  // Li *= visibility.Tr(scene, sampler)
        if (!is_black(Li)) {
            const Spec f = sp3(pl.x, pl.y, pl.z);
            const float w = QS ? vs.p1[0][slot].w : __int_as_float(info.w);
            Ld = Ld + (w < 0 ? (f * Li) / li.w : ((f * Li) * w) / li.w);
        }
    }
    if (info.y >= 0) {
        const float4 ml = vs.misLi[slot], t1 = vs.trAcc[1][slot];
        const Spec Li = sp3(ml.x, ml.y, ml.z);
        if (!is_black(Li)) Ld = Ld + ((((sp3(pm.x, pm.y, pm.z) * Li) * sp3(t1.x, t1.y, t1.z)) * pb.w) / pm.w);
    }
    float4 *Lp = QS ? (info.w >= 0 ? &qsNext.L[info.w] : &st.L[~info.w]) : &st.L[slot];
    float4 L4 = *Lp;
    Spec L = sp3(L4.x, L4.y, L4.z) + sp3(pb.x, pb.y, pb.z) * (Ld / pl.w);
    *Lp = make_float4(L.r, L.g, L.b, L4.w);
}
void launch_resolve_vol(const DScene &sc, PathState st, VolState vs, RayQueue qin, hipStream_t s, int cur) {
    int nblk = PG_REGIONS * (qin.regionCap / PG_BLOCK);
    if (nblk == 0) return;
    if (st.qs[0].L) hipLaunchKernelGGL(k_resolve_vol<true>, dim3(nblk), dim3(PG_BLOCK), 0, s, sc, st, vs, qin, st.qs[cur ^ 1]);
    else hipLaunchKernelGGL(k_resolve_vol<false>, dim3(nblk), dim3(PG_BLOCK), 0, s, sc, st, vs, qin, st.qs[0]);
}
__global__ void k_fill_int(int *p, int value, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = value;
}
void launch_fill_int(int *p, int value, int n, hipStream_t s) {
    if (n > 0) hipLaunchKernelGGL(k_fill_int, dim3((n + 255) / 256), dim3(256), 0, s, p, value, n);
}

// ===========================================================================
// Subsurface scattering: the rest of the BSSRDF branch of PathIntegrator::Li (path.cpp:152-174) behind k_shade<., ., true>.
// SeparableBSSRDF::Sample_Sp (bssrdf.cpp:253-328) intersects the probe segment base -> pTarget again and again from each hit
// (`base = next->si`), keeps every hit on primitives of the entry point's material in a list and picks entry (int)(u1 * nFound).
// Here a chain is walked twice by the traversal kernel -- a list per path has no bound (degenerate soups give hundreds of hits) --:
// pass 1 counts the hits on the material, pass 2 stops at the chosen one.  Each step is one k_trace<0, .> launch over the
// chains still under way followed by k_sss_probe.
// ===========================================================================
// NEST: DScene::hasNest -- the chain's hits can lie under two transforms (a moving shape inside an object definition)
template <int PASS, bool VOL, bool NEST = false>
__global__ __launch_bounds__(PG_BLOCK) void k_sss_probe(DScene sc, SssState sss, RayQueue qin, const float4 *__restrict__ hits, RayQueue qout, int first) {
    const int i = queue_item<>(qin);
    bool push = false;
    float4 no = make_float4(0, 0, 0, 0), nd = make_float4(0, 0, 0, 0);
    if (i >= 0) {
        const float4 o4 = qin.o[i], d4 = qin.d[i];
        const int slot = __float_as_int(d4.w);
        const V3 rayD = mk(d4.x, d4.y, d4.z);
        const float4 h4 = hits[i];
        const int prim = __float_as_int(h4.x);
        int2 cnt = sss.count[slot];
        // volpath: the medium of this probe ray -- base.GetMedium(d) of the previous hit (none for the first ray: Sample_Sp's `base`
        // carries no MediumInterface) -- is what a hit primitive without a transition of its own reports (primitive.cpp:121-125)
        int rayMed = 0;
        if constexpr (VOL) rayMed = first ? 0 : sss.medium[slot].x;
        bool go = prim >= 0;  // bssrdf.cpp:306: no intersection ends the chain
        if (PASS == 2 && cnt.x == 0) go = false;  // nothing to choose from
        if (go) {
            const Tri tri = load_tri(sc, prim);
            if (tri.material == __float_as_int(sss.frame[2][slot].w)) {  // bssrdf.cpp:311: si.primitive->GetMaterial() == this->material
                if (PASS == 1) ++cnt.x;
                else {
                    const float u1 = sss.po[slot].w;
                    int selected = (int)(u1 * cnt.x);  // bssrdf.cpp:321
                    selected = selected < 0 ? 0 : (selected > cnt.x - 1 ? cnt.x - 1 : selected);
                    if (cnt.y == selected) {
                        sss.hit[slot] = h4; sss.hitO[slot] = o4; sss.hitD[slot] = d4;
                        int hInst = sc.hitInst ? sc.hitInst[i] : -1, hInst2 = -1;
                        sss.hitInst[slot] = hInst;  // (NEST: the pair as k_trace wrote it)
                        if constexpr (NEST) nest_decode(sc, hInst, hInst2);
                        if (sss.hitXf && hInst >= 0 && sc.instances[hInst].animated)  // a moving instance: the matrices k_trace interpolated for this probe ray
                            for (int q = 0; q < 33; ++q) sss.hitXf[(size_t)PG_XF_STRIDE * slot + q] = sc.animXf[(size_t)PG_XF_STRIDE * i + q];
                        if (NEST && sss.hitXf && hInst2 >= 0 && sc.instances[hInst2].animated)
                            for (int q = 0; q < 33; ++q) sss.hitXf[(size_t)PG_XF_STRIDE * ((size_t)sss.hitXfNest + slot) + q] = sc.animXf[(size_t)PG_XF_STRIDE * ((size_t)sc.nestXfOff + i) + q];
                        if constexpr (VOL) sss.medium[slot] = make_int2(rayMed, rayMed);
                        go = false;
                    }
                    ++cnt.y;
                }
                sss.count[slot] = cnt;
            }
            if (go) {  // base = next->si; base.SpawnRayTo(pTarget): interaction.h:65-71
                V3 p, pError, n;
                through_point<NEST>(sc, i, o4, rayD, h4, prim, tri, p, pError, n);
                const float4 tg = sss.target[slot];
                const V3 d = mk(tg.x, tg.y, tg.z) - p;
                if (!(d.x == 0 && d.y == 0 && d.z == 0)) {
                    const V3 origin = offset_ray_origin(p, pError, n, d);
                    no = make_float4(origin.x, origin.y, origin.z, 1 - PG_SHADOW_EPS);
                    nd = make_float4(d.x, d.y, d.z, __int_as_float(slot));
                    push = true;
                    if constexpr (VOL) {  // Interaction::GetMedium(d), interaction.h:86-88
                        int mIn, mOut;
                        prim_interface(sc, prim, rayMed, mIn, mOut);
                        sss.medium[slot] = make_int2(dot(d, n) > 0 ? mOut : mIn, 0);
                    }
                }
            }
        }
    }
    int pos;
    block_push<1, false>(&qout, &push, &pos);
    if (push) { qout.o[pos] = no; qout.d[pos] = nd; }
    if (push && sc.rayTimes) PG_QUEUE_TIMES(sc, qout)[pos] = PG_QUEUE_TIMES(sc, qin)[i];
}
void launch_sss_probe(const DScene &sc, SssState sss, int pass, RayQueue qin, const float4 *hits, RayQueue qout, hipStream_t s, bool vol, bool first) {
    int nblk = PG_REGIONS * (qin.regionCap / PG_BLOCK);
    if (nblk == 0) return;
    const int f = first ? 1 : 0;
    if (sc.hasNest) {  // (hits under two transforms: instantiations of their own, the others keep their registers)
        if (vol) {
            if (pass == 1) hipLaunchKernelGGL((k_sss_probe<1, true, true>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, sss, qin, hits, qout, f);
            else hipLaunchKernelGGL((k_sss_probe<2, true, true>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, sss, qin, hits, qout, f);
        } else if (pass == 1) hipLaunchKernelGGL((k_sss_probe<1, false, true>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, sss, qin, hits, qout, f);
        else hipLaunchKernelGGL((k_sss_probe<2, false, true>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, sss, qin, hits, qout, f);
    } else if (vol) {
        if (pass == 1) hipLaunchKernelGGL((k_sss_probe<1, true>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, sss, qin, hits, qout, f);
        else hipLaunchKernelGGL((k_sss_probe<2, true>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, sss, qin, hits, qout, f);
    } else if (pass == 1) hipLaunchKernelGGL((k_sss_probe<1, false>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, sss, qin, hits, qout, f);
    else hipLaunchKernelGGL((k_sss_probe<2, false>), dim3(nblk), dim3(PG_BLOCK), 0, s, sc, sss, qin, hits, qout, f);
}

// The BSDF at a BSSRDF's exit point: one SeparableBSSRDFAdapter (bssrdf.h:209-225), a BxDF of type BSDF_REFLECTION | BSDF_DIFFUSE
// with f = Sw(wi) * eta^2 (TransportMode::Radiance) and BxDF's own cosine-hemisphere Sample_f / Pdf (reflection.cpp:383-394), under
// BSDF::f / ::Pdf / ::Sample_f (reflection.cpp:680-796) for a list of one.
struct AdapterBsdf { V3 ns, ng, ss, ts; float eta; };
PG_DEV V3 ad_to_local(const AdapterBsdf &b, V3 v) { return mk(dot(v, b.ss), dot(v, b.ts), dot(v, b.ns)); }
PG_DEV V3 ad_to_world(const AdapterBsdf &b, V3 v) {
    return mk(b.ss.x * v.x + b.ts.x * v.y + b.ns.x * v.z, b.ss.y * v.x + b.ts.y * v.y + b.ns.y * v.z, b.ss.z * v.x + b.ts.z * v.y + b.ns.z * v.z);
}
PG_DEV Spec ad_lobe_f(const AdapterBsdf &b, V3 wiLocal) { return sp(bssrdf_adapter_f(b.eta, wiLocal.z, fr_dielectric(wiLocal.z, 1, b.eta))); }
PG_DEV Spec ad_f(const AdapterBsdf &b, V3 woW, V3 wiW, int flags) {
    const V3 wi = ad_to_local(b, wiW), wo = ad_to_local(b, woW);
    if (wo.z == 0) return sp(0);
    const int type = PG_BSDF_REFLECTION | PG_BSDF_DIFFUSE;
    const bool reflect = dot(wiW, b.ng) * dot(woW, b.ng) > 0;
    return ((type & flags) == type && reflect) ? ad_lobe_f(b, wi) : sp(0);
}
PG_DEV float ad_pdf(const AdapterBsdf &b, V3 woW, V3 wiW, int flags) {
    const V3 wo = ad_to_local(b, woW), wi = ad_to_local(b, wiW);
    if (wo.z == 0) return 0.f;
    const int type = PG_BSDF_REFLECTION | PG_BSDF_DIFFUSE;
    if ((type & flags) != type) return 0.f;
    return same_hemisphere(wo, wi) ? fabsf(wi.z) * PG_INVPI : 0;
}
PG_DEV Spec ad_sample_f(const AdapterBsdf &b, V3 woW, V3 &wiW, float u0, float u1, float &pdf, int flags, int &sampledType) {
    const int type = PG_BSDF_REFLECTION | PG_BSDF_DIFFUSE;
    sampledType = 0;
    pdf = 0;
    if ((type & flags) != type) return sp(0);  // matchingComps == 0
    const float uR0 = pmin(u0 * 1 - 0, PG_ONE_MINUS_EPS);  // one matching component: comp = 0
    const V3 wo = ad_to_local(b, woW);
    if (wo.z == 0) return sp(0);
    sampledType = type;
    V3 wi = cosine_sample_hemisphere(uR0, u1);
    if (wo.z < 0) wi.z *= -1;
    pdf = same_hemisphere(wo, wi) ? fabsf(wi.z) * PG_INVPI : 0;
    if (pdf == 0) { sampledType = 0; return sp(0); }
    wiW = ad_to_world(b, wi);
    return (dot(wiW, b.ng) * dot(woW, b.ng) > 0) ? ad_lobe_f(b, wi) : sp(0);
}

// The exit vertex pi of a path that went through a BSSRDF (path.cpp:157-174): Sp and its pdf (bssrdf.cpp:322-327), beta *= S / pdf,
// direct lighting at pi under the adapter BSDF (UniformSampleOneLight / EstimateDirect: the shadow and MIS rays and their pending terms,
// indexed by the job's queue position, resolved by k_resolve over the job queue), the next direction, Russian roulette (path.cpp:176-184).
// The path's L, beta and meta wait in their slot (k_shade put them there; k_resolve added the entry vertex's direct lighting to L).
// VOL (volpath.cpp:151-176): UniformSampleOneLight with handleMedia -- the two rays are transmittance rays (k_through), their factors
// stay apart for k_resolve_vol, every pending term and the path's state are indexed by slot -- and the next ray starts in pi.GetMedium(wi).
template <bool VOL>
__global__ __launch_bounds__(PG_SHADE_BLOCK) void k_sss_exit(DScene sc, RenderParams rp, PathState st, SssState sss, RayQueue qnext, RayQueue qshadow,
                                                            RayQueue qmis, unsigned long long *lightTriTests, QueueState qsOut, VolState vs, int phase, float4 *exitVertex) {
    // phase (VOL, a scene with a GridDensityMedium): 1 = up to the exit vertex's direct lighting -- its transmittance rays draw ratio tracking's numbers from the
    // path's sampler BEFORE the next direction is drawn (integrator.cpp:146-150, then path.cpp:165-173), as at k_shade's vertices --, 2 = the next direction and
    // Russian roulette, once those rays are through and k_resolve_vol has added the lighting; exitVertex[slot].w says which jobs phase 1 left alive.  0: all at once.
    const int j = queue_item<PG_SHADE_BLOCK>(sss.qjob);
    __shared__ float4 s_ray[3][2][PG_SHADE_BLOCK];
    const int tid = threadIdx.x;
    bool pushNext = false, pushShadow = false, pushMis = false;
    int slot = 0, lightNum = -1, nextBin = 0;
    unsigned int nLightTests = 0;
    float4 outL = make_float4(0, 0, 0, 0), outB = make_float4(0, 0, 0, 0);
    int4 outM = make_int4(0, 0, 0, 0);
    bool alive = false;
    if (j >= 0) {
        slot = __float_as_int(sss.qjob.d[j].w);
        if (phase != 2) st.pdInfo[VOL ? slot : j] = make_int4(-1, -1, -1, VOL ? 0 : ~slot);
        const int2 cnt = sss.count[slot];
        alive = cnt.x > 0;  // bssrdf.cpp:318: no hit on the material: Sample_Sp returns black and the path ends (its L is in its slot)
        if (phase == 2) alive = __float_as_int(exitVertex[slot].w) == 1;
    }
    const int pdi = VOL ? slot : j;  // index of this vertex's pending direct-light terms
    float volWeight = 0;
    if (alive) {
        const PgRenderDesc &rd = rp.rd;
        const float4 L4 = st.L[slot], B4 = st.beta[slot];
        const int4 meta = st.meta[slot];
        Spec L = sp3(L4.x, L4.y, L4.z), beta = sp3(B4.x, B4.y, B4.z);
        const uint64_t index = (uint64_t)(uint32_t)meta.x | ((uint64_t)(uint32_t)meta.y << 32);
        int dim = (int)((uint32_t)meta.w >> 20);
        const float etaScale = __int_as_float(meta.z);
        const int bounces = meta.w & 0xffff;  // already counts this vertex (k_shade)
        const bool tileSerial = rd.sampler >= PG_SAMPLER_RANDOM;
        auto draw1 = [&]() -> float { return tileSerial ? ts_get1d(sc, slot) : halton_sample(sc, rd, index, dim++); };
        auto draw2 = [&](float &a, float &b) {
            if (tileSerial) ts_get2d(sc, rd.sampler, slot, a, b);
            else { a = halton_sample(sc, rd, index, dim); b = halton_sample(sc, rd, index, dim + 1); dim += 2; }
        };
        // ---- pi: the chosen hit as a SurfaceInteraction (the tail of Triangle::Intersect / Sphere::Intersect, then the instance's transform)
        const float4 h4 = sss.hit[slot], o4 = sss.hitO[slot], d4 = sss.hitD[slot];
        const int prim = __float_as_int(h4.x);
        int inst = sss.hitInst[slot], inst2;
        nest_decode(sc, inst, inst2);  // (DScene::hasNest: the chosen hit may lie under two transforms)
        const float *i2wIn = nullptr, *w2iIn = nullptr;  // the inner TransformedPrimitive's, as below
        bool innerIdentity = true;
        if (inst2 >= 0) {
            const PgInstance &in2 = sc.instances[inst2];
            const bool moving2 = in2.animated && sss.hitXf;
            const float *xf2 = sss.hitXf + (size_t)PG_XF_STRIDE * ((size_t)sss.hitXfNest + slot);
            i2wIn = moving2 ? xf2 : in2.i2w;
            w2iIn = moving2 ? xf2 + 16 : in2.w2i;
            innerIdentity = moving2 ? xf2[32] != 0.f : in2.identity != 0;
        }
        // the chosen hit's instance transform: the instance's own, or -- a moving instance -- what k_sss_probe kept of k_trace's interpolation
        const float *i2w = nullptr, *w2i = nullptr;
        bool instIdentity = true;
        if (inst >= 0) {
            const PgInstance &in = sc.instances[inst];
            const bool moving = in.animated && sss.hitXf;
            i2w = moving ? sss.hitXf + (size_t)PG_XF_STRIDE * slot : in.i2w;
            w2i = moving ? sss.hitXf + (size_t)PG_XF_STRIDE * slot + 16 : in.w2i;
            instIdentity = moving ? sss.hitXf[(size_t)PG_XF_STRIDE * slot + 32] != 0.f : in.identity != 0;
        }
        const V3 rayD = mk(d4.x, d4.y, d4.z);
        const Tri tri = load_tri(sc, prim);
        Isect is;
        V3 shapeRayD = rayD;
        if (inst >= 0) shapeRayD = m4_vec(w2i, rayD);
        if (inst2 >= 0) shapeRayD = m4_vec(w2iIn, shapeRayD);
        if (tri.flags & PG_PRIM_SPHERE) {
            V3 shapeRayO = mk(o4.x, o4.y, o4.z);
            if (inst >= 0) { float dt; instance_ray(w2i, shapeRayO, rayD, shapeRayO, shapeRayD, dt); }
            if (inst2 >= 0) { float dt; instance_ray(w2iIn, shapeRayO, shapeRayD, shapeRayO, shapeRayD, dt); }
            const SphereHit sh = sphere_interaction(sc.spheres[__float_as_int(tri.p0.x)], shapeRayO, shapeRayD, h4.y);
            is.p = sh.p; is.pError = sh.pError; is.wo = sh.wo; is.n = sh.n; is.ns = sh.n; is.sdpdu = sh.dpdu;
            is.sdpdv = sh.dpdv; is.sdndu = sh.dndu; is.sdndv = sh.dndv;
        } else is = make_isect(sc, prim, tri, h4.y, h4.z, h4.w, shapeRayD);
        if (inst2 >= 0 && !innerIdentity) isect_to_world(i2wIn, w2iIn, is);  // (the inner transform first)
        if (inst >= 0 && !instIdentity) {  // InterpolatedPrimToWorld(*isect), transform.cpp:262-297
            Isect w;
            w.p = m4_point_err2(i2w, is.p, is.pError, w.pError);
            w.n = normalize(m4_normal(w2i, is.n));
            w.wo = normalize(m4_vec(i2w, is.wo));
            w.sdpdu = m4_vec(i2w, is.sdpdu);
            w.sdpdv = m4_vec(i2w, is.sdpdv);
            w.sdndu = m4_normal(w2i, is.sdndu); w.sdndv = m4_normal(w2i, is.sdndv);
            w.ns = normalize(m4_normal(w2i, is.ns));
            if (dot(w.ns, w.n) < 0.f) w.ns = -w.ns;
            is = w;
        }
        // ---- Sp(pi) = Sr(|po - pi|) and Pdf_Sp(pi) / nFound (bssrdf.cpp:322-327), beta *= S / pdf (path.cpp:158)
        const float4 po4 = sss.po[slot], f0 = sss.frame[0][slot], f1 = sss.frame[1][slot], f2 = sss.frame[2][slot];
        const V3 poP = mk(po4.x, po4.y, po4.z);
        DBssrdf b = bssrdf_bind(sc.bssrdfs[__float_as_int(f1.w)], sc.bssrdfTables);
        const float4 cs = sss.coef[0][slot], cr = sss.coef[1][slot];
        b.sigma_t[0] = cs.x; b.sigma_t[1] = cs.y; b.sigma_t[2] = cs.z; b.rho[0] = cr.x; b.rho[1] = cr.y; b.rho[2] = cr.z;
        const int nFound = sss.count[slot].x;
        bool through = true;  // S and its pdf let the path through (phase 2: phase 1 found that, and beta has the factor)
        if (phase != 2) {
            const float pdf = bssrdf_pdf_sp(b, mk(f1.x, f1.y, f1.z), mk(f2.x, f2.y, f2.z), mk(f0.x, f0.y, f0.z), poP, is.p, is.n) / nFound;
            const Spec S = bssrdf_sr(b, sqrtf(lensq(poP - is.p)));
            through = !(is_black(S) || pdf == 0);
            if (through) beta = beta * (S / pdf);
        }
        if (!through) alive = false;
        else {
            // Sample_S's BSDF at pi (bssrdf.cpp:243-248): shading frame of pi, the adapter; pi.wo = pi.shading.n
            AdapterBsdf ab;
            ab.ns = is.ns; ab.ng = is.n; ab.ss = normalize(is.sdpdu); ab.ts = cross(ab.ns, ab.ss); ab.eta = f0.w;
            is.wo = is.ns;
            const int nonSpecular = PG_BSDF_ALL & ~PG_BSDF_SPECULAR;
            int mIn = 0, mOut = 0;  // VOL: pi's MediumInterface (the primitive's own, or the medium of the probe ray that found it)
            if constexpr (VOL) prim_interface(sc, prim, sss.medium[slot].y, mIn, mOut);
            // ---- L += beta * UniformSampleOneLight(pi, ...) (path.cpp:161-163; integrator.cpp:85-215)
            if (phase != 2) {
            const float *tab = sc.nLights > 0 ? light_distribution(sc, is.p) : nullptr;
            bool misCand = false;
            V3 misRo = mk(0, 0, 0), misWi = mk(0, 0, 1);
            Spec misF = sp(0);
            float misPdf = 0, misLightArea = 1;
            int misLightPrim = 0;
            bool misInside = false;
            if (tab) {
                float lightSelPdf;
                lightNum = sample_discrete(tab, sc.nLights, draw1(), lightSelPdf);
                if (lightSelPdf != 0) {
                    float uL0, uL1, uS0, uS1;
                    draw2(uL0, uL1);
                    draw2(uS0, uS1);
                    const LightHot lh = load_light_hot(sc, lightNum);
                    const PgLight &light = sc.lights[lightNum];
                    V3 wi = mk(0, 0, 0);
                    float lightPdf = 0, scatteringPdf = 0;
                    float4 pdLight = make_float4(0, 0, 0, 0);
                    LightSample ls;
                    const Spec Li = light_sample_li_hot<true>(sc, lh, light, is.p, is.pError, is.n, uL0, uL1, wi, lightPdf, ls);
                    if (lightPdf > 0 && !is_black(Li)) {
                        const Spec f = ad_f(ab, is.wo, wi, nonSpecular) * absdot(wi, ab.ns);
                        scatteringPdf = ad_pdf(ab, is.wo, wi, nonSpecular);
                        if (!is_black(f)) {
                            const V3 origin = offset_ray_origin(is.p, is.pError, is.n, ls.p - is.p);
                            const V3 target = offset_ray_origin(ls.p, ls.pError, ls.n, origin - ls.p);
                            const V3 shD = target - origin;
                            s_ray[1][0][tid] = make_float4(origin.x, origin.y, origin.z, 1 - PG_SHADOW_EPS);
                            s_ray[1][1][tid] = make_float4(shD.x, shD.y, shD.z, __int_as_float(slot));
                            pushShadow = true;
                            const bool isDelta = PG_LIGHT_IS_DELTA(lh.type);
                            if constexpr (VOL) {
                                volWeight = isDelta ? -1.f : power_heuristic(1, lightPdf, 1, scatteringPdf);
                                pdLight = make_float4(f.r, f.g, f.b, 0);
                                vs.pdLi[slot] = make_float4(Li.r, Li.g, Li.b, lightPdf);
                                vs.p1[0][slot] = make_float4(ls.p.x, ls.p.y, ls.p.z, 0); vs.p1[1][slot] = make_float4(ls.pError.x, ls.pError.y, ls.pError.z, 0);
                                vs.p1[2][slot] = make_float4(ls.n.x, ls.n.y, ls.n.z, 0);
                                vs.trAcc[0][slot] = make_float4(1, 1, 1, __int_as_float(dot(shD, is.n) > 0 ? mOut : mIn));
                            } else {
                                const Spec c = isDelta ? (f * Li) / lightPdf : ((f * Li) * power_heuristic(1, lightPdf, 1, scatteringPdf)) / lightPdf;
                                pdLight = make_float4(c.r, c.g, c.b, 0);
                            }
                        }
                    }
                    V3 wi2 = wi;
                    float sPdf2 = 0;
                    Spec fm = sp(0);
                    if (lh.type == PG_LIGHT_AREA || lh.type == PG_LIGHT_INFINITE) {
                        int st2;
                        fm = ad_sample_f(ab, is.wo, wi2, uS0, uS1, sPdf2, nonSpecular, st2);
                        fm = fm * absdot(wi2, ab.ns);
                    }
                    if (!is_black(fm) && sPdf2 > 0) {
                        misCand = true;
                        spawn_ray(is, wi2, misRo);
                        misWi = wi2; misF = fm; misPdf = sPdf2;
                        misLightPrim = lh.type == PG_LIGHT_INFINITE ? -1 - lightNum : lh.prim; misLightArea = lh.area;
                        if (lh.type == PG_LIGHT_AREA && (lh.tri.flags & PG_PRIM_SPHERE)) {
                            const PgSphere &lsph = sc.spheres[__float_as_int(lh.tri.p0.x)];
                            misInside = lsph.shape != PG_SHAPE_SPHERE || sphere_ref_inside(lsph, is.p, is.pError, is.n);
                        }
                    }
                    pdLight.w = lightSelPdf;
                    st.pdLight[pdi] = pdLight;
                    st.pdBeta[pdi] = make_float4(beta.r, beta.g, beta.b, 0.f);
                }
            }
            int misMedium = 0;
            if (VOL && misCand) misMedium = dot(misWi, is.n) > 0 ? mOut : mIn;
            if (misCand) {  // light.Pdf_Li(pi, wi) of the BSDF-sampled direction (integrator.cpp:175-178): as at the end of k_shade
                float lightPdf2 = 0;
                if (misLightPrim < 0) lightPdf2 = env_pdf_li(sc, sc.lights[-1 - misLightPrim], misWi);
                const Tri lt = load_tri(sc, misLightPrim < 0 ? 0 : misLightPrim);
                float t, lb0, lb1, lb2;
                if (misLightPrim >= 0 && (lt.flags & PG_PRIM_SPHERE)) {
                    const PgSphere &lsp = sc.spheres[__float_as_int(lt.p0.x)];
                    if (!misInside) lightPdf2 = sphere_cone_pdf(lsp, is.p);
                    else if (sphere_test(lsp, misRo, misWi, PG_INF, t)) {
                        const SphereHit sh = sphere_interaction(lsp, misRo, misWi, t);
                        float pdf2 = lensq(is.p - sh.p) / (absdot(sh.n, -misWi) * misLightArea);
                        if (isinf(pdf2)) pdf2 = 0.f;
                        lightPdf2 = pdf2;
                    }
                } else {
                    if (misLightPrim >= 0) ++nLightTests;
                    if (misLightPrim >= 0 && tri_test(lt.p0, lt.p1, lt.p2, misRo, misWi, PG_INF, t, lb0, lb1, lb2) && !(lt.flags & PG_TRI_BOGUS)) {
                        const V3 lp = lt.p0 * lb0 + lt.p1 * lb1 + lt.p2 * lb2;
                        const V3 ln = normalize(cross(lt.p0 - lt.p2, lt.p1 - lt.p2));
                        float pdf2 = lensq(is.p - lp) / (absdot(ln, -misWi) * misLightArea);
                        if (isinf(pdf2)) pdf2 = 0.f;
                        lightPdf2 = pdf2;
                    }
                }
                if (lightPdf2 != 0) {
                    s_ray[2][0][tid] = make_float4(misRo.x, misRo.y, misRo.z, PG_INF);
                    s_ray[2][1][tid] = make_float4(misWi.x, misWi.y, misWi.z, __int_as_float(slot));
                    pushMis = true;
                    st.pdMis[pdi] = make_float4(misF.r, misF.g, misF.b, misPdf);
                    st.pdBeta[pdi].w = power_heuristic(1, misPdf, 1, lightPdf2);
                    if constexpr (VOL) vs.trAcc[1][slot] = make_float4(1, 1, 1, __int_as_float(misMedium));
                }
            }
            }
            int newFlags = phase == 1 ? (meta.w & 0xf0000) : 0;  // (phase 1 leaves the flags as they came)
            if (phase != 1) {
            // ---- indirect illumination from pi (path.cpp:165-173), then Russian roulette (:176-184)
            V3 wi;
            float pdf2, u0, u1;
            draw2(u0, u1);
            int sampledType = 0;
            const Spec f = ad_sample_f(ab, is.wo, wi, u0, u1, pdf2, PG_BSDF_ALL, sampledType);
            if (!(is_black(f) || pdf2 == 0.f)) {
                beta = beta * ((f * absdot(wi, ab.ns)) / pdf2);
                if (sampledType & PG_BSDF_SPECULAR) newFlags |= PG_META_SPECULAR;
                V3 nextO;
                spawn_ray(is, wi, nextO);
                s_ray[0][0][tid] = make_float4(nextO.x, nextO.y, nextO.z, PG_INF);
                s_ray[0][1][tid] = make_float4(wi.x, wi.y, wi.z, __int_as_float(slot));
                nextBin = (wi.x < 0 ? 1 : 0) | (wi.y < 0 ? 2 : 0) | (wi.z < 0 ? 4 : 0);
                pushNext = true;
                if constexpr (VOL) vs.medium[slot] = dot(wi, is.n) > 0 ? mOut : mIn;
                const Spec rrBeta = beta * etaScale;
                if (max_component(rrBeta) < rd.rr_threshold && bounces - 1 > 3) {  // (`bounces` of the reference's loop: this vertex's, before k_shade's increment)
                    const float qq = pmax(.05f, 1 - max_component(rrBeta));
                    if (draw1() < qq) pushNext = false;
                    else beta = beta / (1 - qq);
                }
            }
            }
            outL = make_float4(L.r, L.g, L.b, L4.w);
            outB = make_float4(beta.r, beta.g, beta.b, B4.w);
            outM = make_int4(meta.x, meta.y, __float_as_int(etaScale), (dim << 20) | bounces | newFlags);
        }
    }
    const RayQueue outQ[3] = {qnext, qshadow, qmis};
    const bool outPred[3] = {pushNext, pushShadow, pushMis};
    int outPos[3];
    block_push<3, true, PG_SHADE_BLOCK>(outQ, outPred, outPos, nextBin);
    const int posNext = outPos[0], posShadow = outPos[1], posMis = outPos[2];
    if (pushNext) { qnext.o[posNext] = s_ray[0][0][tid]; qnext.d[posNext] = s_ray[0][1][tid]; }
    if (pushShadow) { qshadow.o[posShadow] = s_ray[1][0][tid]; qshadow.d[posShadow] = s_ray[1][1][tid]; }
    if (pushMis) { qmis.o[posMis] = s_ray[2][0][tid]; qmis.d[posMis] = s_ray[2][1][tid]; }
    if (sc.rayTimes) {  // the rays leaving pi carry the path's time (the job's probe ray has it)
        const float t = j >= 0 ? PG_QUEUE_TIMES(sc, sss.qjob)[j] : 0.f;
        if (pushNext) PG_QUEUE_TIMES(sc, qnext)[posNext] = t;
        if (pushShadow) PG_QUEUE_TIMES(sc, qshadow)[posShadow] = t;
        if (pushMis) PG_QUEUE_TIMES(sc, qmis)[posMis] = t;
    }
    if (alive) {
        if constexpr (VOL) {  // by slot: L is there already (k_resolve_vol adds to it), beta and meta follow
            st.beta[slot] = outB; st.meta[slot] = outM;
            if (phase != 2) st.pdInfo[slot] = make_int4(posShadow, posMis, lightNum, __float_as_int(volWeight));
        } else {
            if (pushNext) { qsOut.L[posNext] = outL; qsOut.beta[posNext] = outB; qsOut.meta[posNext] = outM; }  // (L itself already is in the slot)
            st.pdInfo[j] = make_int4(posShadow, posMis, lightNum, pushNext ? posNext : ~slot);
        }
    }
    unsigned long long nl = wave_sum(nLightTests);
    if (lane_id() == 0 && nl) atomicAdd(lightTriTests + (blockIdx.x & (PG_LIGHT_TEST_SHARDS - 1)) * PG_LIGHT_TEST_STRIDE, nl);
    // "Path length" of the paths that end inside the BSSRDF branch (path.cpp:157-174: no probe hit, a black S or f, Russian roulette): the
    // reference's `bounces` is the entry vertex's, the stored count less k_shade's increment
    // (two phases: a job reports where it ends -- in phase 1 when the exit vertex did not come about, in phase 2 otherwise)
    if (VOL && phase == 1 && j >= 0) exitVertex[slot] = make_float4(0, 0, 0, __int_as_float(alive ? 1 : 0));
    stats_path_end(lightTriTests, j >= 0 && !pushNext && (phase == 0 || (phase == 1) != alive), j >= 0 ? (st.meta[slot].w & 0xffff) - 1 : 0);
}
void launch_sss_exit(const DScene &sc, const RenderParams &rp, PathState st, SssState sss, RayQueue qnext, RayQueue qshadow, RayQueue qmis,
                     unsigned long long *lightTriTests, hipStream_t s, int nxt, bool vol, VolState vs, int phase, float4 *exitVertex) {
    int nblk = PG_REGIONS * (sss.qjob.regionCap / PG_SHADE_BLOCK);
    if (nblk == 0) return;
    if (vol) hipLaunchKernelGGL(k_sss_exit<true>, dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, sss, qnext, qshadow, qmis, lightTriTests, st.qs[nxt], vs, phase, exitVertex);
    else hipLaunchKernelGGL(k_sss_exit<false>, dim3(nblk), dim3(PG_SHADE_BLOCK), 0, s, sc, rp, st, sss, qnext, qshadow, qmis, lightTriTests, st.qs[nxt], vs, 0, (float4 *)nullptr);
}

// ===========================================================================
// Film: one lane per pixel of the batch's tiles; samples are added in sample
// order so a pixel's sum is the reference's (integrator.cpp:276-325).
// ===========================================================================
__global__ __launch_bounds__(PG_BLOCK) void k_film(RenderParams rp, PathState st, PgFilmPixel *film, PgStraySample *strays, int maxStrays,
                                                    int *nStrays) {
    const int tileInBatch = blockIdx.x;
    const int pix = threadIdx.x;
    const PgRenderDesc &rd = rp.rd;
    int px, py, sn;
    const int slot0 = (tileInBatch * rp.sCount) * 256 + pix;
    if (!slot_to_pixel(rp, slot0, px, py, sn)) return;
    const int local = rp.tileLocal0 + tileInBatch;
    const int t = rd.tile_first + local * rd.tile_step;
    const int tx = t % rp.nTilesX, ty = t / rp.nTilesX;
    const int x0 = rd.sample_bounds[0] + tx * 16, y0 = rd.sample_bounds[1] + ty * 16;
    const int x1 = min(x0 + 16, rd.sample_bounds[2]), y1 = min(y0 + 16, rd.sample_bounds[3]);
    const float frx = rd.filter_radius[0], fry = rd.filter_radius[1];
    // FilmTile pixel bounds, film.cpp:95-106
    const int tp0x = max((int)ceilf((float)x0 - 0.5f - frx), rd.cropped_pixel_bounds[0]);
    const int tp0y = max((int)ceilf((float)y0 - 0.5f - fry), rd.cropped_pixel_bounds[1]);
    const int tp1x = min((int)floorf((float)x1 - 0.5f + frx) + 1, rd.cropped_pixel_bounds[2]);
    const int tp1y = min((int)floorf((float)y1 - 0.5f + fry) + 1, rd.cropped_pixel_bounds[3]);
    PgFilmPixel *fp = &film[(size_t)local * 256 + pix];
    float r = fp->rgb[0], g = fp->rgb[1], b = fp->rgb[2], w = fp->weight;
    for (int sIdx = 0; sIdx < rp.sCount; ++sIdx) {
        const int slot = (tileInBatch * rp.sCount + sIdx) * 256 + pix;
        const float4 L4 = st.L[slot];
        const float pFilmX = L4.w, pFilmY = st.beta[slot].w;
        Spec L = sp3(L4.x, L4.y, L4.z);
        // integrator.cpp:294-315
        if (isnan(L.r) || isnan(L.g) || isnan(L.b)) L = sp(0);
        else if ((double)lum(L) < -1e-5) L = sp(0);
        else if (isinf(lum(L))) L = sp(0);
        // FilmTile::AddSample, film.h:121-161; box filter => every filterTable entry is 1, sampleWeight = rayWeight = 1
        if (lum(L) > rd.max_sample_luminance) L = L * (rd.max_sample_luminance / lum(L));
        const float dx = pFilmX - 0.5f, dy = pFilmY - 0.5f;
        const int p0x = max((int)ceilf(dx - frx), tp0x), p0y = max((int)ceilf(dy - fry), tp0y);
        const int p1x = min((int)floorf(dx + frx) + 1, tp1x), p1y = min((int)floorf(dy + fry) + 1, tp1y);
        Spec c = (L * 1.f) * 1.f;
        for (int y = p0y; y < p1y; ++y)
            for (int x = p0x; x < p1x; ++x) {
                if (x == px && y == py) { r += c.r; g += c.g; b += c.b; w += 1.f; }
                else {
                    int k = atomicAdd(nStrays, 1);
                    if (k < maxStrays) {
                        PgStraySample s;
                        s.px = x; s.py = y; s.src_px = px; s.src_py = py;
                        s.rgb[0] = c.r; s.rgb[1] = c.g; s.rgb[2] = c.b; s.weight = 1.f;
                        strays[k] = s;
                    }
                }
            }
    }
    fp->rgb[0] = r; fp->rgb[1] = g; fp->rgb[2] = b; fp->weight = w;
}
// ---------------------------------------------------------------------------------------------------------------
// Film for every other pixel filter (gaussian, mitchell, sinc, triangle, wide box): FilmTile::AddSample with the 16x16
// filter table (film.h:121-161).  A sample now lands in every pixel within the filter radius, including pixels owned by
// neighbouring tiles, so a tile's block is its FilmTile pixel bounds (16 + halo), and the kernel GATHERS: one lane per
// block pixel walks the tile's samples in the reference's order (pixels row-major, samples in order) and adds those whose
// footprint covers it.  Deterministic, no atomics, and a pixel's tile-local sum has the reference's summation order;
// the host merges the overlapping blocks in tile order (Film::MergeFilmTile).  All samples of a tile are in one batch.
__global__ __launch_bounds__(PG_BLOCK) void k_film_general(RenderParams rp, PathState st, PgFilmPixel *film) {
    const int tileInBatch = blockIdx.x;
    const PgRenderDesc &rd = rp.rd;
    const int local = rp.tileLocal0 + tileInBatch;
    const int t = rd.tile_first + local * rd.tile_step;
    const int tx = t % rp.nTilesX, ty = t / rp.nTilesX;
    const int x0 = rd.sample_bounds[0] + tx * 16, y0 = rd.sample_bounds[1] + ty * 16;
    const int x1 = min(x0 + 16, rd.sample_bounds[2]), y1 = min(y0 + 16, rd.sample_bounds[3]);
    const float frx = rd.filter_radius[0], fry = rd.filter_radius[1];
    const float invRx = 1 / frx, invRy = 1 / fry;  // FilmTile::invFilterRadius, film.h:113
    // FilmTile pixel bounds, film.cpp:95-106
    const int tp0x = max((int)ceilf((float)x0 - 0.5f - frx), rd.cropped_pixel_bounds[0]);
    const int tp0y = max((int)ceilf((float)y0 - 0.5f - fry), rd.cropped_pixel_bounds[1]);
    const int tp1x = min((int)floorf((float)x1 - 0.5f + frx) + 1, rd.cropped_pixel_bounds[2]);
    const int tp1y = min((int)floorf((float)y1 - 0.5f + fry) + 1, rd.cropped_pixel_bounds[3]);
    const int tw = 16 + rd.tile_halo[0] + rd.tile_halo[2];
    // source pixels that can reach a pixel: a sample of pixel p has pFilmDiscrete in [p - 0.5, p + 0.5] and covers the pixels
    // ceil(d - r) .. floor(d + r); one more than floor(0.5 + r) against the rounding of those sums -- which a radius <= 0.5 (the box
    // filter on this path: frames whose film positions can round onto the next pixel) cannot need: p + 0.5 + r <= p + 1 is exact
    const int reachX = frx <= 0.5f ? 1 : (int)floorf(0.5f + frx) + 1, reachY = fry <= 0.5f ? 1 : (int)floorf(0.5f + fry) + 1;
    // Radius <= 0.5 (the box filter): almost every sample stays in its own pixel.  Each source pixel's samples are looked at once
    // first -- does one of them cover a pixel besides its own? --, and the gather below visits the samples of OTHER pixels only
    // where that is so (the frame of config 3 at 68 spp: 6.45 ms of film kernel for nine source pixels per pixel)
    __shared__ unsigned char s_reaches[256];
    const bool narrow = frx <= 0.5f && fry <= 0.5f;
    if (narrow) {
        for (int pix = threadIdx.x; pix < 256; pix += PG_BLOCK) {
            const int px = x0 + (pix & 15), py = y0 + (pix >> 4);
            bool reaches = false;
            if (px < x1 && py < y1 && !(px < rd.pixel_bounds[0] || px >= rd.pixel_bounds[2] || py < rd.pixel_bounds[1] || py >= rd.pixel_bounds[3]))
                for (int sIdx = 0; sIdx < rp.sCount && !reaches; ++sIdx) {
                    const int slot = (tileInBatch * rp.sCount + sIdx) * 256 + pix;
                    const float dx = st.L[slot].w - 0.5f, dy = st.beta[slot].w - 0.5f;  // pFilmDiscrete
                    const int p0x = max((int)ceilf(dx - frx), tp0x), p1x = min((int)floorf(dx + frx) + 1, tp1x);
                    const int p0y = max((int)ceilf(dy - fry), tp0y), p1y = min((int)floorf(dy + fry) + 1, tp1y);
                    reaches = p0x < px || p1x > px + 1 || p0y < py || p1y > py + 1;
                }
            s_reaches[pix] = reaches ? 1 : 0;
        }
        __syncthreads();
    }
    for (int e = threadIdx.x; e < rd.tile_pixels; e += PG_BLOCK) {
        const int X = x0 - rd.tile_halo[0] + e % tw, Y = y0 - rd.tile_halo[1] + e / tw;
        if (X < tp0x || X >= tp1x || Y < tp0y || Y >= tp1y) continue;
        float r = 0, g = 0, b = 0, w = 0;
        for (int py = max(y0, Y - reachY); py <= min(y1 - 1, Y + reachY); ++py)
            for (int px = max(x0, X - reachX); px <= min(x1 - 1, X + reachX); ++px) {
                // InsideExclusive(pixel, pixelBounds), integrator.cpp:273
                if (px < rd.pixel_bounds[0] || px >= rd.pixel_bounds[2] || py < rd.pixel_bounds[1] || py >= rd.pixel_bounds[3]) continue;
                const int pix = (py - y0) * 16 + (px - x0);
                if (narrow && !(px == X && py == Y) && !s_reaches[pix]) continue;  // none of that pixel's samples leaves it
                for (int sIdx = 0; sIdx < rp.sCount; ++sIdx) {
                    const int slot = (tileInBatch * rp.sCount + sIdx) * 256 + pix;
                    const float4 L4 = st.L[slot];
                    const float dx = L4.w - 0.5f, dy = st.beta[slot].w - 0.5f;  // pFilmDiscrete
                    const int p0x = max((int)ceilf(dx - frx), tp0x), p1x = min((int)floorf(dx + frx) + 1, tp1x);
                    if (X < p0x || X >= p1x) continue;
                    const int p0y = max((int)ceilf(dy - fry), tp0y), p1y = min((int)floorf(dy + fry) + 1, tp1y);
                    if (Y < p0y || Y >= p1y) continue;
                    Spec L = sp3(L4.x, L4.y, L4.z);
                    // integrator.cpp:294-315
                    if (isnan(L.r) || isnan(L.g) || isnan(L.b)) L = sp(0);
                    else if ((double)lum(L) < -1e-5) L = sp(0);
                    else if (isinf(lum(L))) L = sp(0);
                    if (lum(L) > rd.max_sample_luminance) L = L * (rd.max_sample_luminance / lum(L));
                    const float fx = fabsf(((float)X - dx) * invRx * 16), fy = fabsf(((float)Y - dy) * invRy * 16);
                    const int ifx = min((int)floorf(fx), 15), ify = min((int)floorf(fy), 15);
                    const float fw = rd.filter_table[ify * 16 + ifx];
                    const Spec c = (L * 1.f) * fw;
                    r += c.r; g += c.g; b += c.b; w += fw;
                }
            }
        PgFilmPixel *fp = &film[(size_t)local * rd.tile_pixels + e];
        fp->rgb[0] = r; fp->rgb[1] = g; fp->rgb[2] = b; fp->weight = w;
    }
}
// Tile-serial samplers: the one finished sample of every tile, added to the tile's film block exactly as FilmTile::AddSample
// does it (film.h:121-161) -- one lane per tile walks the sample's footprint, so a block's sums have the reference's order.
__global__ void k_ts_film(DScene sc, RenderParams rp, PathState st, PgFilmPixel *film, PgStraySample *strays, int maxStrays, int *nStrays) {
    const int local = blockIdx.x * blockDim.x + threadIdx.x;
    if (local >= rp.nTilesBatch || !sc.ts[local].active) return;
    const PgRenderDesc &rd = rp.rd;
    const int px = sc.ts[local].px, py = sc.ts[local].py;
    const int t = rd.tile_first + local * rd.tile_step;
    const int tx = t % rp.nTilesX, ty = t / rp.nTilesX;
    const int x0 = rd.sample_bounds[0] + tx * 16, y0 = rd.sample_bounds[1] + ty * 16;
    const int x1 = min(x0 + 16, rd.sample_bounds[2]), y1 = min(y0 + 16, rd.sample_bounds[3]);
    const float frx = rd.filter_radius[0], fry = rd.filter_radius[1];
    const int tp0x = max((int)ceilf((float)x0 - 0.5f - frx), rd.cropped_pixel_bounds[0]);
    const int tp0y = max((int)ceilf((float)y0 - 0.5f - fry), rd.cropped_pixel_bounds[1]);
    const int tp1x = min((int)floorf((float)x1 - 0.5f + frx) + 1, rd.cropped_pixel_bounds[2]);
    const int tp1y = min((int)floorf((float)y1 - 0.5f + fry) + 1, rd.cropped_pixel_bounds[3]);
    const float4 L4 = st.L[local];
    const float pFilmX = L4.w, pFilmY = st.beta[local].w;
    Spec L = sp3(L4.x, L4.y, L4.z);
    if (isnan(L.r) || isnan(L.g) || isnan(L.b)) L = sp(0);  // integrator.cpp:294-315
    else if ((double)lum(L) < -1e-5) L = sp(0);
    else if (isinf(lum(L))) L = sp(0);
    if (lum(L) > rd.max_sample_luminance) L = L * (rd.max_sample_luminance / lum(L));
    const float dx = pFilmX - 0.5f, dy = pFilmY - 0.5f;
    const int p0x = max((int)ceilf(dx - frx), tp0x), p0y = max((int)ceilf(dy - fry), tp0y);
    const int p1x = min((int)floorf(dx + frx) + 1, tp1x), p1y = min((int)floorf(dy + fry) + 1, tp1y);
    if (rd.filter_general) {
        const int tw = 16 + rd.tile_halo[0] + rd.tile_halo[2];
        const float invRx = 1 / frx, invRy = 1 / fry;
        for (int y = p0y; y < p1y; ++y) {
            const float fy = fabsf(((float)y - dy) * invRy * 16);
            const int ify = min((int)floorf(fy), 15);
            for (int x = p0x; x < p1x; ++x) {
                const float fx = fabsf(((float)x - dx) * invRx * 16);
                const int ifx = min((int)floorf(fx), 15);
                const float fw = rd.filter_table[ify * 16 + ifx];
                const Spec c = (L * 1.f) * fw;
                PgFilmPixel *fp = &film[(size_t)local * rd.tile_pixels + (size_t)(y - (y0 - rd.tile_halo[1])) * tw + (x - (x0 - rd.tile_halo[0]))];
                fp->rgb[0] += c.r; fp->rgb[1] += c.g; fp->rgb[2] += c.b; fp->weight += fw;
            }
        }
        return;
    }
    const Spec c = (L * 1.f) * 1.f;
    for (int y = p0y; y < p1y; ++y)
        for (int x = p0x; x < p1x; ++x) {
            if (x == px && y == py) {
                PgFilmPixel *fp = &film[(size_t)local * 256 + (py - y0) * 16 + (px - x0)];
                fp->rgb[0] += c.r; fp->rgb[1] += c.g; fp->rgb[2] += c.b; fp->weight += 1.f;
            } else {
                const int k = atomicAdd(nStrays, 1);
                if (k < maxStrays) {
                    PgStraySample s;
                    s.px = x; s.py = y; s.src_px = px; s.src_py = py;
                    s.rgb[0] = c.r; s.rgb[1] = c.g; s.rgb[2] = c.b; s.weight = 1.f;
                    strays[k] = s;
                }
            }
        }
}
void launch_ts_film(const DScene &sc, const RenderParams &rp, PathState st, PgFilmPixel *film, PgStraySample *strays, int maxStrays, int *nStrays,
                    hipStream_t s) {
    hipLaunchKernelGGL(k_ts_film, dim3((rp.nTilesBatch + 63) / 64), dim3(64), 0, s, sc, rp, st, film, strays, maxStrays, nStrays);
}
void launch_film_general(const RenderParams &rp, PathState st, PgFilmPixel *film, hipStream_t s) {
    if (rp.nTilesBatch == 0) return;
    hipLaunchKernelGGL(k_film_general, dim3(rp.nTilesBatch), dim3(PG_BLOCK), 0, s, rp, st, film);
}
void launch_film(const RenderParams &rp, PathState st, PgFilmPixel *film, PgStraySample *strays, int maxStrays, int *nStrays,
                 hipStream_t s) {
    if (rp.nTilesBatch == 0) return;
    hipLaunchKernelGGL(k_film, dim3(rp.nTilesBatch), dim3(PG_BLOCK), 0, s, rp, st, film, strays, maxStrays, nStrays);
}

// ===========================================================================
// SpatialLightDistribution::ComputeDistribution for every voxel up front
// (lightdistrib.cpp:232-300).  The reference fills voxels lazily through a
// hash table; the distribution is a pure function of the voxel, so a dense
// table holds the same values.  One lane per voxel.
// ===========================================================================
__global__ __launch_bounds__(PG_BLOCK) void k_light_tables(DScene sc, float *table, int nDistributions) {
    const int v = blockIdx.x * PG_BLOCK + threadIdx.x;
    if (v >= nDistributions) return;
    const int nl = sc.nLights;
    float *func = table + (size_t)v * (2 * nl + 2), *cdf = func + nl;
    int pi0 = v % sc.nVoxels[0], pi1 = (v / sc.nVoxels[0]) % sc.nVoxels[1], pi2 = v / (sc.nVoxels[0] * sc.nVoxels[1]);
    V3 p0 = mk((float)pi0 / (float)sc.nVoxels[0], (float)pi1 / (float)sc.nVoxels[1], (float)pi2 / (float)sc.nVoxels[2]);
    V3 p1 = mk((float)(pi0 + 1) / (float)sc.nVoxels[0], (float)(pi1 + 1) / (float)sc.nVoxels[1], (float)(pi2 + 1) / (float)sc.nVoxels[2]);
    V3 a = mk(plerp(p0.x, sc.bmin[0], sc.bmax[0]), plerp(p0.y, sc.bmin[1], sc.bmax[1]), plerp(p0.z, sc.bmin[2], sc.bmax[2]));
    V3 b = mk(plerp(p1.x, sc.bmin[0], sc.bmax[0]), plerp(p1.y, sc.bmin[1], sc.bmax[1]), plerp(p1.z, sc.bmin[2], sc.bmax[2]));
    V3 vmin = mk(pmin(a.x, b.x), pmin(a.y, b.y), pmin(a.z, b.z)), vmax = mk(pmax(a.x, b.x), pmax(a.y, b.y), pmax(a.z, b.z));
    for (int j = 0; j < nl; ++j) func[j] = 0;
    const int nSamples = 128;
    for (int i = 0; i < nSamples; ++i) {
        V3 t = mk(radical_inverse_base2(i), radical_inverse(3, i), radical_inverse(5, i));
        V3 po = mk(plerp(t.x, vmin.x, vmax.x), plerp(t.y, vmin.y, vmax.y), plerp(t.z, vmin.z, vmax.z));
        float u0 = radical_inverse(7, i), u1 = radical_inverse(11, i);
        for (int j = 0; j < nl; ++j) {
            float pdf;
            V3 wi;
            LightSample ls;
            Spec Li = light_sample_li<true>(sc, sc.lights[j], po, mk(0, 0, 0), mk(0, 0, 0), u0, u1, wi, pdf, ls);  // Interaction(po, Normal3f(), Vector3f(), ...)
            if (pdf > 0) func[j] += lum(Li) / pdf;
        }
    }
    float sumContrib = 0;
    for (int j = 0; j < nl; ++j) sumContrib = sumContrib + func[j];
    float avgContrib = sumContrib / (float)((size_t)nSamples * (size_t)nl);
    float minContrib = (avgContrib > 0) ? (float)(.001 * (double)avgContrib) : 1.f;
    for (int j = 0; j < nl; ++j) func[j] = pmax(func[j], minContrib);
    // Distribution1D ctor, sampling.h:57-70
    cdf[0] = 0;
    for (int i = 1; i < nl + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / nl;
    float funcInt = cdf[nl];
    if (funcInt == 0) { for (int i = 1; i < nl + 1; ++i) cdf[i] = (float)i / (float)nl; }
    else { for (int i = 1; i < nl + 1; ++i) cdf[i] /= funcInt; }
    func[2 * nl + 1] = funcInt;
}
// The same for a list of voxels, one BLOCK per voxel (sparse tables: few voxels, possibly thousands of lights): the lights are
// spread over the block's threads -- func[j] is a sum over the 128 sample points in order, independent of the other lights --
// and one thread then runs the reference's sequential sums over the lights (sumContrib, the cdf).
__global__ __launch_bounds__(PG_BLOCK) void k_light_tables_sparse(DScene sc, float *pool, const int *requests, int firstSlot) {
    const int v = requests[blockIdx.x];
    const int nl = sc.nLights;
    float *func = pool + (size_t)(firstSlot + (int)blockIdx.x) * (2 * nl + 2), *cdf = func + nl;
    int pi0 = v % sc.nVoxels[0], pi1 = (v / sc.nVoxels[0]) % sc.nVoxels[1], pi2 = v / (sc.nVoxels[0] * sc.nVoxels[1]);
    V3 p0 = mk((float)pi0 / (float)sc.nVoxels[0], (float)pi1 / (float)sc.nVoxels[1], (float)pi2 / (float)sc.nVoxels[2]);
    V3 p1 = mk((float)(pi0 + 1) / (float)sc.nVoxels[0], (float)(pi1 + 1) / (float)sc.nVoxels[1], (float)(pi2 + 1) / (float)sc.nVoxels[2]);
    V3 a = mk(plerp(p0.x, sc.bmin[0], sc.bmax[0]), plerp(p0.y, sc.bmin[1], sc.bmax[1]), plerp(p0.z, sc.bmin[2], sc.bmax[2]));
    V3 b = mk(plerp(p1.x, sc.bmin[0], sc.bmax[0]), plerp(p1.y, sc.bmin[1], sc.bmax[1]), plerp(p1.z, sc.bmin[2], sc.bmax[2]));
    V3 vmin = mk(pmin(a.x, b.x), pmin(a.y, b.y), pmin(a.z, b.z)), vmax = mk(pmax(a.x, b.x), pmax(a.y, b.y), pmax(a.z, b.z));
    const int nSamples = 128;
    for (int j = threadIdx.x; j < nl; j += PG_BLOCK) {
        float f = 0;
        for (int i = 0; i < nSamples; ++i) {
            V3 t = mk(radical_inverse_base2(i), radical_inverse(3, i), radical_inverse(5, i));
            V3 po = mk(plerp(t.x, vmin.x, vmax.x), plerp(t.y, vmin.y, vmax.y), plerp(t.z, vmin.z, vmax.z));
            float u0 = radical_inverse(7, i), u1 = radical_inverse(11, i);
            float pdf;
            V3 wi;
            LightSample ls;
            Spec Li = light_sample_li<true>(sc, sc.lights[j], po, mk(0, 0, 0), mk(0, 0, 0), u0, u1, wi, pdf, ls);
            if (pdf > 0) f += lum(Li) / pdf;
        }
        func[j] = f;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sumContrib = 0;
        for (int j = 0; j < nl; ++j) sumContrib = sumContrib + func[j];
        float avgContrib = sumContrib / (float)((size_t)nSamples * (size_t)nl);
        float minContrib = (avgContrib > 0) ? (float)(.001 * (double)avgContrib) : 1.f;
        for (int j = 0; j < nl; ++j) func[j] = pmax(func[j], minContrib);
        cdf[0] = 0;  // Distribution1D ctor, sampling.h:57-70
        for (int i = 1; i < nl + 1; ++i) cdf[i] = cdf[i - 1] + func[i - 1] / nl;
        float funcInt = cdf[nl];
        if (funcInt == 0) { for (int i = 1; i < nl + 1; ++i) cdf[i] = (float)i / (float)nl; }
        else { for (int i = 1; i < nl + 1; ++i) cdf[i] /= funcInt; }
        func[2 * nl + 1] = funcInt;
        sc.voxelSlot[v] = firstSlot + (int)blockIdx.x;  // published at the end of the launch (kernel boundary)
    }
}
void launch_light_tables_sparse(const DScene &sc, float *pool, const int *requests, int n, int firstSlot, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_light_tables_sparse, dim3(n), dim3(PG_BLOCK), 0, s, sc, pool, requests, firstSlot);
}
void launch_light_tables(const DScene &sc, float *table, int nDistributions, hipStream_t s) {
    int nblk = (nDistributions + PG_BLOCK - 1) / PG_BLOCK;
    hipLaunchKernelGGL(k_light_tables, dim3(nblk), dim3(PG_BLOCK), 0, s, sc, table, nDistributions);
}

