// ubench_gather.hip -- how should 64 lanes fetch 64 different 64-byte records (a BVH child-pair record per ray)?
//   A  four 16-byte loads per lane, issued together (what k_trace did through round 2)
//   B  the first 16 bytes, wait, then the other three (the line is in L1 by then)
//   C  quad-cooperative: in round r the four lanes of a quad fetch the four 16-byte pieces of quad-lane r's record
//      (one 64-byte access per quad and round); the pieces are NOT exchanged back -- this is the fetch cost alone
//   D  one 16-byte load per lane (a quarter of the data: the floor for "one access per lane")
//   E  C followed by the exchange of the pieces inside the quad (DPP quad broadcasts + selects): what a traversal would pay
// Every lane chases its own chain: the next record index depends on the data just loaded, as in a traversal.
// Built by pbrt-v3_amd/Makefile into pbrt-v3_amd/ubench_gather; bench.py runs it (--json, the scene's working-set size) for the
// record-fetch ceiling it quotes next to k_trace's rate.  (records per second with --json)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline unsigned next_index(unsigned idx, float4 v, unsigned n) {  // a 32-bit mixer (an LCG modulo n cycles through few records for some n)
    unsigned h = idx + 0x9e3779b9u + (unsigned)__float_as_int(v.x);
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h % n;
}

template <int MODE>
__global__ __launch_bounds__(256) void k_gather(const float4 *__restrict__ table, unsigned n, int iters, float *out) {
    unsigned idx = next_index(blockIdx.x * 256u + threadIdx.x, make_float4(0, 0, 0, 0), n);
    float acc = 0;
    const int lane4 = threadIdx.x & 3;
    for (int it = 0; it < iters; ++it) {
        const float4 *rec = table + 4 * (size_t)idx;
        float4 a, b, c, d;
        if (MODE == 0) { a = rec[0]; b = rec[1]; c = rec[2]; d = rec[3]; }
        else if (MODE == 1) {
            a = rec[0];
            unsigned idx2 = idx;
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(idx2) : "v"(a.x) : "memory");
            const float4 *rec2 = table + 4 * (size_t)idx2;
            b = rec2[1]; c = rec2[2]; d = rec2[3];
        } else if (MODE == 2) {
            // round r: every lane of the quad takes piece lane4 of quad-lane r's record
            const unsigned i0 = __shfl(idx, (threadIdx.x & 60) + 0), i1 = __shfl(idx, (threadIdx.x & 60) + 1), i2 = __shfl(idx, (threadIdx.x & 60) + 2),
                           i3 = __shfl(idx, (threadIdx.x & 60) + 3);
            a = table[4 * (size_t)i0 + lane4]; b = table[4 * (size_t)i1 + lane4]; c = table[4 * (size_t)i2 + lane4]; d = table[4 * (size_t)i3 + lane4];
        } else if (MODE == 4) {
            // C + the exchange: afterwards every lane holds the four pieces of ITS OWN record, as a traversal needs them.
            // In round r the quad fetched quad-lane r's record, lane j piece j; lane l takes piece p of its record from lane p's
            // round-l registers (a quad broadcast), selected by its position in the quad.
            const unsigned i0 = __shfl(idx, (threadIdx.x & 60) + 0), i1 = __shfl(idx, (threadIdx.x & 60) + 1), i2 = __shfl(idx, (threadIdx.x & 60) + 2),
                           i3 = __shfl(idx, (threadIdx.x & 60) + 3);
            float4 r[4] = {table[4 * (size_t)i0 + lane4], table[4 * (size_t)i1 + lane4], table[4 * (size_t)i2 + lane4], table[4 * (size_t)i3 + lane4]};
            float4 o[4] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
#define BC(v, P) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), (P) | ((P) << 2) | ((P) << 4) | ((P) << 6), 0xf, 0xf, true))
#define TAKE(L, P) { const bool me = lane4 == (L); const float x = BC(r[L].x, P), y = BC(r[L].y, P), z = BC(r[L].z, P), w = BC(r[L].w, P); \
                     o[P].x = me ? x : o[P].x; o[P].y = me ? y : o[P].y; o[P].z = me ? z : o[P].z; o[P].w = me ? w : o[P].w; }
            TAKE(0, 0) TAKE(0, 1) TAKE(0, 2) TAKE(0, 3) TAKE(1, 0) TAKE(1, 1) TAKE(1, 2) TAKE(1, 3)
            TAKE(2, 0) TAKE(2, 1) TAKE(2, 2) TAKE(2, 3) TAKE(3, 0) TAKE(3, 1) TAKE(3, 2) TAKE(3, 3)
#undef TAKE
#undef BC
            a = o[0]; b = o[1]; c = o[2]; d = o[3];
        } else if (MODE == 5) {
            // E with each select and its quad broadcast fused into ONE v_cndmask_b32_dpp (the compiler emits a v_mov_b32_dpp and a
            // v_cndmask_b32): D = vcc ? old : broadcast(r), vcc = the lanes that are NOT quad-lane L
            const unsigned i0 = __shfl(idx, (threadIdx.x & 60) + 0), i1 = __shfl(idx, (threadIdx.x & 60) + 1), i2 = __shfl(idx, (threadIdx.x & 60) + 2),
                           i3 = __shfl(idx, (threadIdx.x & 60) + 3);
            float4 r[4] = {table[4 * (size_t)i0 + lane4], table[4 * (size_t)i1 + lane4], table[4 * (size_t)i2 + lane4], table[4 * (size_t)i3 + lane4]};
            float4 o[4] = {make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0), make_float4(0, 0, 0, 0)};
#define ROW(P) "v_cndmask_b32_dpp %" #P "0, %16, %" #P "0, vcc quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
               "v_cndmask_b32_dpp %" #P "1, %17, %" #P "1, vcc quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
               "v_cndmask_b32_dpp %" #P "2, %18, %" #P "2, vcc quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
               "v_cndmask_b32_dpp %" #P "3, %19, %" #P "3, vcc quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t"
            // operands: %0..%3 = o[0].xyzw, %4..%7 = o[1], %8..%11 = o[2], %12..%15 = o[3], %16..%19 = r[L].xyzw, %20 = mask
#define TAKE_ROW(L) { const unsigned long long notMe = ~(0x1111111111111111ull << (L)); \
            asm volatile("s_nop 1\n\ts_mov_b64 vcc, %20\n\t" \
                "v_cndmask_b32_dpp %0, %16, %0, vcc quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %1, %17, %1, vcc quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t" \
                "v_cndmask_b32_dpp %2, %18, %2, vcc quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %3, %19, %3, vcc quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t" \
                "v_cndmask_b32_dpp %4, %16, %4, vcc quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %5, %17, %5, vcc quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t" \
                "v_cndmask_b32_dpp %6, %18, %6, vcc quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %7, %19, %7, vcc quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t" \
                "v_cndmask_b32_dpp %8, %16, %8, vcc quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %9, %17, %9, vcc quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t" \
                "v_cndmask_b32_dpp %10, %18, %10, vcc quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %11, %19, %11, vcc quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t" \
                "v_cndmask_b32_dpp %12, %16, %12, vcc quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %13, %17, %13, vcc quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\t" \
                "v_cndmask_b32_dpp %14, %18, %14, vcc quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\tv_cndmask_b32_dpp %15, %19, %15, vcc quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf" \
                : "+v"(o[0].x), "+v"(o[0].y), "+v"(o[0].z), "+v"(o[0].w), "+v"(o[1].x), "+v"(o[1].y), "+v"(o[1].z), "+v"(o[1].w), \
                  "+v"(o[2].x), "+v"(o[2].y), "+v"(o[2].z), "+v"(o[2].w), "+v"(o[3].x), "+v"(o[3].y), "+v"(o[3].z), "+v"(o[3].w) \
                : "v"(r[L].x), "v"(r[L].y), "v"(r[L].z), "v"(r[L].w), "s"(notMe) : "vcc"); }
            TAKE_ROW(0) TAKE_ROW(1) TAKE_ROW(2) TAKE_ROW(3)
#undef TAKE_ROW
#undef ROW
            a = o[0]; b = o[1]; c = o[2]; d = o[3];
        } else { a = rec[0]; b = c = d = a; }
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);  // every dword is used, as in a traversal
        idx = next_index(idx, a, n);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ void k_fill(float4 *table, size_t n4) {  // distinct small integers per dword, so that a wrong exchange changes the sums
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) table[i] = make_float4((float)((4 * i) % 251), (float)((4 * i + 1) % 241), (float)((4 * i + 2) % 239), (float)((4 * i + 3) % 233));
}

// Round 3 questions: what would k_trace gain from (i) serving a fraction of its record fetches from an LDS-resident
// table (the top of the tree), (ii) 8 instead of 7 waves per SIMD, (iii) 32-B half records?  Same chain-chasing loop as above;
//   frac256 / 256 of the steps read a record of an LDS table of `ldsRecs` records (swizzled as in pg_traverse.hip: piece p of
//   record r at slot (p + (r >> 2)) & 3) instead of the global table; REC16 = 16-B pieces per global record (4 = 64 B, 2 = 32 B).
template <int REC16>
__global__ __launch_bounds__(256) void k_gather_lds(const float4 *__restrict__ table, unsigned n, int iters, float *out, int ldsRecs, unsigned frac256) {
    // pointers that carry their address space in the type: from two branches loading through generic pointers the compiler makes one
    // flat_load through a selected pointer (a third of the rate -- round 3's first run of this table measured exactly that)
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ float4 ldsRaw[];
    __attribute__((address_space(3))) f4 *ldsTab = (__attribute__((address_space(3))) f4 *)ldsRaw;
    const __attribute__((address_space(1))) f4 *gtab = (const __attribute__((address_space(1))) f4 *)table;
    for (int i = threadIdx.x; i < 4 * ldsRecs; i += 256) { const int r = i >> 2, p = i & 3; ldsTab[4 * r + ((p + (r >> 2)) & 3)] = gtab[i]; }
    __syncthreads();
    unsigned idx = next_index(blockIdx.x * 256u + threadIdx.x, make_float4(0, 0, 0, 0), n);
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        f4 a, b, c, d;
        const bool fromLds = ((idx * 2654435761u) >> 24) < frac256;
        if (fromLds) {
            const unsigned r = idx % (unsigned)ldsRecs, sw = r >> 2;
            const __attribute__((address_space(3))) f4 *rec = ldsTab + 4 * r;
            a = rec[sw & 3]; b = rec[(sw + 1) & 3]; c = rec[(sw + 2) & 3]; d = rec[(sw + 3) & 3];
        } else {
            const __attribute__((address_space(1))) f4 *rec = gtab + 4 * (size_t)idx;
            a = rec[0]; b = rec[1];
            if (REC16 == 4) { c = rec[2]; d = rec[3]; } else { c = a; d = b; }
        }
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
        idx = next_index(idx, make_float4(a.x, a.y, a.z, a.w), n);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
template <int REC16>
static double run_lds(const float4 *table, unsigned n, int iters, float *out, int blocks, int ldsRecs, unsigned frac256, size_t ldsBytes) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    CHECK(hipFuncSetAttribute((const void *)k_gather_lds<REC16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(k_gather_lds<REC16>, dim3(blocks), dim3(256), ldsBytes, 0, table, n, 8, out, ldsRecs, frac256);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather_lds<REC16>, dim3(blocks), dim3(256), ldsBytes, 0, table, n, iters, out, ldsRecs, frac256);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
}
// --calib: one launch of the 64-B record gather (mode A) and one coalesced 16 B / lane streaming read of the same table, each with an
// exactly known byte count, for calibrating rocprofv3's FETCH_SIZE on this access pattern (tools/pmc_calibrate.sh).
__global__ __launch_bounds__(256) void k_stream(const float4 *__restrict__ table, size_t n4, float *out) {
    float acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = table[i]; acc += v.x + v.y + v.z + v.w; }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// The sibling test: every lane reads its record AND the other 64-B half of the same 128-B line, one after the other.  If an L2
// miss fills the whole 128-B line, the second read hits in the L2 and the memory-side request count stays that of k_gather<0>;
// if the L2 fills 64-B sectors, it doubles.  That tells how many bytes one counted request of a record gather really moves.
__global__ __launch_bounds__(256) void k_gather_pair(const float4 *__restrict__ table, unsigned n, int iters, float *out) {
    unsigned idx = next_index(blockIdx.x * 256u + threadIdx.x, make_float4(0, 0, 0, 0), n);
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        const float4 *rec = table + 4 * (size_t)idx;
        const float4 a = rec[0], b = rec[1], c = rec[2], d = rec[3];
        acc += (a.x + a.y + a.z + a.w) + (b.x + b.y + b.z + b.w) + (c.x + c.y + c.z + c.w) + (d.x + d.y + d.z + d.w);
        unsigned sib = idx ^ 1u;
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sib) : "v"(acc) : "memory");
        const float4 *rec2 = table + 4 * (size_t)sib;
        const float4 e = rec2[0], f = rec2[1], g = rec2[2], h = rec2[3];
        acc += (e.x + e.y + e.z + e.w) + (f.x + f.y + f.z + f.w) + (g.x + g.y + g.z + g.w) + (h.x + h.y + h.z + h.w);
        idx = next_index(idx, a, n);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
static void calib(size_t mb, float *out) {
    const unsigned n = (unsigned)(mb * 1024 * 1024 / 64);
    const int blocks = 256 * 7, iters = 512;
    float4 *table;
    CHECK(hipMalloc(&table, (size_t)n * 64));
    hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)n * 4 + 255) / 256)), dim3(256), 0, 0, table, (size_t)n * 4);
    hipLaunchKernelGGL(k_gather<0>, dim3(blocks), dim3(256), 0, 0, table, n, iters, out);
    hipLaunchKernelGGL(k_stream, dim3(blocks), dim3(256), 0, 0, table, (size_t)n * 4, out);
    hipLaunchKernelGGL(k_gather_pair, dim3(blocks), dim3(256), 0, 0, table, n & ~1u, iters, out);
    CHECK(hipDeviceSynchronize());
    printf("{\"table_MiB\": %zu, \"gather_requested_bytes\": %.0f, \"stream_bytes\": %.0f}\n", mb, (double)blocks * 256 * iters * 64, (double)n * 64);
    CHECK(hipFree(table));
}
// --predict: the table profiles/r03*_gather_predictions.txt is made of.  Rates are record fetches per second (LDS-served ones included).
static void predict(const std::vector<size_t> &sizesMB, float *out) {
    const int iters = 512;
    for (size_t mb : sizesMB) {
        const unsigned n = (unsigned)(mb * 1024 * 1024 / 64);
        float4 *table;
        CHECK(hipMalloc(&table, (size_t)n * 64));
        hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)n * 4 + 255) / 256)), dim3(256), 0, 0, table, (size_t)n * 4);
        printf("table %zu MB, 64-B records unless noted; G records / s\n", mb);
        // (blocks per CU, LDS bytes per block incl. a stack stand-in that only sets the occupancy, records in LDS)
        struct Geo { int perCU; size_t lds; int recs; const char *what; } geos[] = {
            {7, 22 * 1024 + 512, 1, "7 blocks/CU (28 waves), no tree top"},
            {8, 19 * 1024, 1, "8 blocks/CU (32 waves), no tree top"},
            {6, 26 * 1024, 1, "6 blocks/CU (24 waves), no tree top"},
            {7, 22 * 1024 + 512, 127, "7 blocks/CU, 127 records (8 KB) in LDS"},
            {6, 26 * 1024, 255, "6 blocks/CU, 255 records (16 KB) in LDS"},
            {4, 40 * 1024, 511, "4 blocks/CU (16 waves), 511 records (32 KB) in LDS"},
        };
        for (const Geo &g : geos) {
            const int blocks = 256 * g.perCU;
            const double fetches = (double)blocks * 256 * iters;
            printf("  %-52s", g.what);
            for (unsigned f : {0u, 64u, 102u, 128u, 154u}) {
                if (g.recs == 1 && f > 0) continue;
                const double ms = run_lds<4>(table, n, iters, out, blocks, g.recs, f, g.lds);
                printf(" | LDS %2.0f%%: %6.1f", f / 2.56, fetches / ms * 1e-6);
            }
            if (g.recs == 1) { const double ms = run_lds<2>(table, n, iters, out, blocks, g.recs, 0, g.lds); printf(" | 32-B records: %6.1f", fetches / ms * 1e-6); }
            printf("\n");
        }
        CHECK(hipFree(table));
    }
}
static double g_checksum[8];
template <int MODE>
static double run(const float4 *table, unsigned n, int iters, float *out, int blocks) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, table, n, 8, out);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_gather<MODE>, dim3(blocks), dim3(256), 0, 0, table, n, iters, out);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<float> h((size_t)blocks * 256);
    CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    double sum = 0;
    for (float v : h) sum += v;
    g_checksum[MODE] = sum;
    return ms;
}

int main(int argc, char **argv) {
    // usage: ubench_gather [--json] [table sizes in MiB ...]   (default: L2-resident, config 3's working set, the 5 M / 10 M-triangle sets)
    const int blocks = 256 * 7, iters = 512;
    bool json = false, doPredict = false, doCalib = false;
    std::vector<size_t> sizesMB;
    for (int i = 1; i < argc; ++i) {
        if (std::string(argv[i]) == "--json") json = true;
        else if (std::string(argv[i]) == "--predict") doPredict = true;
        else if (std::string(argv[i]) == "--calib") doCalib = true;
        else sizesMB.push_back((size_t)atol(argv[i]));
    }
    if (sizesMB.empty()) sizesMB = {2, 24, 91, 459, 925};
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 256 * 8));
    if (doPredict) { predict(sizesMB, out); return 0; }
    if (doCalib) { for (size_t mb : sizesMB) calib(mb, out); return 0; }
    if (json) printf("{");
    bool first = true;
    for (size_t mb : sizesMB) {
        if (mb < 1) mb = 1;
        const unsigned n = (unsigned)(mb * 1024 * 1024 / 64);
        float4 *table;
        CHECK(hipMalloc(&table, (size_t)n * 64));
        hipLaunchKernelGGL(k_fill, dim3((unsigned)(((size_t)n * 4 + 255) / 256)), dim3(256), 0, 0, table, (size_t)n * 4);
        const double fetches = (double)blocks * 256 * iters;
        const double a = run<0>(table, n, iters, out, blocks), b = run<1>(table, n, iters, out, blocks), c = run<2>(table, n, iters, out, blocks),
                     d = run<3>(table, n, iters, out, blocks), e = run<4>(table, n, iters, out, blocks), f = run<5>(table, n, iters, out, blocks);
        if (json)
            printf("%s\"%zu\": {\"together\": %.4g, \"staged\": %.4g, \"quad_cooperative\": %.4g, \"first_16_bytes_only\": %.4g, \"quad_cooperative_exchanged\": %.4g, \"quad_cooperative_exchanged_fused\": %.4g}",
                   first ? "" : ", ", mb, fetches / a * 1e3, fetches / b * 1e3, fetches / c * 1e3, fetches / d * 1e3, fetches / e * 1e3, fetches / f * 1e3);
        else
            printf("table %4zu MB: A together %.2f ms (%.1f G rec/s, %.0f GB/s) | B staged %.2f ms (%.1f) | C quad-cooperative %.2f ms (%.1f) | D 16 B only %.2f ms (%.1f) | E C + exchange %.2f ms (%.1f) | F fused %.2f ms (%.1f)\n",
                   mb, a, fetches / a * 1e-6, fetches * 64 / a * 1e-6, b, fetches / b * 1e-6, c, fetches / c * 1e-6, d, fetches / d * 1e-6, e, fetches / e * 1e-6, f, fetches / f * 1e-6);
        // modes A, B, E, F deliver the same 64 bytes to every lane: their sums must agree (C and D read other bytes by design)
        if (g_checksum[0] != g_checksum[1] || g_checksum[0] != g_checksum[4] || g_checksum[0] != g_checksum[5]) {
            fprintf(stderr, "ubench_gather: the modes disagree at %zu MiB: A %.17g B %.17g E %.17g F %.17g\n", mb, g_checksum[0], g_checksum[1], g_checksum[4], g_checksum[5]);
            return 1;
        }
        first = false;
        CHECK(hipFree(table));
    }
    if (json) printf("}\n");
    return 0;
}
