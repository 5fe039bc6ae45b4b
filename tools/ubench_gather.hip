// ubench_gather.hip -- how should 64 lanes fetch 64 different 64-byte records (a BVH child-pair record per ray)?
//   A  four 16-byte loads per lane, issued together (what k_trace did through round 2)
//   B  the first 16 bytes, wait, then the other three (the line is in L1 by then)
//   C  quad-cooperative: in round r the four lanes of a quad fetch the four 16-byte pieces of quad-lane r's record
//      (one 64-byte access per quad and round); the pieces are NOT exchanged back -- this is the fetch cost alone
//   D  one 16-byte load per lane (a quarter of the data: the floor for "one access per lane")
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
        } else { a = rec[0]; b = c = d = a; }
        acc += a.y + b.y + c.y + d.y;
        idx = next_index(idx, a, n);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

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
    return ms;
}

int main(int argc, char **argv) {
    // usage: ubench_gather [--json] [table sizes in MiB ...]   (default: L2-resident, config 3's working set, the 5 M / 10 M-triangle sets)
    const int blocks = 256 * 7, iters = 512;
    bool json = false;
    std::vector<size_t> sizesMB;
    for (int i = 1; i < argc; ++i) {
        if (std::string(argv[i]) == "--json") json = true;
        else sizesMB.push_back((size_t)atol(argv[i]));
    }
    if (sizesMB.empty()) sizesMB = {2, 24, 91, 459, 925};
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
    if (json) printf("{");
    bool first = true;
    for (size_t mb : sizesMB) {
        if (mb < 1) mb = 1;
        const unsigned n = (unsigned)(mb * 1024 * 1024 / 64);
        float4 *table;
        CHECK(hipMalloc(&table, (size_t)n * 64));
        CHECK(hipMemset(table, 0, (size_t)n * 64));
        const double fetches = (double)blocks * 256 * iters;
        const double a = run<0>(table, n, iters, out, blocks), b = run<1>(table, n, iters, out, blocks), c = run<2>(table, n, iters, out, blocks),
                     d = run<3>(table, n, iters, out, blocks);
        if (json)
            printf("%s\"%zu\": {\"together\": %.4g, \"staged\": %.4g, \"quad_cooperative\": %.4g, \"first_16_bytes_only\": %.4g}", first ? "" : ", ", mb,
                   fetches / a * 1e3, fetches / b * 1e3, fetches / c * 1e3, fetches / d * 1e3);
        else
            printf("table %4zu MB: A together %.2f ms (%.1f G rec/s, %.0f GB/s) | B staged %.2f ms (%.1f) | C quad-cooperative %.2f ms (%.1f) | D 16 B only %.2f ms (%.1f)\n",
                   mb, a, fetches / a * 1e-6, fetches * 64 / a * 1e-6, b, fetches / b * 1e-6, c, fetches / c * 1e-6, d, fetches / d * 1e-6);
        first = false;
        CHECK(hipFree(table));
    }
    if (json) printf("}\n");
    return 0;
}
