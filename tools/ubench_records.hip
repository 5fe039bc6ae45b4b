// ubench_records.hip -- the ceiling for VERDICT r04 item 1 (two-level 128-B BVH records): how fast do the lanes of the chip chase
// chains of random records of 32 / 64 / 128 bytes (2 / 4 / 8 16-byte loads per lane and step, issued together, every dword used),
// at the resident blocks per CU a traversal kernel of that register budget would have, with all lanes fetching and with the ~55 %
// of the lanes a k_trace interior step has active (profiles/r03w_trace_step_statistics.txt)?  If a 128-B record costs as much as a
// 64-B one (the L2 fills 128-B lines either way: profiles/fetch_size_calibration.json), two tree levels per fetch is a gain; if the
// cost follows the bytes or the load instructions, it is not.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench_records.hip -o ubench_records && ./ubench_records [table MiB ...]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__device__ inline unsigned mix(unsigned h) { h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16; return h; }
// REC16 16-byte pieces per record; `active` of 64 lanes fetch in a step (which ones changes from step to step)
template <int REC16, bool NT = false, int STRIDE16 = REC16, bool MIXED = false>
__global__ __launch_bounds__(256) void k_chase(const float4 *__restrict__ table, unsigned nRec, int iters, float *out, unsigned active, unsigned hot256, unsigned nHot) {
    extern __shared__ float4 pad[];  // sets the resident blocks per CU only
    unsigned idx = mix(blockIdx.x * 256u + threadIdx.x + 0x9e3779b9u) % nRec;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        const bool on = (mix(threadIdx.x * 977u + it * 131u) & 63u) < active;
        if (on) {
            const float4 *rec = table + (size_t)STRIDE16 * idx;
            float4 v[REC16];
            const bool shortFetch = MIXED && (mix(threadIdx.x * 31u + it * 7919u + blockIdx.x) & 1u);  // MIXED: every other step needs the first 32 B only
            if (MIXED) {
                v[0] = rec[0]; v[1] = rec[1];
#pragma unroll
                for (int k = 2; k < REC16; ++k) v[k] = make_float4(0, 0, 0, 0);
                if (!shortFetch) {  // one block for the remaining pieces: the loads leave together
#pragma unroll
                    for (int k = 2; k < REC16; ++k) v[k] = rec[k];
                }
            } else {
#pragma unroll
            for (int k = 0; k < REC16; ++k) if (NT) { typedef float f4 __attribute__((ext_vector_type(4))); const f4 t = __builtin_nontemporal_load((const f4 *)(rec + k)); v[k] = make_float4(t.x, t.y, t.z, t.w); } else v[k] = rec[k];
            }
            float s = 0;
#pragma unroll
            for (int k = 0; k < REC16; ++k) s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
            acc += s;
            // the next record depends on the data (a dependent fetch, as in a traversal) AND on the lane and the step: walks that are a
            // function of the record alone coalesce (two lanes that meet stay together), and after a few hundred steps the whole chip
            // chases a few thousand records -- cache hits that a traversal does not have.  hot256 / 256 of the steps go to the first
            // nHot records instead (the top of a tree: shared by everyone, resident in every L2).
            const unsigned h = mix(idx + 0x9e3779b9u * (blockIdx.x * 256u + threadIdx.x + 1u) + (unsigned)it * 0x85ebca6bu + (unsigned)__float_as_int(v[0].x) + (unsigned)__float_as_int(s));
            idx = ((h >> 24) < hot256) ? (h % nHot) : (h % nRec);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
    if (threadIdx.x == 100000) out[0] = pad[0].x;
}
__global__ void k_fill(float4 *table, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n4) table[i] = make_float4((float)((4 * i) % 251), (float)((4 * i + 1) % 241), (float)((4 * i + 2) % 239), (float)((4 * i + 3) % 233));
}
// STRIDE16: 16-byte pieces between record starts (a 48-B record in a 64-B slot: REC16 3, STRIDE16 4)
static unsigned g_hot256 = 0;
template <int REC16, bool NT = false, int STRIDE16 = REC16, bool MIXED = false>
static double rate(const float4 *table, size_t bytes, float *out, int perCU, unsigned active) {
    const unsigned nRec = (unsigned)(bytes / (16 * STRIDE16));
    const unsigned nHot = (unsigned)((2u << 20) / (16 * STRIDE16)) < nRec ? (unsigned)((2u << 20) / (16 * STRIDE16)) : nRec;  // 2 MiB of hot records
    const int blocks = 256 * perCU, iters = 512;
    const size_t lds = perCU >= 8 ? 19 * 1024 : perCU == 7 ? 22 * 1024 + 512 : perCU == 6 ? 26 * 1024 : perCU == 5 ? 32 * 1024 : 40 * 1024;
    CHECK(hipFuncSetAttribute((const void *)(k_chase<REC16, NT, STRIDE16, MIXED>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_chase<REC16, NT, STRIDE16, MIXED>), dim3(blocks), dim3(256), lds, 0, table, nRec, 8, out, active, g_hot256, nHot);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_chase<REC16, NT, STRIDE16, MIXED>), dim3(blocks), dim3(256), lds, 0, table, nRec, iters, out, active, g_hot256, nHot);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return (double)blocks * 256 * iters * (active / 64.0) / ms * 1e-6;  // G records / s
}
int main(int argc, char **argv) {
    std::vector<size_t> sizesMB;
    for (int i = 1; i < argc; ++i) sizesMB.push_back((size_t)atol(argv[i]));
    if (sizesMB.empty()) sizesMB = {2, 47, 94, 106, 212, 535, 1070};
    float *out;
    CHECK(hipMalloc(&out, sizeof(float) * 256 * 256 * 8));
    printf("G record fetches / s (chains of dependent random records; every lane of an `active` set fetches one record per step)\n");
    for (unsigned hot : {0u, 128u, 192u})
    for (size_t mb : sizesMB) {
        g_hot256 = hot;
        printf("-- %u %% of the steps fetch from a hot 2 MiB, the rest anywhere in the table\n", hot * 100 / 256);
        const size_t bytes = mb * 1024 * 1024;
        float4 *table;
        CHECK(hipMalloc(&table, bytes));
        hipLaunchKernelGGL(k_fill, dim3((unsigned)((bytes / 16 + 255) / 256)), dim3(256), 0, 0, table, bytes / 16);
        for (unsigned active : {64u, 35u}) {
            printf("table %5zu MiB, %2u of 64 lanes:", mb, active);
            for (int perCU : {7, 6, 5, 4}) {
                printf("  | %d blk/CU: 32 B %6.1f  64 B %6.1f  128 B %6.1f", perCU, rate<2>(table, bytes, out, perCU, active), rate<4>(table, bytes, out, perCU, active),
                       rate<8>(table, bytes, out, perCU, active));
            }
            printf("\n");
            // 7 blocks per CU: 48 B of a 64-B slot (three loads), 96 B of a 128-B slot (six), and the 64-B / 128-B records through non-temporal loads
            printf("                                 | 7 blk/CU: 48 of 64 B %6.1f  96 of 128 B %6.1f  | non-temporal loads: 64 B %6.1f  128 B %6.1f\n", rate<3, false, 4>(table, bytes, out, 7, active),
                   rate<6, false, 8>(table, bytes, out, 7, active), rate<4, true>(table, bytes, out, 7, active), rate<8, true>(table, bytes, out, 7, active));
            printf("                                 | 7 blk/CU: 64-B slots, every other step reads the first 32 B only %6.1f\n", rate<4, false, 4, true>(table, bytes, out, 7, active));
        }
        CHECK(hipFree(table));
    }
    return 0;
}
