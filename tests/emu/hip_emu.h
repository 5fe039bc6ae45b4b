// hip_emu.h -- TEST INFRASTRUCTURE: a minimal SIMT emulator under which the product's device translation units
// (pbrt-v3_amd/csrc/pg_abi.hip, pg_kernels.hip, pg_traverse.hip) compile for the HOST and run without a GPU.
//
// Force-included (-include) ahead of every translation unit, compiled with `hipcc --cuda-host-only`:
//   * __device__ / __global__ functions become ordinary host functions;
//   * a kernel launch runs the grid's blocks one after the other; the lanes of a block are cooperatively scheduled fibers
//     (ucontext), so __syncthreads and the wave-level operations (__ballot, __shfl_down, readfirstlane ...) are rendezvous points:
//     a lane that reaches one yields until every live lane of its wave (block) has arrived.  A kernel whose lanes are NOT
//     converged around such a call deadlocks on hardware-independent grounds and the emulator says so;
//   * atomics are plain operations (one OS thread), __shared__ variables are statics (one block at a time);
//   * the HIP runtime calls the host code makes (malloc / memcpy / memset / streams / events) act on host memory.
// The arithmetic is the device build's: -ffp-contract=off, correctly rounded divide and sqrt (x86).  See tests/emu/README.md.
#ifndef HIP_EMU_H
#define HIP_EMU_H
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>

namespace emu {
struct Idx { unsigned x, y, z; };
Idx thread_idx();
Idx block_idx();
Idx block_dim();
Idx grid_dim();
int lane();
// wave-level exchange: every live lane of the calling lane's wave deposits v; returns the 64 deposited values and the mask of lanes that took part
const uint64_t *wave_gather(uint64_t v, uint64_t *mask, int site, bool divergent = false);
void block_barrier();
void launch(const char *name, dim3 grid, dim3 block, size_t dynShared, const std::function<void()> &body);
extern unsigned char *dyn_shared;  // the block's dynamic shared memory
}  // namespace emu

// ---- device-only intrinsics: host overloads (clang overloads on the target attribute) ----
__host__ inline unsigned int __float_as_uint(float f) { unsigned int u; memcpy(&u, &f, 4); return u; }
__host__ inline float __uint_as_float(unsigned int u) { float f; memcpy(&f, &u, 4); return f; }
__host__ inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
__host__ inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
__host__ inline bool isinf(float v) { return __builtin_isinf(v); }
__host__ inline bool isnan(float v) { return __builtin_isnan(v); }
__host__ inline unsigned long long __brevll(unsigned long long v) { return __builtin_bitreverse64(v); }
__host__ inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
__host__ inline int __popc(unsigned int v) { return __builtin_popcount(v); }
__host__ inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
__host__ inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
__host__ inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
__host__ inline unsigned int __umulhi(unsigned int a, unsigned int b) { return (unsigned int)(((unsigned long long)a * b) >> 32); }
__host__ inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
__host__ inline unsigned int __lane_id() { return (unsigned)emu::lane(); }
__host__ inline void __syncthreads() { emu::block_barrier(); }
__host__ inline void __threadfence() {}
__host__ inline void __threadfence_block() {}
// A wave-level call in DIVERGENT code (only some lanes of the wave execute it) cannot be a rendezvous of all live lanes.  The kernels
// have a few such ballots whose result every lane uses for its OWN bit only (k_trace's slab masks, combined bitwise and read back
// through inverse_ballot): the build script rewrites exactly those call sites to this form, which needs no other lane.
__host__ inline unsigned long long emu_ballot_own(int pred) { return pred ? 1ull << emu::lane() : 0ull; }
__host__ inline unsigned long long emu_ballot(int pred, int site, bool divergent) {
    uint64_t mask;
    const uint64_t *v = emu::wave_gather(pred ? 1 : 0, &mask, site, divergent);
    unsigned long long r = 0;
    for (int l = 0; l < 64; ++l) if (((mask >> l) & 1) && v[l]) r |= 1ull << l;
    return r;
}
__host__ inline unsigned long long __ballot(int pred, int site = __builtin_LINE()) { return emu_ballot(pred, site, false); }
// the same among the lanes of a divergent branch (the build script marks those call sites): when every live lane of the wave waits
// somewhere and the sites differ, the lanes at a marked site are served first -- they are the ones the hardware would be executing
// while the others are masked off
__host__ inline unsigned long long emu_ballot_div(int pred, int site = __builtin_LINE()) { return emu_ballot(pred, site, true); }
template <class T> __host__ inline T emu_shfl_down(T val, unsigned delta, int site) {
    static_assert(sizeof(T) <= 8, "shuffle of up to 64 bits");
    uint64_t bits = 0, mask;
    memcpy(&bits, &val, sizeof(T));
    const uint64_t *v = emu::wave_gather(bits, &mask, site);
    const unsigned src = (unsigned)emu::lane() + delta;
    if (src < 64 && ((mask >> src) & 1)) { T r; memcpy(&r, &v[src], sizeof(T)); return r; }
    return val;
}
__host__ inline int __shfl_down(int v, unsigned d, int = 64, int site = __builtin_LINE()) { return emu_shfl_down(v, d, site); }
__host__ inline unsigned __shfl_down(unsigned v, unsigned d, int = 64, int site = __builtin_LINE()) { return emu_shfl_down(v, d, site); }
__host__ inline float __shfl_down(float v, unsigned d, int = 64, int site = __builtin_LINE()) { return emu_shfl_down(v, d, site); }
__host__ inline unsigned long long __shfl_down(unsigned long long v, unsigned d, int = 64, int site = __builtin_LINE()) { return emu_shfl_down(v, d, site); }
__host__ inline long long __shfl_down(long long v, unsigned d, int = 64, int site = __builtin_LINE()) { return emu_shfl_down(v, d, site); }
template <class T> __host__ inline T emu_readfirstlane_impl(T val, int site, bool divergent) {
    uint64_t bits = 0, mask;
    memcpy(&bits, &val, sizeof(T));
    const uint64_t *v = emu::wave_gather(bits, &mask, site, divergent);
    T r; memcpy(&r, &v[__builtin_ctzll(mask)], sizeof(T)); return r;
}
template <class T> __host__ inline T emu_readfirstlane(T val, int site = __builtin_LINE()) { return emu_readfirstlane_impl(val, site, false); }
template <class T> __host__ inline T emu_readfirstlane_div(T val, int site = __builtin_LINE()) { return emu_readfirstlane_impl(val, site, true); }
__host__ inline bool emu_inverse_ballot(unsigned long long m) { return (m >> emu::lane()) & 1; }
#define __builtin_amdgcn_readfirstlane(v) emu_readfirstlane(v)
#define __builtin_amdgcn_inverse_ballot_w64(m) emu_inverse_ballot(m)
// v_mbcnt_lo / _hi: base + the number of set mask bits that belong to lanes below this one (low / high half of the wave)
__host__ inline unsigned emu_mbcnt_lo(unsigned mask, unsigned base) { const int l = emu::lane(); return base + (unsigned)__builtin_popcount(l >= 32 ? mask : (mask & ((1u << l) - 1u))); }
__host__ inline unsigned emu_mbcnt_hi(unsigned mask, unsigned base) { const int l = emu::lane(); return base + (unsigned)(l > 32 ? __builtin_popcount(mask & ((1u << (l - 32)) - 1u)) : 0); }
#define __builtin_amdgcn_mbcnt_lo(m, b) emu_mbcnt_lo(m, b)
#define __builtin_amdgcn_mbcnt_hi(m, b) emu_mbcnt_hi(m, b)
// atomics: one OS thread, lanes switch only at rendezvous points
template <class T> __host__ inline T emu_atomic_add(T *p, T v) { T o = *p; *p = o + v; return o; }
__host__ inline int atomicAdd(int *p, int v) { return emu_atomic_add(p, v); }
__host__ inline unsigned atomicAdd(unsigned *p, unsigned v) { return emu_atomic_add(p, v); }
__host__ inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { return emu_atomic_add(p, v); }
__host__ inline float atomicAdd(float *p, float v) { return emu_atomic_add(p, v); }
__host__ inline int atomicOr(int *p, int v) { int o = *p; *p = o | v; return o; }
__host__ inline unsigned atomicOr(unsigned *p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
__host__ inline int atomicMax(int *p, int v) { int o = *p; if (v > o) *p = v; return o; }
__host__ inline unsigned long long atomicMax(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; if (v > o) *p = v; return o; }
__host__ inline int atomicMin(int *p, int v) { int o = *p; if (v < o) *p = v; return o; }
__host__ inline int atomicExch(int *p, int v) { int o = *p; *p = v; return o; }
__host__ inline int atomicCAS(int *p, int c, int v) { int o = *p; if (o == c) *p = v; return o; }
__host__ inline unsigned atomicCAS(unsigned *p, unsigned c, unsigned v) { unsigned o = *p; if (o == c) *p = v; return o; }
__host__ inline unsigned long long atomicCAS(unsigned long long *p, unsigned long long c, unsigned long long v) { unsigned long long o = *p; if (o == c) *p = v; return o; }

// ---- qualifiers ----
#undef __device__
#define __device__
#undef __global__
#define __global__
#undef __host__
#define __host__
#undef __forceinline__
#define __forceinline__ inline
#undef __launch_bounds__
#define __launch_bounds__(...)
#undef __shared__
#define __shared__ static
#define threadIdx (emu::thread_idx())
#define blockIdx (emu::block_idx())
#define blockDim (emu::block_dim())
#define gridDim (emu::grid_dim())
#define TR_SGPR_ATTR

// ---- kernel launch ----
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernel, grid, block, shm, stream, ...) emu::launch(#kernel, dim3(grid), dim3(block), (size_t)(shm), [&]() { kernel(__VA_ARGS__); })

// ---- the HIP runtime calls of the host code, on host memory ----
namespace emu {
inline hipError_t Malloc(void **p, size_t n) { *p = calloc(n ? n : 1, 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> inline hipError_t Malloc(T **p, size_t n) { return Malloc((void **)p, n); }
inline hipError_t Free(void *p) { free(p); return hipSuccess; }
inline hipError_t Memcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t MemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t MemcpyPeer(void *d, int, const void *s, int, size_t n) { if (n) memmove(d, s, n); return hipSuccess; }
inline hipError_t Memset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t MemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
inline hipError_t Ok() { return hipSuccess; }
inline hipError_t MemGetInfo(size_t *freeB, size_t *totalB) { *freeB = (size_t)64 << 30; *totalB = (size_t)64 << 30; return hipSuccess; }  // (host memory: calloc says no when there is none)
inline hipError_t StreamCreate(hipStream_t *s, unsigned = 0) { *s = (hipStream_t)(uintptr_t)8; return hipSuccess; }  // (never dereferenced)
inline hipError_t EventCreate(hipEvent_t *e, unsigned = 0) { *e = (hipEvent_t)(uintptr_t)8; return hipSuccess; }
inline hipError_t EventElapsed(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
// HIP_EMU_DEVICES=N pretends to N devices (all of them this host): the multi-device code paths run, one host thread per "device"
inline int device_count() { const char *e = getenv("HIP_EMU_DEVICES"); const int n = e ? atoi(e) : 1; return n < 1 ? 1 : n; }
inline int &current_device() { static thread_local int d = 0; return d; }
inline hipError_t DeviceCount(int *n) { *n = device_count(); return hipSuccess; }
inline hipError_t GetDevice(int *d) { *d = current_device(); return hipSuccess; }
inline hipError_t SetDevice(int d) { if (d < 0 || d >= device_count()) return hipErrorInvalidDevice; current_device() = d; return hipSuccess; }
inline hipError_t CanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
}  // namespace emu
#define hipMalloc emu::Malloc
#define hipFree emu::Free
#define hipMemcpy emu::Memcpy
#define hipMemcpyAsync emu::MemcpyAsync
#define hipMemcpyPeer emu::MemcpyPeer
#define hipMemset emu::Memset
#define hipMemGetInfo emu::MemGetInfo
#define hipMemsetAsync emu::MemsetAsync
#define hipStreamCreateWithFlags emu::StreamCreate
#define hipStreamCreate emu::StreamCreate
#define hipStreamDestroy(s) emu::Ok()
#define hipStreamSynchronize(s) emu::Ok()
#define hipStreamWaitEvent(...) emu::Ok()
#define hipDeviceSynchronize() emu::Ok()
#define hipEventCreate emu::EventCreate
#define hipEventCreateWithFlags emu::EventCreate
#define hipEventDestroy(e) emu::Ok()
#define hipEventRecord(...) emu::Ok()
#define hipEventSynchronize(e) emu::Ok()
#define hipEventElapsedTime emu::EventElapsed
#define hipGetDeviceCount emu::DeviceCount
#define hipGetDevice emu::GetDevice
#define hipSetDevice emu::SetDevice
#define hipGetLastError() emu::Ok()
#define hipPeekAtLastError() emu::Ok()
#define hipDeviceCanAccessPeer emu::CanAccessPeer
#define hipDeviceEnablePeerAccess(...) emu::Ok()
#endif
