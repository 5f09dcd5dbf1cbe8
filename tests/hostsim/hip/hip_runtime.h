// FAKE <hip/hip_runtime.h> - TEST INFRASTRUCTURE ONLY (tests/hostsim).
//
// Lets g++ compile the PRODUCT sources (fastp_amd/csrc/*.hip, *.h) unchanged into
// a host library in which every kernel launch is executed by a lock-step SIMT
// emulator (sim.cpp): one coroutine per GPU thread, 64-lane wavefront collectives
// (__ballot/__shfl/wave barrier) and __syncthreads implemented as rendezvous
// points, threads scheduled in a shuffled order between rendezvous points so a
// missing barrier shows up as a wrong answer.  It exists so the device code can be
// checked against the CPU oracle in the CPU-only container; it is never shipped,
// never loaded by fastp_amd, and is NOT a fallback path of the engine.
#pragma once
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <functional>

#define FASTP_HOSTSIM 1

// ---- function / storage qualifiers -------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __shared__
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

namespace sim {
struct ThreadState;
extern ThreadState* cur;
struct Idx { unsigned x, y, z; };
const Idx& thread_idx();
const Idx& block_idx();
const Idx& block_dim();
const Idx& grid_dim();
void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body);
void syncthreads();
void wave_barrier();
void group_barrier(int group, int nthreads);   // barrier over threads [group * nthreads, (group + 1) * nthreads)
unsigned long long ballot(bool pred);
int shfl(int v, int src_lane);
int shfl_xor(int v, int mask);
}  // namespace sim

#define FQ_HOSTSIM 1
#define threadIdx (sim::thread_idx())
#define blockIdx (sim::block_idx())
#define blockDim (sim::block_dim())
#define gridDim (sim::grid_dim())

static inline void __syncthreads() { sim::syncthreads(); }
static inline unsigned long long __ballot(int pred) { return sim::ballot(pred != 0); }
static inline int __shfl(int v, int src, int width = 64) { (void)width; return sim::shfl(v, src); }
static inline int __shfl_xor(int v, int mask, int width = 64) { (void)width; return sim::shfl_xor(v, mask); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __clz(int v) { return v == 0 ? 32 : __builtin_clz((unsigned)v); }
static inline unsigned __brev(unsigned v) {
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
static inline long long clock64() { return 0; }
static inline unsigned __umul24(unsigned a, unsigned b) { return (a & 0xFFFFFFu) * (b & 0xFFFFFFu); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

// ---- gfx950 builtins used by fq_intrin.h -----------------------------------------
static inline unsigned sim_alignbit(unsigned hi, unsigned lo, unsigned s) {
    return (unsigned)((((unsigned long long)hi << 32) | lo) >> (s & 31));
}
static inline unsigned sim_sad_u8(unsigned a, unsigned b, unsigned c) {
    unsigned r = c;
    for (int i = 0; i < 4; i++) {
        int x = (a >> (8 * i)) & 0xFF, y = (b >> (8 * i)) & 0xFF;
        r += (unsigned)(x > y ? x - y : y - x);
    }
    return r;
}
static inline unsigned sim_udot4(unsigned a, unsigned b, unsigned c) {
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xFF) * ((b >> (8 * i)) & 0xFF);
    return c;
}
#define __builtin_amdgcn_udot4(a, b, c, clamp) sim_udot4((a), (b), (c))
static inline unsigned sim_ubfe(unsigned v, unsigned off, unsigned width) {
    off &= 31; width &= 31;
    return width == 0 ? 0u : ((v >> off) & ((1u << width) - 1u));
}
#define __builtin_amdgcn_ubfe(v, off, width) sim_ubfe((v), (off), (width))
// DPP lane exchanges used by fq_intrin.h: quad_perm (ctrl < 0x100), row_half_mirror (0x141), row_mirror (0x140)
static inline int sim_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    (void)old; (void)row_mask; (void)bank_mask; (void)bound_ctrl;
    const int lane = (int)(sim::thread_idx().x & 63u);
    int from = lane;
    if (ctrl < 0x100) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - (lane & 15));
    else abort();
    return sim::shfl(src, from);
}
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rm, bm, bc) sim_update_dpp((old), (src), (ctrl), (rm), (bm), (bc))
#define __builtin_amdgcn_alignbit(hi, lo, s) sim_alignbit((hi), (lo), (s))
#define __builtin_amdgcn_sad_u8(a, b, c) sim_sad_u8((a), (b), (c))
// v_qsad_pk_u16_u8: field i (16 bits) = SAD of the four bytes of s0 >> 8i against the bytes of s1, plus field i of acc
static inline unsigned long long sim_qsad_pk_u16_u8(unsigned long long s0, unsigned int s1, unsigned long long acc) {
    unsigned long long out = 0;
    for (int i = 0; i < 4; i++) {
        unsigned int sad = 0;
        for (int k = 0; k < 4; k++) {
            const int a = (int)((s0 >> (8 * (i + k))) & 0xFFu), b = (int)((s1 >> (8 * k)) & 0xFFu);
            sad += (unsigned int)(a > b ? a - b : b - a);
        }
        out |= (unsigned long long)((sad + (unsigned int)((acc >> (16 * i)) & 0xFFFFu)) & 0xFFFFu) << (16 * i);
    }
    return out;
}
#define __builtin_amdgcn_qsad_pk_u16_u8(s0, s1, acc) sim_qsad_pk_u16_u8((s0), (s1), (acc))
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)   // scheduling hints: nothing to emulate
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_wave_barrier() sim::wave_barrier()

#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
template <class T> static inline T __hip_atomic_load(T* p, int, int) { return *p; }
template <class T, class V> static inline void __hip_atomic_store(T* p, V v, int, int) { *p = (T)v; }
template <class T, class V> static inline T __hip_atomic_fetch_add(T* p, V v, int, int) { T o = *p; *p = (T)(o + (T)v); return o; }
template <class T, class V> static inline T __hip_atomic_fetch_or(T* p, V v, int, int) { T o = *p; *p = (T)(o | (T)v); return o; }
template <class T, class V> static inline T __hip_atomic_exchange(T* p, V v, int, int) { T o = *p; *p = (T)v; return o; }
template <class T, class V> static inline T __hip_atomic_fetch_min(T* p, V v, int, int) { T o = *p; if ((T)v < o) *p = (T)v; return o; }
template <class T, class V> static inline T __hip_atomic_fetch_max(T* p, V v, int, int) { T o = *p; if ((T)v > o) *p = (T)v; return o; }
template <class T> static inline bool __hip_atomic_compare_exchange_strong(T* p, T* expected, T desired, int, int, int) {
    if (*p == *expected) { *p = desired; return true; }
    *expected = *p;
    return false;
}

// ---- the sliver of the HIP runtime API that fastp_gpu.hip calls -------------------
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorNotReady = 600, hipErrorUnknown = 999 };
typedef struct sim_stream* hipStream_t;
typedef struct sim_event* hipEvent_t;
enum { hipStreamNonBlocking = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
struct hipDeviceProp_t {
    int multiProcessorCount;
    size_t sharedMemPerBlock;
};
static inline const char* hipGetErrorString(hipError_t) { return "hostsim error"; }
enum { hipEventDisableTiming = 2 };
hipError_t hipGetDeviceCount(int* n);
hipError_t hipSetDevice(int);
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int dev);
hipError_t hipMalloc(void** p, size_t bytes);
hipError_t hipFree(void* p);
hipError_t hipHostMalloc(void** p, size_t bytes);
hipError_t hipHostFree(void* p);
hipError_t hipEventQuery(hipEvent_t e);
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind k, hipStream_t st);
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st);
hipError_t hipMemset(void* d, int v, size_t n);
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t words, const uint32_t* mask);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags);
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned flags);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipDeviceSynchronize();
hipError_t hipEventCreate(hipEvent_t* e);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void* f, hipFuncAttribute a, int v);
hipError_t hipGetLastError();
inline hipError_t hipMemGetInfo(size_t* free_bytes, size_t* total_bytes) { *free_bytes = (size_t)8 << 30; *total_bytes = (size_t)16 << 30; return hipSuccess; }   // (a small card: the emulator keeps the small lists)

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    sim::launch((grid), (block), (shmem), [&]() { kernel(__VA_ARGS__); })
