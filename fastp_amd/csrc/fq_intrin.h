// fq_intrin.h - the gfx950 primitives the kernels use, behind short names.
// 64-wide wavefronts throughout (CDNA4): ballot masks are 64 bit.
#pragma once
#include <hip/hip_runtime.h>

#include "fq_types.h"

#define FQ_DEV __device__ __forceinline__

#ifdef FQ_HOSTSIM
extern uint32_t fq_lds[];   // the emulator's LDS (tests/hostsim/sim.cpp): what `extern __shared__ u32 fq_lds[]` names in a kernel
#endif

namespace fq {

FQ_DEV int thread_id() { return (int)threadIdx.x; }
FQ_DEV int block_threads() { return (int)blockDim.x; }
FQ_DEV int block_id() { return (int)blockIdx.x; }
FQ_DEV int grid_blocks() { return (int)gridDim.x; }
FQ_DEV int lane_id() { return (int)(threadIdx.x & 63); }
FQ_DEV int wave_id() { return (int)(threadIdx.x >> 6); }

// the kernel's by-value argument block as it sits in the kernarg segment (constant memory)
template <class T> FQ_DEV const T* kernel_args(const T* by_value) {
#ifdef FQ_HOSTSIM
    return by_value;
#else
    (void)by_value;
    return (const T*)__builtin_amdgcn_kernarg_segment_ptr();
#endif
}

// a pointer read from the argument block, pinned to scalar registers: without this a per-lane choice between
// two such pointers becomes ONE vector load of the chosen pointer (a load of a select of addresses) and a wait
// for it in the middle of the loads that should stay in flight
template <class T> FQ_DEV const T* scalar_ptr(const T* p) {
#ifdef FQ_HOSTSIM
    return p;
#else
    const unsigned long long v = (unsigned long long)p;
    const u32 lo = (u32)__builtin_amdgcn_readfirstlane((int)(u32)v), hi = (u32)__builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
    return (const T*)(((unsigned long long)hi << 32) | lo);
#endif
}

// Touch one dword so that its cache line is pulled into L2 / the Infinity Cache: a load the compiler does not
// see as one (no wait is scheduled for it).  The destination register belongs to the load until touch_done()
// has waited for it - call that before the value's register can be given to anything else.
FQ_DEV u32 touch_begin(const u32* p) {
#ifdef FQ_HOSTSIM
    return *p;
#else
    u32 v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
#endif
}
FQ_DEV void touch_done(u32 v) {
#ifdef FQ_HOSTSIM
    (void)v;
#else
    asm volatile("s_waitcnt vmcnt(0)\n; touched %0" : : "v"(v) : "memory");
#endif
}

// 16 bytes per lane from global memory straight into LDS (global_load_lds_dwordx4: no VGPR, asynchronous - counted by vmcnt).
// The LDS destination is wave-uniform base + lane * 16 whatever the lane asks for, so `lds_wave_base` must be uniform and the
// image lane-linear.  glds_wait() before anything reads the image.
FQ_DEV void glds16(const void* g, void* lds_wave_base, int lane) {
#ifdef FQ_HOSTSIM
    memcpy((char*)lds_wave_base + 16 * lane, g, 16);
#else
    (void)lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
#endif
}
// 4 bytes per lane the same way (global_load_lds_dword): what a line prefetch needs - the data is never read
FQ_DEV void glds4(const void* g, void* lds_wave_base, int lane) {
#ifdef FQ_HOSTSIM
    memcpy((char*)lds_wave_base + 4 * lane, g, 4);
#else
    (void)lane;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)lds_wave_base, 4, 0, 0);
#endif
}
FQ_DEV void glds_wait() {
#ifndef FQ_HOSTSIM
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
}

// a value that is the same in every lane of the wavefront, moved to a scalar register
FQ_DEV u32 uniform(u32 v) {
#ifdef FQ_HOSTSIM
    return v;
#else
    return (u32)__builtin_amdgcn_readfirstlane((int)v);
#endif
}
FQ_DEV void block_sync() { __syncthreads(); }
// Barrier over ONE HALF of the workgroup's wavefronts (the waves that share a tile): an arrival counter and a
// generation word in LDS, lane 0 of each wave arrives and then polls the generation with s_sleep between polls.
// gfx950 has one hardware barrier per workgroup; the two halves must be able to wait independently.
FQ_DEV void half_sync(u32* bar, int group, int nthreads, int naps) {
#ifdef FQ_HOSTSIM
    (void)bar;
    (void)naps;
    sim::group_barrier(group, nthreads);
#else
    (void)group;
    const u32 nwaves = (u32)nthreads >> 6;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wave's LDS writes are done before it arrives
    if ((threadIdx.x & 63) == 0) {
        const u32 gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const u32 arrived = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (arrived == nwaves - 1u) {   // last one: reset the count, then open the next generation (LDS executes a wave's operations in order)
            __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            // poll rarely: every poll is an LDS instruction and a few VALU ones taken from the other half's phases
            while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == gen) {
                if (naps <= 0) __builtin_amdgcn_s_sleep(1);
                else if (naps == 1) __builtin_amdgcn_s_sleep(4);
                else if (naps == 2) __builtin_amdgcn_s_sleep(16);
                else __builtin_amdgcn_s_sleep(64);
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#endif
}
FQ_DEV void nap() {   // ~3 us
    __builtin_amdgcn_s_sleep(127);
}
FQ_DEV u64 cycle_counter() { return (u64)clock64(); }
FQ_DEV void g_atomic_add_u64(u64* p, u64 v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// make this wave's LDS writes visible to its own other lanes (no cross-wave effect)
FQ_DEV void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// order this wave's LDS accesses for the compiler only: the LDS pipe executes one wave's DS instructions in issue
// order, so a later read by any lane of the wave sees an earlier write without draining the queue (lgkmcnt) first
FQ_DEV void wave_order() {
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// nothing is scheduled across this point (keeps two independent load sweeps from being interleaved, which doubles
// the registers in flight)
FQ_DEV void sched_fence() {
    __builtin_amdgcn_sched_barrier(0);
}
// no load or store moves across this point (compiler only): bounds how many table reads an unrolled loop keeps in flight
FQ_DEV void memory_fence_compiler() {
    asm volatile("" ::: "memory");
}

FQ_DEV u64 ballot(bool pred) { return __ballot(pred); }
FQ_DEV u32 shfl(u32 v, int src_lane) { return (u32)__shfl((int)v, src_lane, 64); }
// the value lane `src` holds, src the same in every lane (v_readlane_b32: no LDS crossbar trip)
FQ_DEV u32 read_lane(u32 v, u32 src) {
#ifndef FQ_HOSTSIM
    return (u32)__builtin_amdgcn_readlane((int)v, __builtin_amdgcn_readfirstlane((int)src));
#else
    return (u32)__shfl((int)v, (int)src, 64);
#endif
}
FQ_DEV u32 shfl_xor(u32 v, int mask) { return (u32)__shfl_xor((int)v, mask, 64); }
// lane exchanges inside a row of 16 lanes as DPP modifiers (no LDS round trip like ds_bpermute):
// the value of lane ^ 1, lane ^ 2 (quad_perm) and of the mirrored lane of the 8-lane half row
template <int CTRL> FQ_DEV u32 dpp_move(u32 v) {
    return (u32)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
FQ_DEV u32 quad_xor1(u32 v) { return dpp_move<0xB1>(v); }    // quad_perm:[1,0,3,2]
FQ_DEV u32 quad_xor2(u32 v) { return dpp_move<0x4E>(v); }    // quad_perm:[2,3,0,1]
FQ_DEV u32 half_mirror(u32 v) { return dpp_move<0x141>(v); } // row_half_mirror: lane i <-> 7 - i of each 8 lanes
// sum over the 4 (8) lanes of an aligned group, left in every lane of the group
FQ_DEV u32 sum4(u32 v) { v += quad_xor1(v); return v + quad_xor2(v); }
FQ_DEV u32 sum8(u32 v) { v = sum4(v); return v + half_mirror(v); }
FQ_DEV u64 sum4_u64(u64 v) {
    u64 o = (u64)quad_xor1((u32)v) | ((u64)quad_xor1((u32)(v >> 32)) << 32);
    v += o;
    o = (u64)quad_xor2((u32)v) | ((u64)quad_xor2((u32)(v >> 32)) << 32);
    return v + o;
}
FQ_DEV int popc32(u32 v) { return __popc(v); }
FQ_DEV int popc64(u64 v) { return __popcll(v); }
FQ_DEV int ffs64(u64 v) { return __ffsll((unsigned long long)v); }  // 1-based, 0 if none
FQ_DEV int ffs32(u32 v) { return __ffs((int)v); }                       // 1-based, 0 if none
FQ_DEV int clz32(u32 v) { return __clz((int)v); }                       // 32 for v == 0
FQ_DEV int clz64(u64 v) { const u32 hi = (u32)(v >> 32); return hi ? __clz((int)hi) : 32 + __clz((int)(u32)v); }   // 64 for v == 0
FQ_DEV u32 brev32(u32 v) { return __brev(v); }
// low 32 bits of ({hi,lo} >> (s & 31))  -> v_alignbit_b32
FQ_DEV u32 alignbit(u32 hi, u32 lo, u32 s) { return __builtin_amdgcn_alignbit(hi, lo, s); }
// sum of the four bytes of a, plus c  -> v_sad_u8 against zero
// 24-bit x 24-bit -> low 32 bits (full-rate v_mul_u32_u24)
FQ_DEV u32 mul24(u32 a, u32 b) { return __umul24(a, b); }
FQ_DEV u32 sum_bytes(u32 a, u32 c) { return __builtin_amdgcn_sad_u8(a, 0u, c); }
// four sliding 4-byte sums at once: 16-bit field i = bytes i .. i+3 of the 64-bit value {hi, lo} summed, plus field i of acc
// -> v_qsad_pk_u16_u8 against zero
FQ_DEV u64 sum_bytes_sliding4(u32 lo, u32 hi, u64 acc) { return __builtin_amdgcn_qsad_pk_u16_u8((u64)lo | ((u64)hi << 32), 0u, acc); }
// (v >> off) & ((1 << width) - 1)  -> v_bfe_u32 (kept as one instruction next to the shift-add that uses it)
FQ_DEV u32 bfe(u32 v, u32 off, u32 width) { return __builtin_amdgcn_ubfe(v, off, width); }
// a * b + c on 24-bit operands, b uniform (a scalar register or an inline constant), pinned as ONE v_mad_u32_u24: left to itself the
// compiler re-associates an address a * S1 + b * S2 + base into multiplies and three-operand adds that re-add the base's parts per
// element (fq_stats5.h: 6 instead of 4 instructions per base)
FQ_DEV u32 mad24_su(u32 a, u32 b_uniform, u32 c) {
#ifdef FQ_HOSTSIM
    return (a & 0xFFFFFFu) * (b_uniform & 0xFFFFFFu) + c;
#else
    u32 r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
#endif
}
// (x << SH) + y as ONE v_lshl_add_u32
template <int SH> FQ_DEV u32 lshl_add(u32 x, u32 y) {
#ifdef FQ_HOSTSIM
    return (x << SH) + y;
#else
    u32 r;
    asm("v_lshl_add_u32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "n"(SH), "v"(y));
    return r;
#endif
}
// LDS by its 32-bit address (what a DS instruction takes): a generic pointer + an offset costs an add of the LDS aperture's base
// per access even where that base is zero
FQ_DEV u32 lds_addr_of(const void* p) {
#ifdef FQ_HOSTSIM
    return (u32)(size_t)((const char*)p - (const char*)::fq_lds);
#else
    return (u32)(size_t)((__attribute__((address_space(3))) const char*)p);
#endif
}
FQ_DEV void lds_add_u32_at(u32 addr, u32 v) {
#ifdef FQ_HOSTSIM
    __hip_atomic_fetch_add((u32*)((char*)::fq_lds + addr), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#else
    __hip_atomic_fetch_add((__attribute__((address_space(3))) u32*)(size_t)addr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#endif
}
FQ_DEV u32 lds_read_u8_at(u32 addr) {
#ifdef FQ_HOSTSIM
    return (u32)*((const u8*)::fq_lds + addr);
#else
    return (u32)*(const __attribute__((address_space(3))) u8*)(size_t)addr;
#endif
}
FQ_DEV u32 lds_read_u32_at(u32 addr) {
#ifdef FQ_HOSTSIM
    return *(const u32*)((const u8*)::fq_lds + addr);
#else
    return *(const __attribute__((address_space(3))) u32*)(size_t)addr;
#endif
}
// the value as it is, but opaque to the optimiser: what is computed from it stays computed from it
FQ_DEV u32 opaque(u32 v) {
#ifndef FQ_HOSTSIM
    asm("" : "+v"(v));
#endif
    return v;
}
// c + sum of the four byte products a.b[k] * b.b[k]  -> v_dot4_u32_u8
FQ_DEV u32 dot4_u8(u32 a, u32 b, u32 c) { return __builtin_amdgcn_udot4(a, b, c, false); }

// LDS accumulators: fire-and-forget ds_add / ds_or (no return value used)
FQ_DEV void lds_add_u32(u32* p, u32 v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
FQ_DEV u32 lds_add_ret_u32(u32* p, u32 v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
FQ_DEV void lds_add_u64(u64* p, u64 v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
FQ_DEV void lds_min_u32(u32* p, u32 v) {
    __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
FQ_DEV void lds_or_u32(u32* p, u32 v) {
    __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
FQ_DEV void lds_or_i32(int* p, int v) {
    __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// device-scope atomics on global memory
FQ_DEV int g_atomic_add_i32(int* p, int v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FQ_DEV u32 g_atomic_add_u32(u32* p, u32 v) {
    return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FQ_DEV u32 g_atomic_exch_u32(u32* p, u32 v) {
    return __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FQ_DEV u32 g_atomic_min_u32(u32* p, u32 v) {
    return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FQ_DEV u32 g_atomic_max_u32(u32* p, u32 v) {
    return __hip_atomic_fetch_max(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FQ_DEV u32 g_atomic_or_u32(u32* p, u32 v) {
    return __hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FQ_DEV u64 g_atomic_min_u64(u64* p, u64 v) {
    return __hip_atomic_fetch_min(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
FQ_DEV u64 g_atomic_cas_u64(u64* p, u64 expected, u64 desired) {
    __hip_atomic_compare_exchange_strong(p, &expected, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED,
                                         __HIP_MEMORY_SCOPE_AGENT);
    return expected;
}
FQ_DEV void g_atomic_add_i64(int64_t* p, int64_t v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace fq
