// sim.cpp - lock-step SIMT emulator behind the fake hip_runtime.h (TEST INFRASTRUCTURE ONLY).
// One ucontext coroutine per GPU thread; blocks run one after another; inside a block the
// runnable threads are resumed in a shuffled order and run until their next rendezvous
// (__syncthreads, wave barrier, __ballot, __shfl*) or until they return.
#include <stdio.h>
#include <sys/mman.h>
#include <ucontext.h>

#include <algorithm>
#include <chrono>
#include <mutex>
#include <vector>

#include "hip/hip_runtime.h"

// the dynamic LDS of the block being emulated (`extern __shared__ u32 fq_lds[]` in the kernel)
alignas(16) uint32_t fq_lds[(160 * 1024) / 4 + 64];

namespace sim {

enum Wait { RUN = 0, AT_BLOCK = 1, AT_WAVE_BARRIER = 2, AT_BALLOT = 3, AT_SHFL = 4, AT_SHFL_XOR = 5, DONE = 6, AT_GROUP = 7 };

struct ThreadState {
    ucontext_t ctx;
    char* stack;
    Idx tid;
    int wait;
    int val, arg;               // exchange: value / src lane or xor mask / predicate
    unsigned long long res64;
    int res32;
};

ThreadState* cur = nullptr;
static ucontext_t sched_ctx;
static Idx g_block, g_bdim, g_gdim;
static const std::function<void()>* g_body = nullptr;
static unsigned g_seed = 12345;
static const size_t STACK = 192 * 1024;

const Idx& thread_idx() { return cur->tid; }
const Idx& block_idx() { return g_block; }
const Idx& block_dim() { return g_bdim; }
const Idx& grid_dim() { return g_gdim; }

static void yield_to_sched(int why) {
    cur->wait = why;
    ThreadState* me = cur;
    swapcontext(&me->ctx, &sched_ctx);
}

void syncthreads() { yield_to_sched(AT_BLOCK); }
void wave_barrier() { yield_to_sched(AT_WAVE_BARRIER); }
void group_barrier(int group, int nthreads) { cur->val = group; cur->arg = nthreads; yield_to_sched(AT_GROUP); }
unsigned long long ballot(bool pred) { cur->val = pred ? 1 : 0; yield_to_sched(AT_BALLOT); return cur->res64; }
int shfl(int v, int src) { cur->val = v; cur->arg = src & 63; yield_to_sched(AT_SHFL); return cur->res32; }
int shfl_xor(int v, int mask) { cur->val = v; cur->arg = mask; yield_to_sched(AT_SHFL_XOR); return cur->res32; }

static void trampoline() {
    (*g_body)();
    cur->wait = DONE;
    swapcontext(&cur->ctx, &sched_ctx);
}

static unsigned rnd() { g_seed = g_seed * 1664525u + 1013904223u; return g_seed >> 8; }

static void die(const char* msg) {
    fprintf(stderr, "hostsim: %s\n", msg);
    abort();
}

static void run_block(std::vector<ThreadState>& th) {
    const int T = (int)th.size();
    const int waves = (T + 63) / 64;
    for (int t = 0; t < T; t++) {
        th[t].wait = RUN;
        getcontext(&th[t].ctx);
        th[t].ctx.uc_stack.ss_sp = th[t].stack;
        th[t].ctx.uc_stack.ss_size = STACK;
        th[t].ctx.uc_link = &sched_ctx;
        makecontext(&th[t].ctx, trampoline, 0);
    }
    std::vector<int> order(T);
    for (int t = 0; t < T; t++) order[t] = t;
    for (;;) {
        // resume every runnable thread once, in a shuffled order
        for (int i = T - 1; i > 0; i--) std::swap(order[i], order[rnd() % (unsigned)(i + 1)]);
        bool progressed = false;
        for (int k = 0; k < T; k++) {
            ThreadState& s = th[order[k]];
            if (s.wait != RUN) continue;
            cur = &s;
            swapcontext(&sched_ctx, &s.ctx);
            progressed = true;
        }
        // wave-level rendezvous
        for (int w = 0; w < waves; w++) {
            const int lo = w * 64, hi = std::min(T, lo + 64);
            int kind = -1, live = 0, arrived = 0;
            for (int t = lo; t < hi; t++) {
                if (th[t].wait == DONE) continue;
                live++;
                const int wt = th[t].wait;
                if (wt == AT_WAVE_BARRIER || wt == AT_BALLOT || wt == AT_SHFL || wt == AT_SHFL_XOR) {
                    if (kind == -1) kind = wt;
                    else if (kind != wt) die("lanes of one wavefront are in different collectives (divergent collective)");
                    arrived++;
                }
            }
            if (live == 0 || arrived == 0) continue;
            if (arrived < live) {
                // the rest of the wave may be parked at __syncthreads: that is a divergence bug
                for (int t = lo; t < hi; t++)
                    if (th[t].wait == AT_BLOCK) die("wave collective while other lanes of the wave wait at __syncthreads");
                continue;
            }
            if (kind == AT_BALLOT) {
                unsigned long long m = 0;
                for (int t = lo; t < hi; t++)
                    if (th[t].wait == AT_BALLOT && th[t].val) m |= 1ull << (t - lo);
                for (int t = lo; t < hi; t++)
                    if (th[t].wait == AT_BALLOT) th[t].res64 = m;
            } else if (kind == AT_SHFL || kind == AT_SHFL_XOR) {
                for (int t = lo; t < hi; t++) {
                    if (th[t].wait != kind) continue;
                    const int src = kind == AT_SHFL ? th[t].arg : ((t - lo) ^ th[t].arg);
                    const int st = lo + (src & 63);
                    th[t].res32 = (st < hi && th[st].wait == kind) ? th[st].val : th[t].val;
                }
            }
            for (int t = lo; t < hi; t++)
                if (th[t].wait == kind) th[t].wait = RUN;
            progressed = true;
        }
        // group-level rendezvous (half-workgroup barriers): a group is released when all of its live threads wait
        for (int t0 = 0; t0 < T;) {
            if (th[t0].wait != AT_GROUP) { t0++; continue; }
            const int size = th[t0].arg, g = th[t0].val;
            const int lo = g * size, hi = std::min(T, lo + size);
            int glive = 0, gat = 0;
            for (int t = lo; t < hi; t++) {
                if (th[t].wait == DONE) continue;
                glive++;
                if (th[t].wait == AT_GROUP && th[t].val == g && th[t].arg == size) gat++;
            }
            if (glive > 0 && gat == glive) {
                for (int t = lo; t < hi; t++)
                    if (th[t].wait == AT_GROUP) th[t].wait = RUN;
                progressed = true;
            }
            t0 = hi > t0 ? hi : t0 + 1;
        }
        // block-level rendezvous
        int live = 0, at_block = 0;
        for (int t = 0; t < T; t++) {
            if (th[t].wait == DONE) continue;
            live++;
            if (th[t].wait == AT_BLOCK) at_block++;
        }
        if (live == 0) return;
        if (at_block == live) {
            for (int t = 0; t < T; t++)
                if (th[t].wait == AT_BLOCK) th[t].wait = RUN;
            progressed = true;
        }
        if (!progressed) die("deadlock: threads wait at different rendezvous points");
    }
}

void launch(dim3 grid, dim3 block, size_t shmem, const std::function<void()>& body) {
    // one launch at a time: the emulator's state (coroutine pool, LDS, indices) is global, and host code may drive two
    // contexts from two threads (the two ranks of tests/test_comm_stub.py, the stream loop's reader beside its caller)
    static std::mutex launch_mu;
    std::lock_guard<std::mutex> launch_lock(launch_mu);
    if (shmem > sizeof(fq_lds)) die("dynamic LDS request exceeds 160 KiB");
    static std::vector<ThreadState> pool;
    const int T = (int)block.x;
    while ((int)pool.size() < T) {
        ThreadState s;
        memset(&s, 0, sizeof(s));
        s.stack = (char*)malloc(STACK);
        pool.push_back(s);
    }
    std::vector<ThreadState> th(pool.begin(), pool.begin() + T);
    g_bdim = {block.x, 1, 1};
    g_gdim = {grid.x, 1, 1};
    g_body = &body;
    // FASTP_SIM_REVERSE_BLOCKS: workgroups run last to first (execution order != index order, as on a GPU)
    const bool reverse = getenv("FASTP_SIM_REVERSE_BLOCKS") != nullptr;
    for (unsigned bi = 0; bi < grid.x; bi++) {
        const unsigned b = reverse ? grid.x - 1 - bi : bi;
        g_block = {b, 0, 0};
        memset(fq_lds, 0xA5, shmem);  // LDS is NOT zero on entry
        for (int t = 0; t < T; t++) th[t].tid = {(unsigned)t, 0, 0};
        run_block(th);
    }
    g_body = nullptr;
    cur = nullptr;
}

}  // namespace sim

// ---- fake runtime ---------------------------------------------------------------------
struct sim_stream { int dummy; };
struct sim_event { std::chrono::steady_clock::time_point t; };

hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    const char* cu = getenv("FASTP_SIM_CUS");
    p->multiProcessorCount = cu ? atoi(cu) : 3;  // few "CUs" -> the grid-stride loop is exercised
    p->sharedMemPerBlock = 160 * 1024;
    return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) {
    // lazily committed so the 1 GiB duplicate bitmaps cost nothing until touched
    *p = calloc(bytes ? bytes : 1, 1);
    return *p ? hipSuccess : hipErrorUnknown;
}
hipError_t hipFree(void* p) { free(p); return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t bytes) { *p = calloc(bytes ? bytes : 1, 1); return *p ? hipSuccess : hipErrorUnknown; }
hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }   // launches run to completion inside hipLaunchKernelGGL
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t) {
    if (width > dpitch || width > spitch) return hipErrorUnknown;
    for (size_t r = 0; r < height; r++) memcpy((char*)d + r * dpitch, (const char*)s + r * spitch, width);
    return hipSuccess;
}
// Clearing Duplicate's 1 GiB of bitmaps with memset touches every page (a quarter of a million faults per engine that is
// created or reset: minutes of system time over the suite).  A large zero fill gives the pages back instead: the blocks
// come from calloc, i.e. private anonymous mappings, which read as zeros again after MADV_DONTNEED.
static void sim_fill(void* d, int v, size_t n) {
    const size_t page = 4096;
    if (v == 0 && n >= ((size_t)8 << 20)) {
        const uintptr_t a = ((uintptr_t)d + page - 1) & ~(uintptr_t)(page - 1), e = ((uintptr_t)d + n) & ~(uintptr_t)(page - 1);
        if (e > a && madvise((void*)a, e - a, MADV_DONTNEED) == 0) {
            memset(d, 0, a - (uintptr_t)d);
            memset((void*)e, 0, (uintptr_t)d + n - e);
            return;
        }
    }
    memset(d, v, n);
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { sim_fill(d, v, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { sim_fill(d, v, n); return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = new sim_stream(); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { delete s; return hipSuccess; }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { *s = new sim_stream(); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new sim_event(); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = new sim_event(); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
    *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
