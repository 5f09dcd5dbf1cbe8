// fq_comm.cpp - the multi-GPU exchange steps of the engine behind the C ABI (include/fastp_gpu.h,
// "Collectives"): RCCL over xGMI.  What is replaced: the merge loop at the end of a run,
// PairEndProcessor::process src/peprocessor.cpp:217-234 -> Stats::merge src/stats.cpp:877-955 and
// FilterResult::merge src/filterresult.cpp:38-89 (an int64 sum over the workers' counter arrays), and - for the
// one piece of worker-loop state that is shared between workers, Duplicate's bloom bitmaps
// (src/duplicate.h:34-37) - the rank-ordered exclusive prefix-OR the exact sharded protocol needs.
//
// Built on the public entry points only (counter block pointer, bitmap export, prefix set); librccl is
// dlopen'ed at the first use so that a single-GPU host needs no RCCL at all, and so that a process that already
// carries a copy of the library (PyTorch ships one) binds to that same copy instead of loading a second one.
#include <dlfcn.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/fastp_gpu.h"

namespace {

struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};

Rccl g_rccl;
std::mutex g_mu;
thread_local std::string t_err;

bool load_rccl() {
    if (g_rccl.handle) return true;
    // FASTP_GPU_RCCL_LIB names the library explicitly (a site's own build; the test suite's in-process stand-in)
    const char* names[] = {getenv("FASTP_GPU_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
        if (!n || !*n) continue;
        g_rccl.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (g_rccl.handle) break;
    }
    if (!g_rccl.handle) { g_rccl.err = std::string("librccl not loadable: ") + dlerror(); return false; }
#define FQ_SYM(field, name)                                                        \
    *(void**)(&g_rccl.field) = dlsym(g_rccl.handle, name);                         \
    if (!g_rccl.field) { g_rccl.err = std::string("librccl lacks ") + name; dlclose(g_rccl.handle); g_rccl.handle = nullptr; return false; }
    FQ_SYM(GetUniqueId, "ncclGetUniqueId")
    FQ_SYM(CommInitRank, "ncclCommInitRank")
    FQ_SYM(CommInitAll, "ncclCommInitAll")
    FQ_SYM(CommDestroy, "ncclCommDestroy")
    FQ_SYM(AllReduce, "ncclAllReduce")
    FQ_SYM(Send, "ncclSend")
    FQ_SYM(Recv, "ncclRecv")
    FQ_SYM(GroupStart, "ncclGroupStart")
    FQ_SYM(GroupEnd, "ncclGroupEnd")
    FQ_SYM(GetErrorString, "ncclGetErrorString")
#undef FQ_SYM
    return true;
}

struct Comm {
    ncclComm_t comm = nullptr;
    hipStream_t stream = nullptr;
    int nranks = 0, rank = 0, device = 0;
    void* images = nullptr;   // exchange buffers of the duplicate prefix step (allocated on first use)
    void* slices = nullptr;
    int64_t image_bytes = 0;
    int users = 0;            // collectives holding a pointer to this entry outside g_mu (Pin); drop_comm waits for 0
};
std::map<fastp_gpu_ctx*, Comm> g_comms;
std::condition_variable g_idle;   // signalled when a Comm's `users` falls to 0

int fail(int code, const std::string& msg) {
    t_err = msg;
    return code;
}
#define NCCL_TRY(call)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (call);                                                                        \
        if (r_ != ncclSuccess) return fail(FASTP_GPU_E_HIP, std::string(#call) + ": " + g_rccl.GetErrorString(r_)); \
    } while (0)
#define HIPC_TRY(call)                                                                                   \
    do {                                                                                                 \
        hipError_t e_ = (call);                                                                          \
        if (e_ != hipSuccess) return fail(FASTP_GPU_E_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// ncclGroupStart ... ncclGroupEnd with the end guaranteed on every exit path (an open group would swallow the
// process's next collective)
struct Group {
    bool open = false;
    ncclResult_t start() { const ncclResult_t r = g_rccl.GroupStart(); open = r == ncclSuccess; return r; }
    ncclResult_t end() { open = false; return g_rccl.GroupEnd(); }
    ~Group() { if (open) (void)g_rccl.GroupEnd(); }
};

void drop_comm(fastp_gpu_ctx* ctx) {
    std::unique_lock<std::mutex> lk(g_mu);
    auto it = g_comms.find(ctx);
    if (it == g_comms.end()) return;
    // a collective of another thread may be using this entry outside the lock: the entry (and its RCCL communicator,
    // stream and exchange buffers) goes only when that call has returned (std::map nodes do not move meanwhile)
    g_idle.wait(lk, [&] { return it->second.users == 0; });
    Comm& c = it->second;
    (void)hipSetDevice(c.device);
    if (c.comm && g_rccl.handle) (void)g_rccl.CommDestroy(c.comm);
    if (c.stream) (void)hipStreamDestroy(c.stream);
    if (c.images) (void)hipFree(c.images);
    if (c.slices) (void)hipFree(c.slices);
    g_comms.erase(it);
}

Comm* find_comm(fastp_gpu_ctx* ctx) {
    auto it = g_comms.find(ctx);
    return it == g_comms.end() ? nullptr : &it->second;
}

// The communicators of one collective call, looked up and pinned under g_mu, released (and drop_comm woken) when the
// call returns on whatever path.
struct Pin {
    std::vector<Comm*> cs;
    bool take(fastp_gpu_ctx* const* ctxs, int n) {
        std::lock_guard<std::mutex> lk(g_mu);
        for (int i = 0; i < n; i++) {
            Comm* c = ctxs[i] ? find_comm(ctxs[i]) : nullptr;
            if (!c) { release_locked(); return false; }
            c->users++;
            cs.push_back(c);
        }
        return true;
    }
    void release_locked() {
        for (Comm* c : cs) c->users--;
        cs.clear();
        g_idle.notify_all();
    }
    ~Pin() {
        if (cs.empty()) return;
        std::lock_guard<std::mutex> lk(g_mu);
        release_locked();
    }
};

}  // namespace

extern "C" {

extern void (*fastp_gpu_comm_destroy_hook)(fastp_gpu_ctx*);  // fastp_gpu.hip: called by fastp_gpu_destroy

const char* fastp_gpu_comm_last_error(void) { return t_err.c_str(); }

int fastp_gpu_comm_id(uint8_t id[FASTP_GPU_COMM_ID_BYTES]) {
    if (!id) return fail(FASTP_GPU_E_INVALID, "null argument");
    std::lock_guard<std::mutex> lk(g_mu);
    if (!load_rccl()) return fail(FASTP_GPU_E_UNSUPPORTED, g_rccl.err);
    static_assert(sizeof(ncclUniqueId) == FASTP_GPU_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId u;
    NCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, &u, sizeof(u));
    return FASTP_GPU_OK;
}

int fastp_gpu_comm_init(fastp_gpu_ctx* ctx, const uint8_t id[FASTP_GPU_COMM_ID_BYTES], int nranks, int rank) {
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return fail(FASTP_GPU_E_INVALID, "bad argument");
    drop_comm(ctx);
    std::lock_guard<std::mutex> lk(g_mu);
    if (!load_rccl()) return fail(FASTP_GPU_E_UNSUPPORTED, g_rccl.err);
    Comm c;
    c.nranks = nranks;
    c.rank = rank;
    c.device = fastp_gpu_device(ctx);
    HIPC_TRY(hipSetDevice(c.device));
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    NCCL_TRY(g_rccl.CommInitRank(&c.comm, nranks, u, rank));
    if (hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking) != hipSuccess) {
        (void)g_rccl.CommDestroy(c.comm);
        return fail(FASTP_GPU_E_HIP, "hipStreamCreateWithFlags failed");
    }
    g_comms[ctx] = c;
    fastp_gpu_comm_destroy_hook = drop_comm;
    return FASTP_GPU_OK;
}

int fastp_gpu_comm_init_local(fastp_gpu_ctx* const* ctxs, int n) {
    if (!ctxs || n < 1) return fail(FASTP_GPU_E_INVALID, "bad argument");
    std::vector<int> devs(n);
    for (int i = 0; i < n; i++) {
        if (!ctxs[i]) return fail(FASTP_GPU_E_INVALID, "null context");
        devs[i] = fastp_gpu_device(ctxs[i]);
        for (int j = 0; j < i; j++)
            if (devs[j] == devs[i]) return fail(FASTP_GPU_E_INVALID, "two contexts on one device: a communicator needs one GPU per rank");
        drop_comm(ctxs[i]);
    }
    std::lock_guard<std::mutex> lk(g_mu);
    if (!load_rccl()) return fail(FASTP_GPU_E_UNSUPPORTED, g_rccl.err);
    std::vector<ncclComm_t> comms(n);
    NCCL_TRY(g_rccl.CommInitAll(comms.data(), n, devs.data()));
    for (int i = 0; i < n; i++) {
        Comm c;
        c.comm = comms[i];
        c.nranks = n;
        c.rank = i;
        c.device = devs[i];
        HIPC_TRY(hipSetDevice(c.device));
        HIPC_TRY(hipStreamCreateWithFlags(&c.stream, hipStreamNonBlocking));
        g_comms[ctxs[i]] = c;
    }
    fastp_gpu_comm_destroy_hook = drop_comm;
    return FASTP_GPU_OK;
}

void fastp_gpu_comm_destroy(fastp_gpu_ctx* ctx) { drop_comm(ctx); }

// Stats::merge / FilterResult::merge: every rank's counter block becomes the sum over the ranks (the header
// words are restored).  ctxs = the contexts THIS process owns (1 with one process per GPU).
int fastp_gpu_allreduce(fastp_gpu_ctx* const* ctxs, int n) {
    if (!ctxs || n < 1) return fail(FASTP_GPU_E_INVALID, "bad argument");
    // g_mu guards the context -> communicator map only: the collectives below block on the other ranks, and a process
    // that drives its ranks from separate threads (each calling with n = 1) must not serialise them behind one mutex.
    // A context's communicator is used by one collective at a time (the caller's contract, as for every fastp_gpu_* call);
    // destroying it from another thread meanwhile is safe: the entries are pinned for the length of the call.
    Pin pin;   // a concurrent fastp_gpu_comm_destroy / fastp_gpu_destroy of one of the contexts waits for this call
    if (!pin.take(ctxs, n)) return fail(FASTP_GPU_E_INVALID, "context has no communicator (fastp_gpu_comm_init*)");
    const std::vector<Comm*>& cs = pin.cs;
    std::vector<int64_t*> ptr(n);
    std::vector<int64_t> cnt(n);
    for (int i = 0; i < n; i++) {
        int rc = fastp_gpu_synchronize(ctxs[i]);   // every launch and slab fold of this context is complete
        if (rc) return fail(rc, fastp_gpu_last_error(ctxs[i]));
        rc = fastp_gpu_counters_device(ctxs[i], &ptr[i], &cnt[i], nullptr);
        if (rc) return fail(rc, fastp_gpu_last_error(ctxs[i]));
        if (cnt[i] != cnt[0]) return fail(FASTP_GPU_E_INVALID, "contexts with different counter layouts");
    }
    // header words (ABI version, cycles, insert-size bound) are identical on every rank: keep one copy
    int64_t hdr[4];
    HIPC_TRY(hipSetDevice(cs[0]->device));
    HIPC_TRY(hipMemcpy(hdr, ptr[0], sizeof(hdr), hipMemcpyDeviceToHost));
    {
        Group grp;
        if (n > 1) NCCL_TRY(grp.start());
        for (int i = 0; i < n; i++) {
            HIPC_TRY(hipSetDevice(cs[i]->device));
            NCCL_TRY(g_rccl.AllReduce(ptr[i], ptr[i], (size_t)cnt[i], ncclInt64, ncclSum, cs[i]->comm, cs[i]->stream));
        }
        if (n > 1) NCCL_TRY(grp.end());
    }
    for (int i = 0; i < n; i++) {
        HIPC_TRY(hipSetDevice(cs[i]->device));
        HIPC_TRY(hipMemcpyAsync(ptr[i], hdr, sizeof(hdr), hipMemcpyHostToDevice, cs[i]->stream));
        HIPC_TRY(hipStreamSynchronize(cs[i]->stream));
    }
    return FASTP_GPU_OK;
}

// The exact sharded protocol's exchange (between fastp_gpu_submit_pass1_device and ..._pass2_device): rank r
// receives the OR of the duplicate bitmaps of ranks 0..r-1.  Transpose - scan - transpose: slice s of every
// rank's image goes to rank s (send/recv group = all-to-all), rank s scans its slices over the ranks in place
// (fq_or_images_kernel through fastp_gpu_prefix_or_images), the scanned slices go back.  2 (N-1)/N images cross
// the links per rank instead of the N-1 of an all-gather.
int fastp_gpu_exchange_dup_prefix(fastp_gpu_ctx* const* ctxs, int n) {
    if (!ctxs || n < 1) return fail(FASTP_GPU_E_INVALID, "bad argument");
    Pin pin;   // a concurrent fastp_gpu_comm_destroy / fastp_gpu_destroy of one of the contexts waits for this call
    if (!pin.take(ctxs, n)) return fail(FASTP_GPU_E_INVALID, "context has no communicator (fastp_gpu_comm_init*)");
    const std::vector<Comm*>& cs = pin.cs;
    int64_t bytes = 0;
    for (int i = 0; i < n; i++) {
        const int64_t b = fastp_gpu_dup_bitmap_bytes(ctxs[i]);
        if (i && b != bytes) return fail(FASTP_GPU_E_INVALID, "contexts with different duplicate geometry");
        bytes = b;
    }
    const int W = cs[0]->nranks;
    if (bytes == 0 || W == 1) {
        for (int i = 0; i < n; i++) {
            const int rc = fastp_gpu_dup_prefix_set(ctxs[i], nullptr, 0);
            if (rc) return fail(rc, fastp_gpu_last_error(ctxs[i]));
        }
        return FASTP_GPU_OK;
    }
    if (bytes % (16 * (int64_t)W)) return fail(FASTP_GPU_E_INVALID, "bitmap size does not split into 16-byte aligned slices per rank");
    const int64_t slice = bytes / W;
    for (int i = 0; i < n; i++) {
        Comm& c = *cs[i];
        HIPC_TRY(hipSetDevice(c.device));
        if (c.image_bytes != bytes) {
            if (c.images) (void)hipFree(c.images);
            if (c.slices) (void)hipFree(c.slices);
            c.images = c.slices = nullptr;
            if (hipMalloc(&c.images, (size_t)bytes) != hipSuccess || hipMalloc(&c.slices, (size_t)bytes) != hipSuccess)
                return fail(FASTP_GPU_E_NOMEM, "hipMalloc(bitmap exchange buffers) failed");
            c.image_bytes = bytes;
        }
        const int rc = fastp_gpu_dup_bitmap_export(ctxs[i], c.images);   // synchronous: complete before RCCL reads it
        if (rc) return fail(rc, fastp_gpu_last_error(ctxs[i]));
    }
    auto all_to_all = [&](bool back) -> int {
        {
            Group grp;
            NCCL_TRY(grp.start());
            for (int i = 0; i < n; i++) {
                Comm& c = *cs[i];
                HIPC_TRY(hipSetDevice(c.device));
                const char* src = (const char*)(back ? c.slices : c.images);
                char* dst = (char*)(back ? c.images : c.slices);
                for (int peer = 0; peer < W; peer++) {
                    NCCL_TRY(g_rccl.Send(src + (size_t)peer * slice, (size_t)slice, ncclUint8, peer, c.comm, c.stream));
                    NCCL_TRY(g_rccl.Recv(dst + (size_t)peer * slice, (size_t)slice, ncclUint8, peer, c.comm, c.stream));
                }
            }
            NCCL_TRY(grp.end());
        }
        for (int i = 0; i < n; i++) {
            HIPC_TRY(hipSetDevice(cs[i]->device));
            HIPC_TRY(hipStreamSynchronize(cs[i]->stream));
        }
        return FASTP_GPU_OK;
    };
    int rc = all_to_all(false);           // slices[k] = rank k's image, my slice
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        rc = fastp_gpu_prefix_or_images(ctxs[i], cs[i]->slices, W, slice);   // exclusive scan over the ranks, in place
        if (rc) return fail(rc, fastp_gpu_last_error(ctxs[i]));
    }
    rc = all_to_all(true);                // images[s] = my prefix, slice s  -> the whole prefix image
    if (rc) return rc;
    for (int i = 0; i < n; i++) {
        rc = fastp_gpu_dup_prefix_set(ctxs[i], cs[i]->images, 1);
        if (rc) return fail(rc, fastp_gpu_last_error(ctxs[i]));
    }
    return FASTP_GPU_OK;
}

}  // extern "C"
